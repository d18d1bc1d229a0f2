#!/usr/bin/env python3
"""Print value + the per-kernel table of one bench.py JSON line (stdin or file)."""
import json, sys
d = json.loads(open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read())
print("value %.1f %s  ms/step %.3f  matmul %s  launch %s" % (d["value"], d["unit"], d["ms_per_step"], d["config"]["matmul"], d["config"]["launch"]))
ks = d.get("kernels", {})
tot = sum(v["ms_per_step"] for v in ks.values())
print("  sum of kernels %.3f ms/step, launches/step %.0f" % (tot, sum(v["launches_per_step"] for v in ks.values())))
for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms_per_step"])[:14]:
    print("  %-58s %.4f ms x%-3.0f = %5.1f us each" % (k[:58], v["ms_per_step"], v["launches_per_step"], 1e3 * v["ms_per_step"] / max(v["launches_per_step"], 1)))
