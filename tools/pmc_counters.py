#!/usr/bin/env python3
"""Raw hardware counters per kernel launch of one bench.py run (GPU box, repo root): one rocprofv3 --pmc pass per counter SET.

    python tools/pmc_counters.py --out profiles/r05_l2_bf16x3.json --sets "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" [-- bench args]
Averages per launch and kernel; with TCC_HIT_sum / TCC_MISS_sum present also the L2 hit rate (MI355X_MICROARCH.md, L2).  No trace options
besides the counters (gpurun refuses --pmc combined with the trace domains)."""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="profiles/counters.json")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--sets", nargs="+", default=["TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"])
ap.add_argument("--match", default="avd::", help="keep kernels whose name contains this")
ap.add_argument("--script", default=None, help="another script of this repo to run instead of bench.py (its arguments after --), e.g. tools/vae_bench.py")
args, extra = ap.parse_known_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = collections.defaultdict(dict)
for cset in args.sets:
    d = tempfile.mkdtemp(prefix="pmc_set_", dir=os.path.join(root, "gpurun_out"))
    cmd = ["rocprofv3", "--pmc"] + cset.split() + ["--output-format", "csv", "-d", d, "--", "python3"]
    if args.script:
        cmd += [os.path.join(root, args.script)] + [e for e in extra if e != "--"]
    else:
        cmd += [os.path.join(root, "bench.py"), "--steps", str(args.steps), "--warmup", "1",
                "--no-cpu-baseline", "--no-roofline", "--no-alt", "--split-streams", "0"] + [e for e in extra if e != "--"]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    if r.returncode:
        print(f"pass [{cset}] failed (rc {r.returncode}): {r.stderr[-400:]}")
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            a = agg[row["Kernel_Name"]][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    for k, cs in agg.items():
        for c, (n, tot) in cs.items():
            res[k][c] = tot / n
            res[k]["launches_sampled"] = n
out = {k: v for k, v in res.items() if args.match in k}
for v in out.values():
    if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v and v["TCC_HIT_sum"] + v["TCC_MISS_sum"] > 0:
        v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
sys.path.insert(0, root)
from bench import csrc_hash      # noqa: E402
os.makedirs(os.path.dirname(os.path.join(root, args.out)), exist_ok=True)
json.dump({"note": "raw rocprofv3 counters, average per launch", "csrc_sha16": csrc_hash(), "bench_args": extra, "env": {k: v for k, v in os.environ.items() if k.startswith("AVD_")},
           "kernels": out}, open(os.path.join(root, args.out), "w"), indent=1)
cols = sorted({c for v in out.values() for c in v if c != "launches_sampled"})
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", kv[1].get(cols[0], 0.0) if cols else 0.0))[:14]:
    print("  ".join(f"{c} {v.get(c, float('nan')):14.4g}" for c in cols), " ", k[:70])
