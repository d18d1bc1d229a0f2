#!/usr/bin/env python3
"""Per-kernel MFMA / LDS / VALU utilisation of one bench.py run from rocprofv3 derived-metric passes (GPU box, repo root).

    python tools/pmc_util.py --out profiles/r01_util_f32.json [-- --matmul bf16x3]
One counter per pass (derived metrics need several hardware counters each); no trace options besides the counters.
"""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess
import tempfile

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="profiles/util.json")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--script", default=None, help="another script of this repo to run instead of bench.py (its arguments after --), e.g. tools/vae_bench.py")
args, extra = ap.parse_known_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = collections.defaultdict(dict)
for counter in ("MfmaUtil", "LdsUtil", "VALUBusy", "MemUnitStalled"):
    d = tempfile.mkdtemp(prefix=f"pmc_{counter}_", dir=os.path.join(root, "gpurun_out"))
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", "python3"]
    if args.script:
        cmd += [os.path.join(root, args.script)] + [e for e in extra if e != "--"]
    else:
        cmd += [os.path.join(root, "bench.py"), "--steps", str(args.steps), "--warmup", "1",
                "--no-cpu-baseline", "--no-roofline", "--no-alt", "--split-streams", "0"] + [e for e in extra if e != "--"]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if r.returncode:
        print(f"pass {counter} failed (rc {r.returncode})")
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            a = agg[row["Kernel_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    for k, (n, tot) in agg.items():
        res[k][counter] = tot / n
out = {k: v for k, v in res.items() if "avd::" in k}
os.makedirs(os.path.dirname(os.path.join(root, args.out)), exist_ok=True)
json.dump({"note": "rocprofv3 derived metrics (percent), averaged over the launches of each kernel in one bench.py run",
           "kernels": out}, open(os.path.join(root, args.out), "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("MfmaUtil", 0.0)):
    print("  ".join(f"{c} {v.get(c, float('nan')):6.1f}" for c in ("MfmaUtil", "LdsUtil", "VALUBusy", "MemUnitStalled")), " ", k[:80])
