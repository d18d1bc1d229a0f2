#!/usr/bin/env python3
"""HBM traffic per kernel launch from rocprofv3 PMC passes (run on the GPU box, from the repo root).

Two separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; no trace options besides the counters).
Corrections per MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes
of a wide coalesced 16-B/lane stream (our kernels' pattern, LDS-DMA included) -> doubled; WRITE_SIZE is exact for
16-B/lane streaming stores.  Infinity-Cache hits are counted, so this is traffic beyond L2, an upper bound on HBM.

    python tools/pmc_traffic.py --out profiles/r01_traffic.json [-- extra bench.py args]
"""
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
ap.add_argument("--out", default="profiles/traffic.json")
ap.add_argument("--steps", type=int, default=3)
args, extra = ap.parse_known_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = collections.defaultdict(dict)
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = tempfile.mkdtemp(prefix=f"pmc_{counter}_", dir=os.path.join(root, "gpurun_out"))
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--",
           "python3", os.path.join(root, "bench.py"), "--steps", str(args.steps), "--warmup", "1",
           "--no-cpu-baseline", "--no-roofline", "--no-alt", "--split-streams", "0"] + [e for e in extra if e != "--"]
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(cmd, check=True, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for k, (n, tot) in agg.items():
        res[k][counter] = {"launches": n, "avg_kib": tot / n}
out = {}
for k, v in res.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        fetch = 2.0 * v["FETCH_SIZE"]["avg_kib"] * 1024.0
        write = v["WRITE_SIZE"]["avg_kib"] * 1024.0
        out[k] = {"launches_sampled": v["FETCH_SIZE"]["launches"], "fetch_bytes_per_launch": fetch,
                  "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
                  "raw_fetch_kib": v["FETCH_SIZE"]["avg_kib"], "raw_write_kib": v["WRITE_SIZE"]["avg_kib"]}
os.makedirs(os.path.dirname(os.path.join(root, args.out)), exist_ok=True)
sys.path.insert(0, root)
from bench import csrc_hash      # noqa: E402  (identity of the kernel sources this profile belongs to; bench.py drops it when they change)
json.dump({"note": "bytes beyond L2 per launch; FETCH_SIZE doubled per the gfx950 correction; KiB units",
           "csrc_sha16": csrc_hash(), "bench_args": extra, "kernels": out},
          open(os.path.join(root, args.out), "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"])[:12]:
    print(f"{v['traffic_bytes_per_launch'] / 1e6:10.1f} MB/launch  (fetch {v['fetch_bytes_per_launch'] / 1e6:8.1f} write {v['write_bytes_per_launch'] / 1e6:8.1f})  {k[:90]}")
