#!/usr/bin/env python3
"""Copy the judged summaries of one profiling round (gpurun_out/<tag>/, written by tools/profile_round.sh on the GPU box) into
profiles/ under the round's prefix, and print the numbers DESIGN.md quotes.      python tools/collect_profiles.py r02"""
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src, dst = ROOT / "gpurun_out" / tag, ROOT / "profiles"
dst.mkdir(exist_ok=True)
copied = []
for f in sorted(src.glob("*")):
    if f.is_file() and f.suffix in (".json", ".txt") and f.stat().st_size > 0 and not f.name.endswith(".err"):
        if f.name.startswith(("traffic_f32.txt", "traffic_f16x2.txt", "traffic_bf16x3.txt", "traffic_bf16x3_mlpfused.txt", "bench_n2.out", "k.txt")):
            continue
        shutil.copy(f, dst / f"{tag}_{f.name}")
        copied.append(f.name)
for d in sorted(src.glob("prof_*")):
    if d.is_dir():
        st = d / "run_kernel_stats.csv"
        if st.exists():
            shutil.copy(st, dst / f"{tag}_{d.name}_kernel_stats.csv")
            copied.append(d.name + "/run_kernel_stats.csv")
cb = ROOT / "gpurun_out" / f"{tag}_cpu_baselines.json"
if cb.exists():
    shutil.copy(cb, dst / f"{tag}_cpu_baselines.json")
    copied.append(cb.name)
print("copied:", ", ".join(copied))


def line(name):
    f = src / name
    if not f.exists():
        return None
    return json.loads(f.read_text())


for name in sorted(p.name for p in src.glob("bench*.json")):
    d = line(name)
    r = d.get("roofline", {})
    print(f"{name:32s} {d['value']:8.2f} {d['unit']}  {d['ms_per_step']:7.3f} ms  parity {d.get('parity_rel_err_vs_cpu_oracle')}  "
          f"cpu {d.get('cpu_baseline', {}).get('value')}  single {d.get('single_stream_steps_per_s')}  "
          f"roof {r.get('kernel')} {r.get('achieved', 0):.1f}/{r.get('peak', 0):.1f} = {r.get('frac', 0):.3f} traffic {r.get('traffic')} ({r.get('traffic_source')})")
    for a in d.get("alt", []):
        ar = a.get("roofline", {})
        print(f"    alt {a['matmul']:14s} {a['value']:8.2f}  parity {a.get('parity_rel_err_vs_cpu_oracle')}  roof {ar.get('kernel')} {ar.get('frac')}")
