#!/usr/bin/env python3
"""Build-container check that bench.py's cpu_baseline port (oracle/ref_cpu_fast.py) runs at the REFERENCE's CPU speed.

Imports the reference (PYTHONPATH=/root/reference — exists only in the build container, never on the GPU box), loads the same
seeded weights into its MMDiT / MultiModalNoiseHead, and times, on the same threads:
  reference  MMDiT.forward at [B,421,512]                         (avdiff/models/mmdt.py:134-149)
  port       oracle.ref_cpu_fast.mmdit_forward   (what bench.py times as cpu_baseline)
  oracle     oracle.ref_cpu.mmdit_forward        (the adjudicator; slower by design)
and one whole C3 CFG step of port vs oracle.  Writes profiles/r03_cpu_port_speed.json.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tools/cpu_port_speed.py [--batch 32] [--reps 3]
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import ref_cpu as R, ref_cpu_fast as RF  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--out", default=str(ROOT / "profiles" / "r03_cpu_port_speed.json"))
args = ap.parse_args()

from avdiff.models.mmdt import MMDiT  # noqa: E402  (the reference)

torch.manual_seed(0)
ws = R.synth_weights(seed=0)
B, N, d = args.batch, 421, 512
x = torch.randn(B, N, d)
ref = MMDiT(d_model=512, n_layers=8, n_heads=8, mlp_ratio=4.0).eval()
ref.load_state_dict(ws["core"], strict=True)


def timed(fn):
    with torch.no_grad():
        fn()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            out = fn()
        return (time.perf_counter() - t0) / args.reps, out


t_ref, y_ref = timed(lambda: ref(x))
t_port, y_port = timed(lambda: RF.mmdit_forward(x, ws["core"], 8, 8))
t_orc, y_orc = timed(lambda: R.mmdit_forward(x, ws["core"], 8, 8))
abar = R.alpha_bar_table(R.beta_table(1000))
z = torch.randn(B, 8, 12, 32, 32)
za = torch.randn(B, 8, 150)
tn, tp = torch.full((B,), 999), torch.full((B,), 979)
kw = dict(adapt_v=ws["adapt_v"], adapt_a=ws["adapt_a"], core=ws["core"], head=ws["head"], n_layers=8, n_heads=8, guidance=3.5)
ts_port, s_port = timed(lambda: RF.denoise_step_a2v(z, za, tn, tp, abar, **kw))
ts_orc, s_orc = timed(lambda: R.denoise_step_a2v(z, za, tn, tp, abar, **kw))
rec = {
    "threads": torch.get_num_threads(), "cpu_count": os.cpu_count(), "shape": [B, N, d], "reps": args.reps,
    "mmdit_forward_ms": {"reference": 1e3 * t_ref, "port_ref_cpu_fast": 1e3 * t_port, "oracle_ref_cpu": 1e3 * t_orc},
    "port_over_reference": t_port / t_ref, "oracle_over_reference": t_orc / t_ref,
    "max_abs_diff": {"port_vs_reference": float((y_port - y_ref).abs().max()), "oracle_vs_reference": float((y_orc - y_ref).abs().max())},
    "c3_step_ms": {"port_ref_cpu_fast": 1e3 * ts_port, "oracle_ref_cpu": 1e3 * ts_orc},
    "c3_step_rel_diff_port_vs_oracle": float((s_port - s_orc).abs().max() / max(1.0, float(s_orc.abs().max()))),
}
print(json.dumps(rec, indent=1))
Path(args.out).write_text(json.dumps(rec, indent=1) + "\n")
