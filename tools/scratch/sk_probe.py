"""scratch: bf16 (one-plane) mode at C2 — pairwise differences between split-K slice counts, and each against the oracle"""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_parity as T
from conftest import rel_err
from oracle import ref_cpu as R
dev = torch.device("cuda:0")
ws = R.synth_weights(seed=0)
mods = T._full_modules(dev, ws)
T._tune("s3_min_rows", 0)
outs = {}
for ns in (0, 2, 4, 8):
    T._tune("s3_splitk", ns)
    out, ref = T._one_step(dev, mods, ws, 64, 32, 2, matmul="bf16")
    outs[ns] = out
    print("ns", ns, "vs oracle", rel_err(out[:2], ref))
ks = list(outs)
for i in range(len(ks)):
    for j in range(i + 1, len(ks)):
        print(ks[i], ks[j], rel_err(outs[ks[i]], outs[ks[j]]), "first2:", rel_err(outs[ks[i]][:2], outs[ks[j]][:2]))
# where is the difference concentrated?
d = (outs[0] - outs[4]).abs().flatten(1).max(1).values
print("per-sample max diff 0 vs 4:", [round(float(x), 4) for x in d])
