import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_first: a GPU test that launches child processes; ordered before every other test")


def pytest_collection_modifyitems(config, items):
    """Tests marked `gpu_first` start child processes that use the GPU (torch.distributed.run): they must run BEFORE this process
    touches the device (on the GPU boxes a process that has initialised HIP may not fork + exec another program), so they go first."""
    first = [it for it in items if it.get_closest_marker("gpu_first")]
    if first:
        rest = [it for it in items if not it.get_closest_marker("gpu_first")]
        items[:] = first + rest


def load_golden(name):
    with np.load(GOLDEN / name, allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def split_weights(g, dtype=torch.float32):
    """{'core/…': arr} -> {'core': {name: tensor}, …} for the reduced-model fixtures."""
    out = {}
    for k, v in g.items():
        if "/" in k:
            grp, name = k.split("/", 1)
            if grp in ("core", "head", "adapt_v", "adapt_a", "w"):
                out.setdefault(grp, {})[name] = torch.from_numpy(v).to(dtype)
    return out


@pytest.fixture(scope="session")
def small_model():
    g = load_golden("g5_mmdit_small.npz")
    meta = json.loads(str(g["meta"]))
    return g, split_weights(g), meta


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / max(1.0, float(b.abs().max())))
