"""Worker of tests/test_dist_gloo.py::test_bench_setup_under_torchrun: runs bench.py's own pre-GPU setup (argument parsing,
RANK / LOCAL_RANK / WORLD_SIZE handling, process-group init, the ONE conditioning broadcast, shard selection) under
`python -m torch.distributed.run --nproc-per-node 2` with the gloo backend, on CPU."""
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from multimodal_diffusion_amd import dist as D  # noqa: E402

args = bench.parse_args(["--gpus", os.environ["WORLD_SIZE"], "--backend", "gloo", "--batch", "3", "--size", "64", "--sampler-steps", "7"])
ctx = bench.setup_run(args, need_gpu=False)
rank, world = ctx["rank"], ctx["world"]
expect = torch.randn((3 * world, 8, 150), generator=torch.Generator().manual_seed(2))
lo, hi = D.shard_range(3 * world, rank, world)
ok = torch.equal(ctx["z_a0"], expect[lo:hi]) and ctx["z0"].shape == (3, 8, 12, 8, 8) and ctx["sched"].numel() == 8
ok = ok and torch.equal(ctx["z0"], torch.randn(ctx["lat"], generator=torch.Generator().manual_seed(1 + rank)))
# ragged global batch through the optional gather epilogue (7 samples over the ranks)
glo, ghi = D.shard_range(7, rank, world)
whole = torch.arange(7 * 4, dtype=torch.float32).view(7, 4)
ok = ok and torch.equal(D.gather_batch(whole[glo:ghi].clone(), 7), whole)
slow = D.max_over_ranks(0.5 + rank, ctx["comm_dev"])
D.barrier()
Path(os.environ["AVD_TEST_OUT"] + f".{rank}").write_text(json.dumps({"rank": rank, "world": world, "ok": bool(ok), "slow": slow,
                                                                       "nv": ctx["nv"], "global_batch": ctx["global_batch"]}))
torch.distributed.destroy_process_group()
