"""World-size-2 gloo run (CPU) of the multi-GPU plumbing: one broadcast of the conditioning latents from rank 0,
contiguous batch shards, max-over-ranks timing.  The per-rank compute itself is covered by the GPU tests; here the
'step' is a stand-in so the test exercises exactly the collective layout bench.py uses."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from multimodal_diffusion_amd import dist as D
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    B = 3                                                    # per-rank batch
    gshape = (B * world, 8, 150)
    cond = torch.randn(gshape, generator=torch.Generator().manual_seed(2)) if rank == 0 else None
    full = D.broadcast_conditioning(cond, gshape, torch.device("cpu"))
    mine = D.local_conditioning(full, rank, world)
    expect = torch.randn(gshape, generator=torch.Generator().manual_seed(2))
    lo, hi = D.shard_range(B * world, rank, world)
    ok = torch.equal(full, expect) and torch.equal(mine, expect[lo:hi]) and mine.shape[0] == B
    # ragged global batch (7 samples over 2 ranks): every rank gets the whole batch back in rank order
    glo, ghi = D.shard_range(7, rank, world)
    whole = torch.arange(7 * 5, dtype=torch.float32).view(7, 5)
    ok = ok and torch.equal(D.gather_batch(whole[glo:ghi].clone(), 7), whole)
    slow = D.max_over_ranks(1.0 + rank, torch.device("cpu"))
    D.barrier()
    out.put((rank, bool(ok), slow))
    dist.destroy_process_group()


def test_broadcast_and_shard_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, 2.0), (1, True, 2.0)]


def test_bench_setup_under_torchrun(tmp_path):
    """bench.py's own N > 1 setup, launched the way the driver launches it (torch.distributed.run, one process per rank),
    with gloo on CPU: env handling, group init, the conditioning broadcast and shards, per-rank latents, max-over-ranks."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    out = tmp_path / "res"
    env = dict(os.environ, AVD_TEST_OUT=str(out), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(root / "tests" / "_bench_setup_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = [json.loads(Path(f"{out}.{k}").read_text()) for k in range(2)]
    assert [x["rank"] for x in res] == [0, 1] and all(x["ok"] and x["world"] == 2 for x in res)
    assert all(x["slow"] == 1.5 and x["global_batch"] == 6 and x["nv"] == 24 for x in res)


@pytest.mark.gpu
@pytest.mark.gpu_first
def test_bench_two_ranks_share_device():
    """bench.py's N = 2 path on the one-GPU box: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 --backend gloo
    --share-device` as FRESH child processes (both ranks on cuda:0, gloo for the one broadcast) — the launch line the driver uses for
    N > 1 with the collective backend swapped.  Ordered first in the session (conftest: gpu_first): this process has not touched the
    device yet, and it never does here (device_count() does not initialise HIP).  Asserts: one JSON line, n_gpus == 2, both ranks
    finished with finite latents (bench.py asserts that per rank and the launcher propagates a failure), the broadcast conditioning is
    identical on both ranks, the two ranks stepped DIFFERENT shards (different latents), value == 2 x the global-batch rate."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    if torch.cuda.device_count() < 1:
        pytest.skip("needs a GPU")
    if torch.cuda.is_initialized():
        pytest.skip("this process already initialised the GPU: run this test first / alone (conftest orders it first)")
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--backend", "gloo",
           "--share-device", "--no-alt", "--no-cpu-baseline", "--no-roofline", "--verify-ranks"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and abs(d["value"] - 2 * d["global_batch_steps_per_s"]) < 1e-9 * d["value"]
    cs, ls = d["verify"]["conditioning_checksum_per_rank"], d["verify"]["latent_abs_sum_per_rank"]
    assert len(cs) == 2 and cs[0] == cs[1], cs
    assert len(ls) == 2 and all(v == v and v < float("inf") for v in ls) and ls[0] != ls[1], ls
