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


def _shard_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from multimodal_diffusion_amd import dist as D
    D.init_from_env("gloo")
    n = 7                                                    # ragged: 4 + 3 items
    cond_all = torch.arange(n * 6, dtype=torch.float32).view(n, 2, 3)
    seen = []

    def step(cond_part, lo, hi):                             # stand-in for the denoising loop: per-item, no communication
        seen.append((lo, hi, cond_part.clone()))
        return cond_part.sum(-1) * 2.0 + torch.arange(lo, hi, dtype=torch.float32)[:, None]

    got = D.run_sharded(n, cond_all if rank == 0 else None, (n, 2, 3), torch.device("cpu"), step)
    want = cond_all.sum(-1) * 2.0 + torch.arange(n, dtype=torch.float32)[:, None]
    lo, hi = D.shard_range(n, rank, world)
    ok = torch.equal(got, want) and len(seen) == 1 and seen[0][:2] == (lo, hi) and torch.equal(seen[0][2], cond_all[lo:hi])
    # more ranks than items: the empty shard contributes nothing and every rank still gets the whole result
    one = D.run_sharded(1, torch.ones(1, 4) if rank == 0 else None, (1, 4), torch.device("cpu"), lambda c, a, b: c * 3.0)
    ok = ok and torch.equal(one, torch.full((1, 4), 3.0))
    # result="root" — what stream_generate(shard=True) uses since round 5: every rank DECODES its own items (here: a uint8 stand-in of
    # another trailing shape than the conditioning) and ONE gather brings them to rank 0 only; ragged shards; the other ranks get None
    def decode(cond_part, lo, hi):
        return (cond_part.sum((1, 2)).to(torch.int64)[:, None, None] + torch.arange(lo, hi)[:, None, None] + torch.arange(10).view(1, 2, 5)).to(torch.uint8)
    frames = D.run_sharded(n, cond_all if rank == 0 else None, (n, 2, 3), torch.device("cpu"), decode, result="root")
    want_f = (cond_all.sum((1, 2)).to(torch.int64)[:, None, None] + torch.arange(n)[:, None, None] + torch.arange(10).view(1, 2, 5)).to(torch.uint8)
    ok = ok and ((frames is None) if rank else (frames.dtype == torch.uint8 and torch.equal(frames, want_f)))
    # more ranks than items with result="root": rank 1's shard is empty, it returns an empty tensor of the right trailing shape
    one_r = D.run_sharded(1, torch.ones(1, 4) if rank == 0 else None, (1, 4), torch.device("cpu"),
                          lambda c, a, b: (c[:, :2] * 5.0).to(torch.uint8), result="root")
    ok = ok and ((one_r is None) if rank else torch.equal(one_r, torch.full((1, 2), 5, dtype=torch.uint8)))
    # rank 0 fails while preparing the conditioning: every rank raises instead of waiting for a broadcast that never comes
    try:
        D.run_sharded(n, None, (n, 2, 3), torch.device("cpu"), step, error=KeyError("encode failed") if rank == 0 else None)
        raised = None
    except KeyError as e:
        raised = "src:" + str(e)
    except RuntimeError as e:
        raised = "peer:" + str(e)[:6]
    ok = ok and raised == ("src:'encode failed'" if rank == 0 else "peer:rank 0")
    # ... and a conditioning whose shape the other ranks did not expect is reported the same way
    try:
        D.run_sharded(n, cond_all[:, :1] if rank == 0 else None, (n, 2, 3), torch.device("cpu"), step)
        raised = None
    except (ValueError, RuntimeError) as e:
        raised = type(e).__name__
    ok = ok and raised == ("ValueError" if rank == 0 else "RuntimeError")
    D.barrier()
    out.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_run_sharded_world2():
    """dist.run_sharded — the library entry stream_generate(shard=True) is built on: ONE broadcast of the conditioning from rank 0,
    contiguous (ragged) shards, the per-shard function called once per rank with exactly its slice, one all-gather of the results in
    item order — or, result="root" (every rank decodes its own windows), one gather of the decoded uint8
    items to rank 0 only; more ranks than items; a rank-0 failure before the broadcast raises on EVERY rank.
    World size 2 over gloo on CPU with a stand-in step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_run_sharded_single_process():
    from multimodal_diffusion_amd import dist as D
    c = torch.arange(10, dtype=torch.float32).view(5, 2)
    assert torch.equal(D.run_sharded(5, c, (5, 2), torch.device("cpu"), lambda part, lo, hi: part + lo), c)
    with pytest.raises(ValueError):
        D.run_sharded(5, c, (4, 2), torch.device("cpu"), lambda part, lo, hi: part)
    with pytest.raises(ValueError):
        D.run_sharded(5, c, (5, 2), torch.device("cpu"), lambda part, lo, hi: part[:1])
    assert torch.equal(D.run_sharded(5, c, (5, 2), torch.device("cpu"), lambda part, lo, hi: part * 2, result="root"), c * 2)
    with pytest.raises(ValueError):
        D.run_sharded(5, c, (5, 2), torch.device("cpu"), lambda part, lo, hi: part, result="some")
    with pytest.raises(KeyError):
        D.run_sharded(5, c, (5, 2), torch.device("cpu"), lambda part, lo, hi: part, error=KeyError("x"))


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


@pytest.mark.gpu
@pytest.mark.gpu_first
def test_stream_generate_sharded_two_ranks_share_device(tmp_path):
    """The library's data-parallel entry on hardware: `stream_generate(shard=True)` launched as two FRESH
    child ranks under `python -m torch.distributed.run` (gloo for the one broadcast and the one gather, both ranks on cuda:0).  Since
    round 5 every rank also DECODES its own windows (VideoVAE / AudioCodec on each rank) and the gather carries
    decoded uint8 frames / waveforms to rank 0, which only stitches.  Rank 0 then repeats the generation single-process in the same
    worker and compares: the stitched uint8 video (audio prompt) and the stitched waveform (video prompt, the V -> A twin) of the
    sharded runs equal the single-process results BIT FOR BIT (tests/_stream_shard_worker.py).  This process never touches the device."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    if torch.cuda.device_count() < 1:
        pytest.skip("needs a GPU")
    if torch.cuda.is_initialized():
        pytest.skip("this process already initialised the GPU: run this test first / alone (conftest orders it first)")
    root = Path(__file__).resolve().parent.parent
    out = tmp_path / "shard.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", AVD_TEST_OUT=str(out))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(root / "tests" / "_stream_shard_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = json.loads(out.read_text())
    assert d["world"] == 2 and d["windows"] == 4 and d["shards"] == [[0, 2], [2, 4]], d
    assert d["rank1_returned_none"] and d["frames_shape"][1:] == [32, 32, 3]
    assert d["bit_identical"], d
    assert d["rank1_returned_none"], d
    assert d["audio_windows"] == 2 and d["audio_len"] == 52000 and d["audio_finite_nonzero"] and d["audio_bit_identical"], d
