"""GPU (-m gpu): the multi-GPU path on the one GPU this box has -- two ranks (processes) share cuda:0, the real
kernels run with GHF_EMIT_REBASE / bit-phase start offsets, the two collectives go over gloo (staged through the
host, since NCCL/RCCL refuses two ranks on one device).  The 8-GPU RCCL run is the driver's; this pins the
kernel side of it: shard streams OR-merge to the single-stream .crs2, and every rank decodes its own shard."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class HostStagedDist:
    """torch.distributed look-alike whose tensor collectives bounce through CPU tensors (gloo)"""

    def __init__(self, dist):
        self.d = dist
        self.ReduceOp = dist.ReduceOp

    def get_rank(self, group=None):
        return self.d.get_rank()

    def get_world_size(self, group=None):
        return self.d.get_world_size()

    def all_reduce(self, t, op=None, group=None):
        c = t.cpu()
        self.d.all_reduce(c, op=op)
        t.copy_(c)

    def all_gather_into_tensor(self, out, t, group=None):
        parts = [torch.zeros_like(t.cpu()) for _ in range(self.d.get_world_size())]
        self.d.all_gather(parts, t.cpu())
        out.copy_(torch.cat(parts))

    def all_gather_object(self, lst, obj, group=None):
        self.d.all_gather_object(lst, obj)

    def barrier(self):
        self.d.barrier()


def _shard_data(dg, kind, rank, world, lo, hi):
    """"hetero": low-entropy shards and ONE uniform shard -- the global code then costs the uniform shard more than
    9 bits per symbol (what ghf_compress_bound allows for a buffer's own code)"""
    if kind == "hetero":
        return dg.make("uniform" if rank == world - 1 else "sym16", hi - lo, seed=9, offset=lo)
    return dg.make(kind, hi - lo, seed=9, offset=lo)


def _worker(rank, world, port, kind, n_total, q):
    import traceback

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import datagen as dg
        import pkgload

        pkg = pkgload.load()
        from golden_huffman_amd import sharded

        ctx = pkg.ghf.Context(0)
        hd = HostStagedDist(dist)
        lo, hi = rank * n_total // world, (rank + 1) * n_total // world
        data = _shard_data(dg, kind, rank, world, lo, hi)
        shard = torch.from_numpy(data).cuda()
        index = ctx.index_alloc(shard.numel())
        enc = sharded.encode_sharded(ctx, hd, shard, index=index)
        err = ""
        ok = False
        try:
            ctx.sync()
            back, _ = sharded.decode_sharded(ctx, enc, index)
            ctx.sync()
            ok = bool((back[: shard.numel()] == shard).all().item())
        except Exception:  # keep the collectives below in step with the other rank
            err = traceback.format_exc()
        # the same shard once more into a buffer of EXACTLY ghf_shard_bytes(all-gathered totals): K5 must find it large enough
        # (GHF_E_CAP otherwise) and write the same bytes (every rank runs this: the exchanges stay in step)
        try:
            exact = ctx.shard_bytes(enc["code"], enc["totals"], world, rank)
        except Exception:
            exact, ok, err = pkg.ghf.shard_bound(shard.numel()), False, err + traceback.format_exc()
        small = ctx.empty_u8(exact)
        enc2 = sharded.encode_sharded(ctx, hd, shard, out=small)
        try:
            ctx.sync()
            nb = int(enc["end"][1].item())
            assert nb <= exact <= nb + 16, (nb, exact)
            assert int(enc2["end"][1].item()) == nb and bool((small[:nb] == enc["out"][:nb]).all().item())
            assert exact < pkg.ghf.shard_bound(shard.numel())
        except Exception:
            ok, err = False, err + traceback.format_exc()
        stream = sharded.gather_stream(ctx, hd, enc)
        q.put((rank, ok, stream.tobytes() if rank == 0 else None, err))
        dist.barrier()
        ctx.index_free(index)
        ctx.close()
    except Exception:
        q.put((rank, False, None, traceback.format_exc()))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,n_total", [("zipf", 3 * (1 << 20) + 17), ("uniform", 1 << 21), ("sym16", 777777)])
def test_two_ranks_on_one_gpu(kind, n_total):
    import datagen as dg
    from oracle import oracle as orc

    world = 2
    port = 29700 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(g[1] for g in got), "a rank failed to decode its own shard: %s" % [g[3] for g in got]
    stream = next(g[2] for g in got if g[2] is not None)
    ref = orc.compress(dg.make(kind, n_total, seed=9))
    got_stream = np.frombuffer(stream, dtype=np.uint8)
    assert got_stream.size == ref.size and np.array_equal(got_stream, ref)


def test_three_ranks_heterogeneous_shards_on_one_gpu():
    """two 16-symbol shards and one uniform shard: the uniform one is packed at ~10 bits per symbol"""
    import datagen as dg
    from oracle import oracle as orc

    world, n_total, kind = 3, 3 * 400000 + 5, "hetero"
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(g[1] for g in got), "a rank failed to decode its own shard: %s" % [g[3] for g in got]
    whole = np.concatenate([_shard_data(dg, kind, r, world, r * n_total // world, (r + 1) * n_total // world) for r in range(world)])
    ref = orc.compress(whole)
    got_stream = np.frombuffer(next(g[2] for g in got if g[2] is not None), dtype=np.uint8)
    assert got_stream.size == ref.size and np.array_equal(got_stream, ref)
    # the uniform shard really does not fit compress_bound: that is what shard_bound is for
    import pkgload

    ghf = pkgload.load().ghf
    body_bits = 8 * (ref.size - 1040 - 8 * 32)
    assert body_bits > 0 and ghf.shard_bound(n_total // world) >= 4 * (n_total // world)


def test_encode_sharded_c_abi_world1_over_rccl():
    """ghf_encode_sharded with a REAL RCCL communicator of one rank (ncclCommInitRank through the C ABI): the two
    collectives are queued on the context's stream; the result is the single-stream .crs2"""
    import datagen as dg
    import pkgload
    from oracle import oracle as orc

    ghf = pkgload.load().ghf
    assert ghf.rccl_version(), "RCCL not bound"
    ctx = ghf.Context(0)
    comm = ctx.comm_init(ghf.comm_unique_id(), 1, 0)
    try:
        for kind, n in (("zipf", 1 << 20), ("uniform", 300001)):
            data = dg.make(kind, n, seed=5)
            d_in = torch.from_numpy(data).cuda()
            idx = ctx.index_alloc(n)
            enc = ctx.encode_sharded(comm, d_in, index=idx)
            ctx.sync()
            nb = int(enc["end"][1].item())
            ref = orc.compress(data)
            assert nb == ref.size and np.array_equal(enc["out"][:nb].cpu().numpy(), ref)
            back, _ = ctx.decode(enc["out"], nb, enc["code"], idx)
            ctx.sync()
            assert np.array_equal(back[:n].cpu().numpy(), data)
            ctx.index_free(idx)
        # the stand-alone collectives: a no-op sum over one rank, a gather of one value
        h = ctx.histogram(d_in)
        before = h.clone()
        ctx.comm_allreduce_hist(comm, h)
        tot = torch.tensor([12345], dtype=torch.int64, device="cuda")
        tots = torch.zeros(1, dtype=torch.int64, device="cuda")
        ctx.comm_allgather_total(comm, tot, tots)
        ctx.sync()
        assert bool((h == before).all().item()) and int(tots.item()) == 12345
    finally:
        ctx.comm_destroy(comm)
        ctx.close()


def _bench_two_ranks(extra_env=None, timeout=400):
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29900 + os.getpid() % 90), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--backend", "gloo", "--mib", "32"]
    env = dict(os.environ)
    env.update(extra_env or {})
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_two_rank_rehearsal():
    """bench.py's N>1 control flow (offsets, flags, pipelining over rotating buffer sets, index flags, round-trip checks, the
    same-shard single-GPU run in front of the sharded one) with two ranks on this one GPU: `--backend gloo` stages the
    two collectives through the host.  The RCCL run is the driver's."""
    import json

    r = _bench_two_ranks()
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "shard2" and d["config"]["buffer_sets"] >= 3
    assert d["config"]["side_streams"] == 1  # N > 1: collectives in strict step order on one stream
    assert d["same_shard_1gpu_ms_per_step"] > 0 and d["speedup_vs_same_shard"] > 0


def test_bench_watchdog_names_the_stuck_stage():
    """a rank that never joins a collective must end the run within the step timeout, with the stage named -- not sit there
    until the driver's limit (rank 1 stays away from step 1's all-reduce)."""
    import time

    t0 = time.time()
    r = _bench_two_ranks({"GHF_BENCH_INJECT_HANG": "1:1", "GHF_BENCH_STEP_TIMEOUT": "8"}, timeout=300)
    assert r.returncode != 0
    assert "no progress for 8 s" in r.stderr and "allreduce of step 1" in r.stderr, r.stderr[-2000:]
    assert time.time() - t0 < 200


def _foreign_worker(rank, world, port, kind, n_total, q):
    import traceback

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import datagen as dg
        import pkgload
        from oracle import oracle as orc

        pkg = pkgload.load()
        from golden_huffman_amd import sharded

        ctx = pkg.ghf.Context(0)
        data = dg.make(kind, n_total, seed=12)
        stream = orc.compress(data)  # byte-identical to what the reference writes: no side-car
        d_out, n_local, offset = sharded.decode_foreign_sharded(ctx, dist, stream)
        ctx.sync()
        q.put((rank, offset, d_out[:n_local].cpu().numpy().tobytes(), ""))
        dist.barrier()
        ctx.close()
    except Exception:
        q.put((rank, -1, b"", traceback.format_exc()))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,n_total,world", [("zipf", (1 << 21) + 5, 2), ("uniform", 1 << 20, 3), ("text", 1500001, 2)])
def test_foreign_stream_decoded_by_several_ranks_on_one_gpu(kind, n_total, world):
    """SURVEY 8(e), decode without a side-car: every rank re-synchronises on its byte range of the body (K6 in piece
    mode), the ranks exchange where their last codes end until nothing moves, symbol counts give the output offsets,
    and each rank decodes its piece (K7).  Pieces in rank order == the original."""
    import datagen as dg

    port = 29100 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_foreign_worker, args=(r, world, port, kind, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=200) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(not g[3] for g in got), [g[3] for g in got]
    data = dg.make(kind, n_total, seed=12)
    pos = 0
    for rank, offset, chunk, _ in got:
        assert offset == pos, (rank, offset, pos)
        part = np.frombuffer(chunk, dtype=np.uint8)
        assert np.array_equal(part, data[pos : pos + part.size]), rank
        pos += part.size
    assert pos == n_total
