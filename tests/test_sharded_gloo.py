"""CPU, world_size 2 (gloo): the sharded driver golden-huffman_amd/sharded.py -- the all-reduce of the 256 byte
counts, the all-gather of the bit totals, the per-rank global bit offsets, the REBASE local buffers and the
OR-merge of the shared boundary bytes.  The stages themselves are played by a stand-in backend built on the
oracle (this is a no-GPU test of the HOST logic; the real stages are tested against the oracle in -m gpu)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleBackend:
    """same method surface as golden_huffman_amd.ghf.Context, CPU tensors, oracle arithmetic"""

    EMIT_LAST = 1
    EMIT_REBASE = 2
    torch = torch

    def __init__(self):
        from oracle import oracle as orc

        self.orc = orc

    @staticmethod
    def compress_bound(n):
        return ((1040 + 256 + (9 * (n + 1) + 7) // 8 + 15) & ~15) + 16

    def empty_u8(self, n):
        return torch.zeros(max(int(n), 1), dtype=torch.uint8)

    def histogram(self, d_in):
        return torch.from_numpy(self.orc.histogram(d_in.numpy()))

    def build_code(self, hist):
        code = self.orc.build_code(hist.numpy())
        return torch.from_numpy(np.frombuffer(bytes(code), dtype=np.uint8).copy())

    def _code(self, d_code):
        return self.orc.OrcCode.from_buffer_copy(d_code.numpy().tobytes())

    def write_header(self, d_code, out):
        h = self.orc.header_bytes(self._code(d_code))
        out[: h.size] = torch.from_numpy(h)

    def encode_plan(self, d_in, d_code):
        code = self._code(d_code)
        lens = np.array(list(code.length), dtype=np.int64)
        return torch.tensor([int(lens[d_in.numpy()].sum())], dtype=torch.int64)

    def encode_emit(self, d_in, d_code, out, start_bit=None, flags=1, index=None):
        code = self._code(d_code)
        sb = int(start_bit.item()) if start_bit is not None else 8 * (1040 + 8 * code.max_len)
        origin = ((sb >> 7) << 4) if (flags & self.EMIT_REBASE) else 0
        a = np.ascontiguousarray(d_in.numpy())
        buf = out.numpy()
        off = (sb >> 3) - origin
        tmp = np.zeros(buf.size - off, dtype=np.uint8)
        bits = self.orc.lib().orc_pack_at(a.ctypes.data, a.size, C.byref(code), sb & 7, 1 if (flags & self.EMIT_LAST) else 0,
                                          tmp.ctypes.data, tmp.size)
        assert bits != 2**64 - 1
        nb = ((sb & 7) + bits + 7) // 8
        buf[off : off + nb] |= tmp[:nb]
        end = sb + bits
        return torch.tensor([end, (end + 7) // 8 - origin], dtype=torch.int64)


def _worker(rank, world, port, kind, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import datagen as dg
    import pkgload

    pkg = pkgload.load()
    from golden_huffman_amd import sharded

    be = OracleBackend()
    lo, hi = rank * n_total // world, (rank + 1) * n_total // world
    shard = torch.from_numpy(dg.make(kind, hi - lo, seed=5, offset=lo))
    enc = sharded.encode_sharded(be, dist, shard)
    stream = sharded.gather_stream(be, dist, enc)
    if rank == 0:
        q.put((stream.tobytes(), enc["totals"].tolist(), int(enc["start_bit"].item())))
    else:
        q.put((None, None, int(enc["start_bit"].item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,n_total", [("zipf", 200001), ("uniform", 65536 * 2 + 7), ("sym16", 99999)])
def test_two_rank_sharded_stream_equals_single_stream(kind, n_total):
    import datagen as dg
    from oracle import oracle as orc

    world = 2
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    stream = next(g[0] for g in got if g[0] is not None)
    whole = dg.make(kind, n_total, seed=5)
    ref = orc.compress(whole)
    assert np.array_equal(np.frombuffer(stream, dtype=np.uint8), ref)
    # the offsets: rank 1 starts where rank 0's body bits end
    code = orc.build_code(orc.histogram(whole))
    lens = np.array(list(code.length), dtype=np.int64)
    half = n_total // world
    hdr_bits = 8 * (1040 + 8 * code.max_len)
    starts = sorted(g[2] for g in got)
    assert starts == [hdr_bits, hdr_bits + int(lens[whole[:half]].sum())]


def test_header_bits_of_reads_max_len_from_the_tables():
    import pkgload

    pkgload.load()
    from golden_huffman_amd import sharded
    from oracle import oracle as orc

    be = OracleBackend()
    code = orc.build_code(orc.histogram(np.frombuffer(b"aaaabbc", dtype=np.uint8)))
    t = torch.from_numpy(np.frombuffer(bytes(code), dtype=np.uint8).copy())
    assert int(sharded.header_bits_of(be, t).item()) == 8 * 1064
