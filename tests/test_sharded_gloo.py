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


    # ---- the side-car-less sharded decode (sharded.decode_foreign_sharded): bit-serial stand-ins for K6 / K7
    device = "cpu"

    def parse_header(self, host_stream):
        code, hs = self.orc.parse_header(host_stream)
        return code, hs

    def code_to_device(self, code):
        return code

    @staticmethod
    def _decode_from(piece, code, bit, end_bit, out=None):
        """decode codes that START in [bit, end_bit); returns (bit behind the last one, count, saw the end mark)"""
        fc, sp, sym = list(code.first_code), list(code.start_pos), list(code.symbol)
        n = 0
        while bit < end_bit:
            v, l = 0, 0
            while True:
                v = (v << 1) | ((int(piece[bit >> 3]) >> (7 - (bit & 7))) & 1)
                bit += 1
                l += 1
                if l >= code.min_len and l <= code.max_len and fc[l] != 1024 and v >= fc[l]:
                    s = sym[sp[l] + v - fc[l]]
                    break
                if l > code.max_len:
                    return bit, n, True  # garbage of a wrong guess: treat like an end
            if s == 256:
                return bit, n, True
            if out is not None:
                out.append(s)
            n += 1
        return bit, n, False

    def sync_piece(self, d_piece, piece_bytes, first_bit, end_bit, code):
        self._piece = (d_piece.numpy(), first_bit, end_bit, code)
        bit, n, eof = self._decode_from(d_piece.numpy(), code, first_bit, end_bit)
        return (0 if eof else bit - end_bit), n, eof

    def decode(self, d_piece, piece_bytes, code, index, cap=None):
        piece, first_bit, end_bit, _ = self._piece
        out = []
        self._decode_from(piece, code, first_bit, end_bit, out)
        return torch.tensor(out, dtype=torch.uint8), None


def _foreign_worker(rank, world, port, kind, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import datagen as dg
    import pkgload

    pkgload.load()
    from golden_huffman_amd import sharded
    from oracle import oracle as orc

    data = dg.make(kind, n_total, seed=6)
    stream = orc.compress(data)  # what the reference would have written: no side-car
    d_out, n_local, offset = sharded.decode_foreign_sharded(OracleBackend(), dist, stream)
    q.put((rank, offset, d_out[:n_local].numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,n_total,world", [("zipf", 20001, 2), ("text", 30000, 3), ("sym16", 9999, 2)])
def test_foreign_stream_decoded_by_several_ranks(kind, n_total, world):
    import datagen as dg

    port = 29300 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_foreign_worker, args=(r, world, port, kind, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    data = dg.make(kind, n_total, seed=6)
    pos = 0
    for rank, offset, chunk in got:
        assert offset == pos, (rank, offset, pos)
        part = np.frombuffer(chunk, dtype=np.uint8)
        assert np.array_equal(part, data[pos : pos + part.size]), rank
        pos += part.size
    assert pos == n_total


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
