"""GPU (-m gpu): the HIP path through the C ABI against the oracle and the committed golden vectors.
Bit-exact everywhere (integer/byte work: no tolerance)."""
import hashlib

import numpy as np
import pytest

import datagen as dg
import pkgload
from cases import CASES
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module")
def env():
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    pkg = pkgload.load()
    ctx = pkg.ghf.Context(0)
    yield pkg.ghf, ctx, torch
    ctx.close()


def to_dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_compress(ghf, ctx, torch, data, with_index=True):
    d_in = to_dev(torch, data)
    idx = ctx.index_alloc(data.size) if with_index else None
    d_out, nbytes, d_code = ctx.compress(d_in, index=idx)
    ctx.sync()
    nb = int(nbytes.item())
    return d_in, d_out, nb, d_code, idx


@pytest.mark.parametrize("name", list(CASES))
def test_every_stage_matches_reference_fixture(env, golden, name):
    ghf, ctx, torch = env
    g = golden[name]
    data = CASES[name]()
    assert sha(data) == g["input_sha256"]
    d_in = to_dev(torch, data)
    # K1
    hist = ctx.histogram(d_in)
    ctx.sync()
    assert hist.cpu().numpy().tolist() == g["hist"]
    # K2 + K3
    d_code = ctx.build_code(hist)
    ctx.sync()
    d = ctx.code_to_host(d_code).as_dict()
    for k in ("length", "codeword", "symbol", "first_code", "start_pos", "min_len", "max_len"):
        assert d[k] == g[k], k
    # whole pipeline
    _, d_out, nb, d_code2, idx = run_compress(ghf, ctx, torch, data)
    crs = d_out[:nb].cpu().numpy()
    assert nb == g["crs2_bytes"]
    assert sha(crs[: g["header_bytes"]]) == sha(orc.header_bytes(orc.build_code(orc.histogram(data))))
    assert sha(crs) == g["crs2_sha256"], "first diff at %s" % (first_diff(crs, orc.compress(data)),)
    # K7 with the side-car
    back, nout = ctx.decode(d_out, nb, d_code2, idx)
    ctx.sync()
    assert int(nout.item()) == data.size
    assert np.array_equal(back[: data.size].cpu().numpy(), data)
    ctx.index_free(idx)


def first_diff(a, b):
    n = min(a.size, b.size)
    d = np.nonzero(a[:n] != b[:n])[0]
    return (int(d[0]) if d.size else None, a.size, b.size)


@pytest.mark.parametrize("kind,n", [("uniform", (1 << 24) + 5), ("zipf", (1 << 24) - 3), ("sym16", 1 << 24), ("text", 3 << 20)])
def test_mid_size_bit_exact_vs_oracle(env, kind, n):
    ghf, ctx, torch = env
    data = dg.make(kind, n, seed=99)
    _, d_out, nb, d_code, idx = run_compress(ghf, ctx, torch, data)
    ref = orc.compress(data)
    crs = d_out[:nb].cpu().numpy()
    assert nb == ref.size and np.array_equal(crs, ref), first_diff(crs, ref)
    back, _ = ctx.decode(d_out, nb, d_code, idx)
    ctx.sync()
    assert np.array_equal(back[:n].cpu().numpy(), data)
    ctx.index_free(idx)


def test_unaligned_input_pointer(env):
    ghf, ctx, torch = env
    data = dg.zipf_bytes(200001, seed=5)
    big = to_dev(torch, np.concatenate([np.zeros(3, np.uint8), data]))
    d_in = big[3:]
    assert d_in.data_ptr() % 16 == 3
    idx = ctx.index_alloc(data.size)
    d_out, nbytes, d_code = ctx.compress(d_in, index=idx)
    ctx.sync()
    nb = int(nbytes.item())
    assert np.array_equal(d_out[:nb].cpu().numpy(), orc.compress(data))
    ctx.index_free(idx)


def test_plan_without_histogram_uses_direct_pass(env):
    """ghf_encode_plan must not depend on a preceding ghf_histogram of the same buffer."""
    ghf, ctx, torch = env
    data = dg.text_bytes(300000, seed=8)
    other = dg.uniform_bytes(300000, seed=9)
    d_in, d_other = to_dev(torch, data), to_dev(torch, other)
    hist = ctx.histogram(d_in)
    d_code = ctx.build_code(hist)
    ctx.histogram(d_other)  # invalidates the cached per-chunk histograms
    out = ctx.empty_u8(ghf.compress_bound(data.size))
    ctx.write_header(d_code, out)
    total = ctx.encode_plan(d_in, d_code)
    end = ctx.encode_emit(d_in, d_code, out)
    ctx.sync()
    ref = orc.compress(data)
    assert int(end[1].item()) == ref.size
    assert np.array_equal(out[: ref.size].cpu().numpy(), ref)
    h = orc.histogram(data)
    assert int(total.item()) == orc.body_bits(h, orc.build_code(h)) - orc.build_code(h).length[256]


def test_error_statuses(env):
    ghf, ctx, torch = env
    # empty input: refused (the reference is undefined there)
    with pytest.raises(ghf.GhfError) as e:
        ctx.compress(ctx.empty_u8(16), n=0)
    assert e.value.status == 3
    # 33-bit codes: refused (reference limit)
    h = np.zeros(257, dtype=np.int64)
    h[:33] = dg.fib_counts(33)
    h[256] = 1
    ctx.build_code(to_dev(torch, h))
    with pytest.raises(ghf.GhfError) as e:
        ctx.sync()
    assert e.value.status == 4
    # capacity
    data = dg.uniform_bytes(100000, seed=3)
    small = ctx.empty_u8(50000)
    ctx.compress(to_dev(torch, data), d_out=small)
    with pytest.raises(ghf.GhfError) as e:
        ctx.sync()
    assert e.value.status == 5
    # and the context still works afterwards
    d_out, nbytes, _ = ctx.compress(to_dev(torch, data))
    ctx.sync()
    assert np.array_equal(d_out[: int(nbytes.item())].cpu().numpy(), orc.compress(data))


FOREIGN = ["aaaabbc", "single_x", "all256_once", "uniform_64k", "zipf_64k", "sym16_64k", "fib22", "fib32_maxlen32", "geom16",
           "text_1m", "zeros_100000", "uniform_65537", "zipf_n3007", "two_values_skewed"]


@pytest.mark.parametrize("name", FOREIGN)
def test_foreign_stream_without_sidecar(env, name):
    """a .crs2 as the reference writes it (here: the oracle, byte-identical to it) decodes with no side-car:
    the library re-synchronises on the GPU (K6) and then runs the same table decode."""
    ghf, ctx, torch = env
    data = CASES[name]()
    crs = orc.compress(data)
    code, hs = ghf.parse_header(crs)
    d_stream = to_dev(torch, np.concatenate([crs, np.zeros(64, np.uint8)]))
    d_code = ctx.code_to_device(code)
    out, nout = ctx.decode(d_stream, crs.size, d_code, None, cap=data.size + 64)
    ctx.sync()
    assert int(nout.item()) == data.size
    assert np.array_equal(out[: data.size].cpu().numpy(), data)


@pytest.mark.parametrize("kind", ["uniform", "zipf", "sym16"])
def test_foreign_stream_16MiB(env, kind):
    ghf, ctx, torch = env
    n = (1 << 24) + 77
    data = dg.make(kind, n, seed=321)
    crs = orc.compress(data)
    code, hs = ghf.parse_header(crs)
    d_stream = to_dev(torch, crs)
    out, nout = ctx.decode(d_stream, crs.size, ctx.code_to_device(code), None, cap=n + 4096)
    ctx.sync()
    assert int(nout.item()) == n
    assert np.array_equal(out[:n].cpu().numpy(), data)


@pytest.mark.parametrize("kind", ["zipf", "uniform"])
def test_foreign_stream_followed_by_stale_bytes(env, kind):
    """a buffer that goes on behind its stream -- here the tail of a LONGER stream of the same code at another bit phase,
    then zeros, then ones -- is legal input: the first end mark ends the stream (the reference's decoder stops there).  K6's
    fixed-point passes must not wait for what lies behind the mark to settle (runs of zeros or ones need one pass per
    subsequence; round 3's and this round's first "4 GiB Zipf without side-car: 45..59 ms" were bench.py handing over a few
    hundred bytes more than the stream had): the driver stops as soon as nothing in front of the mark moves."""
    ghf, ctx, torch = env
    n = (1 << 22) + 1234
    data = dg.make(kind, n, seed=77)
    crs = orc.compress(data)
    longer = orc.compress(dg.make(kind, n + 40000, seed=78))
    tail = np.concatenate([longer[crs.size + 3: crs.size + 3 + 30000], np.zeros(20000, np.uint8), np.full(20000, 255, np.uint8)])
    buf = np.concatenate([crs, tail])
    code, hs = ghf.parse_header(crs)
    d_stream = to_dev(torch, buf)
    out, nout = ctx.decode(d_stream, buf.size, ctx.code_to_device(code), None, cap=n + 4096)
    ctx.sync()
    assert int(nout.item()) == n
    assert np.array_equal(out[:n].cpu().numpy(), data)


def _bytes_that_do_not_compress(n, rare, seed, values=256):
    """every one of `values` byte values equally often, `rare` a little less: values - 1 codes of log2(values) bits, one bit
    more for `rare` and the end mark"""
    rng = np.random.default_rng(seed)
    per = n // values
    data = np.repeat(np.arange(values, dtype=np.uint8), per)
    data = data[np.logical_or(data != rare, np.arange(data.size) % 8 != 0)]  # `rare`: 7/8 of the others' count
    return rng.permutation(data)


@pytest.mark.parametrize("case", ["shuffled", "sorted", "rare_in_runs", "rare_255", "bytes_behind_the_end_mark", "one_group", "eight_subsequences"])
def test_foreign_stream_of_8_and_9_bit_codes(env, case):
    """K6's byte classes (k_sync_table / k_fn_apply / k_sync_index on class masks instead of symbol steps): runs of 9-bit
    codes (more than a row counts: the confirming pass takes those subsequences), either value as the rare one, a stream
    with bytes behind its end mark (the first end mark ends it, as in the reference's decoder), streams too short for a
    whole group of 64 subsequences."""
    ghf, ctx, torch = env
    n = {"one_group": 5000, "eight_subsequences": 511}.get(case, (3 << 20) + 12345)
    rare = 255 if case == "rare_255" else (0 if case == "sorted" else 77)
    if n == 511:  # every value twice, `rare` once
        data = np.random.default_rng(3).permutation(np.delete(np.repeat(np.arange(256, dtype=np.uint8), 2), 2 * rare))
    else:
        data = _bytes_that_do_not_compress(n, rare, seed=n % 1000)
    if case == "sorted":
        data = np.sort(data)
    if case == "rare_in_runs":
        data = data.copy()
        data[1000:1400] = rare
        data[200000:200090] = rare
        keep = np.flatnonzero(data == rare)
        # (still the least frequent value: take as many of them away elsewhere)
        drop = keep[keep >= 300000][:480]
        data = np.delete(data, drop)
    crs = orc.compress(data)
    code, hs = ghf.parse_header(crs)
    assert (code.min_len, code.max_len) == (8, 9)
    stream = crs
    if case == "bytes_behind_the_end_mark":
        stream = np.concatenate([crs, np.random.default_rng(5).integers(0, 256, 100000, dtype=np.uint8)])
    d_stream = to_dev(torch, np.concatenate([stream, np.zeros(64, np.uint8)]))
    out, nout = ctx.decode(d_stream, stream.size, ctx.code_to_device(code), None, cap=data.size + 4096)
    ctx.sync()
    assert int(nout.item()) == data.size
    assert np.array_equal(out[: data.size].cpu().numpy(), data)


@pytest.mark.parametrize("case", ["shuffled", "sorted", "rare_in_runs", "rare_15", "bytes_behind_the_end_mark", "one_group", "tiny"])
def test_foreign_stream_of_4_and_5_bit_codes(env, case):
    """the same classes one size down (16 equally likely values + the end mark: 15 codes of 4 bits, two of 5): a chain
    alternates between two residue classes mod 8; two segment starts can fall into one 512-bit subsequence."""
    ghf, ctx, torch = env
    n = {"one_group": 9000, "tiny": 31}.get(case, (3 << 20) + 777)
    rare = 15 if case == "rare_15" else (0 if case == "sorted" else 9)
    if n == 31:  # every value twice, `rare` once
        data = np.random.default_rng(3).permutation(np.delete(np.repeat(np.arange(16, dtype=np.uint8), 2), 2 * rare))
    else:
        data = _bytes_that_do_not_compress(n, rare, seed=n % 1000, values=16)
    if case == "sorted":
        data = np.sort(data)
    if case == "rare_in_runs":
        data = data.copy()
        data[1000:1700] = rare
        data[200000:200130] = rare
        keep = np.flatnonzero(data == rare)
        data = np.delete(data, keep[keep >= 300000][:820])
    crs = orc.compress(data)
    code, hs = ghf.parse_header(crs)
    assert (code.min_len, code.max_len) == (4, 5)
    stream = crs
    if case == "bytes_behind_the_end_mark":
        stream = np.concatenate([crs, np.random.default_rng(5).integers(0, 256, 100000, dtype=np.uint8)])
    d_stream = to_dev(torch, np.concatenate([stream, np.zeros(64, np.uint8)]))
    out, nout = ctx.decode(d_stream, stream.size, ctx.code_to_device(code), None, cap=data.size + 4096)
    ctx.sync()
    assert int(nout.item()) == data.size
    assert np.array_equal(out[: data.size].cpu().numpy(), data)


@pytest.mark.parametrize("case", ["shuffled", "sorted", "rare_in_runs", "nine_in_one_segment"])
def test_codes_of_8_and_9_bits_with_runs_of_the_rare_value(env, case):
    """bytes that do not compress (255 codes of 8 bits, two of 9) with the 9-bit byte value alone, in runs, eight and nine
    times in one 64-symbol segment: K5's bytes equal the oracle's, K7 reads them back (written for the byte-phase decoder
    experiment of round 3, profiles/r03/experiments/k7_byte_phases_decoder.hip.txt; kept for the generic decoder)."""
    ghf, ctx, torch = env
    n = (2 << 20) + 4321
    data = _bytes_that_do_not_compress(n, 200, seed=9)
    if case == "sorted":
        data = np.sort(data)
    elif case == "rare_in_runs":
        data = data.copy()
        idx = np.flatnonzero(data == 200)
        data[5000:5300] = 200
        data[700000:700008] = 200   # exactly eight in one segment: still the fast way
        data[900000:900009] = 200   # nine
        data = np.delete(data, idx[idx > 1000000][:320])
    elif case == "nine_in_one_segment":
        data = data.copy()
        idx = np.flatnonzero(data == 200)
        data[64 * 1000: 64 * 1000 + 9] = 200
        data = np.delete(data, idx[idx > 1000000][:9])
    ref = orc.compress(data)
    code, _ = ghf.parse_header(ref)
    assert (code.min_len, code.max_len) == (8, 9)
    d_in, d_out, nb, d_code, idx = run_compress(ghf, ctx, torch, data)
    assert nb == ref.size and np.array_equal(d_out[:nb].cpu().numpy(), ref)
    back, nout = ctx.decode(d_out, nb, d_code, idx)
    ctx.sync()
    assert int(nout.item()) == data.size and np.array_equal(back[: data.size].cpu().numpy(), data)
    # a flipped bit inside a segment is caught by the segment-end check (or decodes to the wrong bytes and is caught there)
    bad = d_out.clone()
    bad[4000] ^= 0x10
    ctx.decode(bad, nb, d_code, idx)
    try:
        ctx.sync()  # (a flip that maps an 8-bit code onto an 8-bit code keeps every length: nothing to detect)
    except ghf.GhfError as e:
        assert e.status == 7
    ctx.index_free(idx)


def test_foreign_stream_truncated_is_reported(env):
    ghf, ctx, torch = env
    data = dg.zipf_bytes(100000, seed=4)
    crs = orc.compress(data)
    code, hs = ghf.parse_header(crs)
    cut = crs[: crs.size - 2000].copy()
    with pytest.raises(ghf.GhfError):
        ctx.decode(to_dev(torch, cut), cut.size, ctx.code_to_device(code), None, cap=data.size + 64)
        ctx.sync()


def test_decode_detects_corruption(env):
    ghf, ctx, torch = env
    data = dg.zipf_bytes(500000, seed=17)
    _, d_out, nb, d_code, idx = run_compress(ghf, ctx, torch, data)
    bad = d_out.clone()
    bad[2000:2064] ^= 0x5A
    ctx.decode(bad, nb, d_code, idx)
    with pytest.raises(ghf.GhfError) as e:
        ctx.sync()
    assert e.value.status == 7
    ctx.index_free(idx)


@pytest.mark.parametrize("kind", ["uniform", "zipf", "sym16"])
def test_full_size_256MiB_properties(env, kind):
    """BASELINE config 2 size: properties that do not need the (slow) CPU oracle on the whole buffer:
    exact histogram, tables == oracle(tables from that histogram), size == sum(freq*len), exact round trip,
    1-padding, and an oracle-exact check of the first and last 1 MiB worth of the body."""
    ghf, ctx, torch = env
    n = 1 << 28
    data = dg.make(kind, n, seed=2024)
    d_in = to_dev(torch, data)
    idx = ctx.index_alloc(n)
    d_out, nbytes, d_code = ctx.compress(d_in, index=idx)
    ctx.sync()
    nb = int(nbytes.item())
    hist = np.bincount(data, minlength=256).astype(np.int64)
    hist = np.concatenate([hist, [1]])
    assert ctx.histogram(d_in).cpu().numpy().tolist() == hist.tolist()
    ocode = orc.build_code(hist)
    gcode = ctx.code_to_host(d_code)
    assert gcode.as_dict() == ocode.as_dict()
    hs = 1040 + 8 * ocode.max_len
    bits = orc.body_bits(hist, ocode)
    assert nb == hs + (bits + 7) // 8
    head = d_out[: hs + (5 << 20)].cpu().numpy()
    assert np.array_equal(head[:hs], orc.header_bytes(ocode))
    # the first 1 Mi symbols of the body == the oracle's packer run with the same (global) code
    import ctypes as C
    cap = 5 << 20
    buf = np.zeros(cap, dtype=np.uint8)
    m = 1 << 20
    orc.lib().orc_encode_body(data.ctypes.data, m, C.byref(ocode), buf.ctypes.data, cap)
    lens = np.array(list(ocode.length), dtype=np.int64)
    pbits = int(lens[data[:m]].sum())
    assert np.array_equal(head[hs : hs + pbits // 8], buf[: pbits // 8])
    # round trip
    back, _ = ctx.decode(d_out, nb, d_code, idx)
    ctx.sync()
    assert bool((back[:n] == d_in).all().item())
    # padding bits are ones (buffer.h:277-280)
    pad = (8 - bits % 8) % 8
    last = int(d_out[nb - 1].item())
    assert last & ((1 << pad) - 1) == (1 << pad) - 1
    ctx.index_free(idx)


def test_random_sweep_sizes_and_alphabets(env):
    """120 seeded cases across awkward sizes (around our 1 KiB tiles / 4 KiB chunks / 64-symbol segments) and
    alphabets with many exact frequency ties: GPU .crs2 == oracle byte for byte; decode with side-car and, for
    every third case, without."""
    ghf, ctx, torch = env
    sizes = [1, 2, 3, 15, 16, 17, 63, 64, 65, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8192, 8193, 12289, 65535, 65536,
             65537, 131071, 262145]
    case = 0
    for n in sizes:
        for kind, mod in (("uniform", 0), ("zipf", 0), ("text", 0), ("uniform", 2), ("uniform", 3)):
            case += 1
            data = dg.make(kind, n, seed=5000 + case)
            if mod:
                data = (data % np.uint8(mod)).astype(np.uint8)
            d_in = to_dev(torch, data)
            idx = ctx.index_alloc(n)
            d_out, nbytes, d_code = ctx.compress(d_in, index=idx)
            ctx.sync()
            nb = int(nbytes.item())
            ref = orc.compress(data)
            got = d_out[:nb].cpu().numpy()
            assert nb == ref.size and np.array_equal(got, ref), (kind, mod, n, first_diff(got, ref))
            back, _ = ctx.decode(d_out, nb, d_code, idx)
            ctx.sync()
            assert np.array_equal(back[:n].cpu().numpy(), data), (kind, mod, n)
            ctx.index_free(idx)
            if case % 3 == 0:
                out2, n2 = ctx.decode(d_out, nb, d_code, None, cap=n + 64)
                ctx.sync()
                assert int(n2.item()) == n and np.array_equal(out2[:n].cpu().numpy(), data), (kind, mod, n, "foreign")


def test_small_alphabets_two_symbols_per_lookup(env):
    """max_len <= 6 switches K7 to its pair table (two symbols per lookup).  Alphabets of 1..48 symbols, flat and
    skewed, sizes that end exactly on / just off a 4096-symbol group, decode with and without the side-car, and a
    flipped body bit must still be reported."""
    ghf, ctx, torch = env
    rng = np.random.default_rng(77)
    seen_pair = 0
    for case, (k, n) in enumerate([(1, 8192), (2, 4096), (2, 70000), (3, 12288), (5, 65536 + 17), (8, 1 << 20),
                                   (16, (1 << 20) + 4095), (16, 1 << 22), (24, 300000), (31, 262144), (48, 1 << 19)]):
        w = rng.random(k) ** (1 + case % 3)
        data = rng.choice(k, size=n, p=w / w.sum()).astype(np.uint8)
        data = (data * np.uint8(7) + np.uint8(case)).astype(np.uint8)  # not just the low byte values
        d_in = to_dev(torch, data)
        idx = ctx.index_alloc(n)
        d_out, nbytes, d_code = ctx.compress(d_in, index=idx)
        ctx.sync()
        nb = int(nbytes.item())
        ref = orc.compress(data)
        assert nb == ref.size and np.array_equal(d_out[:nb].cpu().numpy(), ref), (k, n)
        seen_pair += ctx.code_to_host(d_code).max_len <= 6
        back, nout = ctx.decode(d_out, nb, d_code, idx)
        ctx.sync()
        assert int(nout.item()) == n and np.array_equal(back[:n].cpu().numpy(), data), (k, n)
        out2, n2 = ctx.decode(d_out, nb, d_code, None, cap=n + 64)
        ctx.sync()
        assert int(n2.item()) == n and np.array_equal(out2[:n].cpu().numpy(), data), (k, n, "foreign")
        if n >= 65536 and k >= 3:
            bad = d_out.clone()
            bad[nb // 2] ^= 0x10
            got, _ = ctx.decode(bad, nb, d_code, idx)
            try:
                ctx.sync()
                flagged = False
            except ghf.GhfError as e:
                flagged = e.status == 7
            # a flipped bit either breaks a segment boundary check (reported) or stays inside one segment with the
            # same total length (possible only for equal-length codes): then exactly that segment differs
            if not flagged:
                diff = np.nonzero(got[:n].cpu().numpy() != data)[0]
                assert diff.size and diff.max() - diff.min() < 64, (k, n)
        ctx.index_free(idx)
    assert seen_pair >= 5


@pytest.mark.parametrize("kind,n", [("zipf", 1 << 32), ("sym16", 1 << 32), ("uniform", (1 << 32) + 4099)])
def test_baseline_configs_4GiB_properties(env, kind, n):
    """BASELINE configs 3 and 5 (4 GiB Zipf encode, 4 GiB 16-symbol decode) and a > 4 GiB uniform stream (64-bit
    offsets everywhere), generated on the device.  Size-independent properties only: exact histogram, tables ==
    oracle(histogram), stream size == header + ceil(sum freq*len / 8), first MiB of the body == the oracle's packer,
    exact round trip with the side-car, padding bits."""
    ghf, ctx, torch = env
    import ctypes as C
    import importlib
    pkgload.load()
    synth = importlib.import_module("golden_huffman_amd.synth")
    d_in = synth.make(torch, kind, n, device="cuda", seed=7)
    idx = ctx.index_alloc(n)
    out = ctx.empty_u8(ghf.compress_bound(n))
    d_out, nbytes, d_code = ctx.compress(d_in, d_out=out, index=idx)
    ctx.sync()
    nb = int(nbytes.item())
    hist = torch.zeros(256, dtype=torch.int64, device="cuda")
    for lo in range(0, n, 1 << 30):
        hist += torch.bincount(d_in[lo : lo + (1 << 30)].to(torch.int32), minlength=256)
    hist = np.concatenate([hist.cpu().numpy(), [1]]).astype(np.int64)
    assert ctx.histogram(d_in).cpu().numpy().tolist() == hist.tolist()
    ocode = orc.build_code(hist)
    assert ctx.code_to_host(d_code).as_dict() == ocode.as_dict()
    hs = 1040 + 8 * ocode.max_len
    bits = orc.body_bits(hist, ocode)
    assert nb == hs + (bits + 7) // 8
    m = 1 << 20
    head = d_out[: hs + 2 * m + 64].cpu().numpy()
    assert np.array_equal(head[:hs], orc.header_bytes(ocode))
    first = d_in[:m].cpu().numpy()
    buf = np.zeros(2 * m + 64, dtype=np.uint8)
    orc.lib().orc_encode_body(first.ctypes.data, m, C.byref(ocode), buf.ctypes.data, buf.size)
    pbits = int(np.array(list(ocode.length), dtype=np.int64)[first].sum())
    assert np.array_equal(head[hs : hs + pbits // 8], buf[: pbits // 8])
    back, nout = ctx.decode(d_out, nb, d_code, idx)
    ctx.sync()
    assert int(nout.item()) == n
    for lo in range(0, n, 1 << 30):
        assert bool((back[lo : lo + (1 << 30)] == d_in[lo : lo + (1 << 30)]).all().item()), lo
    pad = (8 - bits % 8) % 8
    assert int(d_out[nb - 1].item()) & ((1 << pad) - 1) == (1 << pad) - 1
    ctx.index_free(idx)
    del d_in, out, back, d_out
    torch.cuda.empty_cache()


def test_nonstationary_stream_takes_the_unstaged_path(env):
    """A run of rare symbols (12+ bit codes) inside a stream of one frequent symbol: the groups in the run span
    more compressed bytes than K7's LDS tile holds, so they are decoded by the unstaged fallback, next to
    ordinary groups; also through K6 (no side-car)."""
    ghf, ctx, torch = env
    rng = np.random.default_rng(11)
    # five frequent symbols with halving frequencies push the 200 rare ones below depth 5: 13-bit codes
    data = rng.choice(np.array([65, 66, 67, 68, 69], dtype=np.uint8), size=3 << 20, p=[16 / 31, 8 / 31, 4 / 31, 2 / 31, 1 / 31])
    run = rng.integers(100, 300, size=96 * 1024).astype(np.uint8)  # 200 rare symbols (wraps past 255: 100..255, 0..43)
    data[(1 << 20) + 777 : (1 << 20) + 777 + run.size] = run
    _, d_out, nb, d_code, idx = run_compress(ghf, ctx, torch, data)
    ref = orc.compress(data)
    assert nb == ref.size and np.array_equal(d_out[:nb].cpu().numpy(), ref)
    code = ctx.code_to_host(d_code)
    assert min(code.length[int(s)] for s in set(run.tolist())) >= 10  # 4096 of them do not fit 4608 bytes
    back, nout = ctx.decode(d_out, nb, d_code, idx)
    ctx.sync()
    assert int(nout.item()) == data.size and np.array_equal(back[: data.size].cpu().numpy(), data)
    out2, n2 = ctx.decode(d_out, nb, d_code, None, cap=data.size + 64)
    ctx.sync()
    assert int(n2.item()) == data.size and np.array_equal(out2[: data.size].cpu().numpy(), data)
    ctx.index_free(idx)


def test_decode_into_an_unaligned_output_pointer(env):
    """K7's fast path stores 16 bytes at a time; an output pointer that is not 16-byte aligned takes the byte-store
    path instead (same for K6 + K7 without the side-car)."""
    ghf, ctx, torch = env
    data = dg.zipf_bytes(300001, seed=21)
    _, d_out, nb, d_code, idx = run_compress(ghf, ctx, torch, data)
    big = torch.zeros(data.size + 64, dtype=torch.uint8, device="cuda")
    for off in (1, 7, 8):
        dst = big[off : off + data.size + 16]
        assert dst.data_ptr() % 16 == off
        back, nout = ctx.decode(d_out, nb, d_code, idx, d_out=dst)
        ctx.sync()
        assert int(nout.item()) == data.size and np.array_equal(back[: data.size].cpu().numpy(), data), off
    back, nout = ctx.decode(d_out, nb, d_code, None, d_out=big[3 : 3 + data.size + 16])
    ctx.sync()
    assert int(nout.item()) == data.size and np.array_equal(back[: data.size].cpu().numpy(), data)
    ctx.index_free(idx)


@pytest.mark.parametrize("k,n", [(3, 300001), (7, 700001), (15, 1500001)])
def test_foreign_stream_with_a_fixed_length_code(env, k, n):
    """k equally frequent symbols + the end mark = 2^m leaves of equal depth: a code that never self-synchronises
    (a decoder started off a boundary stays off it for ever).  K6 must still find every boundary -- the known start of
    the body propagates -- and quickly."""
    ghf, ctx, torch = env
    rng = np.random.default_rng(k)
    data = np.tile(np.arange(k, dtype=np.uint8), n // k + 1)[:n]
    data[: n - n % k] = rng.permutation(data[: n - n % k])
    ref = orc.compress(data)
    code, hs = ghf.parse_header(ref)
    assert code.min_len == code.max_len  # all 2^m leaves at depth m
    d = to_dev(torch, np.concatenate([ref, np.zeros(32, np.uint8)]))
    out, n2 = ctx.decode(d, ref.size, ctx.code_to_device(code), None, cap=n + 64)
    ctx.sync()
    assert int(n2.item()) == n and np.array_equal(out[:n].cpu().numpy(), data)


def test_decode_prepare_is_single_use(env):
    """ghf_decode_prepare builds the tables ahead of time; the next decode consumes them, the one after rebuilds them
    (the tables also carry the decode kernel's work counters), and rebuilding the code drops a prepared state."""
    ghf, ctx, torch = env
    data = dg.zipf_bytes(700003, seed=31)
    _, d_out, nb, d_code, idx = run_compress(ghf, ctx, torch, data)
    for _ in range(2):
        ctx.decode_prepare(d_code)
        back, nout = ctx.decode(d_out, nb, d_code, idx)
        ctx.sync()
        assert int(nout.item()) == data.size and np.array_equal(back[: data.size].cpu().numpy(), data)
        back, nout = ctx.decode(d_out, nb, d_code, idx)  # no prepare: must rebuild by itself
        ctx.sync()
        assert np.array_equal(back[: data.size].cpu().numpy(), data)
    # a prepared state does not survive new tables in the same buffer: prepare, rebuild the code from other data,
    # compress and decode that other data with it
    ctx.decode_prepare(d_code)
    other = dg.make("uniform", 300000, seed=2)
    d_other = to_dev(torch, other)
    idx2 = ctx.index_alloc(other.size)
    d_out2, nb2, _ = ctx.compress(d_other, d_code=d_code, index=idx2)
    back, nout = ctx.decode(d_out2, int(ghf.compress_bound(other.size)), d_code, idx2)
    ctx.sync()
    assert int(nout.item()) == other.size and np.array_equal(back[: other.size].cpu().numpy(), other)
    ctx.index_free(idx2)
    ctx.index_free(idx)


def test_histogram_in_pieces_and_stale_plan_cache(env):
    """ghf_histogram_add over ragged pieces of a refilled buffer == the oracle's counts of the whole, and a plan on
    that buffer afterwards prices what is in it NOW (the per-chunk counts of a piece are not kept)"""
    ghf, ctx, torch = env
    data = dg.text_bytes(1 << 20, seed=21)
    cuts = [0, 1, 65537, 65537 + 262144, 700001, data.size]
    buf = torch.empty(400000, dtype=torch.uint8, device="cuda")
    hist = torch.zeros(ghf.NSYM, dtype=torch.int64, device="cuda")
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        buf[: hi - lo].copy_(to_dev(torch, data[lo:hi]))
        ctx.histogram_add(buf, hist, n=hi - lo)
    ctx.sync()
    h = orc.histogram(data)
    assert np.array_equal(hist.cpu().numpy(), np.asarray(h, dtype=np.int64))
    d_code = ctx.build_code(hist)
    lo, hi = cuts[-2], cuts[-1]  # same pointer and length as the last counted piece, other bytes
    other = dg.uniform_bytes(hi - lo, seed=22)
    buf[: hi - lo].copy_(to_dev(torch, other))
    total = ctx.encode_plan(buf, d_code, n=hi - lo)
    ctx.sync()
    length = np.asarray(orc.build_code(h).length, dtype=np.int64)
    assert int(total.item()) == int(length[other].sum())


def test_events_order_two_contexts(env):
    """ghf_event_record / wait / sync: a second context's stream consumes what the first one produces"""
    import ctypes as C

    ghf, ctx, torch = env
    L = ghf.lib()
    ctx2 = ghf.Context(0)
    ev = C.c_void_p()
    assert L.ghf_event_create(ctx.h, C.byref(ev)) == 0
    try:
        assert L.ghf_event_sync(ev) == 0  # never recorded: complete
        data = dg.zipf_bytes(8 << 20, seed=31)
        d_in = to_dev(torch, data)
        hist = torch.zeros(ghf.NSYM, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ctx.histogram(d_in, out=hist)
        assert L.ghf_event_record(ctx.h, ev) == 0
        assert L.ghf_event_wait(ctx2.h, ev) == 0
        d_code = ctx2.build_code(hist)  # on ctx2's stream, behind the event
        ctx2.sync()
        assert L.ghf_event_sync(ev) == 0
        assert ctx2.code_to_host(d_code).as_dict() == orc.build_code(orc.histogram(data)).as_dict()
        assert L.ghf_event_record(None, ev) != 0 and L.ghf_event_wait(ctx.h, None) != 0
    finally:
        assert L.ghf_event_destroy(ev) == 0
        ctx2.close()


def test_decode_refuses_tables_that_are_not_a_complete_code(env):
    """a caller's own ghf_code (not one ghf_parse_header vetted): lengths that break Kraft's equality, a length
    outside [min_len, max_len], a first code that does not fit its length -- GHF_E_FORMAT, nothing decoded"""
    ghf, ctx, torch = env
    data = dg.zipf_bytes(100000, seed=41)
    d_in, d_out, nb, d_code, idx = run_compress(ghf, ctx, torch, data)
    good = ctx.code_to_host(d_code)

    def attempt(mutate):
        bad = ghf.Code.from_buffer_copy(bytes(good))
        mutate(bad)
        back = ctx.empty_u8(data.size + 64)
        ctx.decode(d_out, nb, ctx.code_to_device(bad), idx, d_out=back)
        with pytest.raises(ghf.GhfError) as e:
            ctx.sync()
        return e.value.status

    used = [b for b in range(256) if good.length[b]]

    def longer(c):
        c.length[used[0]] += 1  # Kraft sum < 1: some bit patterns have no code

    def shorter(c):
        c.length[used[-1]] = c.min_len  # Kraft sum > 1

    def outside(c):
        c.length[used[1]] = c.max_len + 1

    def wide_first_code(c):
        c.first_code[c.min_len] = (1 << c.min_len) + 1

    for m in (longer, shorter, outside, wide_first_code):
        assert attempt(m) == 6, m.__name__
    back, _ = ctx.decode(d_out, nb, d_code, idx)  # the untouched tables still decode
    ctx.sync()
    assert np.array_equal(back[: data.size].cpu().numpy(), data)
    ctx.index_free(idx)


@pytest.mark.parametrize("n", [(64 << 20) + 3, (160 << 20) + 1])
def test_wide_codes_many_chunks_bit_exact(env, n):
    """codes longer than 16 bits (K5's 64-bit table entries, two half-wave passes) over thousands of chunks: counts
    fall off geometrically, so the lengths run from 1 to 25+ bits.  64 MiB = 4096 chunks of 16 KiB; 160 MiB = 5121
    chunks of 32 KiB.  Bit-exact against the oracle, then the round trip with and without the side-car."""
    ghf, ctx, torch = env
    rng = np.random.default_rng(7)
    counts = []
    left = n
    for k in range(40):
        c = max(1, left // 2) if k < 39 and left > 1 else left
        counts.append(c)
        left -= c
        if left == 0:
            break
    vals = np.repeat(np.arange(len(counts), dtype=np.uint8), counts)
    # (a full permutation of 160 M elements is slow) scatter values all over the file instead: the rare, long-coded
    # symbols then sit between the frequent ones in every chunk
    data = vals.copy()
    idx = rng.integers(0, n, size=n // 16, dtype=np.int64)
    data[idx], data[idx[::-1]] = vals[idx[::-1]], vals[idx]
    ref = orc.compress(data)
    code = orc.build_code(orc.histogram(data))
    assert code.max_len > 16
    d_in, d_out, nb, d_code, ix = run_compress(ghf, ctx, torch, data)
    assert nb == ref.size
    assert sha(d_out[:nb].cpu().numpy()) == sha(ref)
    back, _ = ctx.decode(d_out, nb, d_code, ix)
    ctx.sync()
    assert bool((back[:n] == d_in).all().item())
    back2, nbytes = ctx.decode(d_out, nb, d_code, None, cap=n + 64)
    ctx.sync()
    assert int(nbytes.item()) == n and bool((back2[:n] == d_in).all().item())
    ctx.index_free(ix)


def test_runs_of_the_longest_codes_at_max_len_11(env):
    """regression (found by scratch/host_soak.py): with max_len == 11 K7 decoded three symbols per refill check, and six
    11-bit codes in a row at the right bit phase read past its 64-bit window.  Exact counts that give 240 byte values
    (and the end mark) 11-bit codes, with all the rare values in one long stretch: runs of them at every phase -- for
    both formats, with and without the side-car"""
    ghf, ctx, torch = env
    rng = np.random.default_rng(11)
    c = 300
    counts = np.zeros(256, dtype=np.int64)
    counts[0:6], counts[6:8], counts[8:248], counts[248:255] = 256 * c, 128 * c, c, 2 * c  # (byte 255 does not occur)
    freq = np.repeat(np.arange(8, dtype=np.uint8), counts[:8])
    rare = np.repeat(np.arange(8, 256).astype(np.uint8), counts[8:])
    rng.shuffle(freq)
    rng.shuffle(rare)
    data = np.concatenate([freq[: freq.size // 2], rare, freq[freq.size // 2 :]])
    code = orc.build_code(orc.histogram(data))
    assert code.max_len == 11 and sum(1 for x in code.length if x == 11) >= 200
    d_in, d_out, nb, d_code, idx = run_compress(ghf, ctx, torch, data)
    assert sha(d_out[:nb].cpu().numpy()) == sha(orc.compress(data))
    back, _ = ctx.decode(d_out, nb, d_code, idx)
    ctx.sync()
    assert np.array_equal(back[: data.size].cpu().numpy(), data)
    back2, n2 = ctx.decode(d_out, nb, d_code, None, cap=data.size + 64)
    ctx.sync()
    assert int(n2.item()) == data.size and np.array_equal(back2[: data.size].cpu().numpy(), data)
    ctx.index_free(idx)
    # the same bytes as a .crs (tree-order codes, the same kind of lengths), at several bit phases of the body
    for drop in (0, 1, 2, 3):
        part = data[drop:]
        d_part = to_dev(torch, part)
        idx = ctx.index_alloc(part.size)
        c_out, c_nbytes, d_tree = ctx.crs_compress(d_part, index=idx)
        ctx.sync()
        cnb = int(c_nbytes.item())
        ref = orc.crs_compress(part)
        assert cnb == ref.size and sha(c_out[:cnb].cpu().numpy()) == sha(ref)
        tree = ctx.tree_to_host(d_tree)
        assert tree.max_len == 11
        left = int(ref[tree.tree_bytes])
        back3, n3 = ctx.crs_decode(c_out, cnb + (1 if left else 0), left, d_tree, idx)
        ctx.sync()
        assert int(n3.item()) == part.size and np.array_equal(back3[: part.size].cpu().numpy(), part)
        ctx.index_free(idx)


@pytest.mark.parametrize("n", [0, 1, 15, 16, 4096 + 7, (1 << 20) + 3, (64 << 20) + 11])
@pytest.mark.parametrize("nt", [True, False])
def test_copy_probe_kernel_copies(env, n, nt):
    """ghf_copy_d2d (bench.py's bandwidth probe) is a copy: every size, ragged tails, both cache policies"""
    ghf, ctx, torch = env
    src = torch.randint(0, 256, (max(n, 1) + 32,), dtype=torch.uint8, device="cuda")
    dst = torch.full((max(n, 1) + 32,), 0xA5, dtype=torch.uint8, device="cuda")
    ctx.copy_d2d(dst, src, n, non_temporal=nt)
    ctx.sync()
    assert bool((dst[:n] == src[:n]).all().item())
    assert bool((dst[n:] == 0xA5).all().item())  # nothing behind the end is touched


@pytest.mark.parametrize("mib", [117, 130, 146, 200, 230])
def test_histogram_exact_for_every_vector_count_remainder(env, mib):
    """K1's fast path takes a chunk's 4 KiB vector rows four at a time and the last one to three separately: sizes whose
    chunks have 5, 6, 7, 9 and 10 rows per thread (remainders 1, 2, 3, 1, 2), exact against torch's count, and
    the per-chunk counts K4 prices from give the oracle's bit total"""
    ghf, ctx, torch = env
    n = (mib << 20) + 4096 * 3 + 77
    rows = ghf.chunk_symbols(n) // 4096
    assert rows == {117: 5, 130: 6, 146: 7, 200: 9, 230: 10}[mib], rows
    g = torch.Generator(device="cuda")
    g.manual_seed(mib)
    d_in = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)
    d_in[::3] &= 0x1F  # skew: lengths differ
    hist = ctx.histogram(d_in)
    ctx.sync()
    want = torch.bincount(d_in.to(torch.int64), minlength=256)
    assert bool((hist[:256] == want).all().item()) and int(hist[256].item()) == 1
    code = ctx.build_code(hist)
    total = ctx.encode_plan(d_in, code)
    ctx.sync()
    lengths = torch.tensor(list(ctx.code_to_host(code).length)[:256], dtype=torch.int64, device="cuda")
    assert int(total.item()) == int((want * lengths).sum().item())


def _big_cases():
    import json
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_big.json")
    return json.load(open(path))["cases"]


def _sha_device(torch, t, nbytes, piece=1 << 28):
    h = hashlib.sha256()
    for lo in range(0, nbytes, piece):
        h.update(t[lo : min(lo + piece, nbytes)].cpu().numpy().tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("name", sorted(_big_cases()))
def test_full_size_config_equals_the_compiled_reference(env, name):
    """The BASELINE configs at FULL size against the reference itself (tests/golden/golden_big.json: ref_glzip on the same
    seeded stream, tests/golden/make_golden_big.py): the .crs2 the GPU writes has the reference file's size, header bytes and
    SHA-256 -- every byte of up to 4.3 GB -- and decodes back to the input, with the side-car and without it.  The flow is
    compressor_func_test's (unit_tests/test.cc:48-84, 101-141) with the byte compare done through the digest."""
    import base64
    import importlib

    ghf, ctx, torch = env
    pkgload.load()
    synth = importlib.import_module("golden_huffman_amd.synth")
    rec = _big_cases()[name]
    n = rec["n"]
    d_in = synth.make(torch, rec["kind"], n, offset=rec["offset"], device="cuda")
    assert _sha_device(torch, d_in, n) == rec["input_sha256"]  # synth.make (GPU) == tests/datagen.py (what the reference read)
    idx = ctx.index_alloc(n)
    out = ctx.empty_u8(ghf.compress_bound(n))
    d_out, nbytes, d_code = ctx.compress(d_in, d_out=out, index=idx)
    ctx.sync()
    nb = int(nbytes.item())
    assert nb == rec["crs2_bytes"]
    hs = rec["header_bytes"]
    assert d_out[:hs].cpu().numpy().tobytes() == base64.b64decode(rec["header_b64"])
    assert _sha_device(torch, d_out, nb) == rec["crs2_sha256"]
    back, nout = ctx.decode(d_out, nb, d_code, idx)
    ctx.sync()
    assert int(nout.item()) == n
    for lo in range(0, n, 1 << 30):
        assert bool((back[lo : lo + (1 << 30)] == d_in[lo : lo + (1 << 30)]).all().item()), lo
    back.zero_()
    back2, nout2 = ctx.decode(d_out, nb, d_code, None, d_out=back)  # no side-car: K6 rebuilds it (what a reference-written file needs)
    ctx.sync()
    assert int(nout2.item()) == n
    for lo in range(0, n, 1 << 30):
        assert bool((back2[lo : lo + (1 << 30)] == d_in[lo : lo + (1 << 30)]).all().item()), lo
    ctx.index_free(idx)
    del d_in, out, back, back2, d_out
    torch.cuda.empty_cache()
