"""GPU (-m gpu): SURVEY 8(f) N4, the opt-in GHF_CODE_LIMIT.  Default behaviour stays the reference's (GHF_E_CODELEN
above 32 bits); with the flag the tables equal the oracle's package-merge definition, the stream equals the oracle's,
round-trips through K7 and K6, and -- where the compiled reference travels along -- its decoder reads it."""
import os
import tempfile

import numpy as np
import pytest

import datagen as dg
import pkgload
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    pkg = pkgload.load()
    ctx = pkg.ghf.Context(0)
    yield pkg.ghf, ctx, torch
    ctx.close()


def fib_hist(k):
    h = np.zeros(257, dtype=np.int64)
    h[:k] = dg.fib_counts(k)
    h[256] = 1
    return h


@pytest.mark.parametrize("k", [33, 34, 36, 40, 45])
def test_tables_from_a_histogram(env, k):
    ghf, ctx, torch = env
    h = fib_hist(k)
    d_hist = torch.from_numpy(h).cuda()
    ctx.build_code(d_hist)
    with pytest.raises(ghf.GhfError) as e:
        ctx.sync()
    assert e.value.status == 4  # the reference's limit, unchanged by default
    d_code = ctx.build_code(d_hist, flags=ghf.CODE_LIMIT)
    ctx.sync()
    assert ctx.code_to_host(d_code).as_dict() == orc.build_code_limited(h, 32).as_dict()


def test_flag_is_a_no_op_when_the_code_fits(env):
    ghf, ctx, torch = env
    for kind in ("zipf", "uniform", "text"):
        data = dg.make(kind, 200003, seed=8)
        d_in = torch.from_numpy(data).cuda()
        a, na, _ = ctx.compress(d_in)
        b, nb, _ = ctx.compress(d_in, code_flags=ghf.CODE_LIMIT)
        ctx.sync()
        assert int(na.item()) == int(nb.item()) and bool((a[: int(na.item())] == b[: int(nb.item())]).all().item())
    h = fib_hist(32)  # exactly 32 bits: still the reference's own code
    d_code = ctx.build_code(torch.from_numpy(h).cuda(), flags=ghf.CODE_LIMIT)
    ctx.sync()
    assert ctx.code_to_host(d_code).as_dict() == orc.build_code(h).as_dict()


@pytest.mark.parametrize("k", [33, 34])
def test_whole_stream_beyond_the_reference_limit(env, k):
    ghf, ctx, torch = env
    data = dg.counts_to_bytes(dg.fib_counts(k), seed=5)  # 14.9 MB / 24.2 MB
    d_in = torch.from_numpy(data).cuda()
    ctx.compress(d_in)
    with pytest.raises(ghf.GhfError) as e:
        ctx.sync()
    assert e.value.status == 4
    idx = ctx.index_alloc(data.size)
    d_out, nbytes, d_code = ctx.compress(d_in, index=idx, code_flags=ghf.CODE_LIMIT)
    ctx.sync()
    nb = int(nbytes.item())
    crs = d_out[:nb].cpu().numpy()
    ref = orc.compress_limited(data, 32)
    assert nb == ref.size and np.array_equal(crs, ref)
    back, nout = ctx.decode(d_out, nb, d_code, idx)
    ctx.sync()
    assert int(nout.item()) == data.size and bool((back[: data.size] == d_in).all().item())
    ctx.index_free(idx)
    # no side-car: header parsed on the host, K6 + K7
    code, hs = ghf.parse_header(crs)
    assert code.max_len == 32
    out2, n2 = ctx.decode(d_out, nb, ctx.code_to_device(code), None, cap=data.size + 64)
    ctx.sync()
    assert int(n2.item()) == data.size and bool((out2[: data.size] == d_in).all().item())
    if k == 33 and orc.have_ref():  # the reference's own bit-serial decoder reads the limited stream
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            f = os.path.join(td, "lim.crs2")
            crs.tofile(f)
            orc.ref_run(["d", f, f + ".de"], timeout=300)
            assert np.array_equal(np.fromfile(f + ".de", dtype=np.uint8), data)


def test_empty_input_opt_in(env):
    """the other half of N4: n == 0 stays refused by default (undefined in the reference); with GHF_EMPTY_OK it is the
    1049 bytes of the oracle's restatement of our definition (PARITY UNPINNED: no reference output exists), which
    K6 + K7 decode back to nothing.  The flag changes nothing for inputs that are not empty."""
    ghf, ctx, torch = env
    nothing = ctx.empty_u8(16)
    with pytest.raises(ghf.GhfError) as e:
        ctx.compress(nothing, n=0)
    assert e.value.status == 3
    d_out, nbytes, d_code = ctx.compress(nothing, n=0, code_flags=ghf.EMPTY_OK)
    ctx.sync()
    ref = orc.compress_empty()
    nb = int(nbytes.item())
    assert nb == ref.size == 1049 and np.array_equal(d_out[:nb].cpu().numpy(), ref)
    code = ctx.code_to_host(d_code)
    assert code.min_len == 1 and code.max_len == 1 and code.length[256] == 1 and sum(code.length) == 1
    # no side-car: header parsed on the host, K6 finds the end mark first, nothing for K7 to do
    hcode, hs = ghf.parse_header(d_out[:nb].cpu().numpy())
    back, n_out = ctx.decode(d_out, nb, ctx.code_to_device(hcode), None, cap=64)
    ctx.sync()
    assert int(n_out.item()) == 0
    # the staged calls keep refusing, the histogram alone is fine (only the end mark counts)
    h = ctx.histogram(nothing, n=0)
    ctx.build_code(h)
    with pytest.raises(ghf.GhfError) as e:
        ctx.sync()
    assert e.value.status == 3
    data = dg.zipf_bytes(50000, seed=12)
    d_in = torch.from_numpy(data).cuda()
    a, na, _ = ctx.compress(d_in, code_flags=ghf.EMPTY_OK)
    ctx.sync()
    assert np.array_equal(a[: int(na.item())].cpu().numpy(), orc.compress(data))
