"""CPU: SURVEY 8(f) N4, the opt-in 32-bit length limit.  This is NOT reference behaviour (the reference is undefined
when a code would exceed 32 bits), so there are no reference fixtures for the limited tables themselves -- parity of
that definition is "unpinned" by construction.  What is checked: the definition's own invariants (complete prefix
code, optimal = never worse than any other 32-bit-limited code we can think of, untouched when nothing exceeds the
limit), and -- with the compiled reference -- that the reference's DECODER reads limited streams back."""
import ctypes as C
import os
import tempfile
from fractions import Fraction

import numpy as np
import pytest

from oracle import oracle as orc
import datagen as dg


def fib_hist(k):
    h = np.zeros(257, dtype=np.int64)
    h[:k] = dg.fib_counts(k)
    h[256] = 1
    return h


@pytest.mark.parametrize("k", [33, 34, 36, 40, 45])
def test_limited_code_is_complete_and_within_32_bits(k):
    h = fib_hist(k)
    with pytest.raises(ValueError):
        orc.build_code(h)  # the reference's limit (include/canonical_huff_encoder.h:43-44)
    c = orc.build_code_limited(h, 32)
    L = np.array(list(c.length), dtype=np.int64)
    assert c.max_len == 32 and L.max() == 32
    assert ((L > 0) == (h > 0)).all()
    assert sum(Fraction(1, 2 ** int(l)) for l in L if l) == 1
    # more frequent symbols never get longer codes
    present = np.nonzero(h)[0]
    for a in present:
        for b in present:
            if h[a] > h[b]:
                assert L[a] <= L[b]
    # canonical tables are consistent: codes of one length are consecutive from first_code
    for s in present:
        l = int(L[s])
        assert c.first_code[l] <= c.codeword[s] < (1 << l)


def test_limit_changes_nothing_when_the_reference_code_fits():
    rng = np.random.default_rng(5)
    for case in range(50):
        k = int(rng.integers(1, 257))
        h = np.zeros(257, dtype=np.int64)
        h[rng.choice(256, k, replace=False)] = (rng.random(k) ** int(rng.integers(1, 10)) * 1e6).astype(np.int64) + 1
        h[256] = 1
        assert orc.build_code_limited(h, 32).as_dict() == orc.build_code(h).as_dict()


def test_limit_is_optimal_against_brute_force_on_small_alphabets():
    """all length vectors with Kraft sum 1 and max <= limit for 5 symbols: none is cheaper"""
    import itertools

    rng = np.random.default_rng(9)
    for case in range(30):
        w = np.sort((rng.random(5) ** 6 * 1000).astype(np.int64) + 1)
        h = np.zeros(257, dtype=np.int64)
        h[:4] = w[:4]
        h[256] = w[4]
        for limit in (3, 4):
            L = np.zeros(257, dtype=np.uint32)
            L[:4] = 6
            L[256] = 6  # pretend the unconstrained code was deeper than the limit
            hh = h.copy()
            orc.lib().orc_limit_lengths.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_uint32), C.c_int]
            orc.lib().orc_limit_lengths(hh.ctypes.data_as(C.POINTER(C.c_int64)), L.ctypes.data_as(C.POINTER(C.c_uint32)), limit)
            cost = int((h * L.astype(np.int64)).sum())
            best = min(
                sum(int(x) * l for x, l in zip([h[0], h[1], h[2], h[3], h[256]], ls))
                for ls in itertools.product(range(1, limit + 1), repeat=5)
                if sum(Fraction(1, 2 ** l) for l in ls) == 1
            )
            assert cost == best, (case, limit, w.tolist(), L[:4].tolist(), int(L[256]))


@pytest.mark.ref
@pytest.mark.skipif(not orc.have_ref(), reason="compiled reference only exists where oracle/_ref was built")
def test_reference_decoder_reads_a_limited_stream():
    data = dg.counts_to_bytes(dg.fib_counts(33), seed=5)  # 14.9 MB: one bit too deep for the reference's encoder
    crs = orc.compress_limited(data, 32)
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        f = os.path.join(td, "lim.crs2")
        crs.tofile(f)
        orc.ref_run(["d", f, f + ".de"], timeout=300)
        assert np.array_equal(np.fromfile(f + ".de", dtype=np.uint8), data)
    assert np.array_equal(orc.decompress(crs, cap=data.size + 8), data)
