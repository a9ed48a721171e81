"""CPU: the `.crs` restatement (NormalHuffEncoder/Decoder, oracle/huff_oracle.c) against the golden
vectors the compiled reference produced (tests/golden/golden_crs.json, generator tests/golden/make_golden_crs.py)."""
import base64
import hashlib
import os
import tempfile

import numpy as np
import pytest

from cases import CRS_CASES as CASES
from oracle import oracle as orc
import datagen as dg


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.mark.parametrize("name", list(CASES))
def test_crs_oracle_matches_reference_fixture(golden_crs, name):
    g = golden_crs[name]
    data = CASES[name]()
    assert data.size == g["n"] and sha(data) == g["input_sha256"], "input generator drifted"
    if "undefined" in g:
        with pytest.raises(ValueError):
            orc.crs_compress(data)
        return
    tree = orc.crs_tree(np.bincount(data, minlength=256))
    assert orc.crs_code_strings(tree) == g["codes"]
    tb = orc.crs_tree_bytes(tree)
    assert tb.size == g["tree_bytes"] and bytes(tb) == base64.b64decode(g["tree_b64"])
    crs = orc.crs_compress(data)
    assert crs.size == g["crs_bytes"] and sha(crs) == g["crs_sha256"]
    assert [int(crs[tb.size]), int(crs[tb.size + 1])] == g["prefix"]
    if "crs_b64" in g:
        assert bytes(crs) == base64.b64decode(g["crs_b64"])
    back = orc.crs_decompress(crs, cap=data.size + 8)
    assert sha(back) == g["decoded_sha256"] and np.array_equal(back, data)


def test_crs_worked_example():
    # aaaabbc: c(1)+b(2) -> 3, then a(4) is popped first (3 > ... no: 3 < 4), so the parent is (3-node, a):
    # a = "1", c = "00", b = "01"; 4*1 + 2*2 + 1*2 = 10 bits -> one whole byte + 2 bits, left_bits = 6
    data = np.frombuffer(b"aaaabbc", dtype=np.uint8)
    crs = orc.crs_compress(data)
    codes = orc.crs_code_strings(orc.crs_tree(np.bincount(data, minlength=256)))
    assert (codes[ord("a")], codes[ord("b")], codes[ord("c")]) == ("1", "01", "00")
    assert bytes(crs[:10]) == bytes([255, 255, 255, 255, 0, ord("c"), 0, ord("b"), 0, ord("a")])
    assert bytes(crs[10:]) == bytes([6, 0b00000000, 0b11110101])  # 1111 01 01 | 00 + six zero bits


def test_crs_oracle_rejects_what_the_reference_leaves_undefined():
    with pytest.raises(ValueError):
        orc.crs_compress(np.zeros(0, dtype=np.uint8))
    with pytest.raises(ValueError):
        orc.crs_compress(np.full(100, 7, dtype=np.uint8))
    good = orc.crs_compress(np.frombuffer(b"hello world", dtype=np.uint8))
    with pytest.raises(ValueError):
        orc.crs_decompress(good[:5])  # truncated tree
    bad = good.copy()
    bad[0] = 0  # root claims to be a leaf
    with pytest.raises(ValueError):
        orc.crs_decompress(bad)


@pytest.mark.ref
@pytest.mark.skipif(not orc.have_ref(), reason="compiled reference only exists in the build container")
def test_crs_random_inputs_against_the_compiled_reference():
    rng = np.random.default_rng(4242)
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for case in range(24):
            n = int(rng.integers(2, 50000))
            k = int(rng.integers(2, 257))
            w = rng.random(k) ** int(rng.integers(1, 6))
            data = rng.choice(k, size=n, p=w / w.sum()).astype(np.uint8)
            if np.count_nonzero(np.bincount(data, minlength=256)) < 2:
                continue
            fin, fcrs, fde = (os.path.join(td, "%d.%s" % (case, e)) for e in ("bin", "crs", "de"))
            data.tofile(fin)
            orc.ref_run(["nc", fin, fcrs])
            ref = np.fromfile(fcrs, dtype=np.uint8)
            mine = orc.crs_compress(data)
            assert np.array_equal(ref, mine), (case, n, k)
            mine.tofile(fcrs)
            orc.ref_run(["nd", fcrs, fde])  # the reference reads what the oracle wrote
            assert np.array_equal(np.fromfile(fde, dtype=np.uint8), data)
            assert np.array_equal(orc.crs_decompress(ref), data)
