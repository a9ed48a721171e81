"""CPU: the oracle (oracle/huff_oracle.c) against the golden vectors the compiled reference produced
(tests/golden/golden.json, generator tests/golden/make_golden.py) -- every stage, bit-exact."""
import base64
import hashlib
import os
import tempfile

import numpy as np
import pytest

from cases import CASES
from oracle import oracle as orc
import datagen as dg


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_fixture(golden, name):
    g = golden[name]
    data = CASES[name]()
    assert data.size == g["n"] and sha(data) == g["input_sha256"], "input generator drifted"
    hist = orc.histogram(data)
    assert hist.tolist() == g["hist"]
    code = orc.build_code(hist)
    d = code.as_dict()
    for k in ("length", "codeword", "symbol", "first_code", "start_pos", "min_len", "max_len"):
        assert d[k] == g[k], k
    hdr = orc.header_bytes(code)
    assert hdr.size == g["header_bytes"]
    assert bytes(hdr) == base64.b64decode(g["header_b64"])
    crs = orc.compress(data)
    assert crs.size == g["crs2_bytes"]
    assert sha(crs) == g["crs2_sha256"]
    assert sha(crs[hdr.size:]) == g["body_sha256"]
    assert bytes(crs[-8:]).hex() == g["body_tail_hex"]
    assert (orc.body_bits(hist, code) + 7) // 8 == crs.size - hdr.size
    if "crs2_b64" in g:
        assert bytes(crs) == base64.b64decode(g["crs2_b64"])
    back = orc.decompress(crs, cap=data.size + 8)
    assert sha(back) == g["decoded_sha256"] and np.array_equal(back, data)


def test_worked_example_from_survey():
    # SURVEY 8 "wire format": aaaabbc -> a=1,b=2,c=3,EOF=3; codes 1,01,000,001; header 1064 B; body F5 07
    data = np.frombuffer(b"aaaabbc", dtype=np.uint8)
    crs = orc.compress(data)
    dcode, hs = orc.parse_header(crs)
    assert hs == 1064 and bytes(crs[hs:]) == b"\xF5\x07"
    assert (dcode.min_len, dcode.max_len) == (1, 3)
    assert list(dcode.start_pos)[1:4] == [0, 1, 2] and list(dcode.first_code)[1:4] == [1, 1, 0]
    code = orc.build_code(orc.histogram(data))
    assert [code.length[ord(c)] for c in "abc"] + [code.length[256]] == [1, 2, 3, 3]
    assert [code.codeword[ord(c)] for c in "abc"] + [code.codeword[256]] == [1, 1, 0, 1]


def test_oracle_rejects_what_the_reference_leaves_undefined():
    with pytest.raises(ValueError):  # empty input: SURVEY 5.2
        orc.compress(np.zeros(0, dtype=np.uint8))
    with pytest.raises(ValueError):  # 33-bit codes: include/canonical_huff_encoder.h:43-44
        h = np.zeros(257, dtype=np.int64)
        h[:33] = dg.fib_counts(33)
        h[256] = 1
        orc.build_code(h)


@pytest.mark.ref
@pytest.mark.skipif(not orc.have_ref(), reason="compiled reference only exists in the build container")
def test_oracle_vs_compiled_reference_random_sweep():
    """fresh random inputs (not in the fixtures): restatement == compiled reference, byte for byte."""
    rng_seed = 1234
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for i in range(40):
            kind = ["uniform", "zipf", "sym16", "text"][i % 4]
            n = int(dg.splitmix64(np.uint64(rng_seed + i)) % np.uint64(70000)) + 1
            data = dg.make(kind, n, seed=1000 + i)
            if i % 5 == 0:  # few distinct values -> many exact ties
                data = data % np.uint8(3 + i % 7)
            fin, fout, fde = (os.path.join(td, "x." + e) for e in ("bin", "crs2", "de"))
            data.tofile(fin)
            orc.ref_run(["c", fin, fout])
            ref = np.fromfile(fout, dtype=np.uint8)
            mine = orc.compress(data)
            assert mine.size == ref.size and np.array_equal(mine, ref), (i, kind, n)
            orc.ref_run(["d", fout, fde])
            assert np.array_equal(np.fromfile(fde, dtype=np.uint8), orc.decompress(ref, cap=n + 8))
