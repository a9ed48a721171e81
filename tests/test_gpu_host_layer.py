"""GPU (-m gpu): the C++ host layer (golden-huffman_amd/host/glzip_hip.h), i.e. the reference's
Compressor<Encoder>/Decompressor<Decoder> API on the HIP policies, driven exactly like the reference's
unit_tests/test.cc drives its own classes (BASELINE config 1: 1 MiB enwik-style ASCII, file to file)."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from cases import CASES
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
TOOL = os.path.join(ROOT, "golden-huffman_amd", "host", "bin", "ghf_tool")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module")
def tool():
    if not os.path.exists(TOOL):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "golden-huffman_amd", "host")], check=True)
    return TOOL


def test_config1_suite_like_unit_tests_test_cc(tool, tmp_path, golden, golden_crs):
    data = CASES["text_1m"]()
    f = tmp_path / "text_1m.bin"
    data.tofile(f)
    r = subprocess.run([tool, str(f)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "10 tests ran, 0 failed" in r.stdout
    crs = np.fromfile(str(f) + ".crs2", dtype=np.uint8)  # default name: <in>.crs2 (canonical_huff_encoder.cc:19-22)
    assert sha(crs) == golden["text_1m"]["crs2_sha256"]
    de = np.fromfile(str(f) + ".crs2.de", dtype=np.uint8)  # default name: <in>.de (encoder.h:229-230)
    assert np.array_equal(de, data)
    # the normal-Huffman leg of the suite: <in>.crs (normal_huff_encoder.h:84-88) and <in>.crs.de
    crs1 = np.fromfile(str(f) + ".crs", dtype=np.uint8)
    assert sha(crs1) == golden_crs["text_1m"]["crs_sha256"]
    assert np.array_equal(np.fromfile(str(f) + ".crs.de", dtype=np.uint8), data)


@pytest.mark.parametrize("name", ["aaaabbc", "single_x", "all256_once", "zipf_64k", "fib32_maxlen32", "uniform_65537"])
def test_mode_switch_compress_then_three_decoders(tool, tmp_path, golden, name):
    data = CASES[name]()
    f = tmp_path / (name + ".bin")
    data.tofile(f)
    assert subprocess.run([tool, str(f), "3"], timeout=120).returncode == 0
    crs = np.fromfile(str(f) + ".crs2", dtype=np.uint8)
    assert crs.size == golden[name]["crs2_bytes"] and sha(crs) == golden[name]["crs2_sha256"]
    for mode in ("4", "5", "6"):
        out = str(f) + ".crs2.de"
        if os.path.exists(out):
            os.remove(out)
        assert subprocess.run([tool, str(f) + ".crs2", mode], timeout=120).returncode == 0
        assert np.array_equal(np.fromfile(out, dtype=np.uint8), data)


def test_decodes_a_file_written_by_the_reference(tool, tmp_path):
    """the .crs2 here comes from the oracle, byte-identical to the compiled reference's output"""
    data = CASES["text_131073"]()
    f = tmp_path / "ref.crs2"
    orc.compress(data).tofile(f)
    assert subprocess.run([tool, str(f), "4"], timeout=120).returncode == 0
    assert np.array_equal(np.fromfile(str(f) + ".de", dtype=np.uint8), data)


def test_errors_are_reported_not_undefined(tool, tmp_path):
    empty = tmp_path / "empty.bin"
    empty.write_bytes(b"")
    assert subprocess.run([tool, str(empty), "3"], capture_output=True, timeout=60).returncode == 1
    junk = tmp_path / "junk.crs2"
    junk.write_bytes(bytes(range(256)) * 8)
    assert subprocess.run([tool, str(junk), "4"], capture_output=True, timeout=60).returncode == 1
    assert subprocess.run([tool, str(tmp_path / "missing.bin"), "3"], capture_output=True, timeout=60).returncode == 1


def test_streaming_stager_multi_piece_file(tool, tmp_path):
    """a 70 MiB file = three 32 MiB staging pieces each way (double-buffered pinned buffers, two copy streams)"""
    import datagen as dg

    data = dg.zipf_bytes((70 << 20) + 12345, seed=77)
    f = tmp_path / "big.bin"
    data.tofile(f)
    assert subprocess.run([tool, str(f), "3"], timeout=300).returncode == 0
    crs = np.fromfile(str(f) + ".crs2", dtype=np.uint8)
    ref = orc.compress(data)
    assert crs.size == ref.size and sha(crs) == sha(ref)
    assert subprocess.run([tool, str(f) + ".crs2", "6"], timeout=300).returncode == 0
    assert np.array_equal(np.fromfile(str(f) + ".crs2.de", dtype=np.uint8), data)


@pytest.mark.parametrize("name", ["aaaabbc", "ab", "all256_once", "zipf_64k", "fib32_maxlen32", "uniform_65537", "sym16_n1003"])
def test_crs_mode_switch_compress_then_decompress(tool, tmp_path, golden_crs, name):
    """modes 1 / 2 of unit_tests/test.cc:295-301: Compressor<NormalHuffEncoder<>> / Decompressor<NormalHuffDecoder<>>"""
    data = CASES[name]()
    f = tmp_path / (name + ".bin")
    data.tofile(f)
    assert subprocess.run([tool, str(f), "1"], timeout=120).returncode == 0
    crs = np.fromfile(str(f) + ".crs", dtype=np.uint8)
    assert crs.size == golden_crs[name]["crs_bytes"] and sha(crs) == golden_crs[name]["crs_sha256"]
    assert subprocess.run([tool, str(f) + ".crs", "2"], timeout=120).returncode == 0
    assert np.array_equal(np.fromfile(str(f) + ".crs.de", dtype=np.uint8), data)


def test_crs_interop_with_the_reference_algorithm(tool, tmp_path):
    """a .crs written by the reference's algorithm (the oracle; with the compiled reference itself when it is around)
    is decoded by ghf_tool, and what ghf_tool writes is decoded by them"""
    data = CASES["text_131073"]()
    f = tmp_path / "ref.crs"
    orc.crs_compress(data).tofile(f)
    assert subprocess.run([tool, str(f), "2"], timeout=120).returncode == 0
    assert np.array_equal(np.fromfile(str(f) + ".de", dtype=np.uint8), data)
    g = tmp_path / "mine.bin"
    data.tofile(g)
    assert subprocess.run([tool, str(g), "1"], timeout=120).returncode == 0
    mine = np.fromfile(str(g) + ".crs", dtype=np.uint8)
    assert np.array_equal(orc.crs_decompress(mine), data)
    if orc.have_ref():
        orc.ref_run(["nd", str(g) + ".crs", str(tmp_path / "ref.de")])
        assert np.array_equal(np.fromfile(tmp_path / "ref.de", dtype=np.uint8), data)


def test_crs_errors_are_reported_not_undefined(tool, tmp_path):
    one = tmp_path / "one.bin"
    one.write_bytes(b"x" * 1000)  # a single distinct byte: NULL dereference in the reference's decoder, status 9 here
    r = subprocess.run([tool, str(one), "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "error 9" in r.stderr
    junk = tmp_path / "junk.crs"
    junk.write_bytes(b"\xff" * 4000)
    assert subprocess.run([tool, str(junk), "2"], capture_output=True, timeout=60).returncode == 1


def test_code_limit_opt_in_through_the_host_layer(tool, tmp_path):
    """a 14.9 MB file whose reference-exact code would need 33 bits.  Default: refused like the
    reference's own limit (status 4); with GHF_CODE_LIMIT=1: an ordinary .crs2 that we, the oracle and -- where it
    travelled along -- the compiled reference's decoder read back."""
    import datagen as dg

    data = dg.counts_to_bytes(dg.fib_counts(33), seed=5)
    f = tmp_path / "fib33.bin"
    data.tofile(f)
    r = subprocess.run([tool, str(f), "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "error 4" in r.stderr
    env = dict(os.environ, GHF_CODE_LIMIT="1")
    assert subprocess.run([tool, str(f), "3"], env=env, timeout=120).returncode == 0
    crs = np.fromfile(str(f) + ".crs2", dtype=np.uint8)
    ref = orc.compress_limited(data, 32)
    assert crs.size == ref.size and sha(crs) == sha(ref)
    assert subprocess.run([tool, str(f) + ".crs2", "4"], timeout=120).returncode == 0
    assert np.array_equal(np.fromfile(str(f) + ".crs2.de", dtype=np.uint8), data)
    if orc.have_ref():
        orc.ref_run(["d", str(f) + ".crs2", str(tmp_path / "ref.de")], timeout=300)
        assert np.array_equal(np.fromfile(tmp_path / "ref.de", dtype=np.uint8), data)
