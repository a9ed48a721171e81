"""GPU (-m gpu): the C++ host layer (golden-huffman_amd/host/glzip_hip.h), i.e. the reference's
Compressor<Encoder>/Decompressor<Decoder> API on the HIP policies, driven exactly like the reference's
unit_tests/test.cc drives its own classes (BASELINE config 1: 1 MiB enwik-style ASCII, file to file)."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from cases import CASES, CRS_CASES
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


def _env(**kw):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in kw.items()})
    return e


def _shm_dir(tmp_path):
    """multi-GiB files live in /dev/shm (what SURVEY 8f N1's file-to-file rate is quoted on) when it is there"""
    import tempfile

    if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK):
        return tempfile.TemporaryDirectory(dir="/dev/shm", prefix="ghf_")
    return tempfile.TemporaryDirectory(dir=str(tmp_path))


@pytest.mark.parametrize("resident,sink", [(1 << 40, "mmap"), (0, "pwrite"), (0, "mmap")], ids=["resident-mmap", "reread-pwrite", "reread-mmap"])
@pytest.mark.parametrize("name", ["text_1m", "zipf_64k", "fib32_maxlen32", "uniform_65537", "single_x", "aaaabbc"])
def test_pipeline_small_pieces_many_edges(tool, tmp_path, golden, name, resident, sink):
    """64 KiB pieces: every piece boundary falls inside a 16-byte unit of the stream that two pieces share; with
    GHF_RESIDENT_BYTES=0 the encoder reads the file a second time through its ring instead of keeping it in HBM;
    GHF_SINK picks how the output file is filled (shared mapping over fallocate-d pages, or pwrite).  The .crs2 must
    still be the reference's bytes and decode piece by piece"""
    data = CASES[name]()
    f = tmp_path / (name + ".bin")
    data.tofile(f)
    env = _env(GHF_PIECE_BYTES=65536, GHF_RESIDENT_BYTES=resident, GHF_IO_THREADS=3, GHF_SINK=sink)
    assert subprocess.run([tool, str(f), "3"], timeout=120, env=env).returncode == 0
    crs = np.fromfile(str(f) + ".crs2", dtype=np.uint8)
    assert crs.size == golden[name]["crs2_bytes"] and sha(crs) == golden[name]["crs2_sha256"]
    assert subprocess.run([tool, str(f) + ".crs2", "4"], timeout=120, env=env).returncode == 0
    assert np.array_equal(np.fromfile(str(f) + ".crs2.de", dtype=np.uint8), data)


def test_pipeline_estimates_that_miss(tool, tmp_path):
    """both policies size the output file from their first piece before they know better: a file that starts
    incompressible and goes on as zeros makes the encoder's guess 6x too high (the file is cut back) and the decoder's
    too low (its HBM-resident output overflows: what it holds is swept out and the rest streams through the ring)"""
    import datagen as dg

    data = np.concatenate([dg.uniform_bytes(256 << 10, seed=3), np.zeros(16 << 20, dtype=np.uint8), dg.uniform_bytes(64 << 10, seed=4)])
    f = tmp_path / "skew.bin"
    data.tofile(f)
    env = _env(GHF_PIECE_BYTES=65536, GHF_SINK="mmap")
    assert subprocess.run([tool, str(f), "3"], timeout=120, env=env).returncode == 0
    crs = np.fromfile(str(f) + ".crs2", dtype=np.uint8)
    ref = orc.compress(data)
    assert crs.size == ref.size and sha(crs) == sha(ref)
    assert subprocess.run([tool, str(f) + ".crs2", "5"], timeout=120, env=env).returncode == 0
    assert np.array_equal(np.fromfile(str(f) + ".crs2.de", dtype=np.uint8), data)
    # and the other way round: zeros first, noise behind (encoder guesses low, decoder high)
    data = data[::-1].copy()
    data.tofile(f)
    assert subprocess.run([tool, str(f), "3"], timeout=120, env=env).returncode == 0
    crs = np.fromfile(str(f) + ".crs2", dtype=np.uint8)
    ref = orc.compress(data)
    assert crs.size == ref.size and sha(crs) == sha(ref)
    assert subprocess.run([tool, str(f) + ".crs2", "6"], timeout=120, env=env).returncode == 0
    assert np.array_equal(np.fromfile(str(f) + ".crs2.de", dtype=np.uint8), data)


def test_pipeline_truncated_and_trailing_bytes(tool, tmp_path):
    """a body cut before the end mark is an error; bytes behind the end mark are ignored, as in the reference
    (its decoders stop at the mark, include/canonical_huff_encoder.cc:404-411)"""
    import datagen as dg

    data = dg.zipf_bytes(3 << 20, seed=5)
    crs = orc.compress(data)
    env = _env(GHF_PIECE_BYTES=1 << 20)
    cut = tmp_path / "cut.crs2"
    crs[: crs.size - (1 << 20) - 7].tofile(cut)
    r = subprocess.run([tool, str(cut), "4"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 1 and "error 7" in r.stderr
    more = tmp_path / "more.crs2"
    np.concatenate([crs, np.full((2 << 20) + 3, 0xA5, dtype=np.uint8)]).tofile(more)
    assert subprocess.run([tool, str(more), "4"], timeout=120, env=env).returncode == 0
    assert np.array_equal(np.fromfile(str(more) + ".de", dtype=np.uint8), data)


def test_pipeline_multi_gib_file_bit_exact_and_rate(tool, tmp_path):
    """2.2 GiB in /dev/shm: 141 pieces of 16 MiB.  The .crs2 equals the oracle's byte for byte, the round trip gives
    the file back, and ghf_tool's mode 7 reports the file-to-file rates (kept in gpurun_out/ for DESIGN.md)."""
    import json

    import datagen as dg

    n = (2200 << 20) + 4321
    with _shm_dir(tmp_path) as d:
        f = os.path.join(d, "big.bin")
        base = dg.zipf_bytes(32 << 20, seed=2024)  # tiled with a different rotation per tile (the generator is slow)
        data = np.empty(n, dtype=np.uint8)
        for t, lo in enumerate(range(0, n, base.size)):
            hi = min(n, lo + base.size)
            data[lo:hi] = np.roll(base, 4099 * t)[: hi - lo]
        data.tofile(f)
        want = sha(orc.compress(data))
        r = subprocess.run([tool, f, "7"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        rep = json.loads(r.stdout.strip().splitlines()[-1])
        assert rep["round_trip_ok"] is True
        h = hashlib.sha256()
        with open(f + ".crs2", "rb") as fh:
            for blk in iter(lambda: fh.read(1 << 24), b""):
                h.update(blk)
        assert h.hexdigest() == want
        out = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out):
            with open(os.path.join(out, "file_to_file_zipf_2200MiB.json"), "w") as fh:
                json.dump(rep, fh)


def test_empty_file_opt_in_through_the_host_layer(tool, tmp_path):
    """GHF_EMPTY_OK=1 (or set_allow_empty): the empty file becomes the 1049-byte stream of include/ghf.h's definition
    (parity unpinned -- the reference is undefined there) and comes back as an empty file; the default stays a refusal"""
    f = tmp_path / "empty.bin"
    f.write_bytes(b"")
    assert subprocess.run([tool, str(f), "3"], capture_output=True, timeout=60).returncode == 1
    assert subprocess.run([tool, str(f), "3"], timeout=60, env=_env(GHF_EMPTY_OK=1)).returncode == 0
    crs = np.fromfile(str(f) + ".crs2", dtype=np.uint8)
    assert np.array_equal(crs, orc.compress_empty())
    for mode in ("4", "5", "6"):
        assert subprocess.run([tool, str(f) + ".crs2", mode], timeout=60).returncode == 0
        assert os.path.getsize(str(f) + ".crs2.de") == 0


def test_streaming_stager_multi_piece_file(tool, tmp_path):
    """a 70 MiB file = five 16 MiB pieces each way through the pipeline (reader/writer threads, pinned rings, three streams)"""
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


@pytest.mark.parametrize("name", ["aaaabbc", "ab", "all256_once", "zipf_64k", "fib32_maxlen32", "uniform_65537", "sym16_n1003", "fib34_depth33"])
def test_crs_mode_switch_compress_then_decompress(tool, tmp_path, golden_crs, name):
    """modes 1 / 2 of unit_tests/test.cc:295-301: Compressor<NormalHuffEncoder<>> / Decompressor<NormalHuffDecoder<>>
    (fib34_depth33: codes of up to 33 bits -- the long-code kernel and the 64-bit tree walk through the file pipeline)"""
    data = CRS_CASES[name]()
    f = tmp_path / (name + ".bin")
    data.tofile(f)
    assert subprocess.run([tool, str(f), "1"], timeout=120).returncode == 0
    crs = np.fromfile(str(f) + ".crs", dtype=np.uint8)
    assert crs.size == golden_crs[name]["crs_bytes"] and sha(crs) == golden_crs[name]["crs_sha256"]
    assert subprocess.run([tool, str(f) + ".crs", "2"], timeout=120).returncode == 0
    assert np.array_equal(np.fromfile(str(f) + ".crs.de", dtype=np.uint8), data)


@pytest.mark.parametrize("resident,sink", [(1 << 40, "mmap"), (0, "pwrite")], ids=["resident-mmap", "reread-pwrite"])
@pytest.mark.parametrize("name", ["text_1m", "zipf_64k", "uniform_65537", "fib32_maxlen32", "ab", "sym16_n1003"])
def test_crs_pipeline_small_pieces(tool, tmp_path, golden_crs, name, resident, sink):
    """the .crs policies on the same file pipeline: 64 KiB pieces (a code may straddle any piece boundary; the decoder's
    last piece carries the stored last byte and must land on a code boundary), the two-byte prefix patched in at the end"""
    data = CASES[name]()
    f = tmp_path / (name + ".bin")
    data.tofile(f)
    env = _env(GHF_PIECE_BYTES=65536, GHF_RESIDENT_BYTES=resident, GHF_IO_THREADS=3, GHF_SINK=sink)
    assert subprocess.run([tool, str(f), "1"], timeout=120, env=env).returncode == 0
    crs = np.fromfile(str(f) + ".crs", dtype=np.uint8)
    assert crs.size == golden_crs[name]["crs_bytes"] and sha(crs) == golden_crs[name]["crs_sha256"]
    assert subprocess.run([tool, str(f) + ".crs", "2"], timeout=120, env=env).returncode == 0
    assert np.array_equal(np.fromfile(str(f) + ".crs.de", dtype=np.uint8), data)


def test_crs_pipeline_multi_piece_file_and_damage(tool, tmp_path):
    """70 MiB through the .crs policies with the default 16 MiB pieces, bit-exact with the oracle; a body cut short or
    with its prefix changed no longer ends on a code boundary (or holds bits that are no code): an error, not garbage"""
    import datagen as dg

    data = dg.zipf_bytes((70 << 20) + 4321, seed=78)
    f = tmp_path / "big.bin"
    data.tofile(f)
    assert subprocess.run([tool, str(f), "1"], timeout=300).returncode == 0
    crs = np.fromfile(str(f) + ".crs", dtype=np.uint8)
    ref = orc.crs_compress(data)
    assert crs.size == ref.size and sha(crs) == sha(ref)
    assert subprocess.run([tool, str(f) + ".crs", "2"], timeout=300).returncode == 0
    assert np.array_equal(np.fromfile(str(f) + ".crs.de", dtype=np.uint8), data)
    cut = tmp_path / "cut.crs"
    crs[: crs.size - 100003].tofile(cut)
    r = subprocess.run([tool, str(cut), "2"], capture_output=True, text=True, timeout=300)
    if r.returncode == 0:  # (one cut in eight lands on a code boundary: then the prefix's left_bits gives it away or not)
        assert os.path.getsize(str(cut) + ".de") < data.size
    else:
        assert "error 7" in r.stderr


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


def test_sink_reuse_writes_over_an_existing_output_file(tool, tmp_path):
    """GHF_SINK=reuse: the output file is not emptied first, its pages are written over and it is cut to the new size at the
    end -- a LARGER old file must not leave a tail, a SMALLER one must grow; both directions, .crs2 equal to the oracle's"""
    import datagen as dg

    big = dg.zipf_bytes((24 << 20) + 333, seed=5)
    small = dg.uniform_bytes((9 << 20) + 17, seed=6)
    f = tmp_path / "x.bin"
    env = _env(GHF_SINK="reuse", GHF_IO_THREADS=4)
    for data in (big, small, big):
        data.tofile(f)
        assert subprocess.run([tool, str(f), "3"], timeout=300, env=env).returncode == 0
        crs = np.fromfile(str(f) + ".crs2", dtype=np.uint8)
        ref = orc.compress(data)
        assert crs.size == ref.size and np.array_equal(crs, ref)
        assert subprocess.run([tool, str(f) + ".crs2", "4"], timeout=300, env=env).returncode == 0
        back = np.fromfile(str(f) + ".crs2.de", dtype=np.uint8)
        assert back.size == data.size and np.array_equal(back, data)


def test_sink_reuse_grows_inside_the_old_files_last_page(tool, tmp_path):
    """GHF_SINK=reuse, the case between "larger" and "smaller": the new output is 1..4095 bytes longer than the old one and
    ends in the old file's LAST page -- the sink must extend the file before it maps it (stores behind the end of a file
    through a shared mapping are unspecified; tmpfs keeps them, xfs and ext4 zero them).  Decoded files (their size is the
    exact bound) and compressed ones, sizes not page aligned."""
    import datagen as dg

    env = _env(GHF_SINK="reuse", GHF_IO_THREADS=4)
    f = tmp_path / "y.bin"
    base = (5 << 20) + 100  # 100 bytes into a page
    for n in (base, base + 1, base + 700, base + 3995):  # ... every next one ends in the same last page as the one before
        data = dg.uniform_bytes(n, seed=11)
        data.tofile(f)
        assert subprocess.run([tool, str(f), "3"], timeout=300, env=env).returncode == 0
        crs = np.fromfile(str(f) + ".crs2", dtype=np.uint8)
        ref = orc.compress(data)
        assert crs.size == ref.size and np.array_equal(crs, ref), n
        assert subprocess.run([tool, str(f) + ".crs2", "4"], timeout=300, env=env).returncode == 0
        back = np.fromfile(str(f) + ".crs2.de", dtype=np.uint8)
        assert back.size == n and np.array_equal(back, data), n
