"""CPU (-m "not gpu"): the C-ABI library loads, exports every symbol include/ghf.h declares, its
host-only entry points behave, and it refuses to work without a GPU (no CPU fallback in the product)."""
import base64
import ctypes as C
import os
import re

import numpy as np
import pytest

import pkgload
from cases import CASES

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def ghf():
    pkg = pkgload.load()
    if not os.path.exists(pkg.ghf.LIB_PATH):
        pkg.build()
    return pkg.ghf


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "ghf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ghf_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported(ghf):
    L = ghf.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libghf.so does not export %s" % n
    assert sorted(ghf.EXPORTS) == names


def test_struct_layout_matches_header(ghf):
    assert C.sizeof(ghf.Code) == 4 * (3 * 257 + 2 * 64 + 2)
    assert C.sizeof(ghf.Index) == 56


def test_no_cpu_fallback_without_gpu(ghf):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert ghf.lib().ghf_ctx_create(0, C.byref(h)) == 2  # GHF_E_HIP
    assert not h.value


def test_bounds_and_chunking(ghf):
    for n in (1, 7, 4096, 65537, 1 << 20, 1 << 28, 1 << 32):
        b = ghf.compress_bound(n)
        assert b % 16 == 0 and b >= 1040 + 256 + (9 * (n + 1) + 7) // 8
        c = ghf.chunk_symbols(n)
        assert c % 4096 == 0 and 16384 <= c <= (1 << 20)
        assert -(-n // c) <= 6144 or c == (1 << 20)  # one resident round of K5 waves (256 CUs x 3 workgroups x 8 waves)
    assert ghf.lib().ghf_header_bytes(9) == 1112


@pytest.mark.parametrize("name", list(CASES))
def test_parse_header_against_reference_headers(ghf, golden, name):
    g = golden[name]
    hdr = np.frombuffer(base64.b64decode(g["header_b64"]), dtype=np.uint8)
    code, hs = ghf.parse_header(hdr)
    assert hs == g["header_bytes"]
    d = code.as_dict()
    for k in ("symbol", "first_code", "start_pos", "min_len", "max_len", "length", "codeword"):
        assert d[k] == g[k], k


def test_parse_header_rejects_garbage(ghf, golden):
    hdr = np.frombuffer(base64.b64decode(golden["zipf_64k"]["header_b64"]), dtype=np.uint8).copy()
    for pos, val in ((3, 0), (1035, 99), (1039, 40), (8, 7), (1047, 5)):
        bad = hdr.copy()
        bad[pos] = val
        with pytest.raises(ghf.GhfError):
            ghf.parse_header(bad)
    with pytest.raises(ghf.GhfError):
        ghf.parse_header(hdr[:500])


# ---------------------------------------------------------------- .crs host-side entry points
def test_empty_stream_definition_is_accepted_by_the_host_parser(ghf):
    """SURVEY 8(f) N4, opt-in GHF_EMPTY_OK -- PARITY UNPINNED (the reference is undefined at n == 0): the oracle's
    restatement of OUR definition (header of the lone end mark's one-bit code + 0x7F) parses, and the oracle's own
    decoder loop (the reference's, which stops at the end mark) reads it back as nothing"""
    from oracle import oracle as orc

    s = orc.compress_empty()
    assert s.size == 1049 and s[-1] == 0x7F
    code, hs = ghf.parse_header(s)
    assert hs == 1048 and code.min_len == 1 and code.max_len == 1
    assert code.length[256] == 1 and code.codeword[256] == 0 and sum(code.length) == 1
    assert orc.decompress(s).size == 0
    # one symbol that is NOT the end mark stays a format error
    bad = s.copy()
    bad[4:8] = [0, 0, 0, 65]
    with pytest.raises(ghf.GhfError):
        ghf.parse_header(bad)


def test_crs_struct_layout(ghf):
    assert C.sizeof(ghf.Tree) == 2 * 256 * 2 + 4 * 4 + 1024


@pytest.mark.parametrize("name", list(CASES))
def test_crs_parse_header_against_reference_trees(ghf, golden_crs, name):
    g = golden_crs[name]
    if "undefined" in g:
        return
    hdr = np.frombuffer(base64.b64decode(g["tree_b64"]), dtype=np.uint8)
    tree, tb = ghf.crs_parse_header(np.concatenate([hdr, np.zeros(2, np.uint8)]))
    assert tb == g["tree_bytes"] == tree.tree_bytes
    assert tree.code_strings() == g["codes"]
    assert tree.max_len == g["max_len"] and tree.n_leaves == sum(1 for c in g["codes"] if c)
    assert bytes(tree.header[:tb]) == bytes(hdr)


def test_crs_parse_header_rejects_garbage(ghf, golden_crs):
    hdr = np.frombuffer(base64.b64decode(golden_crs["zipf_64k"]["tree_b64"]), dtype=np.uint8)
    with pytest.raises(ghf.GhfError) as e:
        ghf.crs_parse_header(hdr[:-2])  # truncated: the last leaf is missing
    assert e.value.status == 6
    with pytest.raises(ghf.GhfError) as e:
        ghf.crs_parse_header(np.array([0, 65], dtype=np.uint8))  # the root is a leaf
    assert e.value.status == 6
    with pytest.raises(ghf.GhfError) as e:
        ghf.crs_parse_header(np.full(2000, 255, dtype=np.uint8))  # parents only, never closes
    assert e.value.status == 6
    # a 40-deep comb: well-formed, and within the 64 bits the long-code paths handle
    def comb(depth):
        c = []
        for d in range(depth):
            c += [255, 255, 0, d]
        return np.array(c + [0, 200], dtype=np.uint8)

    tree, tb = ghf.crs_parse_header(comb(40))
    assert tree.max_len == 40 and tree.n_leaves == 41 and tb == 2 * (2 * 41 - 1)
    # a 70-deep one: deeper than any code the kernels pack (it would take more than 2^44 input bytes)
    with pytest.raises(ghf.GhfError) as e:
        ghf.crs_parse_header(comb(70))
    assert e.value.status == 4


def _kernel_asm(name):
    """gfx950 ISA text of golden-huffman_amd/csrc/<name>.hip, built with the Makefile's own flags"""
    import shutil
    import subprocess
    import tempfile

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    src = os.path.join(ROOT, "golden-huffman_amd", "csrc", name + ".hip")
    mk = open(os.path.join(ROOT, "golden-huffman_amd", "Makefile")).read()
    flags = re.search(r"^HIPFLAGS \?= (.*)$", mk, flags=re.M).group(1)
    flags = flags.replace("$(ARCH)", "gfx950").replace("$(ROOT)", ROOT).replace("$(HERE)", os.path.join(ROOT, "golden-huffman_amd") + "/")
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        r = subprocess.run([hipcc] + flags.split() + ["--cuda-device-only", "-S", "-o", os.path.join(td, "k.s"), src],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return open(os.path.join(td, "k.s")).read()


def test_emit_main_loops_have_counted_waits_and_no_scratch():
    """K5's main loop keeps four tile loads in flight per wave; the wait in front of a tile's first use is COUNTED by the
    compiler ("all but the 3 younger loads and the stores issued since"), which it can only do because every vector-memory
    operation of the loop body is an unconditional straight-line instruction.  Guard both halves of that: each narrow-code
    instantiation has a loop with vmcnt(15) (side-car) and one with vmcnt(11) (no side-car) waits, four per trip, and no
    loop with such waits touches scratch (a spill inside would add memory operations the design did not plan for; spills
    of loop invariants OUTSIDE the loops are the compiler's own business, and it counts them itself)."""
    text = _kernel_asm("ghf_emit")
    body = text[text.index("_ZN3ghf6k_emitENS_10EmitParamsE:"):]
    body = body[: body.index(".Lfunc_end")].split("\n")
    loops = []  # (label, first line, last line) of innermost loops: header label .. the branch back to it
    for i, line in enumerate(body):
        m = re.match(r"(\.LBB\d+_\d+):.*=>\s*This Inner Loop Header", line)
        if m:
            for j in range(i + 1, len(body)):
                if re.search(r"s_c?branch\S*\s+%s\b" % re.escape(m.group(1)), body[j]):
                    loops.append((m.group(1), i, j))
                    break
    assert loops
    counted = {15: 0, 11: 0}
    for label, a, b in loops:
        blk = body[a:b + 1]
        waits = [int(w) for l in blk for w in re.findall(r"s_waitcnt.*vmcnt\((\d+)\)", l)]
        for n in (15, 11):
            if waits.count(n) >= 4:
                counted[n] += 1
                assert not any("scratch_" in l for l in blk), (label, "scratch traffic inside a main loop")
                assert set(waits) <= {n, 0}, (label, waits)  # the loop's only counted wait is the one in front of a tile
    assert counted[15] == 3 and counted[11] == 3, counted  # modes 1..3 (codes <= 9 / 12 / 16 bits)
    # ... and the shipped geometry (512 threads, 6 waves per SIMD) needs no scratch at all
    meta = re.search(r"\.name:\s+_ZN3ghf6k_emitENS_10EmitParamsE\b(.*?)\.wavefront_size", text, flags=re.S).group(1)
    assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1)) == 0
    assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1)) == 0


def test_histogram_epilogue_keeps_its_replica_loads_in_flight():
    """K1 (k_histogram).  The launch's last workgroup sums 32 replicas of the totals with agent-scope loads and zeroes them.
    Written as one loop over (load, store) pairs the compiler chained them -- a full wait behind every few loads, up to 32 L2
    round trips at the very end of a 55 us launch (8-10 us of it, profiles/r04/experiments/k1_k5_launch_edges.txt).  Now all
    32 loads are issued before the first store: between the first and the last of them no wait drains the queue (the counted
    waits the compiler puts there -- it starts adding while it still issues -- leave at least six loads outstanding)."""
    text = _kernel_asm("ghf_kernels")
    sym = "_ZN3ghf11k_histogramEPKhmjjPjPyS3_j"
    body = text[text.index(sym + ":"):]
    body = body[: body.index(".Lfunc_end")].split("\n")
    loads = [i for i, l in enumerate(body) if re.search(r"global_load_dwordx2 .*\bsc1\b", l)]
    assert len(loads) >= 32, len(loads)
    run = loads[-32:]  # the epilogue's 32 replica loads are the function's last agent-scope loads
    between = body[run[0]: run[-1] + 1]
    assert not any(re.search(r"global_store|global_atomic", l) for l in between), "a store between the replica loads"
    waits = [int(w) for l in between for w in re.findall(r"s_waitcnt.*vmcnt\((\d+)\)", l)]
    assert all(w >= 6 for w in waits), waits


def test_k6_kernels_use_no_scratch():
    """Twice now a build of one of these LDS-heavy kernels that SPILLED gave wrong streams on the GPU where its non-spilling
    twin did not (round 1: k_emit with 6 spilled VGPRs; round 3: k_sync_table with 4, the build that read both candidate
    end-mark masks of the 4/5-bit class walk up front; neither root-caused).  The K6 kernels are kept
    scratch-free: a change that makes one of them spill fails here, on the CPU, before it can fail on a GPU."""
    text = _kernel_asm("ghf_decode")
    for sym in ("_ZN3ghf12k_sync_tableENS_10SyncParamsEjPhPj", "_ZN3ghf11k_sync_passENS_10SyncParamsE", "_ZN3ghf12k_sync_indexENS_10SyncParamsEPmmm"):
        meta = re.search(r"\.name:\s+%s\b(.*?)\.wavefront_size" % re.escape(sym), text, flags=re.S).group(1)
        assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1)) == 0, sym
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1)) == 0, sym


def test_decode_kernel_uses_no_scratch_and_counts_its_waits():
    """K7 (k_decode).  Round 3's build spilled the fifth vector of its prefetch INSIDE the hot loop of the variants uniform
    bytes take: `global_load ... ; s_waitcnt vmcnt(0) ; scratch_store` -- every pass waited for the whole prefetch before it
    decoded a symbol (profiles/r04/k7_spill_r03.txt).  Now: no scratch at all, a register budget with room (a 16-wave
    workgroup at <= 96 registers shares its CU with the one-wave code build of a later step instead of waiting for it), and
    the hot pass's wait for the span prefetched a pass ago is a COUNTED one (all but the youngest six or more vector-memory
    operations: this pass's stores, the next side-car words, the ticket), never a full drain."""
    text = _kernel_asm("ghf_decode")
    sym = "_ZN3ghf8k_decodeENS_9DecParamsE"
    meta = re.search(r"\.name:\s+%s\b(.*?)\.wavefront_size" % re.escape(sym), text, flags=re.S).group(1)
    assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1)) == 0
    assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1)) == 0
    assert int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1)) <= 96
    body = text[text.index(sym + ":") :]
    body = body[: body.index(".Lfunc_end")].split("\n")
    assert not any("scratch_" in l for l in body)
    # the hot pass: from the first non-temporal span load behind a run of counted waits back to those waits
    nt_loads = [i for i, l in enumerate(body) if re.search(r"global_load_dwordx4 .* nt", l)]
    assert len(nt_loads) >= 10  # prologue + hot + cold
    counted = []
    for i, l in enumerate(body):
        w = re.search(r"s_waitcnt vmcnt\((\d+)\)", l)
        if w and int(w.group(1)) >= 6:
            counted.append((i, int(w.group(1))))
    # five spans of a pass are waited for one by one: vmcnt(10), (9), (8), (7), (6) in this order somewhere
    seq = [v for _, v in counted]
    assert any(seq[k : k + 5] == [10, 9, 8, 7, 6] for k in range(len(seq))), seq
