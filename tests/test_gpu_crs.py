"""GPU (-m gpu): SURVEY 8(f) N3, the `.crs` format (NormalHuffEncoder / NormalHuffDecoder) through the C ABI, against
the fixtures the compiled reference produced (tests/golden/golden_crs.json) and the oracle.  Bit-exact."""
import base64
import hashlib

import numpy as np
import pytest

import datagen as dg
import pkgload
from cases import CRS_CASES as CASES
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


def as_decoder_input(torch, crs, tree_bytes):
    """what ghf_crs_decode wants from a .crs FILE: the image, and the stored last byte appended when left_bits != 0"""
    left, last = int(crs[tree_bytes]), int(crs[tree_bytes + 1])
    img = np.concatenate([crs, np.array([last], dtype=np.uint8)]) if left else crs
    pad = np.zeros((-img.size) % 16 + 16, dtype=np.uint8)  # the library never reads past stream_bytes; keep the tail tidy
    return to_dev(torch, np.concatenate([img, pad])), img.size, left


@pytest.mark.parametrize("name", list(CASES))
def test_crs_every_stage_matches_reference_fixture(env, golden_crs, name):
    ghf, ctx, torch = env
    g = golden_crs[name]
    data = CASES[name]()
    assert sha(data) == g["input_sha256"]
    d_in = to_dev(torch, data)
    if "undefined" in g:
        ctx.crs_compress(d_in)
        with pytest.raises(ghf.GhfError) as e:
            ctx.sync()
        assert e.value.status == 9
        return
    # tree + codes
    hist = ctx.histogram(d_in)
    d_tree, d_code = ctx.crs_build_code(hist)
    ctx.sync()
    tree = ctx.tree_to_host(d_tree)
    assert tree.code_strings() == g["codes"]
    assert tree.tree_bytes == g["tree_bytes"] and bytes(tree.header[: tree.tree_bytes]) == base64.b64decode(g["tree_b64"])
    code = ctx.code_to_host(d_code)
    deep = g["max_len"] > 32  # bits 32..63 of a code then sit in symbol[] (include/ghf.h, ghf_crs_build_code)
    for s, cs in enumerate(g["codes"]):
        full = code.codeword[s] | ((code.symbol[s] << 32) if deep else 0)
        assert code.length[s] == len(cs) and (not cs or full == int(cs, 2)), s
    # whole pipeline
    idx = ctx.index_alloc(data.size)
    d_out, nbytes, d_tree2 = ctx.crs_compress(d_in, index=idx)
    ctx.sync()
    nb = int(nbytes.item())
    crs = d_out[:nb].cpu().numpy()
    assert nb == g["crs_bytes"] and sha(crs) == g["crs_sha256"], first_diff(crs, orc.crs_compress(data))
    tb = g["tree_bytes"]
    assert [int(crs[tb]), int(crs[tb + 1])] == g["prefix"]
    left = g["prefix"][0]
    # K7 with the side-car, straight from the encoder's buffer (the partial last byte is still behind the image)
    back, nout = ctx.crs_decode(d_out, nb + (1 if left else 0), left, d_tree2, idx)
    ctx.sync()
    assert int(nout.item()) == data.size and np.array_equal(back[: data.size].cpu().numpy(), data)
    ctx.index_free(idx)
    # K6 + K7 on the file image alone, tree from the host-side header parser
    htree, tb2 = ghf.crs_parse_header(crs)
    assert tb2 == tb
    d_stream, sbytes, left2 = as_decoder_input(torch, crs, tb)
    assert left2 == left
    assert ctx.crs_decoded_size(d_stream, sbytes, left, ctx.tree_to_device(htree)) == data.size
    out2, n2 = ctx.crs_decode(d_stream, sbytes, left, ctx.tree_to_device(htree), None, cap=data.size + 64)
    ctx.sync()
    assert int(n2.item()) == data.size and np.array_equal(out2[: data.size].cpu().numpy(), data)


def first_diff(a, b):
    n = min(a.size, b.size)
    d = np.nonzero(a[:n] != b[:n])[0]
    return (int(d[0]) if d.size else None, a.size, b.size)


@pytest.mark.parametrize("kind,n", [("uniform", (1 << 24) + 5), ("zipf", (1 << 24) - 3), ("sym16", 1 << 24), ("text", 3 << 20)])
def test_crs_mid_size_bit_exact_vs_oracle(env, kind, n):
    ghf, ctx, torch = env
    data = dg.make(kind, n, seed=77)
    d_in = to_dev(torch, data)
    idx = ctx.index_alloc(n)
    d_out, nbytes, d_tree = ctx.crs_compress(d_in, index=idx)
    ctx.sync()
    nb = int(nbytes.item())
    ref = orc.crs_compress(data)
    crs = d_out[:nb].cpu().numpy()
    assert nb == ref.size and np.array_equal(crs, ref), first_diff(crs, ref)
    tree = ctx.tree_to_host(d_tree)
    left = int(crs[tree.tree_bytes])
    back, _ = ctx.crs_decode(d_out, nb + (1 if left else 0), left, d_tree, idx)
    ctx.sync()
    assert np.array_equal(back[:n].cpu().numpy(), data)
    ctx.index_free(idx)
    # a file written by the reference's algorithm (the oracle), no side-car
    htree, tb = ghf.crs_parse_header(ref)
    d_stream, sbytes, left = as_decoder_input(torch, ref, tb)
    out2, n2 = ctx.crs_decode(d_stream, sbytes, left, ctx.tree_to_device(htree), None, cap=n + 64)
    ctx.sync()
    assert int(n2.item()) == n and np.array_equal(out2[:n].cpu().numpy(), data)


def test_crs_random_sweep(env):
    """sizes around tiles/chunks/segments x alphabets: GPU .crs == oracle byte for byte, both decoders"""
    ghf, ctx, torch = env
    rng = np.random.default_rng(99)
    case = 0
    for n in [2, 3, 17, 63, 64, 65, 1023, 1025, 4095, 4096, 4097, 8193, 65535, 65537, 262145]:
        for k in (2, 3, 16, 97, 256):
            case += 1
            w = rng.random(k) ** (1 + case % 4)
            data = rng.choice(k, size=n, p=w / w.sum()).astype(np.uint8)
            if np.count_nonzero(np.bincount(data, minlength=256)) < 2:
                data[0] ^= 1
            d_in = to_dev(torch, data)
            idx = ctx.index_alloc(n)
            d_out, nbytes, d_tree = ctx.crs_compress(d_in, index=idx)
            ctx.sync()
            nb = int(nbytes.item())
            ref = orc.crs_compress(data)
            got = d_out[:nb].cpu().numpy()
            assert nb == ref.size and np.array_equal(got, ref), (n, k, first_diff(got, ref))
            tb = ctx.tree_to_host(d_tree).tree_bytes
            left = int(ref[tb])
            back, _ = ctx.crs_decode(d_out, nb + (1 if left else 0), left, d_tree, idx)
            ctx.sync()
            assert np.array_equal(back[:n].cpu().numpy(), data), (n, k)
            ctx.index_free(idx)
            if case % 2 == 0:
                htree, tb2 = ghf.crs_parse_header(ref)
                d_stream, sbytes, left = as_decoder_input(torch, ref, tb2)
                out2, n2 = ctx.crs_decode(d_stream, sbytes, left, ctx.tree_to_device(htree), None, cap=n + 64)
                ctx.sync()
                assert int(n2.item()) == n and np.array_equal(out2[:n].cpu().numpy(), data), (n, k, "no side-car")


def test_crs_errors(env):
    ghf, ctx, torch = env
    with pytest.raises(ghf.GhfError) as e:
        ctx.crs_compress(to_dev(torch, np.zeros(1, np.uint8)), n=0)
    assert e.value.status == 3
    # a body that does not end on a code boundary is reported
    data = dg.zipf_bytes(300000, seed=3)
    ref = orc.crs_compress(data)
    htree, tb = ghf.crs_parse_header(ref)
    cut = ref[: ref.size - 1000]
    d_stream, sbytes, left = as_decoder_input(torch, cut, tb)
    try:
        out, n2 = ctx.crs_decode(d_stream, sbytes, left, ctx.tree_to_device(htree), None, cap=data.size + 64)
        ctx.sync()
        got = int(n2.item())
        assert got < data.size  # a truncation that happens to end on a boundary decodes to a prefix
        assert np.array_equal(out[:got].cpu().numpy(), data[:got])
    except ghf.GhfError as err:
        assert err.status == 7
    # a flipped body bit with the side-car is caught by the per-segment end check
    d_in = to_dev(torch, data)
    idx = ctx.index_alloc(data.size)
    d_out, nbytes, d_tree = ctx.crs_compress(d_in, index=idx)
    ctx.sync()
    nb = int(nbytes.item())
    bad = d_out.clone()
    bad[5000:5032] ^= 0x5A
    ctx.crs_decode(bad, nb + 1, int(ref[tb]), d_tree, idx)
    with pytest.raises(ghf.GhfError) as e:
        ctx.sync()
    assert e.value.status == 7
    ctx.index_free(idx)


@pytest.mark.parametrize("kind,n,piece", [("text", 1 << 20, 65536), ("zipf", (3 << 20) + 17, 1 << 18), ("uniform", 300001, 4096), ("sym16", 70000, 1000)])
def test_crs_body_in_pieces(env, kind, n, piece):
    """ghf_crs_sync_piece + ghf_crs_decode(index = NULL) piece by piece (what the host layer's file pipeline does with a
    .crs): every piece gets the exact first bit from the landing of the one before; the last piece carries the stored
    last byte and must land on a code boundary; the pieces' outputs concatenate to the input."""
    ghf, ctx, torch = env
    data = dg.make(kind, n, seed=5)
    crs = orc.crs_compress(data)
    htree, tb = ghf.crs_parse_header(crs)
    d_tree = ctx.tree_to_device(htree)
    left, last_byte = int(crs[tb]), int(crs[tb + 1])
    body = crs[tb + 2 :]
    out = []
    first = 0
    npieces = (body.size + piece - 1) // piece
    for k in range(npieces):
        lo, hi = k * piece, min(body.size, (k + 1) * piece)
        own = hi - lo
        buf = np.zeros(((own + 16 + 15) // 16) * 16 + 16, dtype=np.uint8)
        avail = min(body.size, hi + 16) - lo
        buf[:avail] = body[lo : lo + avail]
        end_bit = 8 * own
        if k == npieces - 1 and left:
            buf[own] = last_byte
            end_bit += 8 - left
        d_piece = to_dev(torch, buf)
        landing, nsym = ctx.crs_sync_piece(d_piece, buf.size, first, end_bit, d_tree)
        if nsym:
            dec, nb = ctx.crs_decode(d_piece, buf.size, 0, d_tree, None, cap=nsym + 64)
            ctx.sync()
            assert int(nb.item()) == nsym
            out.append(dec[:nsym].cpu().numpy())
        if k == npieces - 1:
            assert landing == 0
        first = landing
    got = np.concatenate(out) if out else np.zeros(0, dtype=np.uint8)
    assert got.size == data.size and np.array_equal(got, data)


def _comb(depth):
    """a comb of `depth` + 1 leaves (keys 0 .. depth): leaf i hangs at depth i + 1 with the code '1' * i + '0', the last one at
    `depth` with '1' * depth.  -> (header bytes in the reference's preorder form, codes as ints, lengths)"""
    hdr = []
    for i in range(depth):
        hdr += [255, 255, 0, i]  # parent, its left child = leaf i   (huff_tree.cc:174-187: (255,255) parent / (0, key) leaf)
    hdr += [0, depth]            # the last parent's right child
    codes = [(1 << (i + 1)) - 2 for i in range(depth)] + [(1 << depth) - 1]
    lens = [i + 1 for i in range(depth)] + [depth]
    return np.array(hdr, dtype=np.uint8), codes, lens


@pytest.mark.parametrize("depth", [33, 40, 64])
def test_codes_of_up_to_64_bits_by_a_hand_built_tree(env, depth):
    """k_emit_long / deposit64, the 64-bit tree walk of K7's long path, the cursors' double refill and K6's rows of stride 64
    with codes of 40 and 64 bits.  No input makes the reference build such a tree below 2^28 (2^44) bytes, so the tree is built
    by hand -- a comb -- and the few KiB of input use its deepest leaves back to back; expected bytes: the definition of the
    format (every symbol's code, most significant bit first, normal_huff_encoder.h:158-186), packed on the host."""
    ghf, ctx, torch = env
    hdr, codes, lens = _comb(depth)
    tree, tb = ghf.crs_parse_header(hdr)
    assert tb == hdr.size and tree.max_len == depth and tree.n_leaves == depth + 1
    rng = np.random.default_rng(depth)
    # mostly the two deepest leaves (runs of them), every other leaf now and then
    data = rng.choice(np.arange(depth + 1), size=20000 + depth, p=np.array([1.0] * (depth - 1) + [30.0, 30.0]) / (depth + 59.0)).astype(np.uint8)
    data[1000:1300] = depth          # three hundred 64-bit (40-bit) codes in a row
    data[5000:5064] = depth - 1
    code = ghf.Code()
    for s in range(ghf.NSYM):
        code.length[s], code.codeword[s], code.symbol[s] = 0, 0, 0xFFFFFFFF
    for s, (c, l) in enumerate(zip(codes, lens)):
        code.length[s], code.codeword[s], code.symbol[s] = l, c & 0xFFFFFFFF, c >> 32
    code.min_len, code.max_len = 1, depth
    d_code = ctx.code_to_device(code)
    d_tree = ctx.tree_to_device(tree)
    d_in = to_dev(torch, data)
    head = tb + 2
    # expected body, bit by bit
    bits = "".join(format(codes[v], "0%db" % lens[v]) for v in data.tolist())
    nbits = len(bits)
    body = np.frombuffer(int(bits + "0" * ((-nbits) % 8), 2).to_bytes((nbits + 7) // 8, "big"), dtype=np.uint8)
    left = (-nbits) % 8
    cap = ((head + body.size + 64 + 15) // 16) * 16 + 64
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    out[:tb] = to_dev(torch, hdr)
    idx = ctx.index_alloc(data.size)
    ctx.histogram(d_in)
    total = ctx.encode_plan(d_in, d_code)
    start = torch.tensor([8 * head], dtype=torch.int64, device="cuda")
    end = ctx.encode_emit(d_in, d_code, out, start_bit=start, flags=8, index=idx)  # GHF_EMIT_LONG_CODES, no end mark
    ctx.sync()
    assert int(total.item()) == nbits and int(end[0].item()) == 8 * head + nbits
    got = out[head : head + body.size].cpu().numpy()
    assert np.array_equal(got, body), first_diff(got, body)
    sbytes = head + body.size
    # K7's long path with the side-car
    back, nout = ctx.crs_decode(out, sbytes, left, d_tree, idx)
    ctx.sync()
    assert int(nout.item()) == data.size and np.array_equal(back[: data.size].cpu().numpy(), data)
    ctx.index_free(idx)
    # ... and K6 first (rows of stride 64 for depth > 32)
    assert ctx.crs_decoded_size(out, sbytes, left, d_tree) == data.size
    back2, n2 = ctx.crs_decode(out, sbytes, left, d_tree, None, cap=data.size + 64)
    ctx.sync()
    assert int(n2.item()) == data.size and np.array_equal(back2[: data.size].cpu().numpy(), data)
