# GPU box: randomised soak of the whole C ABI against the oracle (both formats, both decoders).  usage: soak.py SECONDS [seed]
import sys, time, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, pkgload
from oracle import oracle as orc
pkg = pkgload.load(); ghf = pkg.ghf
ctx = ghf.Context(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
t0 = time.time(); cases = 0; maxlen_seen = 0
def gen():
    big = os.environ.get("SOAK_BIG") == "1"
    mode = rng.integers(2, 4) if big and rng.random() < 0.6 else rng.integers(0, 6)
    n = int(2 ** (rng.uniform(20, 25.3) if big else rng.uniform(0, 21))) + int(rng.integers(0, 70))
    k = int(rng.integers(1, 257))
    if mode == 0:   # flat
        w = np.ones(k)
    elif mode == 1: # power-law
        w = rng.random(k) ** int(rng.integers(1, 12))
    elif mode == 2: # geometric -> long codes
        w = 0.5 ** np.arange(k) * rng.uniform(0.8, 1.2, k)
    elif mode == 3: # fibonacci-like
        w = 1.618 ** (-np.arange(k, dtype=np.float64))
    elif mode == 4: # two levels
        w = np.where(rng.random(k) < 0.1, 100.0, 1.0)
    else:           # zipf
        w = (np.arange(k) + 1.0) ** -rng.uniform(0.5, 2.5)
    w = np.maximum(w, 1e-300); w = w / w.sum()
    syms = rng.permutation(256)[:k].astype(np.uint8)
    data = syms[rng.choice(k, size=n, p=w)]
    if rng.random() < 0.3 and n > 4096:  # nonstationary: a run of one symbol
        a = int(rng.integers(0, n - 1)); b = min(n, a + int(rng.integers(1, 100000)))
        data[a:b] = syms[0]
    if rng.random() < 0.3 and n > 4096:  # ... or a stretch of the rare symbols only (runs of the longest codes, spans beyond K7's tile)
        a = int(rng.integers(0, n - 1)); b = min(n, a + int(rng.integers(1, 300000)))
        rare = syms[np.argsort(w)[: max(1, k // 2)]]
        data[a:b] = rare[rng.integers(0, rare.size, b - a)]
    return data
while time.time() - t0 < budget:
    data = gen(); n = data.size; cases += 1
    d_in = torch.from_numpy(data).cuda()
    try:
        ref = orc.compress(data)
    except ValueError:
        ref = None  # > 32 bits: must be refused by default, limited with the flag
    idx = ctx.index_alloc(n)
    if ref is None:
        ctx.compress(d_in, index=idx)
        try:
            ctx.sync(); raise SystemExit("case %d: expected GHF_E_CODELEN" % cases)
        except ghf.GhfError as e:
            assert e.status == 4
        ref = orc.compress_limited(data, 32)
        d_out, nbytes, d_code = ctx.compress(d_in, index=idx, code_flags=ghf.CODE_LIMIT)
    else:
        d_out, nbytes, d_code = ctx.compress(d_in, index=idx)
    ctx.sync()
    nb = int(nbytes.item())
    got = d_out[:nb].cpu().numpy()
    assert nb == ref.size and np.array_equal(got, ref), ("crs2 mismatch", cases, n)
    maxlen_seen = max(maxlen_seen, ctx.code_to_host(d_code).max_len)
    back, nout = ctx.decode(d_out, nb, d_code, idx); ctx.sync()
    assert int(nout.item()) == n and np.array_equal(back[:n].cpu().numpy(), data), ("decode", cases, n)
    if cases % 2 == 0:
        code, hs = ghf.parse_header(got)
        out2, n2 = ctx.decode(d_out, nb, ctx.code_to_device(code), None, cap=n + 64); ctx.sync()
        assert int(n2.item()) == n and np.array_equal(out2[:n].cpu().numpy(), data), ("foreign decode", cases, n)
    ctx.index_free(idx)
    # .crs
    if np.count_nonzero(np.bincount(data, minlength=256)) >= 2:
        try:
            cref = orc.crs_compress(data)
        except ValueError:
            cref = None
        t = orc.crs_tree(np.bincount(data, minlength=256))
        if max(len(c) for c in orc.crs_code_strings(t)) <= 32:
            idx = ctx.index_alloc(n)
            c_out, c_nb, d_tree = ctx.crs_compress(d_in, index=idx); ctx.sync()
            cnb = int(c_nb.item())
            cg = c_out[:cnb].cpu().numpy()
            assert cnb == cref.size and np.array_equal(cg, cref), ("crs mismatch", cases, n)
            tb = ctx.tree_to_host(d_tree).tree_bytes
            left = int(cref[tb])
            back, _ = ctx.crs_decode(c_out, cnb + (1 if left else 0), left, d_tree, idx); ctx.sync()
            assert np.array_equal(back[:n].cpu().numpy(), data), ("crs decode", cases, n)
            ctx.index_free(idx)
            if cases % 3 == 0:
                htree, tb2 = ghf.crs_parse_header(cref)
                img = np.concatenate([cref, np.array([cref[tb + 1]], np.uint8)]) if left else cref
                d_s = torch.from_numpy(np.concatenate([img, np.zeros((-img.size) % 16 + 16, np.uint8)])).cuda()
                out3, n3 = ctx.crs_decode(d_s, img.size, left, ctx.tree_to_device(htree), None, cap=n + 64); ctx.sync()
                assert int(n3.item()) == n and np.array_equal(out3[:n].cpu().numpy(), data), ("crs foreign", cases, n)
    if cases % 50 == 0:
        print("cases", cases, "max_len seen", maxlen_seen, "%.0f s" % (time.time() - t0), flush=True)
print("soak ok:", cases, "cases, max_len seen", maxlen_seen)
