import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, pkgload
pkg = pkgload.load(); ghf = pkg.ghf
from golden_huffman_amd import synth
ctx = ghf.Context(0)
for kind, n in (("zipf", (5 << 30) + 12345), ("uniform", (4 << 30) + 7)):
    d_in = synth.make(torch, kind, n, device="cuda")
    idx = ctx.index_alloc(n)
    out = ctx.empty_u8(ghf.compress_bound(n))
    d_out, nbytes, d_code = ctx.compress(d_in, d_out=out, index=idx)
    ctx.sync()
    nb = int(nbytes.item())
    hist = torch.bincount(d_in[: 1 << 30].to(torch.int32), minlength=256)  # spot check of the first GiB only
    back, nout = ctx.decode(out, nb, d_code, idx)
    ctx.sync()
    ok = bool((back[:n] == d_in).all().item())
    code = ctx.code_to_host(d_code)
    print(kind, n, "->", nb, "ratio %.4f" % (nb / n), "max_len", code.max_len, "roundtrip", ok, "decoded", int(nout.item()), flush=True)
    # foreign decode of the same stream (K6) on the first case only (time)
    if kind == "zipf":
        t0 = time.time()
        nn = ctx.decoded_size(out, nb, d_code)
        back2, _ = ctx.decode(out, nb, d_code, None, d_out=back)
        ctx.sync()
        print("  foreign decode n=%d ok=%s %.1f ms" % (nn, bool((back2[:n] == d_in).all().item()), (time.time() - t0) * 1e3), flush=True)
    ctx.index_free(idx)
    del d_in, out, back
    torch.cuda.empty_cache()
