#!/usr/bin/env python3
"""side-car-less decode (ghf_decode(index = NULL): K6 + K7), host wall time per stream, beside the indexed decode of the same
stream; every result compared with the input.   python scratch/foreign_time.py [lib.so|-] [mib ...]
FOREIGN_OFFSETS=0,1,2 times several inputs of each kind (synth offsets in units of the size: different bytes, same statistics);
FOREIGN_KINDS=zipf restricts the kinds."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, pkgload
pkg = pkgload.load(); ghf = pkg.ghf
lib = sys.argv[1] if len(sys.argv) > 1 else "-"
if lib != "-": ghf.LIB_PATH = lib if os.path.isabs(lib) else os.path.join(ROOT, lib)
mibs = [int(x) for x in sys.argv[2:]] or [256, 4096]
from golden_huffman_amd import synth
ctx = ghf.Context(0)
offs = [int(x) for x in os.environ.get("FOREIGN_OFFSETS", "0").split(",")]
kinds = os.environ.get("FOREIGN_KINDS", "uniform,zipf,sym16").split(",")
for mib in mibs:
  for off in offs:
    for kind in kinds:
        n = mib << 20
        d_in = synth.make(torch, kind, n, offset=off * n, device="cuda")
        idx = ctx.index_alloc(n)
        out, nbytes, code = ctx.compress(d_in, index=idx)
        ctx.sync()
        nb = int(nbytes.item())
        dec = ctx.empty_u8(n)
        def timed(fn, reps):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps): fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3
        reps = 5 if mib <= 1024 else 2
        t_idx = timed(lambda: ctx.decode(out, nb, code, idx, d_out=dec), reps)
        t_for = timed(lambda: ctx.decode(out, nb, code, None, d_out=dec, cap=n), reps)
        ctx.sync()
        ok = bool((dec[:n] == d_in).all().item())
        print(json.dumps({"lib": os.path.basename(lib), "mib": mib, "kind": kind, "offset": off, "indexed_ms": round(t_idx, 3), "foreign_ms": round(t_for, 3), "ratio": round(t_for / t_idx, 2), "ok": ok}), flush=True)
        ctx.index_free(idx)
        del d_in, out, dec
        torch.cuda.empty_cache()
