#!/usr/bin/env python3
"""one process, one library, K1 / K5 / K7 twenty times each on one buffer: what `rocprofv3 --kernel-trace --stats -- python3
scratch/prof_one.py [lib.so|-] [mib] [kind]` wraps to get per-kernel durations of an experimental build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, pkgload
pkg = pkgload.load(); ghf = pkg.ghf
lib = sys.argv[1] if len(sys.argv) > 1 else "-"
if lib != "-": ghf.LIB_PATH = lib if os.path.isabs(lib) else os.path.join(ROOT, lib)
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 256
kind = sys.argv[3] if len(sys.argv) > 3 else "uniform"
from golden_huffman_amd import synth
ctx = ghf.Context(0)
n = mib << 20
d_in = synth.make(torch, kind, n, offset=0, device="cuda")
out = ctx.empty_u8(ghf.compress_bound(n)); dec = ctx.empty_u8(n)
idx = ctx.index_alloc(n)
h = ctx.histogram(d_in); c = ctx.build_code(h); ctx.encode_plan(d_in, c)
end = ctx.encode_emit(d_in, c, out, flags=ghf.EMIT_LAST | ghf.EMIT_HEADER, index=idx)
torch.cuda.synchronize()
nb = int(end[1].item())
for _ in range(20):
    ctx.histogram(d_in, out=h)
    ctx.encode_emit(d_in, c, out, flags=ghf.EMIT_LAST | ghf.EMIT_HEADER, index=idx)
    ctx.decode_prepare(c); ctx.decode(out, nb, c, idx, d_out=dec)
ctx.sync(); torch.cuda.synchronize()
assert bool((dec[:n] == d_in).all().item())
print("ok")
