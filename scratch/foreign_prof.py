#!/usr/bin/env python3
"""run the side-car-less decode path (K6 + K7) a few times on one synthetic stream; for rocprofv3 --kernel-trace --stats"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, pkgload
pkg = pkgload.load(); ghf = pkg.ghf
from golden_huffman_amd import synth
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = ghf.Context(0)
n = mib << 20
d_in = synth.make(torch, kind, n, offset=0, device="cuda")
out, nbytes, code = ctx.compress(d_in)
ctx.sync()
nb = int(nbytes.item())
dec = ctx.empty_u8(n)
for _ in range(4):
    ctx.decode(out, nb, code, None, d_out=dec, cap=n)
ctx.sync()
assert bool((dec == d_in).all().item())
print("ok", kind, mib, nb)
