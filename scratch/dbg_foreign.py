#!/usr/bin/env python3
"""foreign decode of 3 MiB of 16 equally likely values with every scratch/exp/libghf_*.so (one subprocess each)"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, pkgload
pkg = pkgload.load(); ghf = pkg.ghf
ghf.LIB_PATH = %(lib)r
from oracle import oracle as orc
rng = np.random.default_rng(5)
n = (3 << 20) + 777
V = int(%(values)d)
data = np.repeat(np.arange(V, dtype=np.uint8), n // V)
data = rng.permutation(data[np.logical_or(data != 9, np.arange(data.size) %% 8 != 0)])
crs = orc.compress(data)
code, hs = ghf.parse_header(crs)
ctx = ghf.Context(0)
d = torch.from_numpy(np.concatenate([crs, np.zeros(64, np.uint8)])).cuda()
try:
    out, nout = ctx.decode(d, crs.size, ctx.code_to_device(code), None, cap=data.size + 4096)
    ctx.sync()
    ok = int(nout.item()) == data.size and np.array_equal(out[:data.size].cpu().numpy(), data)
    print(V, os.path.basename(%(lib)r), "ok" if ok else "WRONG OUTPUT", int(nout.item()), data.size)
except Exception as e:
    print(V, os.path.basename(%(lib)r), "ERROR", e)
'''
for lib in sorted(glob.glob(os.path.join(ROOT, "scratch", "exp", "libghf_*.so"))):
  for values in (16, 256):
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "lib": lib, "values": values}], capture_output=True, text=True, timeout=300)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAILED " + lib + " " + r.stderr[-400:]), flush=True)
