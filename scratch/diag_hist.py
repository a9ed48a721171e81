import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import pkgload
pkg = pkgload.load(); ghf = pkg.ghf
ctx = ghf.Context(0)
g = torch.Generator(device='cuda'); g.manual_seed(1)
for kind in ('uniform', 'sym16'):
    for lg in (24, 25, 26, 27, 28, 29):
        n = 1 << lg
        x = torch.randint(0, 256 if kind == 'uniform' else 16, (n,), dtype=torch.uint8, device='cuda', generator=g)
        ref = torch.bincount(x.to(torch.int32), minlength=256)
        h = ctx.histogram(x); ctx.sync()
        d = (ref - h[:256])
        print(kind, lg, 'chunk', ghf.chunk_symbols(n), 'lost total', int(d.sum()), 'max', int(d.max()), 'min', int(d.min()), 'frac %.5f' % (float(d.sum()) / n), flush=True)
