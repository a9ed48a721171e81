import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, pkgload
from oracle import oracle as orc
pkg=pkgload.load(); ghf=pkg.ghf
ctx=ghf.Context(0)
rng=np.random.default_rng(1)
for n in (7000000,):
    data=rng.integers(0,7,size=n).astype(np.uint8)
    # make counts exactly equal so that a,b,c,EOF all get 2 bits
    data[:n - n%7]=np.tile(np.arange(7,dtype=np.uint8), n//7)
    rng.shuffle(data)
    ref=orc.compress(data)
    code,hs=ghf.parse_header(ref)
    print("n",n,"min/max len",code.min_len,code.max_len, flush=True)
    d=torch.from_numpy(np.concatenate([ref,np.zeros(32,np.uint8)])).cuda()
    t0=time.time()
    out,n2=ctx.decode(d, ref.size, ctx.code_to_device(code), None, cap=n+64)
    ctx.sync()
    dt=time.time()-t0
    print("  decoded",int(n2.item()),"ok",np.array_equal(out[:n].cpu().numpy(),data),"%.3f s"%dt, flush=True)
