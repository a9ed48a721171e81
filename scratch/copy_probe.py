#!/usr/bin/env python3
"""what a kernel that only moves bytes reaches on this box: the library's own streaming copy (ghf_copy_d2d, nt and plain) and
torch's copy_, at the bench sizes.  GB/s of read + write."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, pkgload
ghf = pkgload.load().ghf
ctx = ghf.Context(0)
res = {}
for mib in (256, 1024, 4096):
    n = mib << 20
    a = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    def timeit(fn, reps=10):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return 2.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    res["%dMiB" % mib] = {"ghf_nt": round(timeit(lambda: ctx.copy_d2d(b, a, non_temporal=True)), 1),
                          "ghf_plain": round(timeit(lambda: ctx.copy_d2d(b, a, non_temporal=False)), 1),
                          "torch_copy_": round(timeit(lambda: b.copy_(a)), 1)}
    assert bool((a == b).all().item())
    del a, b
    torch.cuda.empty_cache()
print(json.dumps(res))
