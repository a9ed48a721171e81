#!/usr/bin/env python3
"""GPU-box side of the ablations: time emit / decode of the 256 MiB uniform workload with every scratch/exp/libghf_<name>.so
(one subprocess per library: a process binds one libghf).  Outputs of ablated builds are WRONG by construction; only times count."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, pkgload
pkg = pkgload.load(); ghf = pkg.ghf
ghf.LIB_PATH = %(lib)r
from golden_huffman_amd import synth
ctx = ghf.Context(0)
mib = int(%(mib)d); n = mib << 20
d_in = synth.make(torch, %(kind)r, n, offset=0, device="cuda")
out = ctx.empty_u8(ghf.compress_bound(n)); dec = ctx.empty_u8(n)
idx = ctx.index_alloc(n)
h = ctx.histogram(d_in); c = ctx.build_code(h); ctx.encode_plan(d_in, c)
end = ctx.encode_emit(d_in, c, out, flags=ghf.EMIT_LAST | ghf.EMIT_HEADER, index=idx)
torch.cuda.synchronize()
nb = int(end[1].item())
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
t_emit = timeit(lambda: ctx.encode_emit(d_in, c, out, flags=ghf.EMIT_LAST | ghf.EMIT_HEADER, index=idx))
def dec_once():
    ctx.decode_prepare(c); ctx.decode(out, nb, c, idx, d_out=dec)
t_dec = timeit(dec_once)
t_hist = timeit(lambda: ctx.histogram(d_in, out=h))
print(json.dumps({"lib": os.path.basename(%(lib)r), "emit_ms": round(t_emit, 4), "decode_ms": round(t_dec, 4), "hist_ms": round(t_hist, 4)}))
'''
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
for lib in sorted(glob.glob(os.path.join(ROOT, "scratch", "exp", "libghf_*.so"))):
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "lib": lib, "mib": mib, "kind": kind}], capture_output=True, text=True, timeout=300)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAILED " + lib + " " + r.stderr[-300:]), flush=True)
