#!/usr/bin/env python3
"""GPU side of the K5 timeline diagnostic (scratch/k5_stamps_build.py): one k_emit with the stamped library, then the per-wave
(= per-chunk) cycle counts.  s_memtime is NOT synchronised between CUs: only durations are compared.

    python scratch/k5_stamps_run.py [mib] [kind]
"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, pkgload
import numpy as np
pkg = pkgload.load(); ghf = pkg.ghf
ghf.LIB_PATH = os.path.join(ROOT, "scratch", "exp", "libghf_k5stamps.so")
from golden_huffman_amd import synth
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
ctx = ghf.Context(0)
n = mib << 20
d_in = synth.make(torch, kind, n, offset=0, device="cuda")
cap = ghf.compress_bound(n) + (1 << 20)
out = ctx.empty_u8(cap)
idx = ctx.index_alloc(n)
h = ctx.histogram(d_in); c = ctx.build_code(h); ctx.encode_plan(d_in, c)
for _ in range(3):
    end = ctx.encode_emit(d_in, c, out, flags=ghf.EMIT_LAST | ghf.EMIT_HEADER, index=idx)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); end = ctx.encode_emit(d_in, c, out, flags=ghf.EMIT_LAST | ghf.EMIT_HEADER, index=idx); e1.record()
ctx.sync(); torch.cuda.synchronize()
nb = int(end[1].item())
assert nb + (300 << 10) < cap
off = (cap - (256 << 10)) & ~15
w = out[off:off + 8192 * 32].cpu().numpy().view("<u8").reshape(8192, 4).astype("float64")
nch = int((w[:, 1] > 0).sum())
w = w[:nch]
tab, chunk, blk = w[:, 0], w[:, 1], w[:, 3].astype(int)
tot = tab + chunk
xcd = blk % 8
print(json.dumps({
    "mib": mib, "kind": kind, "emit_event_ms": round(e0.elapsed_time(e1), 4), "chunks": nch,
    "table_build_cycles_min_mean_max": [float(tab.min()), float(tab.mean()), float(tab.max())],
    "chunk_cycles_percentiles_0_5_50_95_100": [float(x) for x in np.percentile(chunk, [0, 5, 50, 95, 100])],
    "wave_total_cycles_percentiles_0_5_50_95_100": [float(x) for x in np.percentile(tot, [0, 5, 50, 95, 100])],
    "mean_over_max": round(float(tot.mean() / tot.max()), 3),
    "wave_total_by_xcd_mean": [round(float(tot[xcd == x].mean()), 0) for x in range(8)],
    "wave_total_by_xcd_max": [round(float(tot[xcd == x].max()), 0) for x in range(8)],
    "chunk_cycles_by_wave_of_workgroup_mean": [round(float(chunk[i::8].mean()), 0) for i in range(8)],
    "slowest_20_blocks": [int(b) for b in blk[np.argsort(-tot)[:20]]],
}))
