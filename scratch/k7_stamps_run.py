#!/usr/bin/env python3
"""GPU side of the K7 timeline diagnostic (scratch/k7_stamps_build.py): one decode with the stamped library, then the
per-wave accumulators from behind the decoded output.  Cycles are s_memtime ticks (shader clock).

    python scratch/k7_stamps_run.py [mib] [kind]
"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, pkgload
pkg = pkgload.load(); ghf = pkg.ghf
ghf.LIB_PATH = os.path.join(ROOT, "scratch", "exp", os.environ.get("STAMP_LIB", "libghf_stamps.so"))
from golden_huffman_amd import synth
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
ctx = ghf.Context(0)
n = mib << 20
d_in = synth.make(torch, kind, n, offset=0, device="cuda")
out = ctx.empty_u8(ghf.compress_bound(n))
dec = ctx.empty_u8(n + (4 << 20))
idx = ctx.index_alloc(n)
h = ctx.histogram(d_in); c = ctx.build_code(h); ctx.encode_plan(d_in, c)
end = ctx.encode_emit(d_in, c, out, flags=ghf.EMIT_LAST | ghf.EMIT_HEADER, index=idx)
torch.cuda.synchronize()
nb = int(end[1].item())
for _ in range(3):
    ctx.decode_prepare(c); ctx.decode(out, nb, c, idx, d_out=dec)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
ctx.decode_prepare(c)
e0.record(); ctx.decode(out, nb, c, idx, d_out=dec); e1.record()
ctx.sync(); torch.cuda.synchronize()
assert bool((dec[:n] == d_in).all().item())
off = (n + 255) & ~255
w = dec[off:off + 4096 * 64].cpu().numpy().view("<u8").reshape(4096, 8).astype("float64")
import numpy as np
a, b, c_, d, np_, life, begin, endt = (w[:, i] for i in range(8))
t0 = begin.min()
print(json.dumps({
    "mib": mib, "kind": kind, "decode_event_ms": round(e0.elapsed_time(e1), 4),
    "passes_per_wave": [float(np_.min()), float(np_.mean()), float(np_.max())],
    "per_pass_cycles": {"wait+stage+issue": round(float(a.sum() / np_.sum()), 1), "decode": round(float(b.sum() / np_.sum()), 1),
                        "copy_out": round(float(c_.sum() / np_.sum()), 1), "descr+meta+ticket": round(float(d.sum() / np_.sum()), 1)},
    "pass_cycles_by_wave_quartile": [round(float(x), 0) for x in np.percentile((a + b + c_ + d) / np_, [5, 25, 50, 75, 95])],
    "wave_life_cycles": [float(life.min()), float(life.mean()), float(life.max())],
    "first_start_to_last_end_cycles": float(endt.max() - t0), "start_spread_cycles": float(begin.max() - t0),
    "end_spread_cycles": float(endt.max() - endt.min()),
    "sum_stamped_over_life": round(float(((a + b + c_ + d) / life).mean()), 3),
    "passes_by_wave_of_workgroup": [round(float(x), 1) for x in np_.reshape(256, 16).mean(axis=0)],
    "passes_per_workgroup_min_mean_max": [float(np_.reshape(256, 16).sum(axis=1).min()), float(np_.reshape(256, 16).sum(axis=1).mean()), float(np_.reshape(256, 16).sum(axis=1).max())],
    "idle_behind_own_end_within_workgroup_cycles_mean_max": [float((endt.reshape(256, 16).max(axis=1, keepdims=True) - endt.reshape(256, 16)).mean()), float((endt.reshape(256, 16).max(axis=1, keepdims=True) - endt.reshape(256, 16)).max())],
    "workgroup_span_cycles_mean": float((endt.reshape(256, 16).max(axis=1) - begin.reshape(256, 16).min(axis=1)).mean()),
    "idle_by_wave_slot": [round(float(x), 0) for x in (endt.reshape(256, 16).max(axis=1, keepdims=True) - endt.reshape(256, 16)).mean(axis=0)],
    "workgroup_span_by_xcd_mean": [round(float(x), 0) for x in (endt.reshape(32, 8, 16).max(axis=2) - begin.reshape(32, 8, 16).min(axis=2)).mean(axis=0)],
    "workgroup_span_by_xcd_max": [round(float(x), 0) for x in (endt.reshape(32, 8, 16).max(axis=2) - begin.reshape(32, 8, 16).min(axis=2)).max(axis=0)],
    "workgroup_span_percentiles": [round(float(x), 0) for x in np.percentile(endt.reshape(256, 16).max(axis=1) - begin.reshape(256, 16).min(axis=1), [0, 10, 50, 90, 100])],
    "latest_workgroups_by_xcd(blockIdx, begin, end, passes; cycles from the XCD's first begin)": [
        sorted([(int(b), float(begin.reshape(256, 16)[b].min() - begin.reshape(256, 16)[x::8].min()), float(endt.reshape(256, 16)[b].max() - begin.reshape(256, 16)[x::8].min()), float(np_.reshape(256, 16)[b].sum())) for b in range(x, 256, 8)], key=lambda t: -t[2])[:3] for x in range(8)],
    "workgroup_begin_offsets_by_xcd_sorted": [sorted(int(v) for v in (begin.reshape(256, 16)[x::8].min(axis=1) - np.median(begin.reshape(256, 16)[x::8].min(axis=1)))) for x in range(8)],
    "passes_by_xcd": [round(float(x), 1) for x in np_.reshape(32, 8, 16).sum(axis=2).mean(axis=0)],
    "decode_cycles_per_pass_by_wave_of_workgroup": [round(float(x), 0) for x in (b.reshape(256, 16).sum(axis=0) / np_.reshape(256, 16).sum(axis=0))],
    "wait_cycles_per_pass_by_wave_of_workgroup": [round(float(x), 0) for x in (a.reshape(256, 16).sum(axis=0) / np_.reshape(256, 16).sum(axis=0))]}))
