#!/usr/bin/env python3
"""GPU-box side of round 4's A/Bs: time K1 / K5 / K7 alone (one buffer, one stream, back to back) for a list of
(library, environment) arms, every arm in a subprocess of its own (a process binds one libghf), all arms of one call on one
box in one session.  Every arm verifies its round trip first -- a wrong build says so instead of printing a time.

    python scratch/k_ab.py [--mib 256,4096] [--kinds uniform,zipf,sym16] [--reps 20] arm [arm ...]
    arm = NAME=path/to/libghf.so[,ENV=VALUE,...]     (NAME alone: the product library)
"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, pkgload
pkg = pkgload.load(); ghf = pkg.ghf
if %(lib)r: ghf.LIB_PATH = %(lib)r
from golden_huffman_amd import synth
ctx = ghf.Context(0)
res = {"arm": %(name)r}
for mib in %(mibs)r:
  for kind in %(kinds)r:
    n = mib << 20
    d_in = synth.make(torch, kind, n, offset=0, device="cuda")
    out = ctx.empty_u8(ghf.compress_bound(n)); dec = ctx.empty_u8(n)
    idx = ctx.index_alloc(n)
    h = ctx.histogram(d_in); c = ctx.build_code(h); ctx.encode_plan(d_in, c)
    end = ctx.encode_emit(d_in, c, out, flags=ghf.EMIT_LAST | ghf.EMIT_HEADER, index=idx)
    torch.cuda.synchronize()
    nb = int(end[1].item())
    ctx.decode_prepare(c); ctx.decode(out, nb, c, idx, d_out=dec); ctx.sync()
    ok = bool((dec[:n] == d_in).all().item())
    def timeit(fn, reps=%(reps)d, pre=None):
        # an event pair around EVERY launch, launches queued back to back (no synchronize between them).  Note what that
        # measures at 256 MiB: k_emit 0.13 ms here against 0.103 ms in bench.py's per-kernel pass, which synchronises after
        # every launch -- a launch that starts on an idle GPU leaves its 268 MB of output dirty in the 256 MB Infinity Cache
        # and somebody else pays the write-back; back to back the next launch pays it.  Neither is the host (this form and the
        # older one-pair-around-the-batch form agree); the pipelined bench sits between the two.
        for _ in range(3):
            if pre: pre()
            fn()
        torch.cuda.synchronize()
        evs = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if pre: pre()
            e0.record(); fn(); e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        return sum(ts[: max(1, len(ts) * 3 // 4)]) / max(1, len(ts) * 3 // 4)  # mean of the fastest three quarters
    t_emit = timeit(lambda: ctx.encode_emit(d_in, c, out, flags=ghf.EMIT_LAST | ghf.EMIT_HEADER, index=idx))
    # (prepared tables are single-use: built in front of every decode, outside its event pair -> k_decode alone)
    t_dec = timeit(lambda: ctx.decode(out, nb, c, idx, d_out=dec), pre=lambda: ctx.decode_prepare(c))
    t_hist = timeit(lambda: ctx.histogram(d_in, out=h))
    ctx.sync()
    ok = ok and bool((dec[:n] == d_in).all().item())
    res["%%d_%%s" %% (mib, kind)] = {"ok": ok, "hist_ms": round(t_hist, 4), "emit_ms": round(t_emit, 4), "decode_ms": round(t_dec, 4),
                                   "decode_TBps": round((nb + n) / t_dec / 1e9, 3), "emit_TBps": round((nb + n) / t_emit / 1e9, 3)}
    ctx.index_free(idx)
    del d_in, out, dec
    torch.cuda.empty_cache()
print(json.dumps(res))
'''
ap = argparse.ArgumentParser()
ap.add_argument("--mib", default="256")
ap.add_argument("--kinds", default="uniform")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("arms", nargs="+")
a = ap.parse_args()
mibs = [int(x) for x in a.mib.split(",")]
kinds = a.kinds.split(",")
for arm in a.arms:
    name, _, rest = arm.partition("=")
    parts = rest.split(",") if rest else []
    lib = ""
    env = dict(os.environ)
    for p in parts:
        if "=" in p:
            k, v = p.split("=", 1)
            env[k] = v
        elif p:
            lib = p if os.path.isabs(p) else os.path.join(ROOT, p)
    src = CHILD % {"root": ROOT, "lib": lib, "name": name, "mibs": mibs, "kinds": kinds, "reps": a.reps}
    r = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=900, env=env)
    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    if r.returncode != 0 or not line.startswith("{"):
        print("FAILED %s rc=%d\n%s" % (name, r.returncode, r.stderr[-2000:]), flush=True)
    else:
        print(line, flush=True)
