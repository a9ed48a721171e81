#!/usr/bin/env python3
"""randomised soak of the side-car-less decode (K6 + K7): random alphabets, distributions, sizes, trailing bytes; every
stream is the oracle's, every decode is compared with the input.  usage: python scratch/k6_soak.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, pkgload
from oracle import oracle as orc
pkg = pkgload.load(); ghf = pkg.ghf
ctx = ghf.Context(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); cases = 0; kinds = {}
while time.time() - t0 < budget:
    kind = rng.choice(["flat256", "flat16", "flatk", "zipf", "geom", "two", "runs256", "runs16"])
    n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 70000), rng.integers(70000, 9_000_000)]))
    if kind in ("flat256", "runs256", "flat16", "runs16"):
        V = 256 if "256" in kind else 16
        n = max(n, 2 * V)
        rare = int(rng.integers(0, V))
        data = np.repeat(np.arange(V, dtype=np.uint8), n // V + 1)
        data = data[np.logical_or(data != rare, np.arange(data.size) % 8 != 0)][:n]
        data = rng.permutation(data)
        if kind.startswith("runs"):
            data = np.sort(data) if rng.random() < 0.3 else data
            for _ in range(int(rng.integers(0, 6))):
                a = int(rng.integers(0, max(1, data.size - 40))); data[a:a + int(rng.integers(1, 40))] = rare
    elif kind == "flatk":
        k = int(rng.integers(2, 40)); data = rng.integers(0, k, n).astype(np.uint8)
    elif kind == "zipf":
        a = float(rng.uniform(1.05, 2.0)); data = (np.minimum(rng.zipf(a, n), 256) - 1).astype(np.uint8)
    elif kind == "geom":
        data = np.minimum(rng.geometric(float(rng.uniform(0.05, 0.6)), n) - 1, 255).astype(np.uint8)
    else:
        data = (rng.random(n) < float(rng.uniform(0.001, 0.5))).astype(np.uint8) * int(rng.integers(1, 256))
    if data.size == 0 or np.unique(data).size < 1:
        continue
    crs = orc.compress(data)
    code, hs = ghf.parse_header(crs)
    stream = crs
    if rng.random() < 0.25:
        stream = np.concatenate([crs, rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8)])
    d = torch.from_numpy(np.concatenate([stream, np.zeros(64, np.uint8)])).cuda()
    try:
        out, nout = ctx.decode(d, stream.size, ctx.code_to_device(code), None, cap=data.size + 4096)
        ctx.sync()
        ok = int(nout.item()) == data.size and np.array_equal(out[: data.size].cpu().numpy(), data)
    except Exception as e:  # noqa: BLE001
        ok = False; print("EXC", e)
    cases += 1; kinds[kind] = kinds.get(kind, 0) + 1
    if not ok:
        print("FAIL kind", kind, "n", data.size, "lens", code.min_len, code.max_len, "trailing", stream.size - crs.size, flush=True)
        np.save(os.path.join(ROOT, "gpurun_out", "k6_soak_fail.npy"), data)
        sys.exit(1)
print("ok", cases, "cases", kinds, "in %.0f s" % (time.time() - t0))
