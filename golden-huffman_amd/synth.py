"""Synthetic byte streams generated ON the device with torch integer ops -- the same counter-based
streams as tests/datagen.py (byte i depends only on (kind, seed, i)), so every rank can produce its own
shard of a BASELINE config without touching the host.  Bench/test plumbing, not the product."""
import numpy as np

_GOLDEN = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB
DEFAULT_SEED = _GOLDEN


def _s64(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(torch, z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)


def _words(torch, seed, k0, k1, device):
    k = torch.arange(k0, k1, dtype=torch.int64, device=device)
    z = k * _s64(_GOLDEN) + _s64(seed)  # int64 arithmetic wraps like uint64
    z = z + _s64(_GOLDEN)
    z = (z ^ _lsr(torch, z, 30)) * _s64(_M1)
    z = (z ^ _lsr(torch, z, 27)) * _s64(_M2)
    z = z ^ _lsr(torch, z, 31)
    return z


def _thresholds(p):
    p = np.asarray(p, dtype=np.float64)
    c = np.cumsum(p / p.sum())
    t = np.floor(c * 4294967296.0).astype(np.int64)
    t[-1] = 4294967296
    return t


def uniform_bytes(torch, n, seed=DEFAULT_SEED, offset=0, device="cuda", mask=0xFF, piece=1 << 27):
    out = torch.empty(n, dtype=torch.uint8, device=device)
    pos = 0
    while pos < n:
        m = min(piece, n - pos)
        lo, hi = offset + pos, offset + pos + m
        w = _words(torch, seed, lo >> 3, (hi + 7) >> 3, device)
        b = w.view(torch.uint8)
        s = lo - ((lo >> 3) << 3)
        out[pos : pos + m] = b[s : s + m]
        pos += m
    if mask != 0xFF:
        out &= mask
    return out


def weighted_bytes(torch, n, probs, seed=DEFAULT_SEED, offset=0, device="cuda", piece=1 << 26):
    thr = torch.from_numpy(_thresholds(probs)).to(device)
    out = torch.empty(n, dtype=torch.uint8, device=device)
    pos = 0
    while pos < n:
        m = min(piece, n - pos)
        lo, hi = offset + pos, offset + pos + m
        w = _words(torch, seed, lo >> 1, (hi + 1) >> 1, device)
        u = torch.stack([w & 0xFFFFFFFF, _lsr(torch, w, 32)], dim=1).reshape(-1)
        s = lo - ((lo >> 1) << 1)
        idx = torch.searchsorted(thr, u[s : s + m].contiguous(), right=True)
        out[pos : pos + m] = idx.to(torch.uint8)
        pos += m
    return out


def make(torch, kind, n, seed=DEFAULT_SEED, offset=0, device="cuda"):
    """kind: uniform (BASELINE configs 2/4) | zipf (config 3, alpha 1.1) | sym16 (config 5)"""
    if kind == "uniform":
        return uniform_bytes(torch, n, seed, offset, device)
    if kind == "sym16":
        return uniform_bytes(torch, n, seed, offset, device, mask=15)
    if kind == "zipf":
        return weighted_bytes(torch, n, np.arange(1, 257, dtype=np.float64) ** -1.1, seed, offset, device)
    raise ValueError(kind)
