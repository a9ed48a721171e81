"""Deterministic, counter-based synthetic byte streams (numpy), shared by the golden-vector generator,
the parity tests and bench.py.  Byte i of a stream depends only on (kind, seed, i), so any rank can
produce any shard independently.  No numpy RNG is used: the streams must never change.

word(k)  = splitmix64(seed + k * GOLDEN)            (k = 0, 1, 2, ...)
uniform  : byte i            = (word(i >> 3) >> (8 * (i & 7))) & 0xFF
weighted : u32  i            = (word(i >> 1) >> (32 * (i & 1))) & 0xFFFFFFFF ; byte = #{thresholds <= u32}
           with thresholds[j] = floor(2^32 * cumsum(p)[j])  (last one forced to 2^32)
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
DEFAULT_SEED = 0x9E3779B97F4A7C15


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _words(seed, k0, k1):
    k = np.arange(k0, k1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return splitmix64(np.uint64(seed) + k * GOLDEN)


def uniform_bytes(n, seed=DEFAULT_SEED, offset=0):
    """n i.i.d. uniform bytes = stream positions [offset, offset+n)."""
    lo, hi = offset, offset + n
    w = _words(seed, lo >> 3, (hi + 7) >> 3)
    b = w.view(np.uint8)  # little-endian host: byte j of word k is (w >> 8j) & 0xFF
    s = lo - ((lo >> 3) << 3)
    return b[s : s + n].copy()


def thresholds_from_probs(p):
    p = np.asarray(p, dtype=np.float64)
    c = np.cumsum(p / p.sum())
    t = np.floor(c * 4294967296.0).astype(np.uint64)
    t[-1] = np.uint64(4294967296)
    return t


def weighted_bytes(n, thresholds, seed=DEFAULT_SEED, offset=0, values=None):
    lo, hi = offset, offset + n
    w = _words(seed, lo >> 1, (hi + 1) >> 1)
    u = w.view(np.uint32)
    s = lo - ((lo >> 1) << 1)
    u = u[s : s + n].astype(np.uint64)
    idx = np.searchsorted(thresholds, u, side="right").astype(np.int64)
    if values is None:
        return idx.astype(np.uint8)
    return np.asarray(values, dtype=np.uint8)[idx]


def zipf_probs(alpha=1.1, k=256):
    r = np.arange(1, k + 1, dtype=np.float64)
    return r ** (-alpha)


def zipf_bytes(n, alpha=1.1, seed=DEFAULT_SEED, offset=0):
    """P(byte = k-1) proportional to k^-alpha, k = 1..256 (BASELINE config 3)."""
    return weighted_bytes(n, thresholds_from_probs(zipf_probs(alpha)), seed, offset)


def sym16_bytes(n, seed=DEFAULT_SEED, offset=0):
    """i.i.d. uniform over the 16 values 0..15 (BASELINE config 5)."""
    return uniform_bytes(n, seed, offset) & np.uint8(15)


# order-0 English-like text model for the "enwik-style ASCII" plumbing case (BASELINE config 1):
# no enwik file exists in the container, so the text is synthesised from letter/space/punctuation
# frequencies (per mille), seed 1.
_TEXT_ALPHABET = " etaoinshrdlcumwfgypbvkjxqzETAOINSHRDLCUMWFGYPBVKJXQZ.,\n0123456789'\"-;:()[]<>/=&|"
_TEXT_WEIGHTS = (
    [180, 102, 75, 65, 61, 57, 56, 51, 50, 49, 35, 33, 23, 23, 20, 19, 18, 16, 16, 15, 12, 8, 6, 1.2, 1.2, 0.8, 0.6]
    + [2.0, 3.0, 2.5, 1.5, 2.5, 1.5, 2.5, 1.5, 1.5, 1.2, 1.2, 1.8, 0.8, 1.6, 1.4, 1.2, 1.0, 0.6, 1.4, 1.2, 0.6, 0.6, 0.4, 0.2, 0.2, 0.2]
    + [10, 9, 12]
    + [3, 3, 2, 1.5, 1.2, 1.2, 1, 1, 1, 1.5]
    + [2, 2, 2, 0.6, 0.6, 1, 1, 3, 3, 2, 2, 1.5, 2, 1, 1.5]
)


def text_bytes(n, seed=1, offset=0):
    assert len(_TEXT_ALPHABET) == len(_TEXT_WEIGHTS), (len(_TEXT_ALPHABET), len(_TEXT_WEIGHTS))
    vals = np.frombuffer(_TEXT_ALPHABET.encode("ascii"), dtype=np.uint8)
    return weighted_bytes(n, thresholds_from_probs(_TEXT_WEIGHTS), seed, offset, values=vals)


def counts_to_bytes(counts, seed=DEFAULT_SEED):
    """a byte stream with exactly counts[v] occurrences of value v, order shuffled deterministically."""
    counts = np.asarray(counts, dtype=np.int64)
    data = np.repeat(np.arange(counts.size, dtype=np.int64), counts).astype(np.uint8)
    key = _words(seed, 0, data.size)
    return data[np.argsort(key, kind="stable")]


def fib_counts(k):
    """1, 2, 3, 5, 8, ... (k terms): together with the implicit end-of-stream count of 1 the weights are
    exactly Fibonacci, the strictly-skewed case: max code length = k."""
    f = [1, 2]
    while len(f) < k:
        f.append(f[-1] + f[-2])
    return f[:k]


def make(kind, n, seed=DEFAULT_SEED, offset=0):
    if kind == "uniform":
        return uniform_bytes(n, seed, offset)
    if kind == "zipf":
        return zipf_bytes(n, 1.1, seed, offset)
    if kind == "sym16":
        return sym16_bytes(n, seed, offset)
    if kind == "text":
        return text_bytes(n, seed, offset)
    raise ValueError(kind)
