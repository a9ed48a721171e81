"""Seeded input cases behind tests/golden/golden.json (SURVEY 8c list).  name -> bytes builder.
The builders are deterministic (tests/datagen.py); golden.json stores the SHA-256 of every input so a
drifting generator is caught before any parity claim is made."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import datagen as dg  # noqa: E402


def _b(s):
    return np.frombuffer(s, dtype=np.uint8).copy()


def _geom_counts(k):
    return [1 << i for i in range(k)]


CASES = {
    # (i) hand-checkable
    "aaaabbc": lambda: _b(b"aaaabbc"),
    "aaaaaaaa": lambda: _b(b"aaaaaaaa"),
    "single_x": lambda: _b(b"x"),
    "ab": lambda: _b(b"ab"),
    "zeros_100000": lambda: np.zeros(100000, dtype=np.uint8),
    "ff_1": lambda: _b(b"\xff"),
    # (ii) every count ties: heap order alone decides the 8/9 split
    "all256_once": lambda: np.arange(256, dtype=np.uint8),
    "all256_x16_shuffled": lambda: dg.counts_to_bytes([16] * 256, seed=7),
    "all256_once_reversed": lambda: np.arange(255, -1, -1).astype(np.uint8),
    # (iii) uniform random: many near-ties
    "uniform_4k": lambda: dg.uniform_bytes(4096, seed=11),
    "uniform_64k": lambda: dg.uniform_bytes(65536, seed=12),
    "uniform_1m": lambda: dg.uniform_bytes(1 << 20, seed=13),
    # (iv) 16-symbol
    "sym16_64k": lambda: dg.sym16_bytes(65536, seed=21),
    # (v) Zipf(1.1)
    "zipf_64k": lambda: dg.zipf_bytes(65536, seed=31),
    "zipf_1m": lambda: dg.zipf_bytes(1 << 20, seed=32),
    # (vi) Fibonacci weights push max_len up: k data symbols -> max_len = k (32 = the reference's limit)
    "fib22": lambda: dg.counts_to_bytes(dg.fib_counts(22), seed=41),
    "fib30": lambda: dg.counts_to_bytes(dg.fib_counts(30), seed=42),
    "fib32_maxlen32": lambda: dg.counts_to_bytes(dg.fib_counts(32), seed=43),
    # (vii) geometric counts: long chains of exact ties between a leaf and a merged node
    "geom16": lambda: dg.counts_to_bytes(_geom_counts(16), seed=51),
    "geom20_sparse_values": lambda: _sparse(dg.counts_to_bytes(_geom_counts(20), seed=52)),
    # (viii) body bits == 0..7 (mod 8): pins the 1-padding
    **{"sym16_n%d" % n: (lambda n=n: dg.sym16_bytes(n, seed=61)) for n in range(1000, 1008)},
    **{"zipf_n%d" % n: (lambda n=n: dg.zipf_bytes(n, seed=62)) for n in range(3001, 3009)},
    # (ix) straddling the reference's 64 KiB buffer edge
    "uniform_65535": lambda: dg.uniform_bytes(65535, seed=71),
    "uniform_65536": lambda: dg.uniform_bytes(65536, seed=71),
    "uniform_65537": lambda: dg.uniform_bytes(65537, seed=71),
    "text_131073": lambda: dg.text_bytes(131073, seed=72),
    # (x) BASELINE config 1: 1 MiB enwik-style ASCII (synthetic order-0 text, seed 1)
    "text_1m": lambda: dg.text_bytes(1 << 20, seed=1),
    # a few awkward sizes around our own kernel tile sizes (1 KiB wave blocks, 32 KiB chunks)
    "zipf_1023": lambda: dg.zipf_bytes(1023, seed=81),
    "zipf_1025": lambda: dg.zipf_bytes(1025, seed=81),
    "uniform_32769": lambda: dg.uniform_bytes(32769, seed=82),
    "sym16_98303": lambda: dg.sym16_bytes(98303, seed=83),
    "two_values_skewed": lambda: dg.counts_to_bytes([70000, 3], seed=84),
}


# `.crs` only: the reference's tree codes are strings of any length (include/huff_tree.cc:157-170), so a tree
# deeper than 32 is reference-defined there -- while the canonical coder stops at 32 bits (canonical_huff_encoder.h:43-44).
# 34 Fibonacci counts (24 157 815 bytes) -> depth 33.
CRS_DEEP_CASES = {
    "fib34_depth33": lambda: dg.counts_to_bytes(dg.fib_counts(34), seed=44),
}
CRS_CASES = dict(CASES)
CRS_CASES.update(CRS_DEEP_CASES)


def _sparse(a):
    """spread the used byte values over 0..255 (value v -> 13*v+5 mod 256)"""
    return ((a.astype(np.int64) * 13 + 5) % 256).astype(np.uint8)


# small enough to commit the whole .crs2
INLINE_CRS2 = {"aaaabbc", "aaaaaaaa", "single_x", "ab", "ff_1", "all256_once", "sym16_n1000", "zipf_n3001"}
