#!/usr/bin/env python3
"""Generate tests/golden/golden_big.json: the BASELINE configs at FULL size, pinned against the compiled reference itself
(oracle/_ref/ref_glzip = /root/reference's own headers, `make -C oracle ref`; build container only).

For every case the input is the seeded stream of tests/datagen.py (the same bytes golden-huffman_amd/synth.py makes on the
GPU: byte i depends only on (kind, seed, i)), written to a file; the reference compresses it (`ref_glzip c`,
Compressor<CanonicalHuffEncoder<>>, unit_tests/test.cc:101-106) and decompresses it with its bit-serial decoder
(`ref_glzip d`, unit_tests/test.cc:108-116), and the round trip is byte-compared the way compressor_func_test does
(unit_tests/test.cc:48-84).  What is committed is data only: sizes, SHA-256 of input / .crs2 / body, the header bytes.

    python tests/golden/make_golden_big.py            # all cases (about 13 GiB of /dev/shm at a time, ~10 min)
    python tests/golden/make_golden_big.py NAME ...   # some (the others are kept from the existing file)
"""
import base64
import hashlib
import json
import os
import struct
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
from oracle import oracle as orc  # noqa: E402

GiB = 1 << 30
# name -> (kind, n, stream offset); offset 0 at the default seed = bench.py's buffer set 0 of rank 0
CASES = {
    "config2_uniform_256MiB": ("uniform", 256 << 20, 0),
    "config3_zipf_4GiB": ("zipf", 4 * GiB, 0),
    "config5_sym16_4GiB": ("sym16", 4 * GiB, 0),
    "config4_shard0_uniform_4GiB": ("uniform", 4 * GiB, 0),
}
PIECE = 1 << 26


def gen(kind, n, offset):
    if kind == "uniform":
        return datagen.uniform_bytes(n, offset=offset)
    if kind == "zipf":
        return datagen.zipf_bytes(n, offset=offset)
    if kind == "sym16":
        return datagen.sym16_bytes(n, offset=offset)
    raise ValueError(kind)


def sha_file(path, skip=0):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        f.seek(skip)
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def main():
    orc.build()
    assert orc.have_ref(), "oracle/_ref/ref_glzip missing: run `make -C oracle ref` in the build container"
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_glzip")
    path = os.path.join(HERE, "golden_big.json")
    out = {"_generator": "tests/golden/make_golden_big.py", "_reference": "chenghuige/golden-huffman @ /root/reference",
           "_compiler": os.popen("g++ --version").read().splitlines()[0],
           "_stream": "tests/datagen.py, seed 0x%X" % datagen.DEFAULT_SEED, "cases": {}}
    if os.path.exists(path):
        out["cases"] = json.load(open(path))["cases"]
    names = sys.argv[1:] or list(CASES)
    td = "/dev/shm/ghf_golden_big"
    os.makedirs(td, exist_ok=True)
    for name in names:
        kind, n, offset = CASES[name]
        fin, fcrs, fde = (os.path.join(td, name + e) for e in (".bin", ".crs2", ".de"))
        t0 = time.time()
        h_in = hashlib.sha256()
        with open(fin, "wb") as f:
            for pos in range(0, n, PIECE):
                b = gen(kind, min(PIECE, n - pos), offset + pos)
                h_in.update(b)
                f.write(b)
        t1 = time.time()
        # (a runaway decoder would fill the disk: bounded file size, bounded time -- SURVEY 5 hazard 1)
        lim = "ulimit -f %d; " % ((n + (64 << 20)) // 512 * 2)
        subprocess.run(["bash", "-c", lim + "exec timeout 1800 %s c %s %s" % (ref, fin, fcrs)], check=True)
        t2 = time.time()
        subprocess.run(["bash", "-c", lim + "exec timeout 1800 %s d %s %s" % (ref, fcrs, fde)], check=True)
        t3 = time.time()
        assert os.path.getsize(fde) == n, (name, os.path.getsize(fde))
        h_de = sha_file(fde)
        assert h_de == h_in.hexdigest(), name  # the reference's own round trip (unit_tests/test.cc:48-84)
        with open(fcrs, "rb") as f:
            head = f.read(1040 + 8 * 32)
        max_len = struct.unpack(">I", head[1036:1040])[0]
        hs = 1040 + 8 * max_len
        rec = {
            "kind": kind, "n": n, "offset": offset,
            "input_sha256": h_in.hexdigest(),
            "min_len": struct.unpack(">I", head[1032:1036])[0], "max_len": max_len,
            "header_bytes": hs,
            "header_b64": base64.b64encode(head[:hs]).decode(),
            "crs2_bytes": os.path.getsize(fcrs),
            "crs2_sha256": sha_file(fcrs),
            "body_sha256": sha_file(fcrs, hs),
            "decoded_sha256": h_de,
            "ref_seconds": {"compress": round(t2 - t1, 1), "decompress": round(t3 - t2, 1)},
        }
        out["cases"][name] = rec
        for p in (fin, fcrs, fde):
            os.remove(p)
        print("%-30s n=%d crs2=%d min=%d max=%d  gen %.0fs c %.0fs d %.0fs" %
              (name, n, rec["crs2_bytes"], rec["min_len"], max_len, t1 - t0, t2 - t1, t3 - t2), flush=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
    os.rmdir(td)
    print("wrote", path)


if __name__ == "__main__":
    main()
