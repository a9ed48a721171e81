#!/usr/bin/env python3
"""Generate tests/golden/golden.json by running the COMPILED REFERENCE (oracle/_ref/ref_glzip, built
from /root/reference in place by `make -C oracle ref`) on the seeded inputs of tests/golden/cases.py.

Only possible in the build container (the reference does not travel).  What is committed is data:
inputs' SHA-256, the reference's per-stage tables, header bytes, the SHA-256/size of the .crs2 it
wrote, the SHA-256 of what its bit-serial decoder returned -- and for the smallest cases the whole
.crs2.  No reference source is stored.

    python tests/golden/make_golden.py
"""
import base64
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import oracle as orc  # noqa: E402
from cases import CASES, INLINE_CRS2  # noqa: E402


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def main():
    orc.build()
    assert orc.have_ref(), "oracle/_ref/ref_glzip missing: run `make -C oracle ref` in the build container"
    out = {"_generator": "tests/golden/make_golden.py", "_reference": "chenghuige/golden-huffman @ /root/reference",
           "_compiler": os.popen("g++ --version").read().splitlines()[0], "cases": {}}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for name, fn in CASES.items():
            data = fn()
            fin = os.path.join(td, name + ".bin")
            fcrs = os.path.join(td, name + ".crs2")
            fde = os.path.join(td, name + ".de")
            data.tofile(fin)
            tables = json.loads(orc.ref_run(["t", fin]))
            orc.ref_run(["c", fin, fcrs])
            orc.ref_run(["d", fcrs, fde])
            crs = np.fromfile(fcrs, dtype=np.uint8)
            de = np.fromfile(fde, dtype=np.uint8)
            assert de.size == data.size and np.array_equal(de, data), name  # reference round trip
            hs = 1040 + 8 * tables["max_len"]
            rec = {
                "n": int(data.size),
                "input_sha256": sha(data),
                **tables,
                "header_bytes": hs,
                "header_b64": base64.b64encode(bytes(crs[:hs])).decode(),
                "crs2_bytes": int(crs.size),
                "crs2_sha256": sha(crs),
                "body_sha256": sha(crs[hs:]),
                "body_tail_hex": bytes(crs[-8:]).hex(),
                "decoded_sha256": sha(de),
            }
            if name in INLINE_CRS2:
                rec["crs2_b64"] = base64.b64encode(bytes(crs)).decode()
            out["cases"][name] = rec
            print("%-24s n=%-8d crs2=%-8d min=%d max=%d" % (name, data.size, crs.size, tables["min_len"], tables["max_len"]))
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", os.path.join(HERE, "golden.json"), os.path.getsize(os.path.join(HERE, "golden.json")), "bytes")


if __name__ == "__main__":
    main()
