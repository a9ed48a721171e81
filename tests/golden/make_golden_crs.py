#!/usr/bin/env python3
"""Generate tests/golden/golden_crs.json (the `.crs` format) by running the COMPILED
REFERENCE (oracle/_ref/ref_glzip nc / nd / nt = Compressor<NormalHuffEncoder<>>, Decompressor<NormalHuffDecoder<>>,
NormalHuffEncoder::encode_map_) on the seeded inputs of tests/golden/cases.py.  Build container only.
Committed: data only (input SHA-256, the 256 code strings, the tree header bytes, SHA-256/size of the .crs,
the SHA-256 of what the reference's decoder returned; whole files for the smallest cases).

    python tests/golden/make_golden_crs.py
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
from cases import CRS_CASES as CASES, INLINE_CRS2  # noqa: E402  (the shared cases + the deep-tree ones)


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def main():
    orc.build()
    assert orc.have_ref(), "oracle/_ref/ref_glzip missing: run `make -C oracle ref` in the build container"
    out = {"_generator": "tests/golden/make_golden_crs.py", "_reference": "chenghuige/golden-huffman @ /root/reference",
           "_compiler": os.popen("g++ --version").read().splitlines()[0], "cases": {}}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for name, fn in CASES.items():
            data = fn()
            nsym = int(np.count_nonzero(np.bincount(data, minlength=256)))
            if nsym < 2:
                # one distinct byte: the lone leaf gets the empty code and the reference's decoder walks off a NULL
                # child (include/huff_tree.cc:255-271) -- undefined there, refused here
                out["cases"][name] = {"n": int(data.size), "input_sha256": sha(data), "undefined": "single symbol"}
                continue
            fin = os.path.join(td, name + ".bin")
            fcrs = os.path.join(td, name + ".crs")
            fde = os.path.join(td, name + ".de")
            data.tofile(fin)
            codes = json.loads(orc.ref_run(["nt", fin]))["codes"]
            orc.ref_run(["nc", fin, fcrs])
            orc.ref_run(["nd", fcrs, fde])
            crs = np.fromfile(fcrs, dtype=np.uint8)
            de = np.fromfile(fde, dtype=np.uint8)
            assert de.size == data.size and np.array_equal(de, data), name  # reference round trip
            hs = 2 * (2 * nsym - 1)
            rec = {
                "n": int(data.size),
                "input_sha256": sha(data),
                "codes": codes,
                "max_len": max(len(c) for c in codes),
                "tree_bytes": hs,
                "tree_b64": base64.b64encode(bytes(crs[:hs])).decode(),
                "prefix": [int(crs[hs]), int(crs[hs + 1])],
                "crs_bytes": int(crs.size),
                "crs_sha256": sha(crs),
                "decoded_sha256": sha(de),
            }
            if name in INLINE_CRS2:
                rec["crs_b64"] = base64.b64encode(bytes(crs)).decode()
            out["cases"][name] = rec
            print("%-24s n=%-8d crs=%-8d max_len=%d" % (name, data.size, crs.size, rec["max_len"]))
    path = os.path.join(HERE, "golden_crs.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
