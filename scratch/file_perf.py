"""File-to-file rate of the C++ host layer on files in /dev/shm: ghf_tool <file> 7.
usage: python scratch/file_perf.py [GiB] [kind ...]   -> gpurun_out/file_perf.json (one record per kind)"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as dg  # noqa: E402

TOOL = os.path.join(ROOT, "golden-huffman_amd", "host", "bin", "ghf_tool")


def make(kind, n):
    base_n = 64 << 20
    if kind == "uniform":
        base = dg.uniform_bytes(base_n, seed=11)
    elif kind == "zipf":
        base = dg.zipf_bytes(base_n, seed=11)
    else:
        base = dg.text_bytes(base_n, seed=11) if hasattr(dg, "text_bytes") else dg.sym16_bytes(base_n, seed=11)
    out = np.empty(n, dtype=np.uint8)
    for t, lo in enumerate(range(0, n, base_n)):
        hi = min(n, lo + base_n)
        out[lo:hi] = np.roll(base, 4099 * t)[: hi - lo]
    return out


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    kinds = sys.argv[2:] or ["uniform", "zipf"]
    n = int(gib * (1 << 30)) + 12345
    res = {}
    for kind in kinds:
        with tempfile.TemporaryDirectory(dir="/dev/shm", prefix="ghf_") as d:
            f = os.path.join(d, kind + ".bin")
            make(kind, n).tofile(f)
            # default sink (a NEW output file every round: bound by the kernel's page allocation), then GHF_SINK=reuse (the
            # output files of the first round are written over: their pages exist), optionally without keeping the input in HBM
            variants = [("", {}), ("_reuse", {"GHF_SINK": "reuse"})]
            if os.environ.get("GHF_PERF_REREAD"):
                variants.append(("_reread", {"GHF_RESIDENT_BYTES": "0"}))
            for suffix, env_extra in variants:
                env = dict(os.environ)
                env.update(env_extra)
                for g in (f + ".crs2", f + ".crs2.de"):
                    if os.path.exists(g):
                        os.remove(g)
                r = subprocess.run([TOOL, f, "7"], capture_output=True, text=True, timeout=1200, env=env)
                key = kind + suffix
                try:
                    res[key] = json.loads(r.stdout.strip().splitlines()[-1])
                except Exception:
                    res[key] = {"error": r.stdout[-500:] + r.stderr[-500:], "rc": r.returncode}
                res[key]["crs2_bytes"] = os.path.getsize(f + ".crs2") if os.path.exists(f + ".crs2") else None
                print(key, json.dumps(res[key]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "file_perf.json"), "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
