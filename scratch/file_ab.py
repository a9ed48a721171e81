"""A/B of an environment switch of the host layer on one file: python scratch/file_ab.py <GiB> <kind> VAR=VALUE"""
import json, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(__file__))
import file_perf as fp
gib, kind, var = float(sys.argv[1]), sys.argv[2], sys.argv[3]
k, v = var.split("=")
with tempfile.TemporaryDirectory(dir="/dev/shm", prefix="ghf_") as d:
    f = os.path.join(d, kind + ".bin")
    fp.make(kind, int(gib * (1 << 30)) + 12345).tofile(f)
    for rnd in range(2):
        for on in (False, True):
            e = dict(os.environ)
            if on:
                e[k] = v
            r = subprocess.run([fp.TOOL, f, "7"], capture_output=True, text=True, env=e, timeout=600)
            try:
                j = json.loads(r.stdout.strip().splitlines()[-1])
                print(kind, (var if on else "default"), "compress", j["compress_ms"], "decompress", j["decompress_ms"], flush=True)
            except Exception:
                print("FAILED", r.stdout[-300:], r.stderr[-300:])
