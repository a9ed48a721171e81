# GPU box: randomised soak of the C++ host layer's file pipeline (ghf_tool modes 3 and 4/5/6) against the oracle:
# random data, sizes, piece sizes, thread counts, residency and sink modes.  usage: host_soak.py SECONDS [seed]
import hashlib, os, subprocess, sys, tempfile, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as orc
TOOL = os.path.join(ROOT, "golden-huffman_amd", "host", "bin", "ghf_tool")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
t0 = time.time(); cases = 0
def gen():
    n = int(2 ** (rng.uniform(25.5, 28.2) if os.environ.get("HOST_SOAK_BIG") else rng.uniform(0, 24.5))) + int(rng.integers(0, 70))
    k = int(rng.integers(1, 257))
    mode = int(rng.integers(0, 5))
    if mode == 0: w = np.ones(k)
    elif mode == 1: w = rng.random(k) ** int(rng.integers(1, 12))
    elif mode == 2: w = 0.5 ** np.arange(k) * rng.uniform(0.8, 1.2, k)
    elif mode == 3: w = (np.arange(k) + 1.0) ** -rng.uniform(0.5, 2.5)
    else: w = np.where(rng.random(k) < 0.1, 100.0, 1.0)
    w = np.maximum(w, 1e-300); w = w / w.sum()
    syms = rng.permutation(256)[:k].astype(np.uint8)
    data = syms[rng.choice(k, size=n, p=w)]
    if rng.random() < 0.3:  # non-stationary: a block of something else in the middle
        a = int(rng.integers(0, n)); b = min(n, a + int(rng.integers(1, max(2, n // 2))))
        data[a:b] = rng.integers(0, 256, b - a, dtype=np.uint8) if rng.random() < 0.5 else 0
    return data
def dump_on_failure(f, tag):
    import shutil
    out = os.path.join(ROOT, "gpurun_out", "host_soak_fail")
    os.makedirs(out, exist_ok=True)
    for ext in ("", ".crs", ".crs2"):
        if os.path.exists(f + ext) and os.path.getsize(f + ext) < (48 << 20):
            shutil.copy(f + ext, os.path.join(out, "x.bin" + ext))
    open(os.path.join(out, "tag.txt"), "w").write(tag + "\n")

with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None, prefix="ghf_soak_") as d:
  try:
    while time.time() - t0 < budget:
          data = gen()
          f = os.path.join(d, "x.bin"); data.tofile(f)
          env = dict(os.environ)
          env["GHF_PIECE_BYTES"] = str(int(rng.choice([4 << 20, 16 << 20, 32 << 20] if os.environ.get("HOST_SOAK_BIG") else [65536, 131072, 1 << 20, 16 << 20])))
          env["GHF_IO_THREADS"] = str(int(rng.choice([1, 2, 5, 12])))
          env["GHF_RESIDENT_BYTES"] = str(int(rng.choice([0, 1 << 40])))
          sink = str(rng.choice(["mmap", "pwrite", ""]))
          if sink: env["GHF_SINK"] = sink
          else: env.pop("GHF_SINK", None)
          tag = "n=%d piece=%s thr=%s res=%s sink=%s" % (data.size, env["GHF_PIECE_BYTES"], env["GHF_IO_THREADS"], env["GHF_RESIDENT_BYTES"], sink or "auto")
          r = subprocess.run([TOOL, f, "3"], capture_output=True, text=True, env=env, timeout=120)
          assert r.returncode == 0, (tag, r.stderr[-300:])
          ref = orc.compress(data)
          got = np.fromfile(f + ".crs2", dtype=np.uint8)
          assert got.size == ref.size and hashlib.sha256(got).digest() == hashlib.sha256(ref).digest(), ("compress differs", tag)
          dm = str(rng.choice(["4", "5", "6"]))
          if os.path.exists(f + ".crs2.de"): os.remove(f + ".crs2.de")
          r = subprocess.run([TOOL, f + ".crs2", dm], capture_output=True, text=True, env=env, timeout=120)
          assert r.returncode == 0, (tag, r.stderr[-300:])
          back = np.fromfile(f + ".crs2.de", dtype=np.uint8)
          assert back.size == data.size and np.array_equal(back, data), ("round trip differs", tag)
          if len(set(data.tolist())) >= 2 and rng.random() < 0.5:  # the .crs pair on the same file (needs two distinct byte values)
              r = subprocess.run([TOOL, f, "1"], capture_output=True, text=True, env=env, timeout=120)
              assert r.returncode == 0, ("crs", tag, r.stderr[-300:])
              ref = orc.crs_compress(data)
              got = np.fromfile(f + ".crs", dtype=np.uint8)
              assert got.size == ref.size and hashlib.sha256(got).digest() == hashlib.sha256(ref).digest(), ("crs compress differs", tag)
              if os.path.exists(f + ".crs.de"): os.remove(f + ".crs.de")
              r = subprocess.run([TOOL, f + ".crs", "2"], capture_output=True, text=True, env=env, timeout=120)
              assert r.returncode == 0, ("crs", tag, r.stderr[-300:])
              back = np.fromfile(f + ".crs.de", dtype=np.uint8)
              assert back.size == data.size and np.array_equal(back, data), ("crs round trip differs", tag)
          cases += 1
          if cases % 20 == 0: print("cases", cases, int(time.time() - t0), "s", flush=True)
  except AssertionError:
    dump_on_failure(f, tag)
    raise
print("host soak ok: %d cases" % cases)
