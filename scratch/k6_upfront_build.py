#!/usr/bin/env python3
"""Round 3's unexplained build, rebuilt: k_sync_table's class walk for L = 4 reading BOTH candidate end-mark
masks together with the F masks (k6_jump<4, true>) instead of the one mask behind the jump.  Two arms:
  upfront          with the may_alias mask pointers the tree has had since efa3f44
  upfront_noalias  with plain uint64_t* (what the failing build of round 3 had)
-> scratch/exp/libghf_k6_<arm>.so + the ISA of k_sync_table beside it.  scratch/k6_upfront_run.py tests them."""
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "golden-huffman_amd", "csrc")
OUT = os.path.join(ROOT, "scratch", "exp")
os.makedirs(OUT, exist_ok=True)
flags0 = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-I" + os.path.join(ROOT, "include")]
for arm in ("shipped", "upfront", "upfront_noalias"):
    td = tempfile.mkdtemp(prefix="ghf_k6_")
    for f in os.listdir(SRC):
        shutil.copy(os.path.join(SRC, f), td)
    p = os.path.join(td, "ghf_decode.hip")
    s = open(p).read()
    if arm != "shipped":
        a = "const uint32_t sh = k6_jump<CLEN, CLEN == 8>("
        assert s.count(a) == 1
        s = s.replace(a, "const uint32_t sh = k6_jump<CLEN, true>(")
        b = "            if (CLEN == 4) eof = found && ((Gl[(q & 7u) * 64] << ((q >> 3) & 63u)) >> 63) != 0;\n"
        assert s.count(b) == 1
        s = s.replace(b, "")
    if arm == "upfront_noalias":
        c = "typedef uint64_t __attribute__((may_alias)) k6_mask_t;"
        assert s.count(c) == 1
        s = s.replace(c, "typedef uint64_t k6_mask_t;")
    open(p, "w").write(s)
    flags = flags0 + ["-I" + td]
    lib = os.path.join(ROOT, "scratch", "exp", "libghf_k6_%s.so" % arm)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    srcs = [os.path.join(td, n + ".hip") for n in ("ghf_kernels", "ghf_emit", "ghf_decode", "ghf_api", "ghf_comm")]
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-shared", "-o", lib] + srcs + ["-ldl"], check=True, stderr=subprocess.DEVNULL)
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", "-o", os.path.join(td, "dec.s"), p], check=True, stderr=subprocess.DEVNULL)
    txt = open(os.path.join(td, "dec.s")).read()
    sym = "_ZN3ghf12k_sync_tableENS_10SyncParamsEjPhPj"
    body = txt[txt.index(sym + ":"):]
    body = body[:body.index(".Lfunc_end")]
    open(os.path.join(OUT, "k_sync_table_%s.s" % arm), "w").write(body)
    i = txt.index(".name:           " + sym)
    meta = [l.strip() for l in txt[i:i + 900].splitlines() if "spill" in l or "vgpr_count" in l or "private_segment" in l]
    print(arm, meta, "scratch ops:", body.count("scratch_"))
    shutil.rmtree(td)
