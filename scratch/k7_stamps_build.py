#!/usr/bin/env python3
"""Build-container side of the K7 timeline diagnostic: a textual patch of a COPY of the kernel sources that stamps every hot
pass of k_decode with s_memtime -- (a) wait for the span + copy into LDS + request for the next span, (b) the 64 lookups,
(c) copy-out, (d) the descriptor / side-car / ticket work behind it -- and leaves eight words per wave behind the decoded
output (the caller allocates 2 MiB of slack).  -> scratch/exp/libghf_stamps.so.  The product sources carry no such switch.
scratch/k7_stamps_run.py is the GPU side.

    python scratch/k7_stamps_build.py
"""
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "golden-huffman_amd", "csrc")
td = tempfile.mkdtemp(prefix="ghf_stamps_")
for f in os.listdir(SRC):
    shutil.copy(os.path.join(SRC, f), td)
p = os.path.join(td, "ghf_decode.hip")
s = open(p).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new)


rep("    DecGroup cur, nxt;\n    uint4 R[kDecVec];",
    "    uint64_t st_a = 0, st_b = 0, st_c = 0, st_d = 0, st_n = 0; const uint64_t st_begin = __builtin_amdgcn_s_memtime();\n"
    "    DecGroup cur, nxt;\n    uint4 R[kDecVec];")
rep("      constexpr bool HOT = decltype(hot_tag)::value;",
    "      constexpr bool HOT = decltype(hot_tag)::value;\n      const uint64_t st0 = __builtin_amdgcn_s_memtime();")
rep("      // ---- 3. decode\n", "      const uint64_t st1 = __builtin_amdgcn_s_memtime();\n      // ---- 3. decode\n")
rep("        // copy-out through the input tile (dead now)",
    "        const uint64_t st2 = __builtin_amdgcn_s_memtime();\n        // copy-out through the input tile (dead now)")
rep("        if (used != cur.expect) acc |= kEntNone;\n        bad_acc |= acc;",
    "        if (used != cur.expect) acc |= kEntNone;\n        bad_acc |= acc;\n        const uint64_t st3 = __builtin_amdgcn_s_memtime();\n"
    "        st_a += st1 - st0; st_b += st2 - st1; st_c += st3 - st2; st_n += 1; st_d -= st3;")
rep("      tk = claim_issue(1);\n    };", "      tk = claim_issue(1);\n      if (HOT) st_d += __builtin_amdgcn_s_memtime();\n    };")
rep("    if (bad_acc & (kEntEnd | kEntNone)) latch_status_here",
    "    if (lane == 0) { uint64_t* dbg = reinterpret_cast<uint64_t*>(P.out + ((P.n_symbols + 255) & ~255ull)) + (uint64_t)wid * 8;"
    " dbg[0] = st_a; dbg[1] = st_b; dbg[2] = st_c; dbg[3] = st_d; dbg[4] = st_n; dbg[5] = __builtin_amdgcn_s_memtime() - st_begin;"
    " dbg[6] = st_begin; dbg[7] = __builtin_amdgcn_s_memtime(); }\n"
    "    if (bad_acc & (kEntEnd | kEntNone)) latch_status_here")
open(p, "w").write(s)
out = os.path.join(ROOT, "scratch", "exp", "libghf_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
         "-I" + os.path.join(ROOT, "include"), "-I" + td]
srcs = [os.path.join(td, n + ".hip") for n in ("ghf_kernels", "ghf_emit", "ghf_decode", "ghf_api", "ghf_comm")]
subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-shared", "-o", out] + srcs + ["-ldl"], check=True)
r = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", "-o", os.path.join(td, "dec.s"), p], check=True)
for line in open(os.path.join(td, "dec.s")):
    if "k_decodeENS_9DecParamsE" in line and ".name:" in line:
        hit = True
    if "vgpr_spill_count" in line or "private_segment_fixed_size:" in line:
        pass
txt = open(os.path.join(td, "dec.s")).read()
i = txt.index(".name:           _ZN3ghf8k_decodeENS_9DecParamsE")
print([l.strip() for l in txt[i:i + 900].splitlines() if "spill" in l or "vgpr_count" in l or "private_segment" in l])
shutil.rmtree(td)
print("built", out)
