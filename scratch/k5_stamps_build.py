#!/usr/bin/env python3
"""Build-container side of the K5 timeline diagnostic: a textual patch of a COPY of the kernel sources in which every wave of
k_emit leaves (table build cycles, chunk cycles) behind the packed stream (the last 256 KiB of the output buffer's capacity;
the runner allocates the slack).  -> scratch/exp/libghf_k5stamps.so.  scratch/k5_stamps_run.py is the GPU side.
"""
import os, shutil, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "golden-huffman_amd", "csrc")
td = tempfile.mkdtemp(prefix="ghf_k5stamps_")
for f in os.listdir(SRC):
    shutil.copy(os.path.join(SRC, f), td)
p = os.path.join(td, "ghf_emit.hip")
s = open(p).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new)


rep("  __shared__ int status0;\n  const int tid = threadIdx.x;\n  EmitGeom G;\n  if (!emit_begin(P, G, &status0, false)) {",
    "  __shared__ int status0;\n  const uint64_t st_k = __builtin_amdgcn_s_memtime();\n  const int tid = threadIdx.x;\n  EmitGeom G;\n  if (!emit_begin(P, G, &status0, false)) {")
rep("  if (G.max_len <= 9) emit_chunk<1>(P, G, c, tab, stage, st, lane);\n  else if (G.max_len <= 12) emit_chunk<2>(P, G, c, tab, stage, st, lane);\n"
    "  else if (!wide) emit_chunk<3>(P, G, c, tab, stage, st, lane);\n  else emit_chunk<0>(P, G, c, tab, stage, st, lane);\n}",
    "  const uint64_t st_0 = __builtin_amdgcn_s_memtime();\n"
    "  if (G.max_len <= 9) emit_chunk<1>(P, G, c, tab, stage, st, lane);\n  else if (G.max_len <= 12) emit_chunk<2>(P, G, c, tab, stage, st, lane);\n"
    "  else if (!wide) emit_chunk<3>(P, G, c, tab, stage, st, lane);\n  else emit_chunk<0>(P, G, c, tab, stage, st, lane);\n"
    "  const uint64_t st_1 = __builtin_amdgcn_s_memtime();\n"
    "  if (lane == 0 && c < 8192u) { uint64_t* dbg = reinterpret_cast<uint64_t*>(P.out + ((P.cap - (256u << 10)) & ~15ull)) + (uint64_t)c * 4;"
    " dbg[0] = st_0 - st_k; dbg[1] = st_1 - st_0; dbg[2] = st_k; dbg[3] = blockIdx.x; }\n}")
open(p, "w").write(s)
out = os.path.join(ROOT, "scratch", "exp", "libghf_k5stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
         "-I" + os.path.join(ROOT, "include"), "-I" + td]
srcs = [os.path.join(td, n + ".hip") for n in ("ghf_kernels", "ghf_emit", "ghf_decode", "ghf_api", "ghf_comm")]
subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-shared", "-o", out] + srcs + ["-ldl"], check=True)
shutil.rmtree(td)
print("built", out)
