#!/usr/bin/env python3
"""Is K5 bound by instruction issue?  A copy of the kernel sources with N dummy VALU instructions (4-byte v_add_u32 on a
register of their own) added to every tile of k_emit's main loop -> scratch/exp/libghf_k5pad<N>.so.  If the launch gets
slower in proportion, every instruction saved is time saved; if not, something else paces it.
    python scratch/k5_pad_build.py 16 32
"""
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "golden-huffman_amd", "csrc")
for n in [int(x) for x in sys.argv[1:]]:
    td = tempfile.mkdtemp(prefix="ghf_k5pad_")
    for f in os.listdir(SRC):
        shutil.copy(os.path.join(SRC, f), td)
    p = os.path.join(td, "ghf_emit.hip")
    s = open(p).read()
    old = "            combine_narrow(e, 16, q, l);\n"
    assert s.count(old) == 1
    pad = "            { uint32_t dummy = (uint32_t)lane; " + " ".join(['asm volatile("v_add_u32 %0, %0, %0" : "+v"(dummy));'] * n) + " asm volatile(\"\" :: \"v\"(dummy)); }\n"
    s = s.replace(old, old + pad)
    open(p, "w").write(s)
    out = os.path.join(ROOT, "scratch", "exp", "libghf_k5pad%d.so" % n)
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
             "-I" + os.path.join(ROOT, "include"), "-I" + td]
    srcs = [os.path.join(td, m + ".hip") for m in ("ghf_kernels", "ghf_emit", "ghf_decode", "ghf_api", "ghf_comm")]
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-shared", "-o", out] + srcs + ["-ldl"], check=True, stderr=subprocess.DEVNULL)
    shutil.rmtree(td)
    print("built", out)
