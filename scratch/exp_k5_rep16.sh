#!/bin/bash
# experiment: K5 with a 16x (instead of 32x) replicated code table -> 33 KiB of LDS per workgroup -> 4 workgroups = 32 waves per CU
set -e
R=$(cd $(dirname $0)/.. && pwd)
P=$R/golden-huffman_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -I$R/include"
mkdir -p $R/scratch/exp
build() {
  T=$(mktemp -d /tmp/ghf_ab.XXXX)
  cp $P/csrc/*.hip $P/csrc/*.h $T/
  [ -n "$2" ] && python3 - $T "$2" <<'XX'
import sys, re
t, mode = sys.argv[1], sys.argv[2]
s = open(t + "/ghf_emit.hip").read()
def rep(a, b, cnt=1):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b)
rep("constexpr int kEmitTabWords = 256 * 32;", "constexpr int kEmitTabWords = 256 * 16;")
rep("e[j] = tab[(b << 5) | r];", "e[j] = tab[(b << 4) | r];")
rep("items_narrow(tab, (uint32_t)lane & 31u, v, cnt, q, l);", "items_narrow(tab, (uint32_t)lane & 15u, v, cnt, q, l);")
rep("const uint32_t e = tab[(byte << 5) | ((uint32_t)lane & 31u)];", "const uint32_t e = tab[(byte << 4) | ((uint32_t)lane & 15u)];")
rep("uint64_t e0 = tab[(b0 << 4) | r], e1 = tab[(b1 << 4) | r];", "uint64_t e0 = tab[(b0 << 3) | r], e1 = tab[(b1 << 3) | r];")
rep("items_wide(reinterpret_cast<const uint64_t*>(tab), (uint32_t)lane & 15u, v, cnt, q, l);", "items_wide(reinterpret_cast<const uint64_t*>(tab), (uint32_t)lane & 7u, v, cnt, q, l);")
rep("const uint64_t e = reinterpret_cast<const uint64_t*>(tab)[(byte << 4) | ((uint32_t)lane & 15u)];", "const uint64_t e = reinterpret_cast<const uint64_t*>(tab)[(byte << 3) | ((uint32_t)lane & 7u)];")
rep("const int s = tid >> 1;\n    const uint32_t code", "const int s = tid & 255;\n    const uint32_t code")
rep("uint4* dst = reinterpret_cast<uint4*>(tab) + (s * 8 + (tid & 1) * 4);", "uint4* dst = reinterpret_cast<uint4*>(tab) + s * 4;")
rep("__launch_bounds__(kEmitThreads, 6)", "__launch_bounds__(kEmitThreads, 8)")
open(t + "/ghf_emit.hip", "w").write(s)
h = open(t + "/ghf_internal.h").read()
assert "constexpr uint32_t kEmitSlots = 256 * 3 * 8;" in h
h = h.replace("constexpr uint32_t kEmitSlots = 256 * 3 * 8;", "constexpr uint32_t kEmitSlots = 256 * 4 * 8;")
open(t + "/ghf_internal.h", "w").write(h)
XX
  /opt/rocm/bin/hipcc $FLAGS -I$T -shared -o $R/scratch/exp/libghf_$1.so $T/ghf_kernels.hip $T/ghf_emit.hip $T/ghf_decode.hip $T/ghf_api.hip $T/ghf_comm.hip -ldl -Rpass-analysis=kernel-resource-usage 2> $T/usage.txt || { tail -20 $T/usage.txt; exit 1; }
  grep -A14 "k_emit" $T/usage.txt | grep -E "VGPRs:|Spill|LDS Size|Occupancy" | head -6
  rm -rf $T
  echo built $1
}
rm -f $R/scratch/exp/libghf_*.so
build base ""
build rep16 yes
ls -la $R/scratch/exp/
