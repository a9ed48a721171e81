#!/bin/bash
# experiment: non-temporal (streaming) stores for K7's output and/or K5's output -- does the next kernel's read still pay
# for the write-back of dirty Infinity-Cache lines?
set -e
R=$(cd $(dirname $0)/.. && pwd)
P=$R/golden-huffman_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -I$R/include"
mkdir -p $R/scratch/exp
rm -f $R/scratch/exp/libghf_*.so
build() {
  T=$(mktemp -d /tmp/ghf_ab.XXXX)
  cp $P/csrc/*.hip $P/csrc/*.h $T/
  python3 - $T "$2" <<'XX'
import sys
t, mode = sys.argv[1], sys.argv[2]
h = open(t + "/ghf_device.h").read()
h = h.replace("namespace ghf {", '''namespace ghf {
typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_nt(void* p, const uint4& v) {
  u32x4_nt x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<u32x4_nt*>(p));
}''', 1)
open(t + "/ghf_device.h", "w").write(h)
if "7" in mode:
    s = open(t + "/ghf_decode.hip").read()
    a = "*reinterpret_cast<uint4*>(og + r * 1024) = *reinterpret_cast<const uint4*>(tile + sl * 16 + piece * 4);"
    assert a in s
    s = s.replace(a, "store_nt(og + r * 1024, *reinterpret_cast<const uint4*>(tile + sl * 16 + piece * 4));")
    open(t + "/ghf_decode.hip", "w").write(s)
if "5" in mode:
    s = open(t + "/ghf_emit.hip").read()
    a = "W.out_units[W.unit_base + j] = v;"
    assert a in s
    s = s.replace(a, "store_nt(&W.out_units[W.unit_base + j], v);")
    open(t + "/ghf_emit.hip", "w").write(s)
XX
  /opt/rocm/bin/hipcc $FLAGS -I$T -shared -o $R/scratch/exp/libghf_$1.so $T/ghf_kernels.hip $T/ghf_emit.hip $T/ghf_decode.hip $T/ghf_api.hip $T/ghf_comm.hip -ldl 2>&1 | grep -v "argument unused" || true
  rm -rf $T
  echo built $1
}
build base "" &
build nt7 7 &
wait
build nt5 5 &
build nt57 57 &
wait
ls $R/scratch/exp
