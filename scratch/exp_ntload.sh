#!/bin/bash
# experiment: non-temporal LOADS for data read exactly once (K7's compressed span, K5's input tiles)
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
h = h.replace("__device__ __forceinline__ void store_stream(", '''__device__ __forceinline__ uint4 load_stream(const void* p) {
  const u32x4_stream x = __builtin_nontemporal_load(reinterpret_cast<const u32x4_stream*>(p));
  return make_uint4(x.x, x.y, x.z, x.w);
}
__device__ __forceinline__ void store_stream(''', 1)
open(t + "/ghf_device.h", "w").write(h)
if "7" in mode:
    s = open(t + "/ghf_decode.hip").read()
    a = "R[k] = *reinterpret_cast<const uint4*>(base + (o + 16u <= lim ? o : 0u));"
    assert s.count(a) == 1
    s = s.replace(a, "R[k] = load_stream(base + (o + 16u <= lim ? o : 0u));")
    open(t + "/ghf_decode.hip", "w").write(s)
if "5" in mode:
    s = open(t + "/ghf_emit.hip").read()
    for a, b in (("A = pv[nx * 64];", "A = load_stream(pv + nx * 64);"), ("B = pv[nx * 64];", "B = load_stream(pv + nx * 64);"),
                 ("A = pv[0];", "A = load_stream(pv);"), ("B = pv[64];", "B = load_stream(pv + 64);")):
        assert s.count(a) == 1, a
        s = s.replace(a, b)
    open(t + "/ghf_emit.hip", "w").write(s)
if "1" in mode:
    s = open(t + "/ghf_kernels.hip").read()
    a = "const uint4 v0 = pv[i];"
    print("K1 sites", s.count(a))
XX
  /opt/rocm/bin/hipcc $FLAGS -I$T -shared -o $R/scratch/exp/libghf_$1.so $T/ghf_kernels.hip $T/ghf_emit.hip $T/ghf_decode.hip $T/ghf_api.hip $T/ghf_comm.hip -ldl 2>&1 | grep -v "argument unused" || true
  rm -rf $T
  echo built $1
}
build base "" &
build ld7 7 &
wait
build ld5 5 &
build ld57 57 &
wait
ls $R/scratch/exp
