#!/bin/bash
# experiment: K1 reads its input with the non-temporal hint
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
if "1" in mode:
    s = open(t + "/ghf_kernels.hip").read()
    for a in ("uint4 A = *vptr(cur, 0), B = *vptr(cur, 1);", "const uint4 C = *vptr(cur, j + 2), D = *vptr(cur, j + 3);",
              "A = *vptr(more ? cur : nxt, more ? j + 4 : 0);", "B = *vptr(more ? cur : nxt, more ? j + 5 : 1);"):
        assert s.count(a) == 1, a
        s = s.replace(a, a.replace("*vptr(", "load_stream(vptr(").replace(");", "));").replace("), B = load_stream", ")), B = load_stream").replace("), D = load_stream", ")), D = load_stream"))
    open(t + "/ghf_kernels.hip", "w").write(s)
XX
  /opt/rocm/bin/hipcc $FLAGS -I$T -shared -o $R/scratch/exp/libghf_$1.so $T/ghf_kernels.hip $T/ghf_emit.hip $T/ghf_decode.hip $T/ghf_api.hip $T/ghf_comm.hip -ldl 2>&1 | grep -v "argument unused" || true
  rm -rf $T
  echo built $1
}
build base "" &
build k1nt 1 &
wait
ls $R/scratch/exp
