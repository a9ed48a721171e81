#!/bin/bash
# Build-container side of round 3's K5/K7 ablations: textual patches of a COPY of the kernel sources ->
# scratch/exp/libghf_<name>.so (the product sources carry no experiment switches).  usage: scratch/ablate_r3.sh [names...]
set -e
R=$(cd $(dirname $0)/.. && pwd)
P=$R/golden-huffman_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -I$R/include"
mkdir -p $R/scratch/exp
build() {  # name, sed script for ghf_decode.hip, sed script for ghf_emit.hip, [git rev to take ghf_emit.hip from], [sed for ghf_device.h]
  T=$(mktemp -d /tmp/ghf_ab.XXXX)
  cp $P/csrc/*.hip $P/csrc/*.h $T/
  [ -n "$4" ] && git -C $R show $4:golden-huffman_amd/csrc/ghf_emit.hip > $T/ghf_emit.hip
  [ -n "$2" ] && sed -i -e "$2" $T/ghf_decode.hip
  [ -n "$3" ] && sed -i -e "$3" $T/ghf_emit.hip
  /opt/rocm/bin/hipcc $FLAGS -I$T -shared -o $R/scratch/exp/libghf_$1.so $T/ghf_kernels.hip $T/ghf_emit.hip $T/ghf_decode.hip $T/ghf_api.hip $T/ghf_comm.hip -ldl 2>&1 | grep -v "argument unused" || true
  rm -rf $T
  echo built $1
}
want() { [ $# -eq 0 ] && return 0; for n in "$@"; do [ "$n" = "$NAME" ] && return 0; done; return 1; }
NAME=base;        want "$@" && build base "" "" &
NAME=emit_r02;    want "$@" && build emit_r02 "" "" 774c86c &
NAME=emit_nostore; want "$@" && build emit_nostore "" 's|      W.out_units\[W.unit_base + j\] = v;|      if (v.x == 0x12345678u \&\& v.y == 0x9abcdef0u) W.out_units[W.unit_base + j] = v;|' &
wait
NAME=emit_noload; want "$@" && build emit_noload "" 's|            buf\[j\] = pv\[nx \* 64\];|            buf[j] = make_uint4(e[0] * 2654435761u + (uint32_t)nx, e[5] + 7u, e[9] ^ e[3], e[13] + 3u);|' &
NAME=emit_nolds;  want "$@" && build emit_nolds "" 's|    if ((uint32_t)k < nw) w\[k\] = r\[k\];|    if ((uint32_t)k < nw \&\& r[k] == 0x12345678u) w[k] = r[k];|; s|  if (sh != 0u) atomicOr(w, r\[0\]);|  if (sh != 0u \&\& r[0] == 0x12345678u) atomicOr(w, r[0]);|' &
NAME=emit_noseg;  want "$@" && build emit_noseg "" 's|      \*seg_dst = seg_base + seg_end;|      if (seg_end == 0x12345678u) *seg_dst = seg_base + seg_end;|' &
wait
ls $R/scratch/exp/
