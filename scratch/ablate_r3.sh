#!/bin/bash
# Build-container side of round 3's ablations: textual patches of a COPY of the kernel sources ->
# scratch/exp/libghf_<name>.so (the product sources carry no experiment switches).  usage: scratch/ablate_r3.sh
set -e
R=$(cd $(dirname $0)/.. && pwd)
P=$R/golden-huffman_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -I$R/include"
rm -rf $R/scratch/exp; mkdir -p $R/scratch/exp
build() {  # name, sed for ghf_decode.hip, sed for ghf_emit.hip, sed for ghf_kernels.hip
  T=$(mktemp -d /tmp/ghf_ab.XXXX)
  cp $P/csrc/*.hip $P/csrc/*.h $T/
  [ -n "$2" ] && sed -i -e "$2" $T/ghf_decode.hip
  [ -n "$3" ] && sed -i -e "$3" $T/ghf_emit.hip
  [ -n "$4" ] && sed -i -e "$4" $T/ghf_kernels.hip
  /opt/rocm/bin/hipcc $FLAGS -I$T -shared -o $R/scratch/exp/libghf_$1.so $T/ghf_kernels.hip $T/ghf_emit.hip $T/ghf_decode.hip $T/ghf_api.hip $T/ghf_comm.hip -ldl 2>&1 | grep -v "argument unused" || true
  rm -rf $T
  echo built $1
}
# K7: what bounds it?  (scratch/ablate_run.py times k_decode alone.)  These five builds were made from the tree that had the
# byte-phase decoder in it (VAR 6, profiles/r03/experiments/k7_byte_phases_decoder.hip.txt; logs: k7_ablate_*.log beside it):
# base = byte phases, k7_generic = the shipped decoder, nodecode / nostore = the memory skeleton.  On the shipped tree only
# k7_nostore still applies.
build base "" "" "" &
build k7_generic 's|  const bool byte_code = P.dt->kind == 0|  const bool byte_code = false \&\& P.dt->kind == 0|' "" "" &
build k7_nodecode 's|          else if (VAR == 6) acc = dec_hot_bytes(lin, la0, T1, thr8, cur.pos, out, used);|          else if (VAR == 6) { for (int d = 0; d < 16; ++d) out[d] = cur.pos + d; used = cur.expect; acc = 0; }|' "" "" &
wait
build k7_nostore 's|            store_stream(og + r \* 1024, \*reinterpret_cast<const uint4\*>(tile + sl \* 16 + piece \* 4));|            if (P.n_symbols == 1) store_stream(og + r * 1024, *reinterpret_cast<const uint4*>(tile + sl * 16 + piece * 4));|' "" "" &
build k7_nodecode_nostore 's|          else if (VAR == 6) acc = dec_hot_bytes(lin, la0, T1, thr8, cur.pos, out, used);|          else if (VAR == 6) { for (int d = 0; d < 16; ++d) out[d] = cur.pos + d; used = cur.expect; acc = 0; }|;s|            store_stream(og + r \* 1024, \*reinterpret_cast<const uint4\*>(tile + sl \* 16 + piece \* 4));|            if (P.n_symbols == 1) store_stream(og + r * 1024, *reinterpret_cast<const uint4*>(tile + sl * 16 + piece * 4));|' "" "" &
wait
ls $R/scratch/exp/
