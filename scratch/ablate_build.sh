#!/bin/bash
# Build-container side of the K7/K5 ablations: textual patches of a COPY of the kernel sources -> scratch/exp/libghf_<name>.so
# (the product sources carry no experiment switches).  usage: scratch/ablate_build.sh
set -e
R=$(cd $(dirname $0)/.. && pwd)
P=$R/golden-huffman_amd
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -I$R/include -I$P/csrc"
mkdir -p $R/scratch/exp
build() {  # name, sed script for ghf_decode.hip, sed script for ghf_emit.hip
  T=$(mktemp -d /tmp/ghf_ab.XXXX)
  cp $P/csrc/*.hip $P/csrc/*.h $T/
  [ -n "$2" ] && sed -i -e "$2" $T/ghf_decode.hip
  [ -n "$3" ] && sed -i -e "$3" $T/ghf_emit.hip
  /opt/rocm/bin/hipcc $FLAGS -I$T -shared -o $R/scratch/exp/libghf_$1.so $T/ghf_kernels.hip $T/ghf_emit.hip $T/ghf_decode.hip $T/ghf_api.hip 2>&1 | grep -v "argument unused" || true
  rm -rf $T
  echo built $1
}
build base "" "" &
# K7: no table lookup (entry from the window bits: 8/9-bit code lengths like uniform data) -> how much is LDS lookup latency
build nolut 's|return \*reinterpret_cast<const uint32_t\*>(T.base + ((v >> T.lsh) << T.ash));|return (v >> 24) \| ((8u + (v >> 31)) << 8);|' "" &
# K7: refills do not read LDS
build norefill 's|    nextw = in_word(lin, la);  \\|    nextw = nextw * 2654435761u + la;  \\|' "" &
wait
# K7: no output stores (copy-out reads stay)
build nostore 's|\*reinterpret_cast<uint4\*>(og + r \* 1024) = \*reinterpret_cast<const uint4\*>(tile + sl \* 16 + piece \* 4);|{ const uint4 t_ = *reinterpret_cast<const uint4*>(tile + sl * 16 + piece * 4); if (t_.x == 0x12345678u \&\& t_.y == 0x9abcdef0u) *reinterpret_cast<uint4*>(og + r * 1024) = t_; }|' "" &
# K7: no decode at all (pipeline only): out = window words
build nodecode 's|        else if (VAR == 2) acc = dec_hot<3>(lin, la0, T1, cur.pos, out, used);|        else if (VAR == 2) { for (int d_ = 0; d_ < 16; ++d_) out[d_] = in_word(lin, la0 + 4u * (uint32_t)(ln * 16 + d_)); used = cur.expect; acc = 0; }|' "" &
# K5: deposits without LDS atomics (plain stores of W0 only) -> LDS conflict cost
build emit_nodep "" 's|  atomicOr(w + 1, alignbit(hi, lo, s));||; s|  atomicOr(w + 2, alignbit(lo, 0u, s));||' &
wait
ls -la $R/scratch/exp/
