#!/bin/bash
# Round 3's unexplained wrong-stream build, rebuilt from git history (results: profiles/r04/k6_upfront/README.md).
# k_sync_table's class walk for L = 4 reading BOTH candidate end-mark masks together with the F masks (k6_jump<4, true>) in the
# tree of commit 21507ac -- the build that spills 4 VGPRs and fails test_foreign_stream_16MiB[sym16] -- plus the arms that
# separate the hypotheses.  Build container:  scratch/k6_r3_repro.sh   then on the GPU box, per arm:
#   python scratch/k6_upfront_run.py <arm>        (arm = the part of the library's name behind libghf_k6_)
set -e
R=$(cd $(dirname $0)/.. && pwd)
T=/tmp/ghf_r3k6
rm -rf $T && mkdir -p $T && git -C $R archive 21507ac golden-huffman_amd/csrc include | tar -x -C $T
S=$T/golden-huffman_amd/csrc
python3 - $S <<'PY'
import re, sys
S = sys.argv[1]
ship = open(S + "/ghf_decode.hip").read()
a = "const uint32_t sh = k6_jump<CLEN, CLEN == 8>("
assert ship.count(a) == 1
up = ship.replace(a, "const uint32_t sh = k6_jump<CLEN, true>(")
up, n = re.subn(r"\n\s*if \(CLEN == 4\) eof = found && \(\(Gl\[\(q & 7u\) \* 64\] << \(\(q >> 3\) & 63u\)\) >> 63\) != 0;", "", up)
assert n == 1
open(S + "/dec_upfront.hip", "w").write(up)
# the SHIPPED logic under artificial register pressure: 42 VGPRs spilled, scratch traffic inside the class-walk loop
b = "        while (any) {\n          any = false;\n"
assert ship.count(b) == 1
regs = ",".join('"v%d"' % i for i in range(40, 128))
open(S + "/dec_shipped_pressure.hip", "w").write(ship.replace(b, b + '          asm volatile("" ::: %s);\n' % regs))
open(S + "/dec_shipped.hip", "w").write(ship)
PY
mkdir -p $R/scratch/exp $R/profiles/r04/k6_upfront
build() {  # arm, decode source, extra flags
  local F="-std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -I$T/include -I$S"
  /opt/rocm/bin/hipcc $3 $F -shared -o $R/scratch/exp/libghf_k6_$1.so $S/ghf_kernels.hip $S/ghf_emit.hip $S/$2 $S/ghf_api.hip $S/ghf_comm.hip -ldl 2>/dev/null
  /opt/rocm/bin/hipcc $3 $F -S --cuda-device-only -o $T/$1.s $S/$2 2>/dev/null
  echo "$1: $(grep -A22 '^    .name:           _ZN3ghf12k_sync_tableENS_10SyncParamsEjPhPj' $T/$1.s | grep 'spill\|vgpr_count\|private_segment' | tr -s ' ' | tr '\n' ' ')"
}
build r3tree_shipped dec_shipped.hip "-O3" &
build r3tree_upfront dec_upfront.hip "-O3" &
build r3tree_shipped_pressure dec_shipped_pressure.hip "-O3" &
build r3tree_upfront_nostrictalias dec_upfront.hip "-O3 -fno-strict-aliasing" &
wait
build r3tree_upfront_O2 dec_upfront.hip "-O2" &
build r3tree_upfront_nomisched dec_upfront.hip "-O3 -mllvm -enable-misched=0" &
build r3tree_upfront_regalloc_basic dec_upfront.hip "-O3 -mllvm -vgpr-regalloc=basic" &
build r3tree_upfront_noaa dec_upfront.hip "-O3 -mllvm -amdgpu-use-aa-in-codegen=0" &
wait
for a in r3tree_upfront r3tree_upfront_nomisched; do
  awk '/^_ZN3ghf12k_sync_tableENS_10SyncParamsEjPhPj:/{f=1} f{print} /^\.Lfunc_end.*k_sync_table/{f=0}' $T/$a.s > $R/profiles/r04/k6_upfront/k_sync_table_$a.s
done
