#!/bin/bash
# scratch/k5_spill_repro.sh -- VERDICT r1 item 7: rebuild round 1's k_emit<PairMode> with its strided loop (the build
# that spilled 6 VGPRs and wrote wrong streams) from the round-1 commit, and list every vector-memory operation and
# every s_waitcnt of the kernel in program order -> stdout.  Build container only (hipcc cross-compiles; no GPU).
set -e
R=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d /tmp/k5spill.XXXX)
for f in ghf_kernels.hip ghf_internal.h; do git -C $R show 63a9840:golden-huffman_amd/csrc/$f > $T/$f; done
git -C $R show 63a9840:include/ghf.h > $T/ghf.h
python3 - $T <<'XX'
import sys
t = sys.argv[1]
s = open(t + "/ghf_kernels.hip").read()
a, b = s.index("struct PairMode {"), s.index("struct WideMode {")
s = s[:a] + s[a:b].replace("static constexpr bool kStrided = false;", "static constexpr bool kStrided = true;") + s[b:]
open(t + "/strided.hip", "w").write(s)
XX
FL="-O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -I$T --cuda-device-only -S"
/opt/rocm/bin/hipcc $FL $T/strided.hip -o $T/strided.s -Rpass-analysis=kernel-resource-usage 2> $T/usage.txt
grep -A12 "k_emitINS_8PairMode" $T/usage.txt | grep -E "Name|VGPRs:|Spill|Scratch" | sed 's/^.*remark: *//'
python3 - $T <<'XX'
import re, sys
s = open(sys.argv[1] + "/strided.s").read()
a = s.index("_ZN3ghf6k_emitINS_8PairModeEEEvNS_10EmitParamsE:")
body = s[a:s.index(".Lfunc_end", a)].split("\n")
print("; %d lines; vector memory operations, waits and labels in program order:" % len(body))
for i, l in enumerate(body):
    t = l.strip()
    if re.match(r"(scratch_|global_load|global_store|global_atomic|s_waitcnt vmcnt|s_waitcnt.*vmcnt|\.LBB\d+_\d+:)", t):
        print("%5d  %s" % (i, t))
XX
rm -rf $T
