#!/bin/bash
# usage: scratch/ablate.sh "<bench args>" exp...   (GPU box) -- decode/emit stage times per ablation build
# build the variants first (build container), e.g. for e in 1 2 3 8 9:
#   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-atomic-optimizer-strategy=None -Iinclude \
#         -Igolden-huffman_amd/csrc -DGHF_EXP=$e -shared -o scratch/exp$e/libghf.so golden-huffman_amd/csrc/*.hip
ARGS=$1; shift
for e in base "$@"; do
  if [ $e = base ]; then unset GHF_LIB; else export GHF_LIB=$GRAFT_REPO_ROOT/scratch/exp$e/libghf.so; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-verify --steps 10 --warmup 3 $ARGS 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e', 'decode', j['stage_ms']['decode'], 'emit', j['stage_ms']['emit'], 'hist', j['stage_ms']['histogram'])
"
done
