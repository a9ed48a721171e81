#!/bin/bash
# usage: scratch/ablate.sh "<bench args>" exp...   (GPU box) -- decode/emit stage times per ablation build
ARGS=$1; shift
for e in base "$@"; do
  if [ $e = base ]; then unset GHF_LIB; else export GHF_LIB=$GRAFT_REPO_ROOT/scratch/exp$e/libghf.so; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-verify --steps 10 --warmup 3 $ARGS 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e', 'decode', j['stage_ms']['decode'], 'emit', j['stage_ms']['emit'], 'hist', j['stage_ms']['histogram'])
"
done
