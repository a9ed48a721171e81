#!/bin/bash
# GPU box: run bench.py (256 MiB, no configs) with every scratch/exp/libghf_<name>.so in place of the product library
R=$GRAFT_REPO_ROOT
cp $R/golden-huffman_amd/lib/libghf.so /tmp/libghf_product.so
for f in $R/scratch/exp/libghf_*.so; do
  cp $f $R/golden-huffman_amd/lib/libghf.so
  echo "== $(basename $f)"
  python3 $R/bench.py --no-configs --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['stage_ms'], d.get('decode_foreign',{}).get('ms'))"
done
cp /tmp/libghf_product.so $R/golden-huffman_amd/lib/libghf.so
