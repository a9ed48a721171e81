#!/bin/bash
# scratch/pipe_trace.sh [MiB] -- rocprofv3 kernel + memory-copy trace of ONE file-to-file compress and ONE decompress
# through the C++ host layer, then scratch/pipe_overlap.py -> gpurun_out/pipe/summary.txt
# (run on the GPU box: /usr/local/graft/bin/gpurun -- 'bash scratch/pipe_trace.sh 1024')
set -e
MIB=${1:-1024}
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/pipe
mkdir -p "$OUT"
F=/dev/shm/ghf_pipe_trace.bin
python - <<XX
import sys
sys.path.insert(0, "$ROOT/scratch")
import file_perf as fp
fp.make("zipf", $MIB << 20).tofile("$F")
XX
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/enc" -o enc -- "$ROOT/golden-huffman_amd/host/bin/ghf_tool" $F 3 > "$OUT/enc.log" 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/dec" -o dec -- "$ROOT/golden-huffman_amd/host/bin/ghf_tool" $F.crs2 4 > "$OUT/dec.log" 2>&1
cmp $F $F.crs2.de && echo "round trip ok" > "$OUT/roundtrip.txt"
rm -f $F $F.crs2 $F.crs2.de
cd "$ROOT"
python scratch/pipe_overlap.py "$OUT/enc" "$OUT/dec" $MIB > "$OUT/summary.txt"
cat "$OUT/summary.txt"
# the raw traces are large: keep the summary only
find "$OUT/enc" "$OUT/dec" -name '*.csv' -size +2M -delete
