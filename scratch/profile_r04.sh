#!/bin/bash
# scratch/profile_r04.sh <tag> [part] -- everything profiles/<tag>/ is made from, on the GPU box (one gpurun call):
#   bench JSON (default command), rocprofv3 kernel stats of the same command, PMC passes (HBM traffic, SQ/LDS) at
#   256 MiB and at 4 GiB uniform, the memory microbenchmark, the file-to-file rates (new output file / reused output file).
TAG=${1:-r04}
PART=${2:-all}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
if [ $PART = all ] || [ $PART = A ]; then
echo "== bench (default command)"; python3 $R/bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json; echo
echo "== kernel stats 256 MiB"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats256 -- python3 $R/bench.py --steps 100 --warmup 3 --no-cpu-baseline --no-configs > $O/stats256.log 2>&1 || echo "stats256 failed"
echo "== kernel stats 4 GiB"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats4096 -- python3 $R/bench.py --mib 4096 --steps 20 --warmup 2 --no-cpu-baseline --no-configs > $O/stats4096.log 2>&1 || echo "stats4096 failed"
for MIB in 256; do
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "$SQ1" "$SQ2"; do
    i=$((i+1))
    echo "== pmc $MIB MiB pass $i"
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc${MIB}_p$i -- python3 $R/bench.py --mib $MIB --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-configs > $O/pmc${MIB}_p$i.log 2>&1 || echo "pmc $MIB pass $i failed"
  done
done
fi
if [ $PART = all ] || [ $PART = B ]; then
for MIB in 4096; do
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "$SQ1" "$SQ2"; do
    i=$((i+1))
    echo "== pmc $MIB MiB pass $i"
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc${MIB}_p$i -- python3 $R/bench.py --mib $MIB --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-configs > $O/pmc${MIB}_p$i.log 2>&1 || echo "pmc $MIB pass $i failed"
  done
done
echo "== side-car-less decode (K6 + K7), per kernel"
for KIND in uniform zipf sym16; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/foreign_$KIND -- python3 $R/scratch/foreign_prof.py $KIND > $O/foreign_$KIND.log 2>&1 || echo "foreign $KIND failed"
done
cd $R
echo "== the other 256 MiB inputs"
for KIND in zipf sym16; do
  python3 bench.py --steps 100 --kind $KIND --no-configs --no-cpu-baseline > $O/bench_256MiB_$KIND.json 2> $O/bench_256MiB_$KIND.err
done
for KIND in uniform zipf; do
  python3 bench.py --mib 4096 --steps 40 --warmup 2 --kind $KIND --no-configs --no-cpu-baseline > $O/bench_4GiB_$KIND.json 2> $O/bench_4GiB_$KIND.err
done
fi
if [ $PART = all ] || [ $PART = C ]; then
cd $R
echo "== stream layouts"
GHF_BENCH_MAINS=1 python3 bench.py --steps 200 --no-configs --no-cpu-baseline > $O/bench_one_main_stream.json 2> $O/bench_one_main_stream.err
GHF_BENCH_NSIDE=2 python3 bench.py --steps 200 --no-configs --no-cpu-baseline > $O/bench_two_side_streams.json 2> $O/bench_two_side_streams.err
GHF_BENCH_NSIDE=1 python3 bench.py --steps 200 --no-configs --no-cpu-baseline > $O/bench_one_side_stream.json 2> $O/bench_one_side_stream.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $O/bench_20_steps.json 2> $O/bench_20_steps.err
echo "== memory microbenchmark"
timeout -k 10 300 ./scratch/membench > $O/membench.txt 2>&1
echo "== file to file"
timeout -k 10 600 python3 scratch/file_perf.py 4 uniform zipf > $O/file_perf.log 2>&1; cp gpurun_out/file_perf.json $O/file_perf.json
timeout -k 10 300 python3 scratch/file_trace_reuse.py uniform > $O/pipe_trace_reuse_uniform.log 2>&1
timeout -k 10 300 python3 scratch/file_trace_reuse.py zipf > $O/pipe_trace_reuse_zipf.log 2>&1
fi
# (no GPU process's stderr goes to /dev/null: every run leaves its .err beside its .json)
# raw per-dispatch traces are large; keep stats and counter files only
find $O -name '*kernel_trace.csv' -size +6M -delete
find $O -name '*.db' -delete
du -sh $O
