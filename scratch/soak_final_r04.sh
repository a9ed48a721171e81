#!/bin/bash
# long randomised soaks of the round's last library (seeds differ from the earlier calls')
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4soak
mkdir -p $O
cd $R
sha256sum golden-huffman_amd/lib/libghf.so > $O/lib.txt
timeout -k 10 330 python3 scratch/soak.py 270 6061 > $O/soak_cabi.log 2> $O/soak_cabi.err; rc=$?; echo "soak.py rc=$rc"; [ $rc -eq 0 ] || exit 1
timeout -k 10 260 python3 scratch/k6_soak.py 200 6062 > $O/soak_k6.log 2> $O/soak_k6.err; rc=$?; echo "k6_soak.py rc=$rc"; [ $rc -eq 0 ] || exit 1
HOST_SOAK_BIG=1 timeout -k 10 240 python3 scratch/host_soak.py 150 6063 > $O/soak_host_big.log 2> $O/soak_host_big.err; rc=$?; echo "host_soak.py (big) rc=$rc"; [ $rc -eq 0 ] || exit 1
tail -n 1 $O/soak_cabi.log; tail -n 1 $O/soak_k6.log | cut -c1-30; tail -n 1 $O/soak_host_big.log
