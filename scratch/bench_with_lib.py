#!/usr/bin/env python3
"""bench.py with another build of the library (scratch/exp/libghf_<name>.so from scratch/ablate_r3.sh): A/B runs of the
PIPELINED bench.  usage: python scratch/bench_with_lib.py <name> [bench.py arguments]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pkgload
ghf = pkgload.load().ghf
ghf.LIB_PATH = os.path.join(ROOT, "scratch", "exp", "libghf_%s.so" % sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench
bench.main()
