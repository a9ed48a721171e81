#!/usr/bin/env python3
"""GPU side of scratch/k6_upfront_build.py: the side-car-less decode tests of 4/5-bit and 8/9-bit codes with one of the rebuilt
libraries.   python scratch/k6_upfront_run.py <arm>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pkgload
pkg = pkgload.load()
pkg.ghf.LIB_PATH = os.path.join(ROOT, "scratch", "exp", "libghf_k6_%s.so" % sys.argv[1])
import pytest
sys.exit(pytest.main(["-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-k", "foreign or 4_and_5 or 8_and_9 or full_size", "-p", "no:cacheprovider"]))
