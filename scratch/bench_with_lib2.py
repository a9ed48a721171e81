#!/usr/bin/env python3
"""bench.py with another library: python scratch/bench_with_lib2.py <lib.so> [bench args]  (a process binds one libghf)"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pkgload
pkg = pkgload.load()
lib = sys.argv[1]
pkg.ghf.LIB_PATH = lib if os.path.isabs(lib) else os.path.join(ROOT, lib)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
