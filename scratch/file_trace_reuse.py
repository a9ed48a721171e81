#!/usr/bin/env python3
"""GHF_PIPE_TRACE of one warm compress + decompress of a 4 GiB file with GHF_SINK=reuse (the output pages exist): where the
time goes when the sink is not the limit.  usage: python scratch/file_trace_reuse.py [kind]"""
import os, subprocess, sys, tempfile
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "scratch"))
import file_perf
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
n = (4 << 30) + 12345
with tempfile.TemporaryDirectory(dir="/dev/shm", prefix="ghf_") as d:
    f = os.path.join(d, kind + ".bin")
    file_perf.make(kind, n).tofile(f)
    env = dict(os.environ, GHF_SINK="reuse", GHF_PIPE_TRACE="1")
    if len(sys.argv) > 2:
        env["GHF_RESIDENT_BYTES"] = sys.argv[2]
    r = subprocess.run([file_perf.TOOL, f, "7"], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-6000:])
    print(r.stderr[-12000:])
