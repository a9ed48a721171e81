"""one traced run of ghf_tool mode 7 on a file in /dev/shm: python scratch/file_trace.py <GiB> <kind>"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__)))
import file_perf as fp
gib, kind = float(sys.argv[1]), sys.argv[2]
with tempfile.TemporaryDirectory(dir="/dev/shm", prefix="ghf_") as d:
    f = os.path.join(d, kind + ".bin")
    fp.make(kind, int(gib * (1 << 30)) + 12345).tofile(f)
    e = dict(os.environ); e["GHF_PIPE_TRACE"] = "1"
    r = subprocess.run([fp.TOOL, f, "7"], capture_output=True, text=True, env=e, timeout=600)
    print(r.stdout[-700:]); print(r.stderr[-6000:])
