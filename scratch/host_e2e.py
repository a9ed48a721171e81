# GPU box: file-to-file timing of the C++ host layer (PCIe- and stdio-inclusive), 256 MiB uniform in /dev/shm
import os, subprocess, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import datagen as dg
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tool = os.path.join(root, "golden-huffman_amd", "host", "bin", "ghf_tool")
d = "/dev/shm/ghf_e2e"
os.makedirs(d, exist_ok=True)
f = os.path.join(d, "u256.bin")
dg.make("uniform", 1 << 28, seed=3).tofile(f)
for rep in range(2):
    r = subprocess.run([tool, f], capture_output=True, text=True)
    print(r.stdout)
for x in os.listdir(d):
    os.remove(os.path.join(d, x))
os.rmdir(d)
