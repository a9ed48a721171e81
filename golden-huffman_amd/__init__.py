"""golden-huffman_amd -- MI355X-native canonical-Huffman hot path (drop-in for glzip's
Compressor<CanonicalHuffEncoder<>> / Decompressor<CanonicalHuffDecoder<>> path).

The product is lib/libghf.so (hand-written gfx950 HIP kernels behind the C ABI of include/ghf.h) and
the C++ host layer in host/.  This Python package is plumbing only: a ctypes binding (ghf.py) used by
the tests and bench.py, and the one-process-per-GPU sharded driver (sharded.py) over torch.distributed.

The directory name contains a '-', so import it through `load()` in pkgload.py at the repo root, or
    importlib.util.spec_from_file_location("golden_huffman_amd", ".../golden-huffman_amd/__init__.py")
"""
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.path.join(PKG_DIR, "lib", "libghf.so")


def build(verbose=False):
    """compile lib/libghf.so for gfx950 (hipcc cross-compiles without a GPU)."""
    import subprocess

    subprocess.run(["make", "-C", PKG_DIR] + ([] if verbose else ["-s"]), check=True)
    return LIB_PATH


from . import ghf  # noqa: E402,F401
