"""import helper: the package directory is called `golden-huffman_amd` (not a Python identifier)."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    name = "golden_huffman_amd"
    if name in sys.modules:
        return sys.modules[name]
    d = os.path.join(_ROOT, "golden-huffman_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(d, "__init__.py"), submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod
