import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the compiled reference oracle/_ref/ref_glzip (build container only)")


@pytest.fixture(scope="session")
def golden():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)["cases"]


@pytest.fixture(scope="session")
def golden_crs():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "golden_crs.json")) as f:
        return json.load(f)["cases"]
