"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import json
import os
import sys

import pytest

try:  # torch first: its bundled HIP runtime must be the one the process loads (a second copy, loaded after
    import torch  # noqa: F401  # libmwf_hip.so pulled in /opt/rocm's, finds no device)
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return [json.loads(line) for line in f if line.strip()]


def golden_inputs(v):
    """(target bytes, query bytes) of one golden vector."""
    if v["kind"] == "literal":
        return bytes.fromhex(v["t"]), bytes.fromhex(v["q"])
    from miniwfa_amd.synth import synth_pair
    t, q = synth_pair(v["seed"], v["tl"], v["p"], v.get("n_long", 0), v.get("long_max", 0))
    assert len(q) == v["ql"], "synthetic generator drifted from the one that made the fixtures"
    return t, q


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()
