import os
import sys

import pytest

# the oracle's OpenMP team: bounded before anything loads libgomp (oracle/awfl_oracle.py: _bound_openmp_team has the story)
os.environ.setdefault("OMP_NUM_THREADS", "8")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """no test of this suite runs longer than two or three minutes on a healthy box: one that stops for 25 minutes fails on its own
    (pytest-timeout, when installed) instead of stalling the whole session"""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(1500))
