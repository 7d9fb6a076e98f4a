import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)                                  # `oracle` (test infra), bench helpers
sys.path.insert(0, os.path.join(ROOT, "rogue-gym_amd"))   # product: rogue_gym, rogue_gym_python


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` via gpurun)")


@pytest.fixture(scope="session")
def goldens():
    with open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")) as f:
        return json.load(f)
