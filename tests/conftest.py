import os
import sys

import pytest
import torch  # noqa: F401  (before libwordpiece_amd.so: torch brings its own HIP runtime, and whichever
#               copy of libamdhip64 is loaded first serves the whole process — torch sees no GPU behind
#               the system copy; encode_tensor and bench.py need torch's GPU view)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
