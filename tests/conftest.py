import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    """the product package (ctypes binding of librt_mi355x.so)"""
    return importlib.import_module("raytracing-in-one-weekend_amd")


@pytest.fixture(scope="session")
def oracle():
    """the CPU oracle (test infrastructure); built on the CPU if the .so is absent"""
    import oracle_lib
    if not os.path.exists(oracle_lib.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return oracle_lib.Oracle()
