import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "zero-latency-yolo_amd")
for p in (PKG, os.path.join(PKG, "tools"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _make(*targets):
    subprocess.run(["make", "-C", ROOT, *targets], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on the CPU oracle (oracle/zly_oracle.c); built on demand with gcc."""
    import oracle_lib
    if not os.path.exists(oracle_lib.LIB_PATH):
        _make("oracle")
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def weights_path():
    p = os.path.join(PKG, "_build", "yolov8n_synth.zlyw")
    if not os.path.exists(p):
        _make("weights")
    return p


@pytest.fixture(scope="session")
def ref_fp32(weights_path):
    import yolov8_ref
    return yolov8_ref.load(weights_path, "fp32")


@pytest.fixture(scope="session")
def ref_bf16(weights_path):
    import yolov8_ref
    return yolov8_ref.load(weights_path, "bf16")
