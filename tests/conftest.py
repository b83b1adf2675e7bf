import importlib
import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- first, so the process holds ONE HIP runtime (torch's bundled libamdhip64.so.7);
#                              libadcensus_hip.so then binds to it by soname and can share torch's streams/memory

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
DATA = os.path.join(ROOT, "tests", "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_bgr(path):
    """PNG -> BGR u8[H][W][3], the channel order cv::imread hands the reference (main.cpp:120-131)."""
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


def load_gray(path):
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path)))


def middlebury2003(name):
    d = os.path.join(DATA, name)
    return load_bgr(os.path.join(d, "im2.png")), load_bgr(os.path.join(d, "im6.png")), load_gray(os.path.join(d, "disp2.png"))


def artroom1():
    d = os.path.join(DATA, "artroom1")
    return load_bgr(os.path.join(d, "im0.png")), load_bgr(os.path.join(d, "im1.png"))


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("stereo-matching_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("stereo-matching_amd.synth")


def fnv1a64(a):
    """64-bit hash of an array's bytes (vectorised multiply-xor fold; stable across platforms)."""
    b = np.frombuffer(np.ascontiguousarray(a).tobytes(), dtype=np.uint8)
    pad = (-len(b)) % 8
    if pad:
        b = np.concatenate([b, np.zeros(pad, np.uint8)])
    w = b.view(np.uint64)
    idx = np.arange(1, len(w) + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        mixed = (w ^ (idx * np.uint64(0x9E3779B97F4A7C15))) * np.uint64(0x100000001B3)
        mixed ^= mixed >> np.uint64(29)
        h = np.bitwise_xor.reduce(mixed) ^ np.uint64(len(b))
        s = np.add.reduce(mixed)
    return "%016x%016x" % (int(h), int(s))
