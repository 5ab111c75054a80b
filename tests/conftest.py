"""Shared fixtures.  `-m "not gpu"` runs anywhere (oracle, host logic, C-ABI surface);
`-m gpu` needs an MI355X and exercises the HIP engine through its C ABI."""
import os
import sys

import numpy as np
import pytest

# PyTorch wheels bundle their own copy of the HIP runtime (torch/lib/libamdhip64.so).  A process must use ONE HIP runtime:
# if liblpcnet_hip.so (linked against /opt/rocm) initialises the GPU first and torch is imported afterwards, torch loads
# its second copy and reports "No HIP GPUs are available".  Importing torch first makes both share torch's copy -- the
# same order bench.py uses.  (C programs such as the reference's lpcnet_demo never load torch and use /opt/rocm's.)
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lpcnet_amd import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run with -m gpu on an MI355X)")


@pytest.fixture(scope="session")
def blob_f32():
    return synth.blob_bytes(synth.make_model(flavour="float"))


@pytest.fixture(scope="session")
def blob_i8():
    return synth.blob_bytes(synth.make_model(flavour="int8"))


@pytest.fixture(scope="session")
def golden():
    if not os.path.exists(GOLDEN):
        pytest.skip("golden fixtures missing (tests/tools/make_golden.py)")
    return np.load(GOLDEN)


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import orc
    return orc.lib()


@pytest.fixture(scope="session")
def hip_lib():
    """The product library.  Built on demand (hipcc cross-compiles without a GPU)."""
    from lpcnet_amd import api, build
    build.build(verbose=False)
    return api.load_library()


def features(seed, n_frames):
    return synth.make_features(seed, n_frames)
