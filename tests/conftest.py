import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _emu_path():
    from motionclone_amd import build
    return build.build_emu()


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    """'emu': kernel sources compiled for the host simulator (tests/hipemu), CPU tensors.
    'hip': the real gfx950 library on cuda:0.  Both go through the same C ABI wrappers."""
    from motionclone_amd import lib
    if request.param == "emu":
        lib.use_library_for_tests(_emu_path())
        yield torch.device("cpu")
        lib._lib = None
        lib._is_emulated = False
    else:
        assert torch.cuda.is_available(), "gpu-marked test on a box without a GPU"
        lib._lib = None
        lib._is_emulated = False
        lib.load()
        yield torch.device("cuda:0")


@pytest.fixture
def emu_device():
    from motionclone_amd import lib
    lib.use_library_for_tests(_emu_path())
    yield torch.device("cpu")
    lib._lib = None
    lib._is_emulated = False


@pytest.fixture
def gpu_device():
    from motionclone_amd import lib
    assert torch.cuda.is_available()
    lib._lib = None
    lib._is_emulated = False
    lib.load()
    return torch.device("cuda:0")
