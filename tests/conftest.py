import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def _bind_emu():
    import build_emu
    from leco_amd import hip
    hip._use_library(build_emu.build())


def _bind_hip():
    from leco_amd import build, hip
    build.build()          # incremental, content-hashed: never test a stale .so
    hip._use_library(hip.LIB_PATH)
    hip._lib_path = hip.LIB_PATH


@pytest.fixture(params=[pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)])
def dev(request):
    """Device the kernels run on: the host emulator of the kernel sources (CPU tier) or the
    real gfx950 build (GPU tier, marked `gpu`)."""
    if request.param == "emu":
        _bind_emu()
        return torch.device("cpu")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _bind_hip()
    return torch.device("cuda:0")


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
