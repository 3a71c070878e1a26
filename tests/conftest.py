import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The package under its import alias; on a fresh checkout (built artefacts are git-ignored) the native library
    and the oracle are compiled first -- hipcc cross-compiles without a GPU, ~25 s."""
    lib = ROOT / "comfyui-video-stabilizer_amd" / "lib" / "libvstab.so"
    if not lib.exists():
        graft.build()
    return graft.load_package()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o

    o.build()
    return o


@pytest.fixture(scope="session")
def ctx(pkg):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vstab_amd import native

    return native.default_context()
