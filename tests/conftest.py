import os
import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the reference built from /root/reference)")


@pytest.fixture(scope="session")
def hip_ctx():
    """A live bu_hip_context; fails loudly (never falls back) when the HIP library or the GPU is missing."""
    from basis_universal_amd import capi
    ctx = capi.Context()
    yield ctx
    ctx.close()
