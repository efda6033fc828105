"""-m gpu: the REFERENCE command line tool with this repository's seam linked in (oracle/_ref/basisu_hip = untouched reference objects +
integration/basisu_hip_shim.cpp + libbasisu_hip.so): `-opencl` now selects the MI355X kernels through the reference's own
accelerator interface (encoder/basisu_opencl.h). The reference only promises its OpenCL path to be close to its CPU path; these kernels
follow the CPU code paths, so the file should be the CPU tool's file byte for byte."""
import subprocess
import tempfile
import pathlib

import numpy as np
import pytest

from helpers import ORACLE_DIR, have_ref_cli, synth, save_png

pytestmark = pytest.mark.gpu
SHIM_TOOL = ORACLE_DIR / "_ref" / "basisu_hip"


def _run(tool, png, *args):
    with tempfile.TemporaryDirectory() as d:
        save_png(pathlib.Path(d) / "in.png", png)
        r = subprocess.run([str(tool), "-basis", "-no_multithreading", *args, "in.png"], cwd=d, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return np.fromfile(pathlib.Path(d) / "in.basis", np.uint8), r.stdout + r.stderr


@pytest.mark.skipif(not (have_ref_cli() and SHIM_TOOL.exists()), reason="oracle/_ref/basisu_hip not present")
@pytest.mark.parametrize("level", [1, 2])
def test_reference_tool_with_hip_seam_writes_the_cpu_tools_file(level):
    img = synth(256, 192, 77)
    cpu, _ = _run(ORACLE_DIR / "_ref" / "basisu", img, "-etc1s", "-q", "128", "-comp_level", str(level))
    hip, log = _run(SHIM_TOOL, img, "-etc1s", "-q", "128", "-comp_level", str(level), "-opencl")
    assert "Using CPU" not in log and "failed" not in log.lower(), log[-1500:]
    assert hip.shape == cpu.shape and (hip == cpu).all()
