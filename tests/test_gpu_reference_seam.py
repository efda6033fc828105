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


RESIDENT_TOOL = ORACLE_DIR / "_ref" / "basisu_hip_resident"


def _run_any(tool, png, out_ext, *args):
    with tempfile.TemporaryDirectory() as d:
        save_png(pathlib.Path(d) / "in.png", png)
        r = subprocess.run([str(tool), "-no_multithreading", *args, "in.png"], cwd=d, capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return np.fromfile(pathlib.Path(d) / ("in." + out_ext), np.uint8), r.stdout + r.stderr


@pytest.mark.skipif(not (have_ref_cli() and RESIDENT_TOOL.exists()), reason="oracle/_ref/basisu_hip_resident not present")
@pytest.mark.parametrize("args,ext,alpha", [
    (("-basis", "-etc1s", "-q", "128"), "basis", False),
    (("-etc1s", "-q", "128"), "ktx2", False),                                   # the tool's default container
    (("-basis", "-etc1s", "-q", "200", "-comp_level", "2"), "basis", False),    # the backend calls back into the frontend
    (("-basis", "-etc1s", "-q", "90", "-comp_level", "4"), "basis", False),
    (("-basis", "-etc1s", "-q", "128", "-mipmap"), "basis", False),             # the reference's own mip generation, eight slices through one frontend
    (("-basis", "-etc1s", "-q", "128"), "basis", True),                         # colour + alpha slices
    (("-basis", "-etc1s", "-q", "128", "-linear"), "basis", False),
    # -validate_etc1s = basisu_frontend::params::m_validate (comp.cpp:3434): the resident frontend's own validate_output() over the state the device delivered
    (("-basis", "-etc1s", "-q", "128", "-comp_level", "2", "-validate_etc1s"), "basis", True),
])
def test_reference_tool_with_resident_frontend_and_backend_writes_the_cpu_tools_file(args, ext, alpha):
    """The reference's basis_compressor / containers / CLI (untouched objects) over integration/basisu_resident_frontend.cpp and
    basisu_resident_backend.cpp: the whole ETC1S path of basis_compressor::process() resident on the MI355X behind the reference's own
    classes. Same file as the stock tool, byte for byte."""
    img = synth(256, 192, 78)
    if alpha:
        yy, xx = np.mgrid[0:192, 0:256]
        img[..., 3] = np.clip(128 + 120 * np.sin(xx / 23.0 - yy / 17.0), 0, 255).astype(np.uint8)
    cpu, _ = _run_any(ORACLE_DIR / "_ref" / "basisu", img, ext, *args)
    res, log = _run_any(RESIDENT_TOOL, img, ext, *args)
    assert "failed" not in log.lower(), log[-1500:]
    assert res.shape == cpu.shape and (res == cpu).all()


@pytest.mark.skipif(not (have_ref_cli() and RESIDENT_TOOL.exists()), reason="oracle/_ref/basisu_hip_resident not present")
def test_reference_tool_default_threads_with_resident_frontend_writes_the_multithreaded_cpu_tools_file():
    """The tool as users run it -- NO -no_multithreading: from 262,144 distinct selector vectors up the stock tool partitions its selector codebook build
    T = min(hardware threads, 8) ways (frontend.cpp:2195-2204, enc.h:2086-2215) and writes a DIFFERENT file than under -no_multithreading. The resident
    integration derives the same T from the same params and reproduces that file."""
    import os
    if (os.cpu_count() or 1) < 2:
        pytest.skip("one hardware thread: the tool's default is the single-threaded configuration")
    img = synth(3072, 3072, 80)   # 589,824 blocks, ~380k distinct selector vectors

    def run(tool, *extra):
        with tempfile.TemporaryDirectory() as d:
            save_png(pathlib.Path(d) / "in.png", img)
            r = subprocess.run([str(tool), "-basis", "-etc1s", "-q", "128", *extra, "in.png"], cwd=d, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            return np.fromfile(pathlib.Path(d) / "in.basis", np.uint8), r.stdout + r.stderr

    cpu_mt, _ = run(ORACLE_DIR / "_ref" / "basisu")
    cpu_st, _ = run(ORACLE_DIR / "_ref" / "basisu", "-no_multithreading")
    assert cpu_mt.shape != cpu_st.shape or (cpu_mt != cpu_st).any(), "the image is too small to reach the reference's partitioned codebook build"
    res, log = run(RESIDENT_TOOL)
    assert "failed" not in log.lower(), log[-1500:]
    assert res.shape == cpu_mt.shape and (res == cpu_mt).all()


UASTC_TOOL = ORACLE_DIR / "_ref" / "basisu_hip_uastc"


@pytest.mark.skipif(not (have_ref_cli() and UASTC_TOOL.exists()), reason="oracle/_ref/basisu_hip_uastc not present")
@pytest.mark.parametrize("args,ext,alpha", [
    (("-basis", "-uastc"), "basis", False),                                      # level 2 (the tool's default)
    (("-basis", "-uastc", "-uastc_level", "1"), "basis", False),
    (("-basis", "-uastc", "-uastc_level", "3"), "basis", True),                  # alpha: the modes with an alpha plane
    (("-basis", "-uastc", "-uastc_rdo_l", "1.0"), "basis", False),               # BASELINE configs[4]'s settings: encode + uastc_rdo, one strip
    (("-uastc", "-uastc_rdo_l", "2.0", "-mipmap"), "ktx2", False),               # the tool's default container (Zstandard over the RDO'd blocks), eight slices
])
def test_reference_tool_with_the_uastc_hot_path_on_the_gpu_writes_the_cpu_tools_file(args, ext, alpha):
    """The reference's basis_compressor / containers / CLI (untouched objects; ONE symbol of basisu_comp.o weakened, oracle/Makefile) over
    integration/basisu_resident_uastc.cpp: encode_slices_to_uastc_4x4_ldr = bu_hip_encode_uastc_blocks (+ bu_hip_uastc_rdo) per slice on the MI355X
    instead of encode_uastc per block on the job pool. Same file as the stock tool, byte for byte."""
    img = synth(256, 192, 79)
    if alpha:
        yy, xx = np.mgrid[0:192, 0:256]
        img[..., 3] = np.clip(128 + 120 * np.sin(xx / 19.0 + yy / 29.0), 0, 255).astype(np.uint8)
    cpu, _ = _run_any(ORACLE_DIR / "_ref" / "basisu", img, ext, *args)
    gpu, log = _run_any(UASTC_TOOL, img, ext, *args, "-debug")
    assert "encode_slices_to_uastc_4x4_ldr (MI355X)" in log, "the stock function ran, not integration/basisu_resident_uastc.cpp"
    assert "failed" not in log.lower(), log[-1500:]
    assert gpu.shape == cpu.shape and (gpu == cpu).all()


@pytest.mark.skipif(not (have_ref_cli() and RESIDENT_TOOL.exists()), reason="oracle/_ref/basisu_hip_resident not present")
@pytest.mark.parametrize("args,alpha", [(("-basis", "-etc1s", "-q", "128"), False), (("-basis", "-etc1s", "-q", "200", "-comp_level", "2"), False),
                                        (("-basis", "-etc1s", "-q", "128", "-mipmap"), True)])
def test_resident_frontend_through_the_frontend_pipeline_writes_the_same_file(args, alpha):
    """BU_RESIDENT_LANES > 0: every basisu_frontend::compress() of the resident integration is submitted to one bu_frontend_pipeline per GPU and collected from it
    (integration/basisu_resident_frontend.cpp) -- the backend, its call back into the frontend (level 2) and the slices of a mip chain then work on a frontend the
    pipeline lent out. Same file as the direct form, which is the stock tool's."""
    import os
    img = synth(256, 192, 31)
    if alpha:
        img = img.copy(); img[..., 3] = (np.arange(256)[None, :] + np.arange(192)[:, None]).astype(np.uint8)
    direct, _ = _run_any(RESIDENT_TOOL, img, "basis", *args)
    old = os.environ.get("BU_RESIDENT_LANES")
    os.environ["BU_RESIDENT_LANES"] = "3"
    try:
        piped, log = _run_any(RESIDENT_TOOL, img, "basis", *args)
    finally:
        if old is None:
            os.environ.pop("BU_RESIDENT_LANES", None)
        else:
            os.environ["BU_RESIDENT_LANES"] = old
    assert "failed" not in log.lower(), log[-1500:]
    assert piped.shape == direct.shape and (piped == direct).all()
    stock, _ = _run_any(ORACLE_DIR / "_ref" / "basisu", img, "basis", *args)
    assert stock.shape == piped.shape and (stock == piped).all()
