"""-m gpu: bench.py's own N > 1 path, end to end, the way the driver launches it (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`),
on the one GPU the test box has: two ranks share cuda:0 and talk through gloo (--dist-backend gloo; with nccl = RCCL the driver gives every rank its own
GPU). Both modes: independent images per rank (the default, weak scaling, no data-path collective) and one image sharded over the ranks (--shard-image:
slabs / cluster shares / TSVQ node shares + all-gather / sum all-reduce). What is checked is the contract of the line rank 0 prints and that the sharded
run ends with the single-GPU result (the codebook sizes of the line; the state itself is held to the reference by tests/test_gpu_etc1s_sharded.py)."""
import json
import os
import pathlib
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent
COMMON = ["--steps", "2", "--warmup", "1", "--size", "512", "--no-cpu-baseline", "--no-pipelined", "--no-uastc", "--no-fast", "--no-big"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(n, extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if n == 1:
        cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", *COMMON, *extra]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               str(ROOT / "bench.py"), "--gpus", str(n), "--dist-backend", "gloo", *COMMON, *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def single():
    return _run(1, [])


def _both_modes(d, n, headline_sharded):
    """N > 1: the line verifies itself -- how many ranks the process group and the sharding communicator really connected -- and carries BOTH modes, the sharded one with
    its per-stage split (what DESIGN.md section 7's scaling estimate is made of)."""
    assert d["comm"]["world_size"] == n and d["comm"]["ranks_seen"] == n
    rep = d["both_modes"]["replicas (one image per GPU, no data-path collective)"]
    sh = d["both_modes"]["one image sharded over the ranks (--shard-image)"]
    assert rep["scaling"] == "weak" and sh["scaling"] == "strong" and rep["value"] > 0 and sh["value"] > 0
    assert bool(sh.get("is_the_headline")) == headline_sharded and bool(rep.get("is_the_headline")) == (not headline_sharded)
    head = sh if headline_sharded else rep
    assert head["value"] == d["value"] and head["ms_per_step"] == d["ms_per_step"]
    assert sh["communicator_ranks_seen"] == n and sh["communicator"]
    split = sh["host_wall_s_per_step"]
    assert split["codebook_builders"] > 0 and split["slab_and_cluster_share_stages"] > 0 and "refine_endpoint_clusterization" in split["by_stage"]
    assert abs(rep["value"] - n * 512 * 512 / 1e6 / (rep["ms_per_step"] / 1e3)) / rep["value"] < 0.01
    assert abs(sh["value"] - 512 * 512 / 1e6 / (sh["ms_per_step"] / 1e3)) / sh["value"] < 0.01


def _contract(d, n):
    assert d["n_gpus"] == n and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "Mpixels/s" and d["higher_is_better"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["kernel_symbol"].startswith("k_")


def test_single_gpu_line_has_no_multi_gpu_objects(single):
    assert single["comm"] is None and single["both_modes"] is None


def test_native_rccl_communicator_two_processes():
    """bu_rccl_comm_create over REAL RCCL between two processes with a GPU each: one all-reduce through the bu_comm view must see both ranks. Needs two devices: skips on
    the one-GPU boxes of the test pool, runs on the first node that has them (bench.py --gpus N then reports the same figure as comm.sharding_communicator_ranks_seen)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU: RCCL cannot connect two ranks here (a communicator needs one device per rank)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(ROOT / "bench.py"), "--gpus", "2", "--shard-image", *COMMON]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["comm"]["ranks_seen"] == 2 and d["comm"]["sharding_communicator_ranks_seen"] == 2 and d["comm"]["sharding_communicator"].startswith("bu_rccl")
    _both_modes(d, 2, True)


def test_weak_scaling_mode_two_ranks(single):
    d = _run(2, [])
    _contract(d, 2)
    _both_modes(d, 2, False)
    assert d["scaling"] == "weak" and "no collective" in d["config"]["parallelism"]
    # value = the pixels of BOTH ranks' images over the slower rank's time
    assert abs(d["value"] - 2 * 512 * 512 / 1e6 / (d["ms_per_step"] / 1e3)) / d["value"] < 0.01
    assert d["config"]["final_endpoint_clusters"] > 0


def test_sharded_image_mode_two_ranks(single):
    d = _run(2, ["--shard-image"])
    _contract(d, 2)
    _both_modes(d, 2, True)
    assert d["scaling"] == "strong" and "sharded" in d["config"]["parallelism"]
    assert abs(d["value"] - 512 * 512 / 1e6 / (d["ms_per_step"] / 1e3)) / d["value"] < 0.01
    # the same image as the single-GPU run (seed 1234): the sharded frontend must end with the same codebooks
    for k in ("final_endpoint_clusters", "final_selector_clusters", "max_endpoint_clusters", "max_selector_clusters"):
        assert d["config"][k] == single["config"][k], k
    assert d["psnr"] == single["psnr"]


def test_weak_scaling_mode_eight_ranks_share_the_gpu(single):
    """The driver's 8-GPU launch in its default mode, on the one GPU there is: eight processes, eight contexts and eight resident frontends side by side, the node's host
    cores divided between the ranks (bench.py sets BU_HOST_THREADS = cores / ranks, 2..8). No collective in the data path; rank 0's line carries all eight images."""
    d = _run(8, [])
    _contract(d, 8)
    _both_modes(d, 8, False)
    assert d["scaling"] == "weak" and "no collective" in d["config"]["parallelism"]
    assert abs(d["value"] - 8 * 512 * 512 / 1e6 / (d["ms_per_step"] / 1e3)) / d["value"] < 0.01
    cpus, per_rank = d["config"]["host_cpus"], d["config"]["host_threads_per_rank"]
    assert per_rank == max(2, min(8, cpus // 8)), (cpus, per_rank)
    for k in ("final_endpoint_clusters", "final_selector_clusters"):   # rank 0's image is the single-GPU run's image (seed 1234 + rank)
        assert d["config"][k] == single["config"][k], k


def test_sharded_image_mode_eight_ranks_share_the_gpu(single):
    """`bench.py --gpus 8 --shard-image` as the driver would launch it: one image over eight ranks (slabs of 16 block rows each), strong scaling, the single-GPU result."""
    d = _run(8, ["--shard-image"])
    _contract(d, 8)
    _both_modes(d, 8, True)
    assert d["scaling"] == "strong" and "sharded" in d["config"]["parallelism"]
    assert abs(d["value"] - 512 * 512 / 1e6 / (d["ms_per_step"] / 1e3)) / d["value"] < 0.01
    for k in ("final_endpoint_clusters", "final_selector_clusters", "max_endpoint_clusters", "max_selector_clusters"):
        assert d["config"][k] == single["config"][k], k
    assert d["psnr"] == single["psnr"]
