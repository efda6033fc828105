#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: ETC1S encoder hot path, Mpixels/s, 4096x4096 synthetic RGBA, -q128 (CLI comp level 1).

A "step" is one full pass of the hot path over one image whose 4x4 tiles are already resident in HBM:
basisu_frontend::init + compress() (per-block ETC1S fit, endpoint TSVQ + codebook + refinement, selector
TSVQ + codebook + assignment) -- nothing is skipped or cached between steps. With --gpus N every rank encodes its own image
(independent objects, like the reference's basis_parallel_compress): weak scaling, no data-path collective.

Prints ONE JSON line (rank 0). Extra objects: "roofline" for the dominant kernel (HIP events on the launch stream, live),
"cpu_baseline" (the real reference frontend from oracle/_ref when present, else the C oracle, on a bounded sample),
"stages" (host wall seconds per frontend stage) and "kernels" (device ms per kernel).
"""
import argparse
import json
import os

# Streams of one process share a small number of hardware queues (ROCm's default: 4), and two streams on one queue run their kernels one after the other. The UASTC
# pipeline's lanes (bu_hip_uastc_pipeline_*) and the images-in-flight modes below want their streams on queues of their own; the runtime reads this once, when it starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md

# Algorithmic bytes per 4x4 block of each kernel (one 64 B tile read + the stage's side traffic)
KERNEL_BYTES_PER_BLOCK = {
    "encode_etc1s_blocks": 64 + 8,
    "generate_endpoint_codebook": 64 + 2 * 4,          # tile + two training-vector indices
    "refine_endpoint_clusterization": 64 + 7 + 4,
    "determine_selectors": 64 + 4 + 8,
    "create_optimized_selector_codebook": 64 + 8 + 4,
    "find_optimal_selector_clusters": 64 + 12 + 4 + 8,
    "selector_training_vectors": 8 + 8,
    "endpoint_training_vectors": 8 + 24,
}


# profile label -> the one kernel it times (csrc/etc1s_kernels.hip); every other label of a step spans several launches
KERNEL_SYMBOL = {
    "encode_etc1s_blocks": "k_encode_etc1s_blocks_by_pixel",
    "generate_endpoint_codebook": "k_generate_endpoint_codebook",
    "refine_endpoint_clusterization": "k_refine_sort_lists + k_refine_sorted",
    "determine_selectors": "k_determine_selectors",
    "create_optimized_selector_codebook": "k_cosc_accumulate + k_cosc_select",
    "find_optimal_selector_clusters": "k_find_optimal_selector_clusters",
    "selector_training_vectors": "k_selector_training_vectors",
    "endpoint_training_vectors": "k_endpoint_training_vectors",
}


def host_cpus():
    """CPUs this process may really use: the cgroup quota if there is one, else the affinity mask."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    return len(os.sched_getaffinity(0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--quality", type=int, default=128)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard-image", action="store_true", help="N > 1: all ranks encode the SAME image with the slab/cluster-sharded frontend "
                    "(RCCL all-gather / sum-merge between stages, strong scaling) instead of one image per rank")
    ap.add_argument("--streams", type=int, default=1, help="images in flight per GPU (one host thread + context + HIP stream each, like the reference's "
                    "basis_parallel_compress); the default 1 is what the headline number uses")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the additional 3-images-in-flight throughput measurement")
    ap.add_argument("--backend-in-flight", type=int, default=12, help="images in flight of the frontend + host backend throughput measurement (one host thread + context each)")
    ap.add_argument("--no-uastc", action="store_true", help="skip the secondary UASTC level-2 measurement (BASELINE config #3)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend of the N > 1 launch: nccl (= RCCL over xGMI, one GPU per "
                    "rank: what the driver runs) or gloo (ranks may share a GPU: the path tests/test_gpu_bench_distributed.py drives on a one-GPU box)")
    ap.add_argument("--no-big", action="store_true", help="skip the secondary 8192x8192 -q255 measurement (BASELINE config #4's single-GPU form)")
    ap.add_argument("--no-fast", action="store_true", help="skip the secondary measurement of the codebook builders' fast mode (SURVEY 8f row f3, not bit-identical)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import helpers
    from basis_universal_amd import capi
    from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and "BU_HOST_THREADS" not in os.environ:
        # the ranks of one node share its host cores: give each frontend its share (2..8 threads) instead of 8 each
        os.environ["BU_HOST_THREADS"] = str(max(2, min(8, host_cpus() // int(os.environ.get("LOCAL_WORLD_SIZE", world)))))
    gloo = args.dist_backend == "gloo"
    if gloo:
        local_rank %= max(1, torch.cuda.device_count())   # ranks share the GPUs there are
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = torch.device("cpu") if gloo else dev   # where the max-over-ranks of the timings is taken
    # The line is self-verifying for N > 1: how many ranks the process group really connected (one all-reduce of 1 per rank, over RCCL when the backend is nccl),
    # and below the same through the communicator the sharded frontend uses (the native bu_comm of libbasisu_rccl.so, or torch's under gloo)
    comm_check = None
    if world > 1:
        one = torch.ones(1, dtype=torch.int64, device=red_dev)
        dist.all_reduce(one)
        comm_check = {"process_group_backend": "gloo (ranks may share a GPU: the one-GPU test path)" if gloo else "nccl (= RCCL)", "world_size": world,
                      "ranks_seen": int(one.item()), "devices": torch.cuda.device_count()}

    # ---- synthetic input (SURVEY 8d recipe), tiled on the host once, resident in HBM before the timed region
    w = h = args.size
    sharded = args.shard_image and world > 1
    img = helpers.synth(w, h, 1234 if sharded else 1234 + rank)
    blocks = helpers.to_pixel_blocks(img)
    n_blocks = blocks.shape[0]
    d_blocks = torch.from_numpy(blocks.reshape(n_blocks, 64)).to(dev)
    max_ep, max_sel = quality_to_clusters(args.quality, n_blocks)

    ctx = capi.Context(local_rank)
    # run on torch's current stream so that torch.cuda.synchronize()/Events bracket our kernels
    ctx.check(ctx.lib.set_stream(ctx.h, torch.cuda.current_stream().cuda_stream), "set_stream")

    def make_comm():
        """the communicator of the sharded mode: native RCCL (C++ collectives on the context's stream); BU_TORCH_COMM=1 or gloo go through torch.distributed instead.
        Returns it with the number of ranks ONE all_reduce_u64 of 1 through it saw."""
        from basis_universal_amd.etc1s import TorchComm, RcclComm
        c = TorchComm() if (gloo or os.environ.get("BU_TORCH_COMM")) else RcclComm(ctx)
        t = torch.ones(1, dtype=torch.int64, device=dev)
        ok = c.struct.all_reduce_u64(c.struct.user, t.data_ptr(), 1)
        torch.cuda.synchronize()
        return c, (int(t.item()) if ok else 0), ("torch.distributed" if isinstance(c, TorchComm) else "bu_rccl (libbasisu_rccl.so: ncclAllReduce on the context's stream)")

    comm = None
    if sharded:
        comm, seen, kind = make_comm()
        comm_check.update({"sharding_communicator": kind, "sharding_communicator_ranks_seen": seen})

    def step():
        fe = Etc1sFrontend(ctx, comm)
        fe.init(d_blocks.data_ptr(), max_ep, max_sel, args.level, True, n_blocks=n_blocks)
        fe.compress()
        return fe

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    extra_ctx = []

    def run_in_flight(streams, images, with_backend=False):
        """`images` whole encodes, `streams` of them in flight: each worker thread owns a context (= a HIP stream) and encodes images one
        after the other (the reference's basis_parallel_compress pattern); ctypes releases the GIL during every library call."""
        import threading
        while len(extra_ctx) < streams - 1:
            c = capi.Context(local_rank)
            w_fe = Etc1sFrontend(c); w_fe.init(d_blocks.data_ptr(), max_ep, max_sel, args.level, True, n_blocks=n_blocks); w_fe.compress(); w_fe.close()  # warm its pool
            extra_ctx.append(c)
        ctxs = [ctx] + extra_ctx[:streams - 1]
        todo, lock, done = list(range(images)), threading.Lock(), []

        def worker(c):
            while True:
                with lock:
                    if not todo:
                        return
                    todo.pop()
                fe = Etc1sFrontend(c)
                fe.init(d_blocks.data_ptr(), max_ep, max_sel, args.level, True, n_blocks=n_blocks)
                fe.compress()
                if with_backend:  # the host backend of this image on this worker while the GPU serves the other workers' frontends
                    from basis_universal_amd.backend import Etc1sBackend
                    be = Etc1sBackend.from_frontend(fe, [(0, w // 4, h // 4)], 1.5, 1.25, args.level)
                    be.encode()
                    be.close()
                with lock:
                    done.append(fe)

        barrier()
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(c,)) for c in ctxs]
        [t.start() for t in th]
        [t.join() for t in th]
        barrier()
        return done, time.perf_counter() - t0

    for _ in range(args.warmup):
        step().close()
    # The timed steps carry HIP events around the kernels that are ONE launch each (profile level 2: the per-block / per-cluster kernels, among them the dominant
    # one whose roofline is reported): the step then runs as it does uninstrumented. Timing every region -- an event pair around each round of the codebook builders,
    # which also keeps a round's two kinds of nodes from sharing the device -- costs the step 0.5-0.7 ms (tools/headline_ab.py), so the full per-region breakdown
    # (kernels_ms_per_step, dominant_stage, host_gap_ms) comes from a second, separately timed pass of the same steps below.
    ctx.profile_enable(2)
    stage_acc = {}
    last = None
    cpu0 = time.process_time()
    if args.streams <= 1:
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if last is not None:
                last.close()
            last = step()
            for name, s in last.stage_times():
                stage_acc[name] = stage_acc.get(name, 0.0) + s
        barrier()
        elapsed = time.perf_counter() - t0
    else:
        finished, elapsed = run_in_flight(args.streams, args.steps)
        for fe in finished:
            for name, s in fe.stage_times():
                stage_acc[name] = stage_acc.get(name, 0.0) + s
        last = finished[-1]
        for fe in finished[:-1]:
            fe.close()
    host_cpu_s = (time.process_time() - cpu0) / args.steps  # all host threads of this rank (spin-waits on the stream included)
    kernels_headline = ctx.profile_read()
    # second pass: every region timed (instrumented: slower than the headline steps by what the events cost)
    instrumented_steps = max(1, min(args.steps, 10))
    ctx.profile_enable(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(instrumented_steps):
        step().close()
    barrier()
    instrumented_elapsed = time.perf_counter() - t0
    kernels = ctx.profile_read()
    ctx.profile_enable(False)

    pipelined = None
    if args.streams <= 1 and not sharded and not args.no_pipelined:
        pipelined = pipelined_bench(local_rank, d_blocks, n_blocks, w, h, max_ep, max_sel, args, barrier)
        if world > 1:
            t = torch.tensor([pipelined.pop("_seconds")], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            pipelined["value"] = round(world * pipelined["images"] * (w * h) / 1e6 / float(t.item()), 3)
            pipelined["images"] *= world
        pipelined.pop("_seconds", None)
        if world == 1:
            # ... with ONE driver thread for all lanes (what bu_frontend_pipeline_create gives): the cheapest form on the host
            one = pipelined_bench(local_rank, d_blocks, n_blocks, w, h, max_ep, max_sel, args, barrier, lanes=6, drivers=1)
            one.pop("_seconds", None); one.pop("note", None)
            pipelined["one_driver_thread"] = one
        # the same images-in-flight idea the round-4 way, one host thread + context per image (what the reference's basis_parallel_compress does): the host cost
        # of the pipeline's driver threads is to be read against this
        cpu0 = time.process_time()
        fes, dt = run_in_flight(6, 12)
        cpu = time.process_time() - cpu0
        for fe in fes:
            fe.close()
        pipelined["thread_per_image"] = {"images_in_flight_per_gpu": 6, "value": round(12 * (w * h) / 1e6 / dt, 3), "unit": "Mpixels/s (this rank)",
                                         "host_cpu_s_per_image": round(cpu / 12, 5)}
    whole_encoder = None
    if args.streams <= 1 and not sharded and not args.no_pipelined and world == 1:
        nf = max(1, args.backend_in_flight)
        fes, dt = run_in_flight(nf, 2 * nf, with_backend=True)
        for fe in fes:
            fe.close()
        whole_encoder = {"images_in_flight_per_gpu": nf, "images": 2 * nf, "value": round(2 * nf * (w * h) / 1e6 / dt, 3), "unit": "Mpixels/s",
                         "note": "frontend (GPU) + backend (host) per image = everything between the tiled input and the file writer; that many images in flight, one "
                                 "host thread / HIP stream each, each backend walking its image on two threads (throughput mode; not the headline value)"}
        # the same through the frontend pipeline: the frontends of all images on its driver threads, `nf` host threads left to the backends alone
        from basis_universal_amd.etc1s import FrontendPipeline
        from basis_universal_amd.backend import Etc1sBackend
        import threading
        pipe = FrontendPipeline(local_rank, PIPELINE_LANES, PIPELINE_DRIVERS)
        n_img = 2 * nf
        lock, tickets = threading.Lock(), []

        def backend_worker():
            while True:
                with lock:
                    if not tickets:
                        return
                    t = tickets.pop(0)
                fe = pipe.wait(t)
                be = Etc1sBackend.from_frontend(fe, [(0, w // 4, h // 4)], 1.5, 1.25, args.level)
                be.encode()
                be.close()
                fe.close()

        barrier()
        t0 = time.perf_counter()
        tickets.extend(pipe.submit(d_blocks.data_ptr(), max_ep, max_sel, args.level, True, n_blocks=n_blocks) for _ in range(n_img))
        th = [threading.Thread(target=backend_worker) for _ in range(nf)]
        [t.start() for t in th]
        [t.join() for t in th]
        barrier()
        dt = time.perf_counter() - t0
        pipe.close()
        whole_encoder["through_the_frontend_pipeline"] = {"lanes": PIPELINE_LANES, "driver_threads": PIPELINE_DRIVERS, "backend_threads": nf, "images": n_img,
                                                          "value": round(n_img * (w * h) / 1e6 / dt, 3), "unit": "Mpixels/s"}
    if world > 1:
        t = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- N > 1: BOTH modes in the same line. The headline is the mode asked for (default: one image per GPU, the throughput mode); the other one is measured
    #      here with the same steps / barrier / max-over-ranks rule, the sharded one with its per-stage split, so that DESIGN.md section 7's estimates are checked by the
    #      first run on a multi-GPU node.
    both_modes = None
    if world > 1 and args.streams <= 1:
        def mode_leg(shard):
            if shard:
                c, seen, kind = (comm, comm_check.get("sharding_communicator_ranks_seen"), comm_check.get("sharding_communicator")) if comm is not None else make_comm()
                blk = d_blocks if (sharded or rank == 0) else torch.from_numpy(helpers.to_pixel_blocks(helpers.synth(w, h, 1234)).reshape(n_blocks, 64)).to(dev)
            else:
                c, seen, kind = None, None, None
                blk = d_blocks if (not sharded or rank == 0) else torch.from_numpy(helpers.to_pixel_blocks(helpers.synth(w, h, 1234 + rank)).reshape(n_blocks, 64)).to(dev)

            def one():
                fe = Etc1sFrontend(ctx, c)
                fe.init(blk.data_ptr(), max_ep, max_sel, args.level, True, n_blocks=n_blocks)
                fe.compress()
                return fe
            one().close()
            calls0 = dict(getattr(c, "calls", {}) or {})
            acc, fe = {}, None
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                if fe is not None:
                    fe.close()
                fe = one()
                for name, sec in fe.stage_times():
                    acc[name] = acc.get(name, 0.0) + sec
            barrier()
            tt = torch.tensor([time.perf_counter() - t0], device=red_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            leg = {"value": round((1 if shard else world) * args.steps * (w * h) / 1e6 / dt, 3), "unit": "Mpixels/s", "ms_per_step": round(dt / args.steps * 1e3, 2),
                   "scaling": "strong" if shard else "weak"}
            if shard:
                leg.update({"communicator": kind, "communicator_ranks_seen": seen, "host_wall_s_per_step": stage_split(acc, args.steps),
                            "identical_to_reference": headline_identical(fe, w, h, args) if rank == 0 else None})
                if calls0 or getattr(c, "calls", None):
                    leg["collectives_per_step"] = {k: round((c.calls[k] - calls0.get(k, 0)) / args.steps, 1) for k in c.calls}
            fe.close()
            if shard and comm is None and hasattr(c, "close"):
                c.close()
            return leg
        other = None   # measured LAST, under a watchdog (below): the headline must not depend on a mode no hardware has run yet
        mine = {"value": round((1 if sharded else world) * args.steps * (w * h) / 1e6 / elapsed, 3), "unit": "Mpixels/s", "ms_per_step": round(elapsed / args.steps * 1e3, 2),
                "scaling": "strong" if sharded else "weak", "is_the_headline": True}
        if sharded:
            mine.update({"communicator": comm_check.get("sharding_communicator"), "communicator_ranks_seen": comm_check.get("sharding_communicator_ranks_seen"),
                         "host_wall_s_per_step": stage_split(stage_acc, args.steps)})
            if getattr(comm, "calls", None):
                mine["collectives_total"] = dict(comm.calls)
        OTHER_KEY = "replicas (one image per GPU, no data-path collective)" if sharded else "one image sharded over the ranks (--shard-image)"
        both_modes = {"replicas (one image per GPU, no data-path collective)": other if sharded else mine,
                      "one image sharded over the ranks (--shard-image)": mine if sharded else other}

    if rank == 0:
        mpix = (1 if sharded else world) * args.steps * (w * h) / 1e6
        value = mpix / elapsed
        # ---- roofline of the dominant KERNEL: the profile labels that are one kernel launch each (the tsvq_* / unique_* / map_* labels cover many launches
        #      of several kernels -- a codebook build is reported as a stage below, not as a kernel)
        single = {k: v for k, v in kernels_headline.items() if k in KERNEL_SYMBOL}   # measured inside the timed steps
        dom = max(single.items(), key=lambda kv: kv[1][0]) if single else None
        roofline = None
        if dom:
            name, (ms, launches) = dom
            avg_s = ms / 1e3 / launches
            alg_bytes = KERNEL_BYTES_PER_BLOCK.get(name, 64) * n_blocks
            achieved = alg_bytes / avg_s / 1e9
            roofline = {"bound": "hbm", "kernel": name, "kernel_symbol": KERNEL_SYMBOL[name], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                        "avg_launch_ms": round(avg_s * 1e3, 4), "algorithmic_bytes_per_launch": alg_bytes,
                        "note": "integer-ALU bound kernel (SURVEY 8d expected it): the bound that applies is VALU issue, see valu_bound and DESIGN.md section 4"}
        # the whole step against SURVEY 8d's figure: 443 algorithmic bytes per block over the step's wall time
        step_s = elapsed / args.steps
        step_roofline = {"bound": "hbm", "algorithmic_bytes_per_step": 443 * n_blocks, "achieved": round(443 * n_blocks / step_s / 1e9 / (1 if sharded else 1), 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(443 * n_blocks / step_s / 1e9 / HBM_PEAK_GBS, 5)}
        # the heaviest multi-launch stage (a TSVQ codebook build), by label
        stages_multi = {k: v for k, v in kernels.items() if k not in KERNEL_SYMBOL}
        dom_stage = None
        if stages_multi:
            k, (ms, launches) = max(stages_multi.items(), key=lambda kv: kv[1][0])
            dom_stage = {"label": k, "ms_per_step": round(ms / instrumented_steps, 3), "timed_regions_per_step": round(launches / instrumented_steps, 1),
                         "note": "a profile label over many kernel launches (not a kernel): device time between the HIP events around the region (instrumented pass)"}
        final_ep = int(last.get("endpoint_clusters", np.uint32)[0])
        final_sel = int(last.get("selector_cluster_block_indices", np.uint32)[0])
        out = {
            "metric": "encoder Mpixels/s (ETC1S frontend hot path)", "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "u8/int32", "data": "synthetic",
            "config": {"workload": f"{w}x{h} synthetic RGBA (SURVEY 8d recipe, seed 1234), ETC1S -q{args.quality} comp_level {args.level}, "
                                   f"basisu_frontend init+compress with tiles resident in HBM",
                       "blocks": n_blocks, "max_endpoint_clusters": max_ep, "max_selector_clusters": max_sel,
                       "final_endpoint_clusters": final_ep, "final_selector_clusters": final_sel,
                       "images_in_flight_per_gpu": args.streams, "host_threads_per_rank": int(os.environ.get("BU_HOST_THREADS", "8")), "host_cpus": host_cpus(),
                       "parallelism": (f"one image sharded over {world} GPUs: block-row slabs + cluster shares, RCCL all_gather / all_reduce between stages, TSVQ replicated"
                                       if sharded else f"{world} x one image per GPU (no collective)")},
            "comm": comm_check,
            "both_modes": both_modes,
            "roofline": roofline,
            "step_roofline": step_roofline,
            "dominant_stage": dom_stage,
            "pipelined": pipelined,
            "pipelined_with_backend": whole_encoder,
            "host_cpu_s_per_step": round(host_cpu_s, 4),
            "stages_s_per_step": {k: round(v / args.steps, 4) for k, v in stage_acc.items()},
            "kernels_ms_per_step": {k: round(v[0] / instrumented_steps, 3) for k, v in kernels.items()},
            "single_launch_kernels_ms_in_timed_steps": {k: round(v[0] / args.steps, 3) for k, v in kernels_headline.items()},
            "instrumented_pass": {"steps": instrumented_steps, "ms_per_step": round(instrumented_elapsed / instrumented_steps * 1e3, 2),
                                  "what": "the same steps with HIP events around EVERY region (kernels_ms_per_step, dominant_stage and host_gap_ms come from here); the timed "
                                          "steps above carry events around the single-launch kernels only"},
        }
        # the state the LAST timed step left (every per-block and per-cluster result the reference's getters serve) against the digests of the reference's
        # own run on this image (tests/golden/: tools/gen_golden_big.py / gen_golden_etc1s.py ran oracle/_ref; nothing under oracle/ is touched here)
        out["identical_to_reference"] = headline_identical(last, w, h, args)
        # what is NOT device time inside a step: host bookkeeping, copies and synchronising calls between the kernels
        out["host_gap_ms"] = round(instrumented_elapsed / instrumented_steps * 1e3 - sum(v[0] for v in kernels.values()) / instrumented_steps, 2)
        if roofline:
            roofline["traffic"] = pmc_traffic(roofline["kernel"])
        valu = valu_bound(out["kernels_ms_per_step"])
        if valu:
            out["valu_bound"] = valu
        # BASELINE's metric is "Mpixels/s + output PSNR": the texture the step's output decodes to against the source image
        out["psnr"] = dict(frontend_psnr(last, img), of="the ETC1 texture the frontend's output blocks decode to; the file's (after the backend's RDO) is backend.psnr",
                           semantics="image_metrics::calc, enc.cpp:2155-2226: RGB / RGBA average over channels, dB")
        if world == 1:  # what follows the hot path on the host: the ETC1S backend on the frontend just timed
            out["backend"] = backend_bench(last, w, h, args, elapsed / args.steps, img)
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline and the secondary workloads are N=1 measurements
            out["cpu_baseline"] = cpu_baseline(helpers, args)
            if helpers.have_ref():
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(helpers, args)
                out["end_to_end"] = end_to_end(helpers, args, img)
            ref_be = out["cpu_baseline"].pop("backend", None)
            if ref_be and "backend" in out:
                out["backend"]["reference_s"] = ref_be["seconds"]
                out["backend"]["identical_to_reference"] = ref_be["sha256"] == out["backend"].pop("sha256")
                out["backend"]["speedup_vs_reference"] = round(ref_be["seconds"] / (out["backend"]["ms_per_image"] / 1e3), 2)
        if "backend" in out:
            out["backend"].pop("sha256", None)
        if world == 1:
            out["reference_default_threads"] = multithreaded_config_bench(ctx, d_blocks, n_blocks, w, h, max_ep, max_sel, args)
            out["h2d_inclusive"] = h2d_inclusive_bench(ctx, blocks, n_blocks, w, h, max_ep, max_sel, args)
            out["mipmaps"] = mip_bench(ctx, img)
        if not args.no_uastc and world == 1:
            out["uastc"] = uastc_bench(ctx, d_blocks, n_blocks, w, h, helpers, args)
        if not args.no_fast and world == 1:
            out["fast_codebooks"] = fast_codebooks_bench(ctx, d_blocks, n_blocks, w, h, max_ep, max_sel, args)
            out["uastc_rdo"] = uastc_rdo_bench(ctx, helpers, args)
        if not args.no_big and world == 1 and args.size == 4096:
            out["etc1s_8192_q255"] = etc1s_8192_bench(ctx, helpers, args)
            # the other input distributions at the headline's size: beside the headline, never instead of it
            out["etc1s_noise4096_q128"] = other_distribution_bench(ctx, helpers, "noise4096_q128", lambda: helpers.uniform_random(4096, 4096, 42), "uniform-random RGB (seed 42)")
            out["etc1s_kodak4096_q128"] = other_distribution_bench(ctx, helpers, "kodak4096_q128", lambda: helpers.kodak_mosaic(4096, 4096), "mosaic of the 24 Kodak images")
            out["etc1s_cube4096_q128"] = other_distribution_bench(ctx, helpers, "cube4096_q128", lambda: helpers.endpoint_cube(4096, 4096, 7),
                                                                  "endpoint cube (nearly every ETC1S endpoint occurs: 228,656 distinct endpoint vectors)")
    # ---- N > 1: the OTHER mode's leg, after everything the line needs has been measured, and under a watchdog: if a rank fails in it or a collective never returns
    #      (the native RCCL path has not met a multi-GPU node yet), rank 0 still prints the line -- with the leg marked -- and every rank leaves.
    if both_modes is not None:
        import threading
        limit = float(os.environ.get("BU_BENCH_OTHER_MODE_TIMEOUT", "240"))
        finished = threading.Event()

        def bail():
            if finished.is_set():
                return
            if rank == 0:
                out["both_modes"][OTHER_KEY] = {"skipped": f"the leg did not finish within {limit:.0f} s; the headline and everything else in this line were measured before it"}
                print(json.dumps(out), flush=True)
            os._exit(0)
        timer = threading.Timer(limit, bail)
        timer.daemon = True
        timer.start()
        failed = None
        try:
            other = mode_leg(not sharded)
        except Exception as e:   # this rank failed: the others are (or will be) stuck in a collective and leave through their watchdogs
            failed = f"{type(e).__name__}: {e}"[:400]
            other = {"error": failed}
        finished.set()
        timer.cancel()
        if rank == 0:
            out["both_modes"][OTHER_KEY] = other
            print(json.dumps(out), flush=True)
        if failed is not None:
            os._exit(0)
    elif rank == 0:
        print(json.dumps(out))
    if last is not None:
        last.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


PIPELINE_LANES = 8      # images in flight of the `pipelined` leg ...
PIPELINE_DRIVERS = 2    # ... over this many driver threads (side by side on one box, 4096^2: 4 / 6 / 8 lanes on 1 / 1 / 2 threads = 1,402 / 1,564 / 1,801 Mpix/s; a host thread
                        # per image, 4 / 6 in flight: 1,703 / 1,837 at three times the host CPU: docs/HISTORY.md R5.10)


def stage_split(acc, steps):
    """The sharded step's host wall time by kind of stage (bu_frontend_stage_times: every stage ends with its results in place, collectives included): the codebook
    builders (top tree / node shares per round or whole trees per rank + their exchanges: the part DESIGN.md section 7 calls serial) against the slab / cluster-share
    stages. ('~gsc/...' entries are parts of generate_selector_clusters and are not counted twice.)"""
    top = {k: v for k, v in acc.items() if not k.startswith("~")}
    builders = ("generate_endpoint_clusters", "generate_selector_clusters")
    return {"codebook_builders": round(sum(v for k, v in top.items() if k in builders) / steps, 5),
            "slab_and_cluster_share_stages": round(sum(v for k, v in top.items() if k not in builders) / steps, 5),
            "by_stage": {k: round(v / steps, 5) for k, v in acc.items()}}


def pipelined_bench(device, d_blocks, n_blocks, w, h, max_ep, max_sel, args, barrier, lanes=PIPELINE_LANES, drivers=PIPELINE_DRIVERS):
    """Throughput mode of ONE GPU as library behaviour: bu_frontend_pipeline_* (include/basisu_hip_frontend.h) -- PIPELINE_LANES images in flight as cooperative
    tasks on the library's one driver thread; the caller only submits and collects. Every image's state is checked against the reference's digests (outside
    the timed region); host CPU seconds per image cover ALL threads of this process while the images were in flight."""
    from basis_universal_amd.etc1s import FrontendPipeline
    import torch
    images = 16   # results held at once below = contexts out at once (the library parks 16)
    pipe = FrontendPipeline(device, lanes, drivers)
    submit = lambda: pipe.submit(d_blocks.data_ptr(), max_ep, max_sel, args.level, True, n_blocks=n_blocks)
    for _ in range(2):   # as many warm contexts as results are held at once below
        for fe in [pipe.wait(t) for t in [submit() for _ in range(images)]]:
            fe.close()
    s0 = pipe.stats()
    barrier()
    cpu0, t0 = time.process_time(), time.perf_counter()
    done = [pipe.wait(t) for t in [submit() for _ in range(images)]]
    barrier()
    dt, cpu = time.perf_counter() - t0, time.process_time() - cpu0
    s1 = pipe.stats()
    same = [headline_identical(fe, w, h, args) for fe in done]
    for fe in done:
        fe.close()
    pipe.close()
    return {"api": f"bu_frontend_pipeline_* (cooperative tasks on {drivers} driver thread{'s' if drivers > 1 else ''})", "images_in_flight_per_gpu": lanes, "driver_threads": drivers, "images": images,
            "value": round(images * (w * h) / 1e6 / dt, 3), "unit": "Mpixels/s", "ms_per_image": round(dt / images * 1e3, 3),
            "host_cpu_s_per_image": round(cpu / images, 5), "driver_threads_cpu_s_per_image": round((s1["driver_cpu_s"] - s0["driver_cpu_s"]) / images, 5),
            "driver_s_per_image": {"in_tasks": round((s1["driver_busy_s"] - s0["driver_busy_s"]) / images, 5), "idle": round((s1["driver_idle_s"] - s0["driver_idle_s"]) / images, 5),
                                   "yields": round((s1["yields"] - s0["yields"]) / images), "naps": round((s1["idle_naps"] - s0["idle_naps"]) / images, 1)},
            "identical_to_reference": (all(same) if all(x is not None for x in same) else None), "_seconds": dt,
            "note": "same work per image as the headline step, tiles resident; throughput mode, not the headline value"}


def psnr_pair(decoded_rgb, img):
    """image_metrics::calc (encoder/basisu_enc.cpp:2155-2226) as basis_compressor calls it for its m_basis_rgb / rgba_avg_psnr stats (comp.cpp:4210-4221):
    mean squared error over the 3 / 4 channels (the decoded ETC1 texture has alpha 255), 20 log10(255 / rms), clamped to 100 dB."""
    import helpers
    h, w = img.shape[:2]
    rgba = np.concatenate([decoded_rgb, np.full((h, w, 1), 255, np.uint8)], axis=2)
    return round(helpers.psnr(decoded_rgb, img[..., :3]), 4), round(helpers.psnr(rgba, img), 4)


def frontend_psnr(fe, img):
    """PSNR of the texture the frontend's output blocks decode to (its encoded_blocks: endpoint + selector codebook entry of every block)"""
    import helpers
    h, w = img.shape[:2]
    rgb, rgba = psnr_pair(helpers.decode_etc1s_blocks(fe.get("encoded_blocks").reshape(-1, 8), w // 4, h // 4), img)
    return {"rgb": rgb, "rgba": rgba}


def headline_identical(last, w, h, args):
    """True / False: all eight frontend state digests of the step just timed = the committed digests of the reference's run on the same image and settings;
    None when no golden exists for this size / quality / level (the 4096^2 -q128 level-1 headline has one; rank 0's image is the seed-1234 image in every mode)."""
    if args.quality != 128 or args.level != 1 or (w, h) != (4096, 4096):
        return None
    g = ROOT / "tests" / "golden" / "etc1s_big_digests.json"
    rec = json.loads(g.read_text()).get("synth4096_q128") if g.exists() else None
    want = (rec or {}).get("frontend_digests")
    if not want:
        return None
    import test_gpu_etc1s_frontend as T      # the canonical byte form of each state array (tests/: checker code, outside the timed region)
    got = T._digest({k: last.get(k) for k in want})
    return got == want


def etc1s_8192_bench(ctx, helpers, args):
    """BASELINE configs[3] in its single-GPU form: 8192x8192 synthetic RGBA (seed 5678), ETC1S -q 255 (8192 / 16128 clusters), level 1, tiles resident.
    The same full init + compress per step as the headline; the result is compared with the committed digest of the reference's output
    (tests/golden/etc1s_big_digests.json, tools/gen_golden_big.py), whose single-core seconds are quoted beside it."""
    import hashlib
    import torch
    from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters
    img = helpers.synth(8192, 8192, 5678)
    blocks = helpers.to_pixel_blocks(img)
    n = blocks.shape[0]
    d = torch.from_numpy(blocks.reshape(n, 64)).cuda()
    max_ep, max_sel = quality_to_clusters(255, n)

    def step():
        fe = Etc1sFrontend(ctx)
        fe.init(d.data_ptr(), max_ep, max_sel, 1, True, n_blocks=n)
        fe.compress()
        return fe

    step().close()
    torch.cuda.synchronize()
    steps, last = 3, None
    t0 = time.perf_counter()
    for _ in range(steps):   # the timed steps run uninstrumented; the per-region breakdown comes from one more step with every region timed (see main)
        if last is not None:
            last.close()
        last = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ctx.profile_enable(1)
    t0 = time.perf_counter()
    step().close()
    torch.cuda.synchronize()
    dt_instrumented = time.perf_counter() - t0
    kern = ctx.profile_read()
    ctx.profile_enable(False)
    out = {"workload": "8192x8192 synthetic RGBA (SURVEY 8d recipe, seed 5678), ETC1S -q255 comp_level 1 (8192 / 16128 clusters), init+compress, tiles resident",
           "value": round(8192 * 8192 / 1e6 / dt, 3), "unit": "Mpixels/s", "ms_per_step": round(dt * 1e3, 2), "steps": steps,
           "kernels_ms_per_step": {k: round(v[0], 3) for k, v in kern.items()}, "instrumented_step_ms": round(dt_instrumented * 1e3, 2),
           "host_gap_ms": round(dt_instrumented * 1e3 - sum(v[0] for v in kern.values()), 2), "psnr": frontend_psnr(last, img)}
    g = ROOT / "tests" / "golden" / "etc1s_big_digests.json"
    if g.exists():
        rec = json.loads(g.read_text()).get("synth8192_q255")
        if rec:
            out["identical_to_reference"] = hashlib.sha256(np.ascontiguousarray(last.get("encoded_blocks")).tobytes()).hexdigest() == rec["frontend_digests"]["encoded_blocks"]
            out["reference_frontend_seconds_1_core"] = rec["reference_seconds"]["frontend"]
            out["speedup_vs_reference_1_core"] = round(rec["reference_seconds"]["frontend"] / dt, 1)
    last.close()
    del d
    torch.cuda.empty_cache()
    return out


def other_distribution_bench(ctx, helpers, case, img_fn, what):
    """The headline step on another input distribution at full size (uniform-random RGB seed 42 = the worst case for every clustering stage, the honest floor of
    the Mpixels/s figure; the reference's Kodak images as one 4096^2 mosaic = photographic statistics; the endpoint cube = the most distinct endpoint vectors an image can
    have). Same full init + compress per step, -q128 level 1, tiles resident, every state digest compared with the reference's run on the same image
    (tests/golden/etc1s_big_digests.json: tools/gen_golden_big.py), which also recorded the distinct training vectors each codebook builder saw."""
    import torch
    from basis_universal_amd.etc1s import Etc1sFrontend
    g = ROOT / "tests" / "golden" / "etc1s_big_digests.json"
    rec = json.loads(g.read_text()).get(case) if g.exists() else None
    if not rec:
        return None
    img = img_fn()
    blocks = helpers.to_pixel_blocks(img)
    n = blocks.shape[0]
    d = torch.from_numpy(blocks.reshape(n, 64)).cuda()

    def step():
        fe = Etc1sFrontend(ctx)
        fe.init(d.data_ptr(), rec["max_endpoint_clusters"], rec["max_selector_clusters"], rec["level"], True, n_blocks=n)
        fe.compress()
        return fe

    step().close()
    torch.cuda.synchronize()
    steps, last = 5, None
    t0 = time.perf_counter()
    for _ in range(steps):
        if last is not None:
            last.close()
        last = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ctx.profile_enable(1)
    t0 = time.perf_counter()
    step().close()
    torch.cuda.synchronize()
    dt_instrumented = time.perf_counter() - t0
    kern = ctx.profile_read()
    ctx.profile_enable(False)
    import test_gpu_etc1s_frontend as T
    same = T._digest({k: last.get(k) for k in rec["frontend_digests"]}) == rec["frontend_digests"]
    h, w = img.shape[:2]
    out = {"workload": f"{w}x{h} {what}, ETC1S -q{rec['quality']} comp_level {rec['level']} ({rec['max_endpoint_clusters']} / {rec['max_selector_clusters']} clusters), init+compress, tiles resident",
           "value": round(w * h / 1e6 / dt, 3), "unit": "Mpixels/s", "ms_per_step": round(dt * 1e3, 2), "steps": steps, "identical_to_reference": same,
           "distinct_training_vectors": rec.get("distinct_vectors"), "final_clusters": [rec["final_endpoint_clusters"], rec["final_selector_clusters"]],
           "kernels_ms_per_step": {k: round(v[0], 3) for k, v in kern.items()}, "instrumented_step_ms": round(dt_instrumented * 1e3, 2),
           "psnr": frontend_psnr(last, img), "reference_frontend_seconds_1_core": rec["reference_seconds"]["frontend"],
           "speedup_vs_reference_1_core": round(rec["reference_seconds"]["frontend"] / dt, 1)}
    last.close()
    del d
    torch.cuda.empty_cache()
    return out


def multithreaded_config_bench(ctx, d_blocks, n_blocks, w, h, max_ep, max_sel, args, threads=8):
    """The headline workload under the reference's DEFAULT (multi-threaded) configuration: codebook builders handed `threads` threads (frontend.cpp:2195-2198), so
    that the selector codebook (674,691 distinct vectors >= 262,144) comes from a T-way partitioned tree build (enc.h:2086-2215) whose T trees share every device
    round. Same full init + compress per step; the output is compared with the committed digest of the reference run with a job pool of that many threads."""
    import hashlib
    import torch
    from basis_universal_amd.etc1s import Etc1sFrontend

    def step():
        fe = Etc1sFrontend(ctx, max_threads=threads)
        fe.init(d_blocks.data_ptr(), max_ep, max_sel, args.level, True, n_blocks=n_blocks)
        fe.compress()
        return fe

    step().close()
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    steps, last = max(3, args.steps // 2), None
    t0 = time.perf_counter()
    for _ in range(steps):
        if last is not None:
            last.close()
        last = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kern = ctx.profile_read()
    ctx.profile_enable(False)
    out = {"workload": f"the headline workload with the reference's multi-threaded codebook builders, {threads} threads (the tool's default on a machine with >= {threads} hardware threads)",
           "codebook_threads": threads, "value": round(w * h / 1e6 / dt, 3), "unit": "Mpixels/s", "ms_per_step": round(dt * 1e3, 2), "steps": steps,
           "kernels_ms_per_step": {k: round(v[0] / steps, 3) for k, v in kern.items() if k.startswith("tsvq") or k.startswith("unique")},
           "host_gap_ms": round(dt * 1e3 - sum(v[0] for v in kern.values()) / steps, 2),
           "final_selector_clusters": int(last.get("selector_cluster_block_indices", np.uint32)[0])}
    g = ROOT / "tests" / "golden" / "etc1s_big_digests.json"
    if g.exists() and (w, h, args.quality, args.level) == (4096, 4096, 128, 1):
        rec = json.loads(g.read_text()).get(f"synth4096_q128_t{threads}")
        if rec:
            out["identical_to_reference"] = hashlib.sha256(np.ascontiguousarray(last.get("encoded_blocks")).tobytes()).hexdigest() == rec["frontend_digests"]["encoded_blocks"]
            out["reference_frontend_seconds"] = {"threads": threads, "seconds": rec["reference_seconds"]["frontend"], "where": "the build container (8 hardware threads), tools/gen_golden_big.py"}
    last.close()
    return out


def h2d_inclusive_bench(ctx, blocks, n_blocks, w, h, max_ep, max_sel, args):
    """SURVEY 8d figure (i) as written there -- the hot path INCLUDING the host-to-device transfer of the tiles: every step hands bu_frontend_init a HOST pointer
    (the boundary the reference's opencl_set_pixel_blocks has), once pageable (what a caller holding a std::vector gives) and once page-locked."""
    import torch
    from basis_universal_amd.etc1s import Etc1sFrontend
    res = {"what": "init (64 B/block H2D inside the step) + compress; the headline `value` starts with the tiles resident. The upload is pipelined with the first kernel "
                   "(bu_hip_k_upload_and_encode_etc1s_blocks: 4 MiB pieces on the side stream, piece i encoded behind piece i's copy, pageable memory staged by helper threads)",
           "bytes_per_step": int(n_blocks) * 64}
    pinned = torch.from_numpy(blocks.reshape(n_blocks, 64)).pin_memory()
    for name, host in (("pageable", np.ascontiguousarray(blocks).reshape(n_blocks, 64)), ("pinned", pinned.numpy())):
        def step():
            fe = Etc1sFrontend(ctx)
            fe.init(host, max_ep, max_sel, args.level, True)
            fe.compress()
            return fe
        step().close()
        torch.cuda.synchronize()
        steps = max(3, min(args.steps, 10))
        t0 = time.perf_counter()
        for _ in range(steps):
            step().close()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res[name] = {"value": round(w * h / 1e6 / dt, 3), "unit": "Mpixels/s", "ms_per_step": round(dt * 1e3, 2), "steps": steps}
    return res


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this same command (profiles/pmc_traffic.json, written by
    tools/rocprof_summary.py traffic from separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled per the gfx950 note in
    MI355X_MICROARCH.md). Counters cannot be collected inside a normal run, so this is null when the file has no entry."""
    f = ROOT / "profiles" / "pmc_traffic.json"
    if not f.exists():
        return None
    rec = json.loads(f.read_text()).get(kernel)
    return None if not rec else int(rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"])


def valu_bound(kernels_ms):
    """The per-block ETC1S kernels are bound by VALU instruction issue, not by HBM or the matrix cores. For the kernels of this step that
    the committed SQ counter pass covers (profiles/valu_busy.json = tools/valu_table.py --json over a separate rocprofv3 --pmc run of this same command):
    the share of all SIMD issue cycles their VALU instructions occupy -- instructions executed (SQ_INSTS_VALU) x the measured cost of their opcodes
    (profiles/valu_calibration.json, tools/valu_calib.hip) over the launch's shader cycles (SQ_BUSY_CYCLES / 32) -- next to this run's own launch time."""
    f = ROOT / "profiles" / "valu_busy.json"
    if not f.exists():
        return None
    rec = json.loads(f.read_text())
    keep = ("valu_busy_frac", "valu_instructions_per_launch", "mean_cycles_per_instruction", "static_mix", "waves_stalled_on_issue_frac")
    rows = {k: dict({"ms_this_run": kernels_ms[k]}, **{kk: v[kk] for kk in keep if kk in v}) for k, v in rec.items() if k in kernels_ms}
    meta = rec.get("_meta", {})
    return {"kernels": rows, "ceiling": meta.get("ceiling"), "method": meta.get("busy"), "source": "profiles/valu_busy.json"} if rows else None


MFMA_F16_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md


def fast_codebooks_bench(ctx, d_blocks, n_blocks, w, h, max_ep, max_sel, args):
    """the same step with both codebooks from the matrix-core k-means instead of the TSVQ. NOT bit-identical to the reference
    (held to its size / PSNR tolerances by tests/test_gpu_fast_codebooks.py); reported beside the headline value, never as it. The roofline
    of its dominant kernel is an MFMA one: flops of the assignment GEMMs (2 x vectors x padded centroids x 16 dims x 2 operand halves per round)."""
    import torch
    from basis_universal_amd.etc1s import Etc1sFrontend

    def step():
        fe = Etc1sFrontend(ctx, fast_codebooks=True)
        fe.init(d_blocks.data_ptr(), max_ep, max_sel, args.level, True, n_blocks=n_blocks)
        fe.compress()
        return fe

    step().close()
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    steps = max(args.steps, 3)
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        if last is not None:
            last.close()
        last = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kernels = ctx.profile_read()
    ctx.profile_enable(False)
    enc = last.get("orig_encoded_blocks", np.uint8).reshape(-1, 8)
    u_sel = int(np.unique(np.ascontiguousarray(enc[:, 4:]).view(np.uint32)).size)
    final = (int(last.get("endpoint_clusters", np.uint32)[0]), int(last.get("selector_cluster_block_indices", np.uint32)[0]))
    last.close()
    out = {"what": "ETC1S frontend with both codebooks from k-means on the matrix cores (4 Lloyd rounds + the final assignment) instead of the TSVQ; "
                   "NOT bit-identical to the reference, gated by its 4.5 % size / 0.3 dB tolerances (tests/test_gpu_fast_codebooks.py)",
           "value": round(w * h / 1e6 / dt, 3), "unit": "Mpixels/s", "ms_per_step": round(dt * 1e3, 2), "final_clusters": final,
           "kernels_ms_per_step": {k: round(v[0] / steps, 3) for k, v in kernels.items() if k.startswith("kmeans")}}
    if "kmeans_selectors" in kernels:
        ms, launches = kernels["kmeans_selectors"]
        k_pad = (max_sel + 63) // 64 * 64
        rounds = 5
        flops = 2.0 * u_sel * k_pad * 16 * 2 * rounds
        avg_s = ms / 1e3 / launches
        out["roofline"] = {"bound": "mfma", "kernel": "kmeans_selectors", "achieved": round(flops / avg_s / 1e12, 2), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(flops / avg_s / 1e12 / MFMA_F16_PEAK_TFLOPS, 5), "traffic": None, "avg_launch_ms": round(avg_s * 1e3, 3),
                           "flops_per_launch": flops, "distinct_vectors": u_sel,
                           "note": "one timed region = unpack + seeding + 5 assignment GEMMs (k_km_assign: LDS-staged centroid tiles, 4 column tiles per wave, min-only epilogue of 17 VALU "
                                   "instructions per MFMA pair, LDS accumulators for the centroid sums) + centroid updates + the reseeding of empty clusters"}
    return out


def uastc_bench(ctx, d_blocks, n_blocks, w, h, helpers, args):
    """BASELINE config #3: the same resident tiles through encode_uastc level 2 (rows a16-a19). One step = all four phases."""
    import torch
    from basis_universal_amd import uastc
    d_out = torch.empty((n_blocks, 16), dtype=torch.uint8, device=d_blocks.device)
    flags = uastc.LEVEL_DEFAULT
    uastc.encode_uastc_blocks(ctx, d_blocks.data_ptr(), flags, n_blocks=n_blocks, out_device=d_out.data_ptr())
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    steps = max(args.steps, 3)
    t0 = time.perf_counter()
    for _ in range(steps):
        uastc.encode_uastc_blocks(ctx, d_blocks.data_ptr(), flags, n_blocks=n_blocks, out_device=d_out.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kern = ctx.profile_read()
    ctx.profile_enable(False)
    name, (ms, launches) = max(kern.items(), key=lambda kv: kv[1][0])
    avg_s = ms / 1e3 / launches
    alg = 80 * n_blocks  # 64 B tile in + 16 B block out
    res = {"metric": "UASTC LDR 4x4 level 2 encoder Mpixels/s", "value": round(w * h / 1e6 / dt, 2), "unit": "Mpixels/s", "ms_per_step": round(dt * 1e3, 2),
           "kernels_ms_per_step": {k: round(v[0] / steps, 3) for k, v in kern.items()},
           "roofline": {"bound": "hbm", "kernel": name, "achieved": round(alg / avg_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / avg_s / 1e9 / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(name), "avg_launch_ms": round(avg_s * 1e3, 3)}}
    # output PSNR over the whole image (the host build of the same core decodes the 1,048,576 blocks)
    src = d_blocks.cpu().numpy().reshape(-1, 4, 4, 4)
    res["psnr_rgba"] = round(helpers.psnr(helpers.host_decode_uastc(d_out.cpu().numpy()), src), 4)
    g = ROOT / "tests" / "golden" / "uastc_big_digests.json"
    if g.exists() and w == 4096 and h == 4096:
        import hashlib
        res["identical_to_reference"] = hashlib.sha256(d_out.cpu().numpy().tobytes()).hexdigest() == json.loads(g.read_text())["synth4096_l2"]["sha256"]
    if not args.no_cpu_baseline and helpers.have_ref():
        sample = d_blocks[:: max(1, n_blocks // 65536)][:65536].cpu().numpy().reshape(-1, 4, 4, 4)
        t0 = time.perf_counter()
        helpers.ref_encode_uastc(sample, flags)
        cdt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(sample.shape[0] * 16 / 1e6 / cdt, 4), "unit": "Mpixels/s", "cores": 1, "kind": "reference",
                               "sample": f"{sample.shape[0]} blocks strided over the bench image, reference encode_uastc level 2 (oracle/_ref), {cdt:.2f} s"}
    return res


def uastc_rdo_bench(ctx, helpers, args):
    """BASELINE config #5 on one GPU: the 24 Kodak images (tests/golden/kodak24.npz, the reference's own test_files/kodim01..24) as ONE resident batch through
    encode_uastc level 2 and uastc_rdo (lambda 1.0, 4 strips per image = comp.cpp:2078's min(4, threads)): 96 strips of 6144 blocks walk
    concurrently, one workgroup each (row a20). One step = encode + RDO of the whole batch. Every image's output is compared with the sha256 of the
    reference's (tests/golden/kodak24_digests.json) and its RGBA PSNR is printed (BASELINE: "PSNR-gated vs. reference")."""
    import hashlib
    import numpy as np
    import torch
    from basis_universal_amd import uastc
    n_images = 24
    npz, dig = ROOT / "tests" / "golden" / "kodak24.npz", ROOT / "tests" / "golden" / "kodak24_digests.json"
    golden = json.loads(dig.read_text())["images"] if dig.exists() else None
    if npz.exists():
        z = np.load(npz)
        names = sorted(z.files)
        images = [np.concatenate([z[k], np.full(z[k].shape[:2] + (1,), 255, np.uint8)], axis=2) for k in names]
        what = "the 24 Kodak images (kodim01..24, 768x512 / 512x768 RGB, alpha 255)"
    else:   # fixture missing: synthetic stand-ins of the same size, said so in the line
        names = [f"s{k:02d}" for k in range(n_images)]
        images = [helpers.synth(768, 512, 500 + k) if k % 2 == 0 else helpers.synth_smooth(768, 512, 500 + k) for k in range(n_images)]
        what = "24 x 768x512 synthetic RGBA (12 noisy + 12 smooth): tests/golden/kodak24.npz not found"
    parts = [helpers.to_pixel_blocks(im) for im in images]
    ofs = np.cumsum([0] + [p.shape[0] for p in parts])
    blocks = np.concatenate(parts)
    n = blocks.shape[0]
    d_px = torch.from_numpy(blocks.reshape(n, 64)).cuda()
    d_out = torch.empty((n, 16), dtype=torch.uint8, device=d_px.device)
    params = uastc.RdoParams(m_lambda=1.0)
    flags, jobs = uastc.LEVEL_DEFAULT, 4 * n_images

    def step():
        uastc.encode_uastc_blocks(ctx, d_px.data_ptr(), flags, n_blocks=n, out_device=d_out.data_ptr())
        return uastc.uastc_rdo(ctx, d_out.data_ptr(), d_px.data_ptr(), params, flags, jobs, n_blocks=n)[1]

    step()
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    steps = max(args.steps, 3)
    t0 = time.perf_counter()
    for _ in range(steps):
        info = step()
    torch.cuda.synchronize()
    sync_steps = steps
    dt_one = (time.perf_counter() - t0) / steps   # one batch start to finish, the host waiting for each: the latency of a batch, and where the per-kernel times come from
    kern = ctx.profile_read()
    ctx.profile_enable(False)
    # The timed steps: every step SUBMITS the batch to the library's pipeline (bu_hip_uastc_pipeline_*: `lanes` private streams + workspaces, nothing between submissions
    # waits for the host), the timed region ends when the last one is complete. One batch's strips kernel is 96 serial chains on 96 of 256 CUs for ~20 ms; the next
    # submission's encode kernels and the previous one's hint refit run beside it. Same bytes per step as the synchronous form above (checked below on the last lanes).
    lanes = int(os.environ.get("BU_UASTC_LANES", "4"))   # (3 / 4 lanes side by side, 12 steps each: 18.4 / 16.4 ms per batch -- profiles/r06_rdo_lanes.txt)
    # 64 of the 256 CUs carry the lanes' strip walks and nothing else (bu_hip_tuning::uastc_walk_cus, a setting of the pipeline's context, read when the lanes are made;
    # off by default in the library: a pipeline that only encodes would lose those CUs). Three runs of tools/rdo_lanes.py, 4 lanes: 529 / 549 / 537 Mpix/s without, 557 / 572 / 569 with.
    walk_cus = int(os.environ.get("BU_UASTC_WALK_CUS", "64"))
    ctx.set_tuning(uastc_walk_cus=walk_cus)
    pipe = uastc.UastcPipeline(ctx, lanes, n, flags, jobs)
    ctx.set_tuning()
    outs = [d_out] + [torch.empty_like(d_out) for _ in range(lanes - 1)]
    for k in range(lanes):
        pipe.submit(d_px.data_ptr(), n, outs[k].data_ptr(), params, flags, jobs)
    pipe.wait(0)
    torch.cuda.synchronize()
    # (at least 12 submissions: with as many steps as lanes the figure is ONE wave of batches start to finish -- 17.9 and 24.3 ms per batch in two runs of the same build)
    steps = max(steps, 12)
    t0 = time.perf_counter()
    for k in range(steps):
        last_ticket = pipe.submit(d_px.data_ptr(), n, outs[k % lanes].data_ptr(), params, flags, jobs)
    pipe.wait(0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    lanes_identical = bool(all((o == outs[0]).all().item() for o in outs[1:])) if lanes > 1 else True
    pipe.close()
    out_blocks = d_out.cpu().numpy()
    psnrs, same = [], []
    for i, (name, im) in enumerate(zip(names, images)):
        mine = out_blocks[ofs[i]:ofs[i + 1]]
        h, w = im.shape[:2]
        psnrs.append(helpers.psnr(helpers.host_decode_uastc(mine, w // 4, h // 4), im))
        if golden and name in golden:
            same.append(hashlib.sha256(mine.tobytes()).hexdigest() == golden[name]["uastc_l2_rdo1_jobs4"])
    res = {"metric": "UASTC LDR 4x4 level 2 + RDO (lambda 1.0) Mpixels/s", "value": round(n * 16 / 1e6 / dt, 2), "unit": "Mpixels/s",
           "ms_per_step": round(dt * 1e3, 2), "steps": steps, "workload": f"{what}, {jobs} strips of {n // jobs} blocks per step",
           "submission": f"every step is one bu_hip_uastc_pipeline_submit of the batch ({lanes} lanes: that many steps in flight on the device, no host wait between them); "
                         "timed from the first submission to the completion of the last",
           "lanes": lanes, "reserved_walk_cus": walk_cus, "lanes_identical": lanes_identical,
           "one_batch_start_to_finish": {"ms": round(dt_one * 1e3, 2), "value": round(n * 16 / 1e6 / dt_one, 2), "unit": "Mpixels/s",
                                         "note": "bu_hip_k_encode_uastc_blocks + bu_hip_k_uastc_rdo with the host waiting for each batch (round 3's `value`)"},
           "modified_blocks": int(info["modified"]), "kernels_ms_per_step": {k: round(v[0] / sync_steps, 3) for k, v in kern.items()},
           "serial_step_us": round(kern["uastc_rdo_strips"][0] / sync_steps * 1e3 / (n // jobs), 3),
           "psnr_rgba": {"mean": round(float(np.mean(psnrs)), 4), "min": round(float(np.min(psnrs)), 4), "max": round(float(np.max(psnrs)), 4)},
           "images_identical_to_reference": (f"{sum(same)}/{len(same)}" if same else None)}
    if golden and same:
        ref_p = [golden[nm]["uastc_psnr_rgba_rdo1_jobs4"] for nm in names if nm in golden]
        res["psnr_rgba"]["reference_mean"] = round(float(np.mean(ref_p)), 4)
    if not args.no_cpu_baseline and helpers.have_ref():
        one = parts[2]  # kodim03
        t0 = time.perf_counter()
        packed = helpers.ref_encode_uastc(one, flags)
        t1 = time.perf_counter()
        helpers.ref_uastc_rdo(packed, one, flags, 0, lam=1.0)
        t2 = time.perf_counter()
        res["cpu_baseline"] = {"value": round(one.shape[0] * 16 / 1e6 / (t2 - t0), 4), "unit": "Mpixels/s", "cores": 1, "kind": "reference",
                               "sample": f"one image of the batch ({names[2]}): reference encode_uastc {t1 - t0:.2f} s + uastc_rdo {t2 - t1:.2f} s on one thread (oracle/_ref)"}
    return res


def mip_bench(ctx, img):
    """The whole mip chain below the bench image on the device (row f4: kaiser, sRGB, wrapping, each level from the one above), raster resident."""
    from basis_universal_amd import mipmap
    L = mipmap._lib()
    h, w = img.shape[:2]
    sizes = mipmap.level_sizes(w, h)
    bufs = [(ctx.upload(img), w, h)] + [(ctx.alloc(lw * lh * 4), lw, lh) for lw, lh in sizes]
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        for (src, sw, sh), (dst, dw, dh) in zip(bufs[:-1], bufs[1:]):
            ctx.check(L.bu_generate_mipmap_level(ctx.h, src, sw, sh, dst, dw, dh, 1, b"kaiser", 1.0, 1, 3), "bu_generate_mipmap_level")
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    for d, _, _ in bufs:
        ctx.free(d)
    return {"what": f"{len(sizes)} levels below {w}x{h}, host-built filter plans + HIP resampling, raster resident in HBM", "ms": round(best * 1e3, 2),
            "mpix_s_of_base_image": round(w * h / 1e6 / best, 1)}


def _payload_digest(get):
    import hashlib
    h = hashlib.sha256()
    for k in ("endpoint_palette", "selector_palette", "slice_image_tables", "slice_image_data", "slice_image_crcs"):
        h.update(np.ascontiguousarray(get(k)).tobytes())
    return h.hexdigest()


def backend_bench(fe, w, h, args, frontend_s, img=None):
    """bu::etc1s_backend (host, one thread) on the finished frontend: endpoint prediction + RDO, selector history RDO, Huffman coding -> the
    compressed payloads of the .basis file. Default basis_compressor thresholds (1.5 / 1.25)."""
    from basis_universal_amd.backend import Etc1sBackend
    best, n, psnr = None, 0, None
    for _ in range(2):  # the first call also fetches the host copy of the resident tiles
        be = Etc1sBackend.from_frontend(fe, [(0, w // 4, h // 4)], 1.5, 1.25, args.level)
        t0 = time.perf_counter()
        n = be.encode()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        digest = _payload_digest(be.get)
        stages = {k: round(v, 4) for k, v in be.stage_times()}
        if img is not None and psnr is None:
            import helpers
            rgb, rgba = psnr_pair(helpers.decode_backend_output(fe, be, w // 4, h // 4), img)
            psnr = {"rgb": rgb, "rgba": rgba, "of": "the ETC1 texture the .basis file decodes to = basis_compressor's m_basis_rgb / rgba_avg_psnr"}
        be.close()
        if args.level > 1:
            break  # above level 1 the backend changes the frontend: once only
    return {"what": "ETC1S backend on the host after the frontend (1 thread), endpoint/selector RDO thresholds 1.5/1.25", "ms_per_image": round(best * 1e3, 1),
            "compressed_bytes": n, "bits_per_texel": round(n * 8 / (w * h), 3), "stages_s": stages,
            "frontend_plus_backend_mpix_s": round(w * h / 1e6 / (frontend_s + best), 2), "psnr": psnr, "sha256": digest}


def cpu_baseline(helpers, args):
    """The same stage set (basisu_frontend init + compress) on one host core = the parity-pinned configuration of the reference.
    With oracle/_ref present the sample is the bench workload itself (the whole 4096x4096 image, ~20 s of CPU); without it, the C
    oracle's per-block fit on a crop (a lower bound of the reference's cost)."""
    from basis_universal_amd.etc1s import quality_to_clusters
    if helpers.have_ref():
        side = args.size
        sample = helpers.to_pixel_blocks(helpers.synth(args.size, args.size, 1234))
        n = sample.shape[0]
        max_ep, max_sel = quality_to_clusters(args.quality, n)
        t0 = time.perf_counter()
        fe = helpers.RefFrontend(sample, max_ep, max_sel, args.level, True)
        fe.call("compress")
        dt = time.perf_counter() - t0
        ref_total, ref_be_s = fe.backend_run([(0, side // 4, side // 4)], 1.5, 1.25)
        ref_backend = {"seconds": round(ref_be_s, 3), "bytes": ref_total, "sha256": _payload_digest(fe.backend_get)}
        fe.close()
        kind, what = "reference", "reference basisu_frontend::init+compress (oracle/_ref, built from /root/reference, -O3, no SSE, single thread)"
    else:
        side = min(1024, args.size)
        sample = helpers.to_pixel_blocks(helpers.synth(args.size, args.size, 1234)[:side, :side])
        n = sample.shape[0]
        max_ep, max_sel = quality_to_clusters(args.quality, n)
        t0 = time.perf_counter()
        helpers.orc_encode_blocks(sample, args.level, True)
        dt = time.perf_counter() - t0
        kind, what = "port", "oracle per-block ETC1S fit only (reference build not present)"
    res = {"value": round(side * side / 1e6 / dt, 4), "unit": "Mpixels/s", "cores": 1, "kind": kind,
           "sample": f"{side}x{side} of the bench image, {n} blocks, {max_ep}/{max_sel} clusters, {what}, {dt:.2f} s",
           "host": f"{os.cpu_count()} logical CPUs on the GPU box"}
    if kind == "reference":
        res["backend"] = ref_backend
    return res


def cpu_baseline_all_cores(helpers, args):
    """The same stage set on ALL host cores: the reference frontend with its job pool sized like the tool sizes it by default (one thread per
    hardware thread, basisu_tool.cpp:2331-2348). Not the parity configuration: above 262,144 distinct selector vectors the reference's
    output depends on the thread count (SURVEY hazard H1)."""
    from basis_universal_amd.etc1s import quality_to_clusters
    threads = host_cpus()
    sample = helpers.to_pixel_blocks(helpers.synth(args.size, args.size, 1234))
    n = sample.shape[0]
    max_ep, max_sel = quality_to_clusters(args.quality, n)
    t0 = time.perf_counter()
    fe = helpers.RefFrontend(sample, max_ep, max_sel, args.level, True, threads=threads)
    fe.call("compress")
    dt = time.perf_counter() - t0
    fe.close()
    return {"value": round(args.size * args.size / 1e6 / dt, 4), "unit": "Mpixels/s", "cores": threads, "kind": "reference",
            "sample": f"{args.size}x{args.size} bench image, {n} blocks, {max_ep}/{max_sel} clusters, reference basisu_frontend::init+compress with a "
                      f"{threads}-thread job pool (oracle/_ref), {dt:.2f} s"}


def end_to_end(helpers, args, img):
    """SURVEY 8d figure (ii): wall time of the reference's own driver, basis_compressor::init + process (raw RGBA in, .basis bytes out; PNG decode
    and file writing outside), linked three ways from one source (integration/process_bench.cpp, oracle/Makefile): the stock reference (1 thread =
    the parity configuration, and all host cores), the reference frontend calling the kernels through its accelerator seam, and the whole ETC1S
    path resident on the GPU behind the reference's classes. `identical` compares the .basis bytes with the stock single-threaded run."""
    import subprocess
    import tempfile
    ref_dir = ROOT / "oracle" / "_ref"
    threads = host_cpus()
    mpix = args.size * args.size / 1e6
    pool = min(threads, 8)
    runs = [("stock_1_thread", "process_bench", 1, 0, 1), ("stock_all_cores", "process_bench", threads, 0, 1), ("seam_1_thread", "process_bench_hip", 1, 1, 1),
            ("seam_all_cores", "process_bench_hip", threads, 1, 1), ("resident_1_thread", "process_bench_resident", 1, 1, 3), ("resident", "process_bench_resident", pool, 1, 3)]
    out = {"what": "basis_compressor::init + process(), raw RGBA in, .basis bytes out, tool defaults (ETC1S comp level 1, sRGB metrics). `identical`: same .basis bytes as the "
                   "STOCK reference in the same thread configuration (1 thread = the tool under -no_multithreading; more = the tool's default, whose codebook builders "
                   "partition their trees min(hardware threads, 8, pool) ways, enc.h:2086-2215)", "host_cpus": threads}
    with tempfile.TemporaryDirectory() as d:
        raw = pathlib.Path(d) / "img.rgba"
        np.ascontiguousarray(img).tofile(raw)
        base_hash = {}
        for name, exe, thr, ocl, reps in runs:
            tool = ref_dir / exe
            if not tool.exists():
                continue
            try:
                r = subprocess.run([str(tool), str(raw), str(args.size), str(args.size), str(args.quality), str(args.level), str(thr), str(ocl), str(reps)],
                                   capture_output=True, text=True, timeout=600)
                rec = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:  # a failed variant is reported, not fatal
                out[name] = {"error": str(e)[:200]}
                continue
            best = min(a + b for a, b in zip(rec["init_s"], rec["process_s"]))
            # the reference's codebook-builder thread count for this run (frontend.cpp:2195-2198): the partition count of its selector codebook
            t_codebook = 0 if thr <= 1 else min(os.cpu_count() or 1, 8, thr)
            config = "stock_1_thread" if t_codebook <= 1 else "stock_all_cores"
            if name.startswith("stock"):
                base_hash[name] = (rec["fnv1a64"], t_codebook)
            want = base_hash.get(config)
            out[name] = {"seconds": round(best, 4), "mpix_s": round(mpix / best, 3), "threads": thr, "codebook_threads": t_codebook, "bytes": rec["bytes"],
                         "identical_to": config, "identical": (rec["fnv1a64"] == want[0]) if want and want[1] == t_codebook else None}
        # throughput under the reference's own driver: basis_parallel_compress (comp.cpp:5466-5559) over the resident integration, one basis_compressor + one accelerator
        # context per image in flight -- the host backend of image i runs while the GPU serves the frontends of the others
        tool = ref_dir / "process_bench_resident"
        if tool.exists():
            n_par = max(2, min(16, threads))   # (a sweep on the GPU box's 16 cores: 12 / 16 / 24 images in flight -> 375 / 405 / 390 Mpix/s before the allocator setting below)
            want = base_hash.get("stock_1_thread")   # every task of basis_parallel_compress owns a one-thread pool: the single-threaded codebooks
            # Sixteen compressors allocate and release ~300 MB each per image (the reference's image copies and tile arrays above all); with glibc's defaults that is ~75,000
            # page faults per image, all serialised on one address space's lock -- the faults, not the GPU or the backend, capped this mode at ~390 Mpix/s in round 4
            # (tools/scratch/ab_parallel2.sh). The benchmark APPLICATION sets its own malloc policy (mallopt in integration/process_bench.cpp's main, all link variants alike; the
            # libraries never touch the process's allocator). Second run: the same on transparent huge pages (GLIBC_TUNABLES=glibc.malloc.hugetlb=1), round 4's setting.
            # One host thread per frontend / backend: the images are the parallelism here.
            for key, tun in (("resident_parallel", None), ("resident_parallel_glibc_hugetlb", "glibc.malloc.hugetlb=1")):
                env = dict(os.environ)
                env.setdefault("BU_HOST_THREADS", "1")
                if tun:
                    env.setdefault("GLIBC_TUNABLES", tun)
                try:
                    r = subprocess.run([str(tool), str(raw), str(args.size), str(args.size), str(args.quality), str(args.level), str(n_par), "1", "4", "-", str(n_par)],
                                       capture_output=True, text=True, timeout=600, env=env)
                    rec = json.loads(r.stdout.strip().splitlines()[-1])
                    best = min(rec["call_s"])
                    out[key] = {"driver": "basis_parallel_compress", "images_in_flight": n_par, "images_per_call": n_par, "seconds_per_call": round(best, 4),
                                "mpix_s": round(n_par * mpix / best, 3), "bytes": rec["bytes"], "host_threads_per_frontend": int(env["BU_HOST_THREADS"]),
                                "glibc_tunables": env.get("GLIBC_TUNABLES"),
                                "identical_to": "stock_1_thread", "identical": bool(rec["all_images_identical"]) and (rec["fnv1a64"] == want[0] if want else None)}
                except Exception as e:
                    out[key] = {"error": str(e)[:200]}
    if "resident" in out and "stock_1_thread" in out and "seconds" in out["resident"] and "seconds" in out["stock_1_thread"]:
        out["resident_vs_stock_1_thread"] = round(out["stock_1_thread"]["seconds"] / out["resident"]["seconds"], 1)
        if "seconds" in out.get("stock_all_cores", {}):
            out["resident_vs_stock_all_cores"] = round(out["stock_all_cores"]["seconds"] / out["resident"]["seconds"], 1)
    return out


if __name__ == "__main__":
    main()
