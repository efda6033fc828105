#!/usr/bin/env python3
"""Throughput of N ETC1S frontends in flight on one GPU (4096^2 q128 level 1, tiles resident), for a list of N, in one process: one
host thread + context (= HIP stream) per image in flight, as bench.py's `pipelined` leg. Prints one JSON line per N: Mpix/s, host CPU
seconds per image, every image's state against the reference's digest. Environment (GPU_MAX_HW_QUEUES, BU_TSVQ_POLL, ...) comes from
the caller, so A/B runs are separate invocations.
    inflight_probe.py [--streams 1,2,3,4,6] [--per-stream 4] [--size 4096] [--null-stream]"""
import argparse
import json
import os
import pathlib
import sys
import threading
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", default="1,2,3,4,6")
    ap.add_argument("--per-stream", type=int, default=4)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--quality", type=int, default=128)
    ap.add_argument("--null-stream", action="store_true", help="context 0 on torch's current (null) stream as bench.py's headline context is")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--pipeline", action="store_true", help="bu_frontend_pipeline_* (one driver thread, cooperative tasks) instead of one host thread per image in flight")
    ap.add_argument("--drivers", type=int, default=1, help="driver threads of the pipeline (bu_frontend_pipeline_create_n)")
    ap.add_argument("--threads", type=int, default=0, help="the reference's codebook thread configuration (bu_frontend_set_max_threads)")
    args = ap.parse_args()
    os.environ.setdefault("BU_HIP_PARKED_CONTEXTS", "32")
    import numpy as np
    import torch
    import helpers
    from basis_universal_amd import capi
    from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters
    import test_gpu_etc1s_frontend as T

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    w = h = args.size
    blocks = helpers.to_pixel_blocks(helpers.synth(w, h, 1234))
    n_blocks = blocks.shape[0]
    d_blocks = torch.from_numpy(blocks.reshape(n_blocks, 64)).to(dev)
    max_ep, max_sel = quality_to_clusters(args.quality, n_blocks)
    want = None
    if not args.no_check and (w, args.quality) == (4096, 128):
        want = json.loads((ROOT / "tests" / "golden" / "etc1s_big_digests.json").read_text())["synth4096_q128"]["frontend_digests"]
    counts = [int(s) for s in args.streams.split(",")]
    if args.pipeline:
        from basis_universal_amd.etc1s import FrontendPipeline
        if args.threads:
            want = json.loads((ROOT / "tests" / "golden" / "etc1s_big_digests.json").read_text())[f"synth4096_q128_t{args.threads}"]["frontend_digests"] if want else None
        for n in counts:
            pipe = FrontendPipeline(0, n, min(args.drivers, n))
            images = n * args.per_stream
            for rep in range(2):   # warm as many contexts as results will be held at once below (a job's context stays with its frontend until that is closed)
                for fe in [pipe.wait(t) for t in [pipe.submit(d_blocks.data_ptr(), max_ep, max_sel, 1, True, n_blocks=n_blocks, max_threads=args.threads) for _ in range(images)]]:
                    fe.close()
            s0 = pipe.stats()
            torch.cuda.synchronize()
            cpu0, t0 = time.process_time(), time.perf_counter()
            tickets = [pipe.submit(d_blocks.data_ptr(), max_ep, max_sel, 1, True, n_blocks=n_blocks, max_threads=args.threads) for _ in range(images)]
            done = [pipe.wait(t) for t in tickets]
            torch.cuda.synchronize()
            dt, cpu = time.perf_counter() - t0, time.process_time() - cpu0
            s1 = pipe.stats()
            same = all(T._digest({k: fe.get(k) for k in want}) == want for fe in done) if want else None
            for fe in done:
                fe.close()
            pipe.close()
            print(json.dumps({"mode": "pipeline", "drivers": min(args.drivers, n), "in_flight": n, "images": images, "value": round(images * w * h / 1e6 / dt, 1), "unit": "Mpixels/s", "ms_per_image": round(dt / images * 1e3, 2),
                              "host_cpu_s_per_image": round(cpu / images, 4), "identical_to_reference": same,
                              "driver": {k: round((s1[k] - s0[k]) / images, 5) for k in s1 if k != "jobs"},
                              "env": {k: os.environ[k] for k in ("GPU_MAX_HW_QUEUES", "BU_PIPELINE_SPIN_US", "BU_PIPELINE_SLEEP_US", "BU_HOST_THREADS") if k in os.environ}}), flush=True)
        return
    ctxs = []
    for i in range(max(counts)):
        c = capi.Context(0)
        if i == 0 and args.null_stream:
            c.check(c.lib.set_stream(c.h, torch.cuda.current_stream().cuda_stream), "set_stream")
        for _ in range(2):   # warm its pools
            fe = Etc1sFrontend(c)
            fe.init(d_blocks.data_ptr(), max_ep, max_sel, 1, True, n_blocks=n_blocks)
            fe.compress()
            fe.close()
        ctxs.append(c)
    for n in counts:
        images = n * args.per_stream
        todo, lock, done = list(range(images)), threading.Lock(), []

        def worker(c):
            while True:
                with lock:
                    if not todo:
                        return
                    todo.pop()
                fe = Etc1sFrontend(c)
                fe.init(d_blocks.data_ptr(), max_ep, max_sel, 1, True, n_blocks=n_blocks)
                fe.compress()
                with lock:
                    done.append(fe)

        torch.cuda.synchronize()
        cpu0, t0 = time.process_time(), time.perf_counter()
        th = [threading.Thread(target=worker, args=(c,)) for c in ctxs[:n]]
        [t.start() for t in th]
        [t.join() for t in th]
        torch.cuda.synchronize()
        dt, cpu = time.perf_counter() - t0, time.process_time() - cpu0
        same = None
        if want:
            same = all(T._digest({k: fe.get(k) for k in want}) == want for fe in done)
        for fe in done:
            fe.close()
        print(json.dumps({"in_flight": n, "images": images, "value": round(images * w * h / 1e6 / dt, 1), "unit": "Mpixels/s", "ms_per_image": round(dt / images * 1e3, 2),
                          "host_cpu_s_per_image": round(cpu / images, 4), "identical_to_reference": same,
                          "env": {k: os.environ[k] for k in ("GPU_MAX_HW_QUEUES", "BU_TSVQ_POLL", "BU_HOST_THREADS") if k in os.environ}}), flush=True)
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
