"""ctypes binding of libbasisu_frontend.so: the host-side mirror of the reference's basisu_frontend
(include/basisu_hip_frontend.h -> basis_universal_amd/csrc/host/etc1s_frontend.{h,cpp}).

    ctx = capi.Context()
    fe = Etc1sFrontend(ctx)
    fe.init(blocks, max_endpoint_clusters, max_selector_clusters, compression_level=1, perceptual=True)   # basisu_frontend::init
    fe.compress()                                                                                           # basisu_frontend::compress
    fe.get("encoded_blocks"), fe.get_csr("endpoint_clusters"), ...

`blocks` is either an (n, 4, 4, 4) uint8 numpy array of 4x4 RGBA tiles (uploaded once) or an int device pointer together with
n_blocks= (tiles already resident in HBM, e.g. a torch tensor's data_ptr()).
"""
import ctypes as C
import pathlib

import numpy as np

from . import capi

FRONTEND_LIB_PATH = capi.LIB_DIR / "libbasisu_frontend.so"
_vp = C.c_void_p
_lib = None


def load_frontend_library():
    global _lib
    if _lib is None:
        if not FRONTEND_LIB_PATH.exists():
            raise capi.HipError(f"{FRONTEND_LIB_PATH} is missing: run __graft_entry__.build()")
        capi.load_library()  # libbasisu_hip.so first (the frontend links against it via $ORIGIN rpath)
        L = C.CDLL(str(FRONTEND_LIB_PATH))
        L.bu_frontend_create.restype = _vp
        L.bu_frontend_destroy.argtypes = [_vp]
        L.bu_frontend_init.restype = C.c_int
        L.bu_frontend_init.argtypes = [_vp, _vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.bu_frontend_set_comm.restype = C.c_int
        L.bu_frontend_set_comm.argtypes = [_vp, _vp]
        L.bu_frontend_compress.restype = C.c_int
        L.bu_frontend_compress.argtypes = [_vp]
        L.bu_frontend_call.restype = C.c_int
        L.bu_frontend_call.argtypes = [_vp, C.c_char_p, C.c_uint32]
        L.bu_frontend_get.restype = C.c_uint64
        L.bu_frontend_get.argtypes = [_vp, C.c_char_p, _vp, C.c_uint64]
        L.bu_frontend_error.restype = C.c_char_p
        L.bu_frontend_error.argtypes = [_vp]
        L.bu_frontend_stage_times.restype = C.c_uint32
        L.bu_frontend_stage_times.argtypes = [_vp, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.c_uint32]
        L.bu_etc1s_quality_to_clusters.argtypes = [C.c_int, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.bu_host_tsvq.restype = C.c_int
        L.bu_host_tsvq.argtypes = [C.c_uint32, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint64, _vp, C.c_uint64]
        L.bu_device_tsvq.restype = C.c_int
        L.bu_device_tsvq.argtypes = [_vp, C.c_uint32, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint64, _vp, C.c_uint64, _vp]
        L.bu_host_tsvq_mt.restype = C.c_int
        L.bu_host_tsvq_mt.argtypes = [C.c_uint32, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint64, _vp, C.c_uint64]
        L.bu_device_tsvq_mt.restype = C.c_int
        L.bu_device_tsvq_mt.argtypes = [_vp, C.c_uint32, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint64, _vp, C.c_uint64, _vp]
        L.bu_frontend_set_max_threads.restype = C.c_int
        L.bu_frontend_set_max_threads.argtypes = [_vp, C.c_uint32]
        L.bu_frontend_reference_max_threads.restype = C.c_uint32
        L.bu_frontend_reference_max_threads.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
        L.bu_frontend_pipeline_create.restype = _vp
        L.bu_frontend_pipeline_create.argtypes = [C.c_int, C.c_uint32]
        L.bu_frontend_pipeline_create_n.restype = _vp
        L.bu_frontend_pipeline_create_n.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
        L.bu_frontend_pipeline_submit.restype = C.c_uint64
        L.bu_frontend_pipeline_submit.argtypes = [_vp, _vp, C.c_uint32]
        L.bu_frontend_pipeline_wait.restype = _vp
        L.bu_frontend_pipeline_wait.argtypes = [_vp, C.c_uint64]
        L.bu_frontend_pipeline_poll.restype = C.c_int
        L.bu_frontend_pipeline_poll.argtypes = [_vp, C.c_uint64]
        L.bu_frontend_pipeline_context.restype = _vp
        L.bu_frontend_pipeline_context.argtypes = [_vp, _vp]
        L.bu_frontend_pipeline_release.restype = C.c_int
        L.bu_frontend_pipeline_release.argtypes = [_vp, _vp]
        L.bu_frontend_pipeline_destroy.argtypes = [_vp]
        L.bu_frontend_pipeline_error.restype = C.c_char_p
        L.bu_frontend_pipeline_error.argtypes = [_vp]
        L.bu_frontend_pipeline_stats.restype = C.c_uint32
        L.bu_frontend_pipeline_stats.argtypes = [_vp, C.POINTER(C.c_double), C.c_uint32]
        L.bu_host_last_exception.restype = C.c_char_p
        _lib = L
    return _lib


def reference_max_threads(multithreaded=True, hardware_threads=0, job_pool_threads=0):
    """frontend.cpp:873-876 / 2195-2198: the thread count the reference's codebook builders are handed (0 = single-threaded)."""
    return int(load_frontend_library().bu_frontend_reference_max_threads(int(bool(multithreaded)), int(hardware_threads), int(job_pool_threads)))


def quality_to_clusters(quality_level, total_blocks):
    """comp.cpp:3325-3379: ETC1S quality (1..255) -> (max endpoint clusters, max selector clusters)."""
    ep, sel = C.c_uint32(), C.c_uint32()
    load_frontend_library().bu_etc1s_quality_to_clusters(int(quality_level), int(total_blocks), C.byref(ep), C.byref(sel))
    return ep.value, sel.value


_GATHER_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint64)
_REDUCE_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_uint64)


class _BuComm(C.Structure):  # = bu_comm, include/basisu_hip_frontend.h
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32), ("user", _vp), ("all_gather", _GATHER_FN), ("all_reduce_u64", _REDUCE_FN),
                ("stream_ordered", C.c_uint32), ("reserved", C.c_uint32)]


class _DevicePtr:
    """Zero-copy view of raw device memory for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, nbytes, typestr="|u1", itemsize=1):
        self.__cuda_array_interface__ = {"shape": (nbytes // itemsize,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class TorchComm:
    """The two collectives the sharded frontend needs (bu_comm), on torch.distributed: backend "nccl" is RCCL over xGMI on ROCm;
    "gloo" works too (tests). One instance per process group; keep it alive as long as frontends use it."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.calls = {"all_gather": 0, "all_reduce_u64": 0, "bytes": 0}

        def gather(_user, d_buf, bytes_per_rank):
            try:
                n = int(bytes_per_rank)
                t = torch.as_tensor(_DevicePtr(d_buf, n * self.world), device=self.device)
                outs = [t[r * n:(r + 1) * n] for r in range(self.world)]
                dist.all_gather(outs, outs[self.rank].clone(), group=self.group)
                torch.cuda.synchronize(self.device)
                self.calls["all_gather"] += 1
                self.calls["bytes"] += n * self.world
                return 1
            except Exception as e:  # never let an exception cross the C boundary
                self.error = repr(e)
                return 0

        def reduce(_user, d_buf, count):
            try:
                t = torch.as_tensor(_DevicePtr(d_buf, int(count) * 8, "<i8", 8), device=self.device)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)  # two's complement: the i64 sum is the u64 sum
                torch.cuda.synchronize(self.device)
                self.calls["all_reduce_u64"] += 1
                self.calls["bytes"] += int(count) * 8
                return 1
            except Exception as e:
                self.error = repr(e)
                return 0

        self.error = ""
        self._gather, self._reduce = _GATHER_FN(gather), _REDUCE_FN(reduce)
        self.struct = _BuComm(self.rank, self.world, None, self._gather, self._reduce, 0, 0)   # blocking convention: torch owns its streams


_rccl_lib = None


def load_rccl_library():
    """libbasisu_rccl.so (include/basisu_hip_comm.h): bu_comm on RCCL, no Python in the collective path."""
    global _rccl_lib
    if _rccl_lib is None:
        capi.load_library()   # libbasisu_hip.so first: the communicator library links against it
        path = pathlib.Path(__file__).resolve().parent / "lib" / "libbasisu_rccl.so"
        if not path.exists():
            raise capi.HipError(f"{path} not found: build it (make -C basis_universal_amd/csrc)")
        L = C.CDLL(str(path))
        L.bu_rccl_get_unique_id.restype = C.c_int
        L.bu_rccl_get_unique_id.argtypes = [_vp]
        L.bu_rccl_comm_create.restype = _vp
        L.bu_rccl_comm_create.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32]
        L.bu_rccl_comm_destroy.argtypes = [_vp]
        L.bu_rccl_comm_fill.restype = C.c_int
        L.bu_rccl_comm_fill.argtypes = [_vp, C.POINTER(_BuComm)]
        L.bu_rccl_last_error.restype = C.c_char_p
        _rccl_lib = L
    return _rccl_lib


class RcclComm:
    """bu_comm on a NATIVE RCCL communicator (libbasisu_rccl.so): the collectives are enqueued by C++ on the context's HIP stream. Python only
    hands the communicator's unique id from rank 0 to the other ranks once, through torch.distributed's object broadcast (any channel would do)."""

    def __init__(self, ctx, group=None):
        import torch.distributed as dist
        self.h = None   # close() / __del__ are safe however far __init__ got
        self.L = load_rccl_library()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        ident = C.create_string_buffer(128)
        made = self.rank != 0 or bool(self.L.bu_rccl_get_unique_id(ident))
        box = [bytes(ident.raw) if made else None]   # rank 0 broadcasts even when it has nothing: the other ranks must not wait for an id that never comes
        dist.broadcast_object_list(box, src=0, group=group)
        if box[0] is None:
            raise capi.HipError("bu_rccl_get_unique_id failed on rank 0" + (": " + self.L.bu_rccl_last_error().decode() if self.rank == 0 else ""))
        self.h = self.L.bu_rccl_comm_create(ctx.h, box[0], self.rank, self.world)
        if not self.h:
            raise capi.HipError("bu_rccl_comm_create: " + self.L.bu_rccl_last_error().decode())
        self.struct = _BuComm()
        self.L.bu_rccl_comm_fill(self.h, C.byref(self.struct))
        self.error = ""

    def close(self):
        if getattr(self, "h", None):
            self.L.bu_rccl_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Etc1sFrontend:
    def __init__(self, ctx, comm=None, video=False, fast_codebooks=False, fast_iterations=0, max_threads=0):
        """comm: a TorchComm to shard the device stages over the ranks of its process group (every rank must drive an identical
        frontend on identical tiles); None = single GPU. fast_codebooks: SURVEY 8f row f3 -- both codebooks from a k-means on the matrix
        cores instead of the TSVQ: NOT bit-identical to the reference (see include/basisu_hip_frontend.h), off by default.
        max_threads: the reference's codebook-builder thread count (reference_max_threads()); > 1 reproduces the multi-threaded tool's output
        (T-way partitioned codebooks from 262,144 distinct training vectors up), 0 / 1 the tool under -no_multithreading."""
        self.ctx = ctx
        self.L = load_frontend_library()
        self.h = self.L.bu_frontend_create()
        ctx.adopt(self)  # the frontend's device buffers belong to ctx: it must go first
        self._keep = None
        self.comm = comm
        if video:  # cBASISTexTypeVideoFrames: a different order of stages, see include/basisu_hip_frontend.h
            self.L.bu_frontend_set_video.argtypes = [_vp, C.c_int]
            self._check(self.L.bu_frontend_set_video(self.h, 1), "bu_frontend_set_video")
        if fast_codebooks:
            self.L.bu_frontend_set_fast_codebooks.argtypes = [_vp, C.c_int, C.c_uint32]
            self._check(self.L.bu_frontend_set_fast_codebooks(self.h, 1, int(fast_iterations)), "bu_frontend_set_fast_codebooks")
        if max_threads:
            self._check(self.L.bu_frontend_set_max_threads(self.h, int(max_threads)), "bu_frontend_set_max_threads")
        if comm is not None:
            self.L.bu_frontend_set_comm_sized.argtypes = [_vp, _vp, C.c_uint32]
            self._check(self.L.bu_frontend_set_comm_sized(self.h, C.byref(comm.struct), C.sizeof(comm.struct)), "bu_frontend_set_comm_sized")

    def _check(self, ok, what):
        if not ok:
            raise capi.HipError(f"{what} failed: {self.L.bu_frontend_error(self.h).decode()}")

    def init(self, blocks, max_endpoint_clusters, max_selector_clusters, compression_level=1, perceptual=True, n_blocks=None):
        if isinstance(blocks, np.ndarray):
            blocks = np.ascontiguousarray(blocks, np.uint8)
            self._keep = blocks
            n = blocks.size // 64
            self._check(self.L.bu_frontend_init(self.h, self.ctx.h, blocks.ctypes.data_as(_vp), None, n, max_endpoint_clusters, max_selector_clusters,
                                                compression_level, int(perceptual)), "bu_frontend_init")
        else:
            self._check(self.L.bu_frontend_init(self.h, self.ctx.h, None, _vp(int(blocks)), int(n_blocks), max_endpoint_clusters, max_selector_clusters,
                                                compression_level, int(perceptual)), "bu_frontend_init")

    def compress(self):
        self._check(self.L.bu_frontend_compress(self.h), "bu_frontend_compress")

    def call(self, stage, arg=0):
        self._check(self.L.bu_frontend_call(self.h, stage.encode(), arg), stage)

    def get(self, name, dtype=np.uint8):
        need = self.L.bu_frontend_get(self.h, name.encode(), None, 0)
        if need == 2 ** 64 - 1:
            raise KeyError(name)
        buf = np.zeros(need, np.uint8)
        self.L.bu_frontend_get(self.h, name.encode(), buf.ctypes.data_as(_vp), need)
        return buf.view(dtype)

    def get_csr(self, name):
        blob = self.get(name, np.uint32)
        n = int(blob[0])
        offs = blob[1:n + 2].copy()
        return offs, blob[n + 2:n + 2 + int(offs[-1])].copy()

    def stage_times(self):
        names = (C.c_char_p * 64)()
        secs = (C.c_double * 64)()
        n = min(self.L.bu_frontend_stage_times(self.h, names, secs, 64), 64)
        return [(names[i].decode(), secs[i]) for i in range(n)]

    def close(self):
        if self.h:
            self.L.bu_frontend_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _FrontendJob(C.Structure):  # = bu_frontend_job, include/basisu_hip_frontend.h
    _fields_ = [("h_blocks", _vp), ("d_blocks", _vp), ("n_blocks", C.c_uint32), ("max_endpoint_clusters", C.c_uint32), ("max_selector_clusters", C.c_uint32),
                ("compression_level", C.c_uint32), ("perceptual", C.c_int32), ("max_threads", C.c_uint32), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class PipelinedFrontend(Etc1sFrontend):
    """A finished frontend handed out by FrontendPipeline.wait(): every getter of Etc1sFrontend (and Etc1sBackend.from_frontend) works on it;
    close() gives it -- and the context it ran on -- back to the pipeline."""

    def __init__(self, pipeline, handle, keep):
        self.L = pipeline.L
        self.pipeline = pipeline
        self.h = handle
        self.ctx = None
        self._keep = keep
        self.comm = None

    def close(self):
        if self.h:
            if self.pipeline.h:
                self.L.bu_frontend_pipeline_release(self.pipeline.h, self.h)
            self.h = None


class FrontendPipeline:
    """bu_frontend_pipeline_*: `lanes` ETC1S frontends in flight on one GPU, all driven by the library's ONE driver thread (cooperative tasks; no host thread
    per image as in the reference's basis_parallel_compress, comp.cpp:5466-5559). submit() never blocks; wait(ticket) returns the finished frontend.

        pipe = FrontendPipeline(device=0, lanes=4)
        tickets = [pipe.submit(d_ptr, max_ep, max_sel, n_blocks=n) for d_ptr in images]
        for t in tickets:
            fe = pipe.wait(t); ...fe.get("encoded_blocks")...; fe.close()
        pipe.close()"""

    def __init__(self, device=0, lanes=3, drivers=1):
        lib = capi.load_library()
        if not lib.init(0):
            raise capi.HipError("bu_hip_init failed: " + lib.last_error(None))
        self.L = load_frontend_library()
        self.h = self.L.bu_frontend_pipeline_create_n(int(device), int(lanes), int(drivers))
        if not self.h:
            raise capi.HipError("bu_frontend_pipeline_create failed: " + (self.L.bu_host_last_exception() or b"").decode())
        self.lanes = int(lanes)
        self._keep = {}
        import weakref
        self._out = weakref.WeakSet()   # frontends handed out and not yet closed: destroying the pipeline takes them with it

    def submit(self, blocks, max_endpoint_clusters, max_selector_clusters, compression_level=1, perceptual=True, n_blocks=None, max_threads=0, video=False):
        """blocks: an (n, 4, 4, 4) uint8 array of tiles (uploaded by the job; kept alive until the frontend is closed) or a device pointer with n_blocks=."""
        job = _FrontendJob()
        keep = None
        if isinstance(blocks, np.ndarray):
            keep = np.ascontiguousarray(blocks, np.uint8)
            job.h_blocks, job.n_blocks = keep.ctypes.data, keep.size // 64
        else:
            job.d_blocks, job.n_blocks = int(blocks), int(n_blocks)
        job.max_endpoint_clusters, job.max_selector_clusters = int(max_endpoint_clusters), int(max_selector_clusters)
        job.compression_level, job.perceptual, job.max_threads, job.flags = int(compression_level), int(bool(perceptual)), int(max_threads), 1 if video else 0
        t = int(self.L.bu_frontend_pipeline_submit(self.h, C.byref(job), C.sizeof(job)))
        if not t:
            raise capi.HipError("bu_frontend_pipeline_submit refused the job: " + (self.L.bu_host_last_exception() or b"").decode())
        self._keep[t] = keep
        return t

    def poll(self, ticket):
        return self.L.bu_frontend_pipeline_poll(self.h, int(ticket)) == 1

    def wait(self, ticket):
        h = self.L.bu_frontend_pipeline_wait(self.h, int(ticket))
        keep = self._keep.pop(int(ticket), None)
        if not h:
            raise capi.HipError("frontend pipeline: " + self.L.bu_frontend_pipeline_error(self.h).decode())
        fe = PipelinedFrontend(self, h, keep)
        self._out.add(fe)
        return fe

    def stats(self):
        v = (C.c_double * 7)()
        self.L.bu_frontend_pipeline_stats(self.h, v, 7)
        return dict(zip(("jobs", "task_switches", "yields", "idle_naps", "driver_busy_s", "driver_idle_s", "driver_cpu_s"), [float(x) for x in v]))

    def close(self):
        if getattr(self, "h", None):
            for fe in list(self._out):   # bu_frontend_pipeline_destroy releases what nobody released: those handles are dead afterwards
                fe.h = None
            self.L.bu_frontend_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
