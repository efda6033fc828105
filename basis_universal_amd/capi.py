"""ctypes binding of libbasisu_hip.so (include/basisu_hip.h).

The binding mirrors the C ABI one-to-one; nothing here computes anything. `HipLibrary` can be constructed without a GPU
(so that CPU-only CI can check that the library loads and exports every declared symbol); creating a context without a GPU
raises HipError.
"""
import weakref
import ctypes as C
import os
import pathlib
import re

import numpy as np

PKG_DIR = pathlib.Path(__file__).resolve().parent
LIB_DIR = pathlib.Path(os.environ.get("BU_HIP_LIB_DIR", PKG_DIR / "lib"))  # override: developer experiments with variant builds
LIB_PATH = LIB_DIR / "libbasisu_hip.so"
HEADER_PATH = PKG_DIR.parent / "include" / "basisu_hip.h"

_vp = C.c_void_p
_u32 = C.c_uint32
_int = C.c_int


class HipError(RuntimeError):
    pass


def declared_symbols(header=HEADER_PATH):
    """Every BU_HIP_API function name declared in include/basisu_hip.h."""
    txt = pathlib.Path(header).read_text()
    return sorted(set(re.findall(r"BU_HIP_API[^;(]*?\b(bu_hip_\w+)\s*\(", txt)))


_SIGNATURES = {
    # section 1
    "bu_hip_init": (_int, [_int]),
    "bu_hip_deinit": (None, []),
    "bu_hip_is_available": (_int, []),
    "bu_hip_create_context": (_vp, []),
    "bu_hip_destroy_context": (None, [_vp]),
    "bu_hip_set_pixel_blocks": (_int, [_vp, C.c_size_t, _vp]),
    "bu_hip_encode_etc1s_blocks": (_int, [_vp, _vp, _int, _u32]),
    "bu_hip_encode_etc1s_pixel_clusters": (_int, [_vp, _vp, _u32, _vp, C.c_uint64, _vp, _vp, _int, _u32]),
    "bu_hip_refine_endpoint_clusterization": (_int, [_vp, _vp, _u32, _vp, _vp, _vp, _int]),
    "bu_hip_find_optimal_selector_clusters_for_each_block": (_int, [_vp, _vp, _u32, _vp, _vp, _vp, _int]),
    "bu_hip_determine_selectors": (_int, [_vp, _vp, _vp, _int]),
    "bu_hip_encode_uastc_blocks": (_int, [_vp, _vp, _u32]),
    # section 2
    "bu_hip_create_context_on": (_vp, [_int]),
    "bu_hip_on_destroy": (_int, [_vp, _vp, _vp]),
    "bu_hip_memcpy_d2d": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "bu_hip_k_map_blocks_from_groups": (_int, [_vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bu_hip_k_map_rank_blocks": (_int, [_vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "bu_hip_k_map_endpoint_csr": (_int, [_vp, _vp, _vp, _u32, _vp, _vp]),
    "bu_hip_k_map_remap": (_int, [_vp, _vp, _vp, _u32, _vp, _vp]),
    "bu_hip_k_map_count_differences": (_int, [_vp, _vp, _vp, _u32, _vp]),
    "bu_hip_k_map_membership": (_int, [_vp, _vp, _vp, _u32, _u32, _u32, _vp]),
    "bu_hip_k_map_gather": (_int, [_vp, _vp, _vp, _u32, _vp]),
    "bu_hip_tsvq_scatter_spans": (_int, [_vp, _vp, _vp, _u32, _vp]),
    "bu_hip_tsvq_exchange_pack": (_int, [_vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp]),
    "bu_hip_tsvq_exchange_unpack": (_int, [_vp, _vp, _vp, _vp, _vp, _u32]),
    "bu_hip_kmeans_codebook": (_int, [_vp, _int, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp]),
    "bu_hip_cancel_on_destroy": (None, [_vp, _vp, _vp]),
    "bu_hip_context_device": (_int, [_vp]),
    "bu_hip_set_stream": (_int, [_vp, _vp]),
    "bu_hip_get_stream": (_vp, [_vp]),
    "bu_hip_sync": (_int, [_vp]),
    "bu_hip_set_wait_hook": (_int, [_vp, _vp, _vp]),
    "bu_hip_get_tuning": (None, [_vp, _vp, _u32]),
    "bu_hip_set_tuning": (_int, [_vp, _vp]),
    "bu_hip_last_error": (C.c_char_p, [_vp]),
    "bu_hip_profile_enable": (_int, [_vp, _int]),
    "bu_hip_profile_read": (_u32, [_vp, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(_u32), _u32]),
    "bu_hip_malloc": (_vp, [_vp, C.c_size_t]),
    "bu_hip_free": (None, [_vp, _vp]),
    "bu_hip_memcpy_h2d": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "bu_hip_memcpy_h2d_async": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "bu_hip_memcpy_d2h": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "bu_hip_memset": (_int, [_vp, _vp, _int, C.c_size_t]),
    "bu_hip_set_pixel_blocks_device": (_int, [_vp, C.c_size_t, _vp]),
    "bu_hip_get_pixel_blocks_device": (_vp, [_vp, C.POINTER(C.c_size_t)]),
    "bu_hip_k_extract_blocks": (_int, [_vp, _vp, _u32, _u32, _u32, _vp]),
    "bu_hip_k_encode_etc1s_blocks": (_int, [_vp, _vp, _u32, _int, _int, _vp]),
    "bu_hip_k_endpoint_training_vectors": (_int, [_vp, _vp, _u32, _vp]),
    "bu_hip_k_generate_endpoint_codebook": (_int, [_vp, _vp, _u32, _vp, _vp, _vp, _int, _int, _u32, _vp, _vp, _vp]),
    "bu_hip_k_generate_endpoint_codebook_part": (_int, [_vp, _vp, _u32, _vp, _vp, _vp, _int, _int, _u32, _vp, _vp, _vp, _u32, _u32]),
    "bu_hip_k_refit_endpoints_given_selectors": (_int, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp]),
    "bu_hip_k_resample_rgba8": (_int, [_vp, _vp, _u32, _u32, _vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _u32]),
    "bu_hip_k_refit_endpoints_given_selectors_q": (_int, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp]),
    "bu_hip_k_subblock_errors": (_int, [_vp, _vp, _u32, _vp, _vp, _int, _vp]),
    "bu_hip_k_backend_block_errors": (_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _int, _int, _vp, _vp]),
    "bu_hip_k_refine_endpoint_clusterization": (_int, [_vp, _vp, _u32, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _int, _vp]),
    "bu_hip_k_determine_selectors": (_int, [_vp, _vp, _u32, _vp, _vp, _int, _vp]),
    "bu_hip_k_selector_training_vectors": (_int, [_vp, _vp, _u32, _int, _vp, _vp]),
    "bu_hip_k_create_optimized_selector_codebook": (_int, [_vp, _vp, _vp, _u32, _vp, _vp, _int, _vp]),
    "bu_hip_k_find_optimal_selector_clusters": (_int, [_vp, _vp, _vp, _u32, _vp, _u32, _u32, _vp, _vp, _vp, _int, _u32, _vp]),
    "bu_hip_k_encode_uastc_blocks": (_int, [_vp, _vp, _u32, _u32, _vp]),
    "bu_hip_uastc_workspace_bytes": (C.c_size_t, [_u32, _u32]),
    "bu_hip_tsvq_create_packed16_device": (_vp, [_vp, _vp, _vp, _u32, _vp]),
    "bu_hip_k_unique_endpoint_vectors": (_int, [_vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "bu_hip_k_unique_selector_vectors": (_int, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp]),
    "bu_hip_uastc_rdo_default_params": (None, [_vp]),
    "bu_hip_k_uastc_rdo": (_int, [_vp, _vp, _vp, _u32, _vp, _u32, _u32, _vp]),
    "bu_hip_uastc_rdo": (_int, [_vp, _vp, _vp, _u32, _u32, _vp]),
    "bu_hip_tsvq_create": (_vp, [_vp, _u32, _vp, _vp, _u32, _vp]),
    "bu_hip_tsvq_create_packed16": (_vp, [_vp, _vp, _vp, _u32, _vp]),
    "bu_hip_k_cluster_colour_means": (_int, [_vp, _vp, _u32, _vp, _vp, _vp]),
    "bu_hip_k_upload_and_encode_etc1s_blocks": (_int, [_vp, _vp, _vp, _u32, _int, _int, _vp]),
    "bu_hip_host_alloc": (_vp, [C.c_size_t]),
    "bu_hip_host_free": (None, [_vp]),
    "bu_hip_download_begin": (_vp, [_vp, _vp, _vp, C.c_size_t]),
    "bu_hip_download_wait": (_int, [_vp]),
    "bu_hip_tsvq_split": (_int, [_vp, _vp, _vp, _u32, _vp]),
    "bu_hip_tsvq_split_deep": (_int, [_vp, _vp, _vp, _u32, _vp, _u32, _vp]),
    "bu_hip_tsvq_roots": (_int, [_vp, _vp, _vp, _u32, _vp]),
    "bu_hip_tsvq_finish_spans": (_int, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp]),
    "bu_hip_tsvq_create_endpoint_device": (_vp, [_vp, _vp, _vp, _u32, _vp]),
    "bu_hip_uastc_pipeline_create": (_vp, [_vp, _u32, _u32, _u32, _u32]),
    "bu_hip_uastc_pipeline_submit": (_int, [_vp, _vp, _u32, _vp, _vp, _u32, _u32, _vp]),
    "bu_hip_uastc_pipeline_wait": (_int, [_vp, C.c_uint64, _vp]),
    "bu_hip_uastc_pipeline_destroy": (None, [_vp]),
    "bu_hip_tsvq_read_members": (_int, [_vp, _vp, _u32, _u32, _u32, _vp]),
    "bu_hip_tsvq_destroy": (None, [_vp, _vp]),
}


class Tuning(C.Structure):  # = bu_hip_tuning, include/basisu_hip.h
    _fields_ = [(n, C.c_uint32) for n in ("struct_bytes", "tsvq_wide_min", "tsvq_wide6_min", "tsvq_wide_cov_min", "tsvq_windows", "tsvq_dense_min", "tsvq_zero_copy",
                                          "tsvq_chained_only", "tsvq_poll", "refine_unsorted", "debug", "tsvq_deep_levels", "uastc_walk_cus", "codebook_wide_min")]


class HipLibrary:
    def __init__(self, path=LIB_PATH):
        path = pathlib.Path(path)
        if not path.exists():
            raise HipError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"(there is no CPU fallback)")
        self.path = path
        self.dll = C.CDLL(str(path))
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(self.dll, name)  # AttributeError = symbol missing = broken build
            fn.restype = res
            fn.argtypes = args
            if name != "bu_hip_last_error":
                setattr(self, name[len("bu_hip_"):], fn)

    def last_error(self, ctx=None):
        s = self.dll.bu_hip_last_error(ctx)
        return s.decode() if s else ""


_lib = None


def load_library():
    global _lib
    if _lib is None:
        _lib = HipLibrary()
    return _lib


class Context:
    """One bu_hip_context (= one HIP stream + resident buffers). Raises HipError if no GPU is usable."""

    def __init__(self, device=None, lib=None):
        self.lib = lib or load_library()
        if not self.lib.init(0):
            raise HipError("bu_hip_init failed: " + self.lib.last_error(None))
        self.h = self.lib.create_context() if device is None else self.lib.create_context_on(int(device))
        if not self.h:
            raise HipError("bu_hip_create_context failed: " + self.lib.last_error(None))
        self._dependants = weakref.WeakSet()

    def adopt(self, obj):
        """Registers an object that owns device memory of this context (it must have close()): closing the context closes it first,
        so that a dependant collected later never frees through a dead context."""
        self._dependants.add(obj)

    def check(self, ok, what=""):
        if not ok:
            raise HipError(f"{what} failed: {self.lib.last_error(self.h)}")

    def tuning(self):
        """bu_hip_get_tuning: this context's path-selection knobs as a dict (all paths are bit-identical; see include/basisu_hip.h)."""
        t = Tuning()
        self.lib.get_tuning(self.h, C.byref(t), C.sizeof(t))
        return {n: getattr(t, n) for n, _ in Tuning._fields_ if n != "struct_bytes"}

    def set_tuning(self, **fields):
        """bu_hip_set_tuning: the process defaults with `fields` replaced (no arguments = back to the defaults); codebook builds started on this context afterwards take them."""
        t = Tuning()
        self.lib.get_tuning(None, C.byref(t), C.sizeof(t))
        for k, v in fields.items():
            if k not in dict(Tuning._fields_) or k == "struct_bytes":
                raise KeyError(k)
            setattr(t, k, int(v))
        t.struct_bytes = C.sizeof(t)
        self.check(self.lib.set_tuning(self.h, C.byref(t)), "bu_hip_set_tuning")

    def close(self):
        if getattr(self, "h", None):
            for d in list(getattr(self, "_dependants", ())):
                d.close()
            self.lib.destroy_context(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- raw device memory helpers (used by tests and the host frontend; torch tensors can be passed by data_ptr instead)
    def alloc(self, nbytes):
        p = self.lib.malloc(self.h, int(nbytes))
        if not p:
            raise HipError("bu_hip_malloc failed: " + self.lib.last_error(self.h))
        return p

    def free(self, p):
        self.lib.free(self.h, p)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.alloc(max(arr.nbytes, 1))
        if arr.nbytes:
            self.check(self.lib.memcpy_h2d(self.h, p, arr.ctypes.data_as(_vp), arr.nbytes), "memcpy_h2d")
        return p

    def download(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        self.check(self.lib.memcpy_d2h(self.h, out.ctypes.data_as(_vp), p, out.nbytes), "memcpy_d2h")
        return out

    def profile_enable(self, on=True):
        self.check(self.lib.profile_enable(self.h, int(on)), "profile_enable")

    def profile_read(self):
        """{kernel name: (total ms, launches)} from HIP events on the launch stream since profile_enable(True)."""
        names = (C.c_char_p * 32)(); ms = (C.c_double * 32)(); cnt = (_u32 * 32)()
        n = min(self.lib.profile_read(self.h, names, ms, cnt, 32), 32)
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(n)}

    def sync(self):
        self.check(self.lib.sync(self.h), "sync")
