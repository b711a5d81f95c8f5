"""ctypes binding of ``lib/libdewi_hip.so`` (C ABI: ``include/dewi_hip.h``).

This is the only door between the Python host layer and the HIP kernels.  There is no
CPU implementation behind it: if the shared library is missing or no GPU is visible,
every hot-path call raises ``NativeLibraryError``.

torch is imported first on purpose: PyTorch-ROCm ships its own ``libamdhip64.so`` and
our library must bind to that same runtime instance (same SONAME), otherwise stream
handles and device pointers from torch would belong to a different HIP runtime.
"""
from __future__ import annotations

import ctypes
import os
import threading
from pathlib import Path
from typing import Optional

PKG_ROOT = Path(__file__).resolve().parent.parent
LIB_PATH = PKG_ROOT / "lib" / "libdewi_hip.so"

OK = 0
ERR_INVALID_ARG = -1
ERR_K_OUT_OF_BOUNDS = -2
ERR_WORKSPACE = -3
ERR_HIP = -4
ERR_UNSUPPORTED = -5

SPACE_CODES = {"cosine": 0, "l2": 1}
MODE_CODES = {"standard": 0, "conditional": 1}
#: similarity the ANN-semantics re-rank blends (include/dewi_hip.h DEWI_SIM_*; reference backends.py:229-231, 335-338)
SIM_CODES = {"ip": 0, "one_minus_dist": 1, "inv_one_plus_dist": 2}
NUM_SIGNALS = 7
ABI_VERSION = 5

#: every symbol include/dewi_hip.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = (
    "dewi_abi_version", "dewi_last_error", "dewi_device_info", "dewi_normalize_rows_f32",
    "dewi_row_cosine_f32", "dewi_convert_f32_to_bf16", "dewi_payload_soa_f64", "dewi_knn_workspace_bytes", "dewi_knn_scan_kernel", "dewi_knn_refusal_flags", "dewi_knn_rerank_f32", "dewi_knn_rerank_f32_shadow",
    "dewi_knn_rerank_bf16", "dewi_knn_rerank_candidates", "dewi_prepare_queries_bf16", "dewi_knn_scan", "dewi_knn_finish", "dewi_knn_candidates", "dewi_merge_workspace_bytes", "dewi_merge_rerank", "dewi_robust_fit_workspace_bytes",
    "dewi_robust_fit_f32", "dewi_robust_fit_begin", "dewi_robust_fit_hist_f32", "dewi_robust_fit_region",
    "dewi_robust_fit_pick", "dewi_robust_fit_finish", "dewi_score_f64", "dewi_score_f64_dev", "dewi_timing_enable", "dewi_timing_read", "dewi_tuning_set",
)


class NativeLibraryError(RuntimeError):
    """The HIP extension is missing, stale, or no MI355X is visible."""


class DewiCandidate(ctypes.Structure):
    _fields_ = [("sim", ctypes.c_float), ("dewi", ctypes.c_float), ("ent", ctypes.c_float), ("id", ctypes.c_int32)]


_lock = threading.Lock()
_lib: Optional[ctypes.CDLL] = None


def _declare(lib: ctypes.CDLL) -> None:
    c = ctypes
    vp, i32, i64, f64, sz = c.c_void_p, c.c_int, c.c_int64, c.c_double, c.c_size_t
    lib.dewi_abi_version.restype = i32
    lib.dewi_abi_version.argtypes = []
    lib.dewi_last_error.restype = c.c_char_p
    lib.dewi_last_error.argtypes = []
    lib.dewi_device_info.restype = i32
    lib.dewi_device_info.argtypes = [c.POINTER(i32), c.POINTER(i32), c.POINTER(sz)]
    lib.dewi_normalize_rows_f32.restype = i32
    lib.dewi_normalize_rows_f32.argtypes = [vp, vp, i64, i32, vp]
    lib.dewi_row_cosine_f32.restype = i32
    lib.dewi_row_cosine_f32.argtypes = [vp, vp, vp, i64, i32, vp]
    lib.dewi_convert_f32_to_bf16.restype = i32
    lib.dewi_convert_f32_to_bf16.argtypes = [vp, vp, i64, vp]
    lib.dewi_payload_soa_f64.restype = i32
    lib.dewi_payload_soa_f64.argtypes = [vp, vp, vp, vp, vp, i64, vp]
    lib.dewi_knn_workspace_bytes.restype = sz
    lib.dewi_knn_workspace_bytes.argtypes = [i64, i32, i32, i32]
    knn = [vp, i64, i32, vp, i32, vp, vp, i32, f64, f64, i32, vp, vp, vp, sz, vp]
    lib.dewi_knn_rerank_f32.restype = i32
    lib.dewi_knn_rerank_f32.argtypes = knn
    lib.dewi_knn_rerank_bf16.restype = i32
    lib.dewi_knn_rerank_bf16.argtypes = knn
    lib.dewi_knn_rerank_candidates.restype = i32
    lib.dewi_knn_rerank_candidates.argtypes = [vp, i32, i64, i32, vp, i32, vp, vp, i32, i32, f64, f64, i32, i32, vp, vp, vp, sz, vp]
    lib.dewi_prepare_queries_bf16.restype = i32
    lib.dewi_prepare_queries_bf16.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.dewi_knn_scan.restype = i32
    lib.dewi_knn_scan.argtypes = [vp, i32, i64, i32, vp, i32, i32, i32, vp, sz, vp]
    lib.dewi_knn_finish.restype = i32
    lib.dewi_knn_finish.argtypes = [vp, sz, vp, i32, i64, i32, vp, i32, i32, i32, i32, f64, f64, vp, vp, i64, vp, vp, vp, vp]
    lib.dewi_knn_refusal_flags.restype = i32
    lib.dewi_knn_refusal_flags.argtypes = [i32, i32, i64, i32, i32, i32, i32, i32, c.POINTER(sz)]
    lib.dewi_knn_scan_kernel.restype = i32
    lib.dewi_knn_scan_kernel.argtypes = [i32, i64, i32, i32, i32, i32, c.c_char_p, sz]
    lib.dewi_knn_candidates.restype = i32
    lib.dewi_knn_candidates.argtypes = [vp, i32, i64, i32, vp, i32, vp, vp, i32, i32, i64, vp, vp, sz, vp]
    lib.dewi_merge_rerank.restype = i32
    lib.dewi_merge_rerank.argtypes = [vp, i32, i32, i32, i32, i32, f64, f64, vp, vp, vp, sz, vp]
    lib.dewi_merge_workspace_bytes.restype = sz
    lib.dewi_merge_workspace_bytes.argtypes = [i32, i32, i32, i32]
    lib.dewi_robust_fit_workspace_bytes.restype = sz
    lib.dewi_robust_fit_workspace_bytes.argtypes = [i32]
    lib.dewi_robust_fit_f32.restype = i32
    lib.dewi_robust_fit_f32.argtypes = [vp, i64, i64, i32, vp, vp, vp, sz, vp]
    lib.dewi_robust_fit_begin.restype = i32
    lib.dewi_robust_fit_begin.argtypes = [i32, vp, sz, vp]
    lib.dewi_robust_fit_hist_f32.restype = i32
    lib.dewi_robust_fit_hist_f32.argtypes = [vp, i64, i64, i32, i32, i32, vp, vp, sz, vp]
    lib.dewi_robust_fit_region.restype = i32
    lib.dewi_robust_fit_region.argtypes = [i32, i32, i32, i32, c.POINTER(sz), c.POINTER(sz)]
    lib.dewi_robust_fit_pick.restype = i32
    lib.dewi_robust_fit_pick.argtypes = [i64, i32, i32, i32, vp, sz, vp]
    lib.dewi_robust_fit_finish.restype = i32
    lib.dewi_robust_fit_finish.argtypes = [i64, i32, i32, vp, sz, vp, vp]
    lib.dewi_score_f64.restype = i32
    lib.dewi_score_f64.argtypes = [vp, i32, i64, i64, c.POINTER(f64), c.POINTER(f64), c.POINTER(f64), f64, i32, vp,
                                   vp, vp]
    lib.dewi_knn_rerank_f32_shadow.restype = i32
    lib.dewi_knn_rerank_f32_shadow.argtypes = [vp, vp, i64, i32, vp, i32, vp, vp, i32, f64, f64, i32, vp, vp, vp, sz, vp]
    lib.dewi_score_f64_dev.restype = i32
    lib.dewi_score_f64_dev.argtypes = [vp, i32, i64, i64, vp, vp, c.POINTER(f64), f64, i32, vp, vp, vp]
    lib.dewi_timing_enable.restype = i32
    lib.dewi_timing_enable.argtypes = [i32]
    lib.dewi_timing_read.restype = i32
    lib.dewi_timing_read.argtypes = [c.POINTER(f64), c.POINTER(i32)]
    lib.dewi_tuning_set.restype = i32
    lib.dewi_tuning_set.argtypes = [i32, i32, i32, i32]


def load_library(require_gpu: bool = True) -> ctypes.CDLL:
    """Load (once) and return the C-ABI library.

    ``require_gpu=False`` is for the CPU-only checks that the library loads and exports
    its symbols; every compute entry point needs a device and is never called there.
    """
    global _lib
    with _lock:
        if _lib is None:
            path = Path(os.environ.get("DEWI_HIP_LIB", str(LIB_PATH)))
            if not path.exists():
                raise NativeLibraryError(
                    f"{path} not found: build it with `make -C {PKG_ROOT / 'csrc'}` (or `python -c 'import "
                    f"__graft_entry__ as g; g.build()'`).  There is no CPU fallback for the DEWI hot path.")
            import torch  # noqa: F401  (loads torch's libamdhip64.so first, see module docstring)
            try:
                lib = ctypes.CDLL(str(path), mode=ctypes.RTLD_GLOBAL)
            except OSError as e:  # pragma: no cover
                raise NativeLibraryError(f"cannot load {path}: {e}") from e
            missing = [s for s in EXPORTED_SYMBOLS if not hasattr(lib, s)]
            if missing:
                raise NativeLibraryError(f"{path} is stale: missing symbols {missing}; rebuild it")
            _declare(lib)
            if lib.dewi_abi_version() != ABI_VERSION:
                raise NativeLibraryError(f"{path}: ABI version {lib.dewi_abi_version()} != {ABI_VERSION}; rebuild it")
            _lib = lib
    if require_gpu:
        import torch
        if not torch.cuda.is_available():
            raise NativeLibraryError("no GPU visible to PyTorch-ROCm: the DEWI hot path runs only on an MI355X "
                                     "(there is no CPU fallback)")
    return _lib


def last_error() -> str:
    return load_library(require_gpu=False).dewi_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    """Translate a status code into the exception the reference would raise."""
    if rc == OK:
        return
    msg = last_error()
    if rc in (ERR_K_OUT_OF_BOUNDS, ERR_INVALID_ARG):
        raise ValueError(msg)            # reference: ValueError out of NumPy / shape checks
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise NativeLibraryError(f"dewi_hip error {rc}: {msg}")


def is_device_tensor(x) -> bool:
    """True for a CUDA torch tensor (without importing torch for plain arrays)."""
    return bool(getattr(x, "is_cuda", False))


def stream_ptr() -> int:
    import torch
    return int(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> int:
    """Device pointer of a contiguous torch tensor (0 for None)."""
    if t is None:
        return 0
    # (pinned host tensors are device-visible at their own address on ROCm: result buffers may live there)
    assert (t.is_cuda or t.is_pinned()) and t.is_contiguous()
    return int(t.data_ptr())
