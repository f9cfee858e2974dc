"""ctypes binding of ``libmf_hip.so`` (C ABI: ``include/mf_hip.h``).

The product path has no CPU fallback: if the library is missing, or a tensor is not
a contiguous fp32/int64 tensor on the GPU, the call raises.
"""
from __future__ import annotations

import ctypes
import os
import pathlib
import subprocess

import torch

_PKG = pathlib.Path(__file__).resolve().parent
LIB_PATH = pathlib.Path(os.environ.get("MF_HIP_LIB", _PKG / "lib" / "libmf_hip.so"))   # override: A/B builds

c_i64, c_int, c_f32, c_vp, c_sz = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/mf_hip.h declares
SIGNATURES = {
    "mf_last_error": (ctypes.c_char_p, []),
    "mf_version": (c_int, []),
    "mf_timing_enable": (None, [c_int]),
    "mf_timing_reset": (None, []),
    "mf_timing_get": (c_i64, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)]),
    "mf_gather_rows": (c_int, [c_vp, c_i64, c_int, c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    "mf_gather_hashed": (c_int, [c_vp, c_i64, c_int, c_vp, c_i64, c_int, ctypes.c_uint64, c_int, c_vp, c_vp, c_vp]),
    "mf_hash_buckets": (c_int, [c_vp, c_i64, c_int, ctypes.c_uint64, c_i64, c_vp, c_vp]),
    "mf_normalize_backward": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp]),
    "mf_row_sqnorm": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp]),
    "mf_scores": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_vp]),
    "mf_sort_ws_bytes": (c_sz, [c_i64]),
    "mf_sort_keys": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mf_group_keys": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp]),
    "mf_loss_ws_bytes": (c_sz, [c_i64, c_i64, c_int, c_int, c_int]),
    "mf_loss_masks": (c_int, [c_i64, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mf_loss_fwd": (c_int, [c_i64, c_i64, c_int, c_int, c_int, c_f32, c_f32, c_int, c_vp, c_vp, c_vp, c_vp,
                            c_vp, c_vp, c_i64, c_int, c_vp, c_sz, c_vp, c_vp, c_vp]),
    "mf_loss_masks_csr": (c_int, [c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_sz, c_vp]),
    "mf_loss_fwd_csr": (c_int, [c_i64, c_i64, c_int, c_int, c_f32, c_f32, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64,
                                c_vp, c_i64, c_int, c_vp, c_sz, c_vp, c_vp, c_vp]),
    "mf_loss_bwd": (c_int, [c_i64, c_i64, c_int, c_int, c_int, c_f32, c_f32, c_int, c_vp, c_vp, c_int,
                            c_vp, c_sz, c_vp, c_vp, c_vp, c_vp]),
    "mf_negative_masks_ws_bytes": (c_sz, [c_i64, c_i64, c_int]),
    "mf_negative_masks": (c_int, [c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_sz, c_vp, c_vp]),
    "mf_set_mining_prefilter": (None, [c_int]),
    "mf_mine_logits": (c_int, [c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp]),
    "mf_update_ws_bytes": (c_sz, [c_i64, c_int]),
    "mf_update_sgd": (c_int, [c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_int, c_f32, c_f32, c_vp, c_sz, c_vp]),
    "mf_update_adam": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_int, c_i64, c_vp, c_f32, c_f32,
                               c_f32, c_f32, c_f32, c_vp, c_sz, c_vp]),
    "mf_update_pair": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_vp, c_sz, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64,
                               c_vp, c_int, c_vp, c_sz, c_i64, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp]),
    "mf_step_small_ws_bytes": (c_sz, [c_int]),
    "mf_step_small": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_vp,
                              c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_f32, c_f32, c_vp, c_i64, c_int, c_i64, c_vp, c_f32, c_f32,
                              c_f32, c_f32, c_f32, c_vp, c_sz, c_vp, c_vp]),
    "mf_topk_ws_bytes": (c_sz, [c_i64, c_i64, c_int, c_int]),
    "mf_topk": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_i64, c_vp, c_sz, c_vp, c_vp, c_vp]),
    "mf_sample_batch": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, ctypes.c_uint64, c_i64, c_i64, c_int, c_vp, c_vp, c_vp,
                                c_vp, c_vp]),
    "mf_retrieval_metrics": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mf_topk_merge": (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_vp, c_vp, c_vp]),
    "mf_init_rows": (c_int, [c_vp, c_i64, c_int, c_i64, c_i64, ctypes.c_uint64, c_f32, c_vp]),
    "mf_comm_unique_id": (c_int, [c_vp]),
    "mf_comm_create": (c_int, [c_int, c_int, c_vp, ctypes.POINTER(c_vp)]),
    "mf_comm_destroy": (c_int, [c_vp]),
    "mf_comm_world": (c_int, [c_vp]),
    "mf_comm_source": (ctypes.c_char_p, []),
    "mf_topk_chunks": (c_int, [c_i64, c_i64, c_int, c_int, ctypes.POINTER(c_i64)]),
    "mf_comm_all_to_all_rows": (c_int, [c_vp, c_vp, ctypes.POINTER(c_i64), c_vp, ctypes.POINTER(c_i64), c_i64, c_vp]),
    "mf_comm_all_gather": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "mf_topk_blocked_bytes": (c_sz, [c_i64, c_int]),
    "mf_topk_blocked_build": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp]),
    "mf_topk_small_ws_bytes": (c_sz, [c_i64, c_i64, c_int, c_int]),
    "mf_topk_small": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_i64, c_vp, c_sz, c_vp, c_vp, c_vp]),
    "mf_topk_pack": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]),
    "mf_topk_merge_packed": (c_int, [c_vp, c_int, c_i64, c_int, c_vp, c_vp, c_vp]),
    "mf_topk_bf3_index_bytes": (c_sz, [c_i64, c_int]),
    "mf_topk_bf3_build": (c_int, [c_vp, c_i64, c_int, c_vp, c_sz, c_vp]),
    "mf_topk_bf3_ws_bytes": (c_sz, [c_i64, c_i64, c_int, c_int]),
    "mf_topk_bf3": (c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_i64, c_vp, c_sz, c_vp, c_vp, c_vp]),
}

MF_OK, MF_EINVAL, MF_ENOSPC, MF_ELAUNCH, MF_ENOTSUP = 0, -1, -2, -3, -4       # return codes (include/mf_hip.h)
LOSS_TARGET_I64, LOSS_ROWC, LOSS_MASKS_READY = 1, 2, 4     # flags of mf_loss_fwd / mf_loss_bwd (include/mf_hip.h)

_lib: ctypes.CDLL | None = None


class MfHipError(RuntimeError):
    pass


def build(force: bool = False) -> pathlib.Path:
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", str(_PKG / "csrc"), "-j4", "-s"] + (["-B"] if force else [])
    subprocess.run(cmd, check=True)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """The loaded library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise MfHipError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(this package has no CPU fallback)")
        handle = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise MfHipError(f"libmf_hip error {rc}: {lib().mf_last_error().decode()}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def dev_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    """Contiguous fp32 CUDA view of ``t`` (copy only if needed); raises on CPU tensors."""
    if not t.is_cuda:
        raise MfHipError(f"{name} must live on the GPU (got {t.device}); this package has no CPU path")
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t.contiguous()


def dev_i64(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise MfHipError(f"{name} must live on the GPU (got {t.device}); this package has no CPU path")
    if t.dtype != torch.int64:
        t = t.to(torch.int64)
    return t.contiguous()


def ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


_ws_pool: dict = {}


class LeasedWorkspace:
    """A loss workspace (1 GB at B = 8192: it holds the logit stashes) taken from a small pool and given
    back when the last reference goes (autograd drops it with the graph).  The loss path uses two streams
    (masks are built on a side stream), and handing such blocks back to torch's caching allocator every step
    made it allocate fresh ones for the first dozens of steps (a block recorded on two streams is reused
    late).  A pooled block is only ever reused behind work that is already ordered before the new use: the
    same stream, or the side stream after ``wait_stream``."""

    def __init__(self, nbytes: int, device) -> None:
        dev = torch.device(device)
        index = dev.index if dev.index is not None else torch.cuda.current_device()
        # keyed by the leasing stream as well: a block is only handed to work that is ordered behind its last use
        self.key = (index, torch.cuda.current_stream(index).cuda_stream, max(int(nbytes), 256))
        free = _ws_pool.setdefault(self.key, [])
        self.tensor = free.pop() if free else torch.empty(self.key[2], dtype=torch.uint8, device=device)

    def __del__(self) -> None:
        try:
            free = _ws_pool.setdefault(self.key, [])
            if len(free) < 4:  # noqa: PLR2004
                free.append(self.tensor)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


SUPPORTED_WIDTHS = (32, 64, 128, 256)


def padded_width(d: int) -> int:
    for w in SUPPORTED_WIDTHS:
        if d <= w:
            return w
    raise MfHipError(f"embedding width {d} > {SUPPORTED_WIDTHS[-1]} is not supported")
