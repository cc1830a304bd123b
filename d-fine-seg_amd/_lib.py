"""ctypes binding of libdfine_hip.so (the C ABI declared in include/dfine_hip.h).

There is no fallback: if the library is missing, or a tensor is not on the GPU, the call
raises. Nothing here imports the CPU oracle.
"""
import ctypes
import os

import torch  # imported first so the HIP runtime torch ships is the one the .so binds to

from ._build import LIB_PATH

_c_void_p, _c_int = ctypes.c_void_p, ctypes.c_int
_i32p = ctypes.POINTER(ctypes.c_int32)

# name -> argtypes; must list EVERY symbol include/dfine_hip.h declares (tests check this)
SIGNATURES = {
    "dfine_abi_version": [],
    "dfine_msda_fwd": [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p]
    + [_c_int] * 6
    + [_i32p, _i32p, _c_void_p],
    "dfine_msda_bwd": [_c_void_p, _c_int] + [_c_void_p] * 4 + [_c_int] + [_c_void_p] * 2 + [_c_int] * 6 + [_i32p, _i32p, _c_void_p],
    "dfine_match_cost": [_c_void_p] * 6 + [_c_int] * 5 + [ctypes.c_float] * 5 + [_c_void_p],
    "dfine_lsap": [_c_void_p] * 4 + [_c_int] * 4 + [_c_void_p],
    "dfine_go_union": [_c_void_p] * 2 + [_c_int] * 3 + [_c_void_p],
    "dfine_dwconv_fwd": [_c_void_p] * 3 + [_c_int] * 7 + [_c_void_p, _c_void_p],
    "dfine_dwconv_bwd_data": [_c_void_p] * 3 + [_c_int] * 7 + [_c_void_p],
    "dfine_dwconv_bwd_weight": [_c_void_p] * 4 + [ctypes.c_longlong] + [_c_int] * 7 + [_c_void_p],
    "dfine_dwconv_bwd_weight_workspace": [_c_int] * 6,
    "dfine_bn_act_fwd": [_c_void_p] * 3 + [_c_int] + [_c_void_p] * 9 + [_c_int] * 4 + [ctypes.c_float] * 2 + [_c_int, _c_int, _c_void_p, _c_int, _c_void_p],
    "dfine_criterion_head": [_c_void_p, _c_int] + [_c_void_p] * 17 + [_c_int] * 5 + [ctypes.c_float] * 4 + [_c_void_p, _c_void_p],
    "dfine_criterion_heads": [_c_void_p, _c_int, _c_int, _c_void_p],
    "dfine_criterion_head_scale": [_c_void_p, _c_int] + [_c_void_p] * 12 + [ctypes.c_longlong] * 3 + [_c_void_p],
    "dfine_linear_wgrad": [_c_void_p] * 4 + [_c_int] * 6 + [_c_void_p],
    "dfine_conv_wgrad": [_c_void_p] * 3 + [_c_int] * 13 + [_c_void_p],
    "dfine_fdr_head_fwd": [_c_void_p] * 6 + [ctypes.c_longlong, _c_int, _c_int, ctypes.c_float, _c_void_p],
    "dfine_fdr_head_bwd": [_c_void_p] * 7 + [ctypes.c_longlong, _c_int, _c_int, ctypes.c_float, _c_void_p],
    "dfine_upsample2x_fwd": [_c_void_p, _c_int, _c_void_p] + [_c_int] * 5 + [_c_void_p],
    "dfine_upsample2x_bwd": [_c_void_p, _c_int, _c_int, _c_void_p] + [_c_int] * 4 + [_c_void_p],
    "dfine_stem_pool_fwd": [_c_void_p, _c_int, _c_void_p] + [_c_int] * 5 + [_c_void_p],
    "dfine_stem_pool_bwd": [_c_void_p, _c_void_p, _c_int, _c_int, _c_void_p] + [_c_int] * 4 + [_c_void_p],
    "dfine_mask_loss_stats": [_c_void_p, _c_int] + [_c_void_p] * 3 + [_c_int] * 4 + [_c_void_p],
    "dfine_mask_loss_grad": [_c_void_p, _c_int] + [_c_void_p] * 4 + [_c_int] * 4 + [_c_void_p],
    "dfine_bn_pair_act_fwd": [_c_void_p] * 3 + [_c_int] + [_c_void_p] * 12 + [_c_int] * 4 + [ctypes.c_float] * 2 + [_c_int, _c_void_p],
    "dfine_bn_pair_act_bwd": [_c_void_p] * 3 + [_c_int, _c_void_p, _c_int, _c_int] + [_c_void_p] * 13 + [_c_int] * 4 + [ctypes.c_float, _c_void_p],
    "dfine_mask_rows_stats": [_c_void_p] * 5 + [_c_int] * 5 + [_c_void_p],
    "dfine_mask_rows_dz": [_c_void_p] * 6 + [_c_int] * 6 + [_c_void_p],
    "dfine_mask_rows_dfeat": [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p] + [_c_int] * 4 + [_c_void_p],
    "dfine_add_layernorm_fwd": [_c_void_p, _c_void_p, _c_int] + [_c_void_p] * 6 + [ctypes.c_longlong, _c_int, ctypes.c_float, ctypes.c_float, _c_void_p],
    "dfine_add_layernorm_bwd": [_c_void_p, _c_void_p, _c_int] + [_c_void_p] * 6 + [ctypes.c_longlong, _c_int, ctypes.c_float, _c_void_p],
    "dfine_grad_sumsq": [_c_void_p, ctypes.c_longlong, _c_void_p, _c_void_p],
    "dfine_adamw_ema_step": [_c_void_p] * 8 + [ctypes.POINTER(ctypes.c_longlong), _c_int, ctypes.c_longlong, _c_void_p],
    "dfine_igemm_bf16": [_c_void_p] * 7,
    "dfine_igemm_bf16_bnred": [_c_void_p] * 6,
    "dfine_igemm_bf16_relumask": [_c_void_p] * 5 + [_c_int, _c_void_p],
    "dfine_msda_bwd_points": [_c_void_p, _c_int] + [_c_void_p] * 5 + [_c_int] * 6 + [_i32p, _i32p, _c_void_p],
    "dfine_msda_bwd_value": [_c_int] + [ctypes.POINTER(_c_void_p)] * 3 + [_c_int, _c_void_p] + [_c_int] * 6 + [_i32p, _i32p, _c_void_p],
    "dfine_msda_bwd_value_sorted": [_c_int] + [ctypes.POINTER(_c_void_p)] * 3 + [_c_int, _c_void_p, _c_int, _c_void_p, ctypes.c_longlong]
                                   + [_c_int] * 6 + [_i32p, _i32p, _c_void_p],
    "dfine_msda_sorted_records": [_c_int] * 6,
    "dfine_mask_targets_nearest": [_c_void_p, _c_void_p, ctypes.c_longlong] + [_c_int] * 6 + [_c_void_p],
    "dfine_gate_mix_fwd": [_c_void_p] * 3 + [_c_int, _c_void_p, ctypes.c_longlong, _c_int, _c_void_p],
    "dfine_gate_mix_bwd": [_c_void_p] * 4 + [_c_int] + [_c_void_p] * 3 + [ctypes.c_longlong, _c_int, _c_void_p],
    "dfine_msda_prepare_fwd": [_c_void_p, _c_void_p, _c_int] + [_c_void_p] * 4 + [ctypes.c_longlong, _c_int, _c_int, _c_void_p],
    "dfine_msda_prepare_bwd": [_c_void_p] * 5 + [_c_int, _c_void_p, _c_void_p, ctypes.c_longlong, _c_int, _c_int, _c_void_p],
    "dfine_bilinear_add_fwd": [_c_void_p, _c_void_p, _c_int, _c_void_p] + [_c_int] * 6 + [_c_void_p],
    "dfine_bilinear_bwd": [_c_void_p, _c_int, _c_void_p] + [_c_int] * 6 + [_c_void_p],
    "dfine_row_max": [_c_void_p, _c_int, _c_void_p, ctypes.c_longlong, _c_int, _c_void_p],
    "dfine_stage_images": [_c_void_p, _c_void_p] + [_c_int] * 6 + [_c_void_p] * 3,
    "dfine_postprocess_topk": [_c_void_p, _c_int] + [_c_void_p] * 6 + [_c_int] * 7 + [_c_void_p],
    "dfine_postprocess_masks": [_c_void_p, _c_int] + [_c_void_p] * 3 + [_c_int] * 8 + [ctypes.c_float, _c_void_p],
    "dfine_attention_fwd": [_c_void_p] * 3 + [_c_int] * 3 + [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p] + [_c_int] * 4 + [ctypes.c_float, _c_void_p],
    "dfine_attention_bwd": [_c_void_p] * 3 + [_c_int] * 3 + [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int,
                            _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p] + [_c_int] * 4 + [ctypes.c_float, _c_void_p],
    "dfine_transpose_weights": [_c_void_p] * 4 + [_c_int, _c_void_p],
    "dfine_embedding_wgrad": [_c_void_p, _c_void_p, _c_int, _c_void_p, ctypes.c_longlong, _c_int, _c_int, _c_int, _c_void_p],
    "dfine_wgrad_group_eligible": [_c_void_p],
    "dfine_wgrad_group_plan": [_c_void_p, _c_int, _c_void_p, ctypes.c_longlong],
    "dfine_wgrad_group_launch": [_c_void_p, _c_void_p, _c_void_p],
    "dfine_bn_act_bwd": [_c_void_p] * 3 + [_c_int, _c_void_p, _c_int, _c_int] + [_c_void_p] * 12 + [_c_int] * 4 + [ctypes.c_float] + [_c_int, _c_void_p],
}

ABI_VERSION = 24
_lib = None

IGEMM_MAX_TAPS = 16


class IGemmGeom(ctypes.Structure):
    """dfine_igemm_geom of include/dfine_hip.h (host struct, passed by pointer)."""

    _fields_ = [(n, _c_int) for n in ("M", "N", "K", "lda", "ldb", "ldc", "gather", "PH", "PW", "IH", "IW", "CI", "ntaps",
                                      "isy", "isx", "OHf", "OWf", "osy", "osx", "ooy", "oox", "act")] + [
        ("tap_dy", ctypes.c_byte * IGEMM_MAX_TAPS), ("tap_dx", ctypes.c_byte * IGEMM_MAX_TAPS),
        ("tap_w", ctypes.c_byte * IGEMM_MAX_TAPS)]


class CriterionHeadArgs(ctypes.Structure):
    """dfine_criterion_head_args of include/dfine_hip.h (one head of a batched dfine_criterion_heads launch)."""

    _fields_ = ([(n, _c_void_p) for n in ("logits", "boxes", "corners", "ref_points", "teacher_corners", "teacher_logits", "own_assign",
                                          "go_assign", "tgt_labels", "tgt_boxes", "Wn", "scalars", "terms", "d_logits", "d_boxes_l1",
                                          "d_boxes_giou", "d_corners_fgl", "d_corners_ddf", "out")]
                + [(n, _c_int) for n in ("B", "Q", "C", "G", "n_bins", "has_strides")]
                + [(n, ctypes.c_float) for n in ("alpha", "gamma", "reg_scale", "temperature")]
                + [("batch_strides", ctypes.c_longlong * 6)])


class IGemmBnRed(ctypes.Structure):
    """dfine_igemm_bnred of include/dfine_hip.h (host struct, passed by pointer)."""

    _fields_ = [(n, _c_void_p) for n in ("x", "dy2", "mean", "rstd", "gamma", "beta", "lab_scale", "red")] + [
        (n, _c_int) for n in ("ldx", "ld_dy2", "act")]


class WgradProblem(ctypes.Structure):
    """dfine_wgrad_problem of include/dfine_hip.h (one queued weight gradient of a grouped launch)."""

    _fields_ = [("dY", _c_void_p), ("X", _c_void_p), ("dW", _c_void_p), ("db", _c_void_p)] + [
        (n, _c_int) for n in ("conv", "M", "N", "K", "ldw", "B", "H", "W", "CI", "OH", "OW", "CO", "KH", "KW", "stride", "pad",
                              "ldy", "ldx")]


class DfineHipError(RuntimeError):
    pass


def load():
    """Load libdfine_hip.so once; raise loudly if it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DfineHipError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950). The HIP path has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = ctypes.c_longlong if name.endswith(("_workspace", "_plan", "_records")) else _c_int
    got = lib.dfine_abi_version()
    if got != ABI_VERSION:
        raise DfineHipError(f"libdfine_hip.so ABI {got} != expected {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib


_ERRORS = {-1: "bad argument", -2: "unsupported shape/dtype"}


def check(rc, what):
    if rc != 0:
        msg = _ERRORS.get(rc, f"hipError_t {rc}")
        raise DfineHipError(f"{what} failed: {msg}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DfineHipError(
                "HIP op called with a CPU tensor: the product path runs on MI355X only "
                "(the CPU restatement lives in oracle/ and is test infrastructure)"
            )


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def i32_array(values):
    flat = [int(v) for v in values]
    return (ctypes.c_int32 * len(flat))(*flat)


DT_F32, DT_BF16 = 0, 1


def dtype_code(t):
    if t.dtype == torch.float32:
        return DT_F32
    if t.dtype == torch.bfloat16:
        return DT_BF16
    raise DfineHipError(f"unsupported dtype {t.dtype} (float32 or bfloat16 only)")


# --------------------------------------------------------------------------------------
# optional per-kernel timing (bench.py): HIP events on the stream the kernel is launched on
# --------------------------------------------------------------------------------------
class KernelTimer:
    """Collects (start, stop) event pairs per kernel name; ``summary()`` after a device sync."""

    def __init__(self, by_detail=False):
        self.events = {}
        self.flops = {}  # name -> algorithmic FLOPs summed over its launches (MFMA kernels)
        self.by_detail = by_detail  # key by "name detail" (e.g. per shape) instead of by name

    def summary(self):
        out = {}
        for name, recs in self.events.items():
            ms = [a.elapsed_time(b) for a, b, _ in recs]
            nbytes = sum(n for _, _, n in recs)
            out[name] = {"launches": len(ms), "avg_us": 1e3 * sum(ms) / max(len(ms), 1), "total_ms": sum(ms),
                         "avg_bytes": nbytes / max(len(ms), 1), "total_flops": self.flops.get(name, 0.0)}
        return out


_timer = None


def set_kernel_timer(timer):
    global _timer
    _timer = timer


class timed:
    """``with timed("msda_fwd", algorithmic_bytes): launch`` -- a no-op unless a KernelTimer is
    installed. ``nbytes`` is the launch's ALGORITHMIC traffic ("Kernels")."""

    def __init__(self, name, nbytes=0, detail=None, flops=0.0):
        self.name, self.nbytes, self.detail, self.flops = name, nbytes, detail, flops

    def __enter__(self):
        if _timer is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record(torch.cuda.current_stream())

    def __exit__(self, *exc):
        if _timer is not None:
            stop = torch.cuda.Event(enable_timing=True)
            stop.record(torch.cuda.current_stream())
            key = f"{self.name} {self.detail}" if (_timer.by_detail and self.detail) else self.name
            _timer.events.setdefault(key, []).append((self.start, stop, self.nbytes))
            if self.flops:
                _timer.flops[key] = _timer.flops.get(key, 0.0) + self.flops
        return False
