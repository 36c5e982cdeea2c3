"""ctypes binding of ``libmmfusion.so`` (C ABI: ``include/mmfusion.h``).

The library is the product: there is no CPU or eager fallback.  ``load()`` raises if the shared
object is missing, and every wrapper raises ``RuntimeError(mmf_last_error())`` on a negative
return code.  Wrappers take raw device pointers (``tensor.data_ptr()``) and enqueue on the HIP
stream that torch currently uses, so launches are ordered with torch's own work and can be
captured by ``torch.cuda.graph``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMF_LIB_PATH") or os.path.join(_HERE, "libmmfusion.so")     # MMF_LIB_PATH: A/B of two builds in one gpurun call

GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_BIAS, EPI_RELU, EPI_MASK_AUX, EPI_ADD_AUX, EPI_ACCUM, EPI_COLSUM_A, EPI_DROPOUT = 1, 2, 4, 8, 16, 32, 64
GEMM_MAX_PROBLEMS, ATTN_MAX_PROBLEMS, LN_MAX_PROBLEMS, COLSUM_MAX_PROBLEMS = 48, 12, 8, 24

# every symbol include/mmfusion.h declares (tests check the .so exports all of them)
SYMBOLS = (
    "mmf_version", "mmf_last_error", "mmf_device_cu_count", "mmf_gemm_grouped", "mmf_gemm_grouped_ex", "mmf_gemm_select_impl", "mmf_gemm_last_impl", "mmf_gemm_set_persistent_workgroups",
    "mmf_attn_fwd_grouped_ex", "mmf_attn_bwd_grouped_ex", "mmf_dropout", "mmf_attn_select_impl",
    "mmf_attn_fwd_grouped", "mmf_attn_bwd_grouped", "mmf_layernorm_fwd_grouped",
    "mmf_layernorm_bwd_grouped", "mmf_layernorm_bwd_workspace_bytes", "mmf_cast_f32_to_bf16", "mmf_cast_bf16_to_f32", "mmf_cast_bf16_to_f32_scaled", "mmf_cast_f32_to_bf16_2d", "mmf_add3_bf16", "mmf_add3_grouped", "mmf_addn_bf16", "mmf_addn_grouped",
    "mmf_meanpool_fwd", "mmf_meanpool_bwd", "mmf_meanpool_cat_fwd", "mmf_meanpool_cat_bwd", "mmf_colsum_bf16", "mmf_colsum_grouped", "mmf_relu_bwd_bf16", "mmf_relu_bwd_mixed",
    "mmf_sqnorm_f32", "mmf_adamw_step", "mmf_adamw_advance", "mmf_skinny_linear_fwd", "mmf_skinny_linear_dgrad", "mmf_skinny_linear_fwd_ex", "mmf_skinny_linear_dgrad_ex",
    "mmf_gat3_dense_fwd", "mmf_gat3_dense_bwd", "mmf_infonce_fwd", "mmf_infonce_bwd", "mmf_adaptive_combine_fwd",
    "mmf_adaptive_combine_bwd", "mmf_adaptive_attn_weights", "mmf_linear_narrow_fwd", "mmf_linear_narrow_bwd", "mmf_stack3_embed_fwd",
    "mmf_stack3_embed_bwd", "mmf_rowmask_apply", "mmf_zero_ranges_f32", "mmf_fusion_loss", "mmf_modality_dropout",
    "mmf_attn_weights_mean", "mmf_gemm_f32_grouped", "mmf_gemm_f32_batched", "mmf_softmax_rows_f32", "mmf_softmax_bwd_rows_f32",
    "mmf_layernorm_f32_fwd", "mmf_layernorm_f32_bwd", "mmf_bilstm_workspace_bytes", "mmf_bilstm_layer_fwd", "mmf_bilstm_layer_bwd", "mmf_swap01",
)


class GemmProblem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p),
                ("aux", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32), ("ldaux", C.c_int32)]


class AttnProblem(C.Structure):
    _fields_ = [("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p),
                ("LSE", C.c_void_p), ("dO", C.c_void_p), ("delta", C.c_void_p), ("dQ", C.c_void_p),
                ("dK", C.c_void_p), ("dV", C.c_void_p), ("B", C.c_int32), ("H", C.c_int32),
                ("Tq", C.c_int32), ("Tk", C.c_int32), ("ldq", C.c_int32), ("ldk", C.c_int32),
                ("ldv", C.c_int32), ("ldo", C.c_int32)]


class LnProblem(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("mean", C.c_void_p), ("rstd", C.c_void_p), ("dy", C.c_void_p), ("dx", C.c_void_p),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("rows", C.c_int32)]


class GemmExtra(C.Structure):
    _fields_ = [("alpha", C.c_float), ("dropout_p", C.c_float), ("rng_state", C.c_void_p), ("site", C.c_uint32)]


class SkinnyProblem(C.Structure):
    _fields_ = [("X", C.c_void_p), ("W", C.c_void_p), ("Y", C.c_void_p), ("bias", C.c_void_p), ("aux", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("ldx", C.c_int32), ("ldw", C.c_int32),
                ("ldy", C.c_int32), ("ldaux", C.c_int32)]


class SkinnyProblemEx(C.Structure):
    _fields_ = [("p", SkinnyProblem), ("Y2", C.c_void_p), ("gate", C.c_void_p), ("dz", C.c_void_p),
                ("ldy2", C.c_int32), ("ldgate", C.c_int32), ("lddz", C.c_int32), ("reserved", C.c_int32)]


class SkinnyExtra(C.Structure):
    _fields_ = [("x_f32", C.c_int32), ("gate_f32", C.c_int32), ("gate_scale", C.c_float), ("dropout_p", C.c_float),
                ("rng_state", C.c_void_p), ("site", C.c_uint32), ("reserved", C.c_uint32)]


SKINNY_MAX_M, SKINNY_MAX_PROBLEMS = 64, 24
ADD3_MAX = 8
ADDN_MAX = 8
ADDN_GROUP_MAX = 4


class AddNProblem(C.Structure):
    _fields_ = [("x", C.c_void_p * 8), ("y", C.c_void_p), ("numel", C.c_int64), ("n", C.c_int32)]


class Add3Problem(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p), ("y", C.c_void_p), ("n", C.c_int64)]

ZERO_MAX_RANGES = 48
POOL_MAX = 4


class Gat3Params(C.Structure):
    _fields_ = [("B", C.c_int32), ("heads", C.c_int32), ("C", C.c_int32), ("relu", C.c_int32),
                ("negative_slope", C.c_float), ("dropout_p", C.c_float), ("rng_state", C.c_void_p), ("site", C.c_uint32)]


class BiLstmArgs(C.Structure):
    _fields_ = [("gx", C.c_void_p), ("w_hh", C.c_void_p * 2), ("b_ih", C.c_void_p * 2), ("b_hh", C.c_void_p * 2),
                ("y", C.c_void_p), ("gates", C.c_void_p), ("cell", C.c_void_p), ("dy", C.c_void_p), ("dgates", C.c_void_p),
                ("T", C.c_int32), ("B", C.c_int32), ("H", C.c_int32)]


class ColsumProblem(C.Structure):
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32), ("ldx", C.c_int32)]


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libmmfusion.so (built by ``__graft_entry__.build()`` / ``make -C csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the MI355X fusion path has no fallback. Build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C simple-multimodal_amd/csrc`.")
    # torch first: its wheel bundles its own libamdhip64; libmmfusion.so must bind to THAT copy (same SONAME), or the
    # process ends up with two HIP runtimes and this library's launches fail with "no ROCm-capable device"
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    lib.mmf_last_error.restype = C.c_char_p
    lib.mmf_version.restype = C.c_int
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    lib.mmf_gemm_grouped.argtypes = [C.POINTER(GemmProblem), i32, i32, i32, i32, vp]
    lib.mmf_gemm_grouped_ex.argtypes = [C.POINTER(GemmProblem), i32, i32, i32, i32, C.POINTER(GemmExtra), vp]
    lib.mmf_attn_fwd_grouped_ex.argtypes = [C.POINTER(AttnProblem), i32, i32, f32, f32, vp, C.c_uint32, vp]
    lib.mmf_attn_bwd_grouped_ex.argtypes = [C.POINTER(AttnProblem), i32, i32, f32, f32, vp, C.c_uint32, vp]
    lib.mmf_dropout.argtypes = [vp, vp, i64, i32, f32, vp, C.c_uint32, vp]
    lib.mmf_attn_fwd_grouped.argtypes = [C.POINTER(AttnProblem), i32, i32, f32, vp]
    lib.mmf_attn_select_impl.argtypes = [i32]
    lib.mmf_attn_bwd_grouped.argtypes = [C.POINTER(AttnProblem), i32, i32, f32, vp]
    lib.mmf_layernorm_fwd_grouped.argtypes = [C.POINTER(LnProblem), i32, i32, f32, vp]
    lib.mmf_layernorm_bwd_grouped.argtypes = [C.POINTER(LnProblem), i32, i32, vp, C.c_size_t, vp]
    lib.mmf_layernorm_bwd_workspace_bytes.argtypes = [i32]
    lib.mmf_layernorm_bwd_workspace_bytes.restype = C.c_size_t
    lib.mmf_cast_f32_to_bf16.argtypes = [vp, vp, i64, vp]
    lib.mmf_cast_bf16_to_f32.argtypes = [vp, vp, i64, vp]
    lib.mmf_cast_bf16_to_f32_scaled.argtypes = [vp, vp, i64, f32, vp]
    lib.mmf_cast_f32_to_bf16_2d.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.mmf_add3_bf16.argtypes = [vp, vp, vp, vp, i64, vp]
    lib.mmf_add3_grouped.argtypes = [C.POINTER(Add3Problem), i32, vp]
    lib.mmf_addn_bf16.argtypes = [C.POINTER(C.c_void_p), i32, vp, i64, i32, vp]
    lib.mmf_addn_grouped.argtypes = [C.POINTER(AddNProblem), i32, vp]
    lib.mmf_meanpool_fwd.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    lib.mmf_meanpool_bwd.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    lib.mmf_meanpool_cat_fwd.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), i32, vp, i32, i32, i32, vp]
    lib.mmf_meanpool_cat_bwd.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_int), i32, i32, i32, i32, vp]
    lib.mmf_colsum_bf16.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.mmf_colsum_grouped.argtypes = [C.POINTER(ColsumProblem), i32, vp]
    lib.mmf_relu_bwd_bf16.argtypes = [vp, vp, vp, i64, vp]
    lib.mmf_relu_bwd_mixed.argtypes = [vp, C.c_int, vp, C.c_int, vp, i64, vp]
    lib.mmf_skinny_linear_fwd.argtypes = [C.POINTER(SkinnyProblem), i32, i32, i32, vp]
    lib.mmf_skinny_linear_dgrad.argtypes = [C.POINTER(SkinnyProblem), i32, i32, f32, i32, vp]
    lib.mmf_fusion_loss.argtypes = [vp, i32, vp, i32, i32, f32, C.POINTER(vp), C.POINTER(f32), i32, vp, vp, vp]
    lib.mmf_modality_dropout.argtypes = [C.POINTER(vp), C.POINTER(vp), vp, i32, i32, f32, vp, C.c_uint32, i32, vp]
    lib.mmf_skinny_linear_fwd_ex.argtypes = [C.POINTER(SkinnyProblemEx), i32, i32, i32, C.POINTER(SkinnyExtra), vp]
    lib.mmf_skinny_linear_dgrad_ex.argtypes = [C.POINTER(SkinnyProblemEx), i32, i32, f32, i32, C.POINTER(SkinnyExtra), vp]
    lib.mmf_sqnorm_f32.argtypes = [vp, i64, vp, vp]
    P3 = C.c_void_p * 3
    lib.mmf_gat3_dense_fwd.argtypes = [vp] * 8 + [C.POINTER(Gat3Params), vp]
    lib.mmf_gat3_dense_bwd.argtypes = [vp] * 12 + [C.POINTER(Gat3Params), vp]
    lib.mmf_infonce_fwd.argtypes = [P3, P3, vp, vp, vp, i32, i32, f32, vp]
    lib.mmf_infonce_bwd.argtypes = [P3, vp, vp, P3, P3, P3, i32, i32, f32, vp]
    lib.mmf_adaptive_combine_fwd.argtypes = [vp] * 6 + [i32, i32, vp]
    lib.mmf_adaptive_combine_bwd.argtypes = [vp] * 10 + [i32, i32, vp]
    lib.mmf_adaptive_attn_weights.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.mmf_attn_weights_mean.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    lib.mmf_linear_narrow_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    lib.mmf_linear_narrow_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.mmf_stack3_embed_fwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.mmf_stack3_embed_bwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.mmf_rowmask_apply.argtypes = [vp, vp, vp, i32, i32, vp]
    lib.mmf_zero_ranges_f32.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), i32, vp]
    lib.mmf_adamw_step.argtypes = [vp, vp, vp, vp, vp, i64, vp, vp, vp]
    lib.mmf_adamw_advance.argtypes = [vp, vp, vp, vp]
    lib.mmf_bilstm_workspace_bytes.restype = C.c_size_t
    lib.mmf_bilstm_layer_fwd.argtypes = [C.POINTER(BiLstmArgs), vp, C.c_size_t, vp]
    lib.mmf_bilstm_layer_bwd.argtypes = [C.POINTER(BiLstmArgs), vp, C.c_size_t, vp]
    lib.mmf_swap01.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    S2 = C.c_int64 * 2
    lib.mmf_gemm_f32_grouped.argtypes = [C.POINTER(GemmProblem), i32, i32, i32, f32, vp]
    lib.mmf_gemm_f32_batched.argtypes = [C.POINTER(GemmProblem), i32, i32, f32, i32, i32, S2, S2, S2, vp]
    lib.mmf_softmax_rows_f32.argtypes = [vp, i64, i32, f32, vp]
    lib.mmf_softmax_bwd_rows_f32.argtypes = [vp, vp, i64, i32, f32, vp]
    lib.mmf_layernorm_f32_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]
    lib.mmf_layernorm_f32_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    for name in SYMBOLS:
        getattr(lib, name)          # AttributeError here = header and .so disagree
    if lib.mmf_version() != 1:
        raise RuntimeError(f"libmmfusion ABI version {lib.mmf_version()} != 1")
    _lib = lib
    for env, fn in (("MMF_ATTN_IMPL", lib.mmf_attn_select_impl), ("MMF_GEMM_IMPL", lib.mmf_gemm_select_impl)):
        if os.environ.get(env):            # A/B runs: pin a kernel generation for the whole process
            check(fn(int(os.environ[env])))
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"libmmfusion error {rc}: {load().mmf_last_error().decode()}")


def stream_ptr() -> int:
    """The HIP stream torch is currently enqueuing on (raw hipStream_t)."""
    import torch
    return torch.cuda.current_stream().cuda_stream


# ---- optional launch timing (bench.py's roofline leg) --------------------------------------------
# When PROFILE is a list, every grouped GEMM / attention launch is bracketed by HIP events recorded
# on the launch stream and appended as (kernel label, algorithmic flops, start, end).
PROFILE: Optional[list] = None
_LAYOUT_NAME = {GEMM_NT: "NT", GEMM_NN: "NN", GEMM_TN: "TN"}


class _Timed:
    def __init__(self, label: str, flops: float, detail=None):
        self.label, self.flops, self.detail = label, flops, detail

    def __enter__(self):
        if PROFILE is not None:
            import torch
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            import torch
            self.e1.record(torch.cuda.current_stream())
            PROFILE.append((self.label, self.flops, self.e0, self.e1, self.detail))


def gemm_grouped(problems: Sequence[GemmProblem], layout: int, epilogue: int, out_f32: bool,
                 alpha: float = 1.0, dropout_p: float = 0.0, rng_state_ptr: Optional[int] = None, site: int = 0) -> None:
    arr = (GemmProblem * len(problems))(*problems)
    flops = sum(2.0 * p.M * p.N * p.K for p in problems) if PROFILE is not None else 0.0
    detail = [(p.M, p.N, p.K) for p in problems] if PROFILE is not None else None
    with _Timed(f"gemm_grouped_kernel<{_LAYOUT_NAME[layout]},{'f32' if out_f32 else 'bf16'}>", flops, detail) as tm:
        if alpha == 1.0 and not (epilogue & EPI_DROPOUT):
            check(load().mmf_gemm_grouped(arr, len(problems), layout, epilogue, int(out_f32), stream_ptr()))
        else:
            ex = GemmExtra(alpha, dropout_p, rng_state_ptr, site)
            check(load().mmf_gemm_grouped_ex(arr, len(problems), layout, epilogue, int(out_f32), C.byref(ex), stream_ptr()))
        if PROFILE is not None:            # name the kernel generation that ran: the dominant kernel is a kernel symbol
            tm.label = f"gemm{load().mmf_gemm_last_impl()}_grouped_kernel<{_LAYOUT_NAME[layout]},{'f32' if out_f32 else 'bf16'}>"


def attn_fwd_grouped(problems: Sequence[AttnProblem], head_dim: int, scale: float, dropout_p: float = 0.0,
                     rng_state_ptr: Optional[int] = None, site: int = 0) -> None:
    arr = (AttnProblem * len(problems))(*problems)
    flops = sum(4.0 * p.B * p.H * p.Tq * p.Tk * head_dim for p in problems) if PROFILE is not None else 0.0
    with _Timed(f"attn_fwd_kernel<{head_dim}>", flops, [(p.Tq, p.Tk) for p in problems] if PROFILE is not None else None):
        check(load().mmf_attn_fwd_grouped_ex(arr, len(problems), head_dim, scale, dropout_p, rng_state_ptr, site,
                                             stream_ptr()))


def attn_bwd_grouped(problems: Sequence[AttnProblem], head_dim: int, scale: float, dropout_p: float = 0.0,
                     rng_state_ptr: Optional[int] = None, site: int = 0) -> None:
    arr = (AttnProblem * len(problems))(*problems)
    # algorithmic backward work: 4 products of 2*Tq*Tk*dh (dV, dP, dQ, dK); the recomputed S is not credited
    flops = sum(8.0 * p.B * p.H * p.Tq * p.Tk * head_dim for p in problems) if PROFILE is not None else 0.0
    with _Timed(f"attn_bwd_kernels<{head_dim}>", flops, [(p.Tq, p.Tk) for p in problems] if PROFILE is not None else None):
        check(load().mmf_attn_bwd_grouped_ex(arr, len(problems), head_dim, scale, dropout_p, rng_state_ptr, site,
                                             stream_ptr()))


def layernorm_fwd_grouped(problems: Sequence[LnProblem], d: int, eps: float) -> None:
    arr = (LnProblem * len(problems))(*problems)
    check(load().mmf_layernorm_fwd_grouped(arr, len(problems), d, eps, stream_ptr()))


def layernorm_bwd_grouped(problems: Sequence[LnProblem], d: int, workspace) -> None:
    """workspace: a float32 torch tensor of >= mmf_layernorm_bwd_workspace_bytes(d) bytes."""
    arr = (LnProblem * len(problems))(*problems)
    check(load().mmf_layernorm_bwd_grouped(arr, len(problems), d, workspace.data_ptr(),
                                           workspace.numel() * workspace.element_size(), stream_ptr()))


def colsum_grouped(problems: Sequence[ColsumProblem]) -> None:
    for i in range(0, len(problems), COLSUM_MAX_PROBLEMS):
        chunk = problems[i:i + COLSUM_MAX_PROBLEMS]
        arr = (ColsumProblem * len(chunk))(*chunk)
        check(load().mmf_colsum_grouped(arr, len(chunk), stream_ptr()))


def skinny_fwd(problems: Sequence[SkinnyProblem], flags: int, out_f32: bool) -> None:
    for i in range(0, len(problems), SKINNY_MAX_PROBLEMS):
        chunk = problems[i:i + SKINNY_MAX_PROBLEMS]
        arr = (SkinnyProblem * len(chunk))(*chunk)
        check(load().mmf_skinny_linear_fwd(arr, len(chunk), flags, int(out_f32), stream_ptr()))


def skinny_fwd_ex(problems: Sequence[SkinnyProblemEx], flags: int, out_f32: bool, extra: SkinnyExtra) -> None:
    """one launch: the group must fit (a dropout group's problem index keys its masks)"""
    if len(problems) > SKINNY_MAX_PROBLEMS:
        raise ValueError("a fused row-linear group must fit one launch")
    arr = (SkinnyProblemEx * len(problems))(*problems)
    check(load().mmf_skinny_linear_fwd_ex(arr, len(problems), flags, int(out_f32), C.byref(extra), stream_ptr()))


def skinny_dgrad_ex(problems: Sequence[SkinnyProblemEx], flags: int, alpha: float, out_f32: bool, extra: SkinnyExtra) -> None:
    if len(problems) > SKINNY_MAX_PROBLEMS:
        raise ValueError("a fused row-linear group must fit one launch")
    arr = (SkinnyProblemEx * len(problems))(*problems)
    check(load().mmf_skinny_linear_dgrad_ex(arr, len(problems), flags, alpha, int(out_f32), C.byref(extra), stream_ptr()))


def skinny_dgrad(problems: Sequence[SkinnyProblem], flags: int, alpha: float, out_f32: bool) -> None:
    for i in range(0, len(problems), SKINNY_MAX_PROBLEMS):
        chunk = problems[i:i + SKINNY_MAX_PROBLEMS]
        arr = (SkinnyProblem * len(chunk))(*chunk)
        check(load().mmf_skinny_linear_dgrad(arr, len(chunk), flags, alpha, int(out_f32), stream_ptr()))
