"""ctypes binding of the C-ABI HIP library (include/gemnet_hip.h).

There is deliberately NO fallback: if ``csrc/libgemnet_hip.so`` is missing or a call fails the
product path raises.  PyTorch is used only for device memory and streams: every entry point
receives raw device pointers + sizes + ``torch.cuda.current_stream()``.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GEMNET_HIP_LIB") or os.path.join(_HERE, "csrc", "libgemnet_hip.so")

_vp, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float


class GemmArgs(ctypes.Structure):
    """Mirror of `gn_gemm_args` (include/gemnet_hip.h)."""
    _fields_ = [
        ("A", _vp), ("B", _vp), ("C", _vp),
        ("M", _i), ("N", _i), ("K", _i),
        ("lda", _i), ("ldb", _i), ("ldc", _i),
        ("trans_a", _i), ("trans_b", _i),
        ("a_dact_pre", _vp),
        ("act", _i),
        ("pre_out", _vp),
        ("mul", _vp), ("ldmul", _i),
        ("alpha", _f),
        ("res", _vp), ("ldres", _i),
        ("beta", _f),
        ("gadd1", _vp), ("gidx1", _vp),
        ("gadd2", _vp), ("gidx2", _vp),
        ("ldg", _i),
        ("ridx", _vp),
        ("res2", _vp), ("ldres2", _i),
        ("beta2", _f),
        ("splitk_ws", _vp), ("splitk", _i),
    ]


GN_OP_LOAD, GN_OP_SCALE, GN_OP_GEMM, GN_OP_STORE = 0, 1, 2, 3
GN_CHAIN_MAX_OPS = 20


class ChainOp(ctypes.Structure):
    """Mirror of `gn_chain_op` (include/gemnet_hip.h)."""
    _fields_ = [
        ("kind", _i), ("slot", _i), ("a_slot", _i), ("width", _i), ("ld", _i),
        ("src", _vp), ("rows", _vp),
        ("W", _vp), ("N", _i), ("K", _i),
        ("act", _i), ("alpha", _f),
        ("gadd1", _vp), ("gidx1", _vp), ("gadd2", _vp), ("gidx2", _vp),
        ("pre_out", _vp),
        ("mul_slot", _i), ("mul_g", _vp),
        ("res_slot", _i), ("res_g", _vp), ("res_rows", _vp), ("beta", _f),
        ("res2_slot", _i), ("res2_g", _vp), ("beta2", _f),
        ("out", _vp),
        ("mul_mode", _i), ("y2_slot", _i), ("y2_src", _i), ("mode2", _i), ("alpha2", _f), ("Z2", _vp), ("out2", _vp),
        ("src_stage", _i), ("src_mode", _i), ("src_alpha", _f), ("srcP", _vp), ("srcQ", _vp),
    ]


class ChainArgs(ctypes.Structure):
    """Mirror of `gn_chain_args`."""
    _fields_ = [("M", _i), ("n_ops", _i), ("ops", ChainOp * GN_CHAIN_MAX_OPS)]


# name -> argtypes; every function returns int (0 = ok)
SIGNATURES = {
    "gn_gemm_f32": [ctypes.POINTER(GemmArgs), _vp],
    "gn_gemm_f32_cfg": [ctypes.POINTER(GemmArgs), _i, _vp],
    "gn_chain_f32": [_vp, _vp],               # const gn_chain_args* (kernels.chain packs the block with `struct`)
    "gn_chain_split_f32": [_vp, _i, _vp],
    "gn_pack_weight_split": [_vp, _i, _i, _i, _i, _vp, _vp],
    "gn_pack_weight_split_fmt": [_vp, _i, _i, _i, _i, _i, _vp, _vp],
    "gn_pack_weight_split_grouped": [_vp, _i, _i, _vp],
    "gn_gemm_tn_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp],
    "gn_gemm_tn_splitk": [_i, _i, _i],
    "gn_gemm_tn_grouped_f32": [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp],
    "gn_index_gpu_stage1": [_vp, _i, _vp, _vp, _i, _i, _i, _i64, ctypes.c_double, ctypes.c_double, _i, _vp,
                            _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "gn_index_gpu_padded_t": [_vp, _i, _vp, _vp, _i, _i, _i, _i64, ctypes.c_double, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp,
                              _vp, _vp, _vp, _vp, _vp],
    "gn_index_poison_f32": [_vp, _i64, _vp, _vp],
    "gn_expanded_csr_i32": [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp],
    "gn_cbf_project_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "gn_cbf_project_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "gn_force_loss_f32": [_vp, _vp, _i64, _vp, _vp, _i64, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _vp, _vp],
    "gn_index_gpu_padded_q": [_vp, _i, _vp, _vp, _i, _i, _i, _i64, ctypes.c_double, ctypes.c_double, _vp, _i, _i, _i, _i, _i,
                              _i, _i, _i, _vp, _vp, _vp, _vp],
    "gn_index_gpu_stage2": [_vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _i64, _i64,
                            _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "gn_pm_f32": [_vp, _i, _vp, _vp, _vp, _f, _vp, _i64, _vp],
    "gn_adamw_ema_step_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _f, _i, _f, _vp, _vp, _i, _vp],
    "gn_nonfinite_flag_f32": [_vp, _i64, _vp, _i, _vp],
    "gn_bil_expand_f32": [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "gn_bil_dy_multi_f32": [_vp, _vp, _i, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "gn_bmm_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "gn_rbf_aggregate_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _f, _vp],
    "gn_rbf_aggregate_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _f, _i, _vp],
    "gn_quad_angles_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "gn_quad_angles_bwd_ld_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i64, _vp],
    "gn_bil_reduce_project_ang_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp],
    "gn_bil_expand_ang_f32": [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp],
    "gn_bil_expand_atoms_ang_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "gn_bil_expand_rows_ang_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i, _i, _i, _vp],
    "gn_bil_dy_multi_ang_f32": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "gn_quad_angles_jvp_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "gn_bil_reduce_project_ang_tan_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "gn_bil_expand_ang_tan_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "gn_bil_expand_rows_ang_tan_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i, _i, _i, _vp],
    "gn_csr_build_i32": [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp],
    "gn_seg_offsets_i32": [_vp, _i64, _i64, _vp, _vp],
    "gn_gather_rows_f32": [_vp, _vp, _vp, _i64, _i, _vp],
    "gn_gather_mul_f32": [_vp, _vp, _vp, _vp, _i64, _i, _f, _vp],
    "gn_dist_fwd_f32": [_vp, _vp, _vp, _vp, _i64, _vp],
    "gn_dist_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "gn_dist_jvp_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "gn_angle_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "gn_angle_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "gn_angle_jvp_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "gn_segsum_rows_f32": [_vp, _vp, _vp, _vp, _i64, _i, _vp],
    "gn_bil_fused_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _f, _i, _vp],
    "gn_segsum_multi_f32": [_i, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp],
    "gn_bil_reduce_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "gn_bil_reduce_t_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "gn_bil_reduce_t_grouped_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "gn_bil_dot_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "gn_bil_reduce_project_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "gn_bil_reduce_project2_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "gn_bil_fused_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _f, _vp],
    "gn_bil_project_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "gn_bil_project_bwd_acc_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp],
    "gn_bessel_rbf_f32": [_vp, _vp, _vp, _i64, _i, _f, _i, _i, _i, _vp],
    "gn_sph_radial_f32": [_vp, _vp, _vp, _vp, _i64, _i, _i, _f, _i, _i, _vp],
    "gn_ylm0_f32": [_vp, _vp, _i64, _i, _i, _vp],
    "gn_ylm_f32": [_vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "gn_edge_basis_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _f, _i, _vp],
    "gn_edge_basis_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _f, _i, _vp],
    "gn_trip_basis_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp],
    "gn_trip_basis_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp],
    "gn_quad_basis_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp],
    "gn_quad_basis_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp],
    "gn_quad_basis_bwd_ld_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp],
    "gn_ssilu_f32": [_vp, _vp, _i64, _i, _vp],
    "gn_dact_mul_f32": [_vp, _vp, _i, _vp, _f, _vp, _vp, _i64, _vp],
}

_lib = None
# hbcheck.Recorder while a captured step is being checked for unordered memory accesses (debug tool, hbcheck.py): `ptr`
# reports the tensors handed to a launch, `load` returns a proxy that reports the launch itself.  None otherwise.
TRACE = None


def load():
    """Load the HIP library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib if TRACE is None else TRACE.proxy(_lib)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.gn_abi_version.restype = _i
    lib.gn_chain_wide_tile_rows.restype = _i
    lib.gn_chain_wide_tile_rows.argtypes = [_i]
    # (no setters: since ABI 13 the arithmetic of the angle-form kernels and the tuning of the wide chain layout are arguments
    #  of each launch — kernels.ANG_F16_MASK / WIDE_TILE_ROWS / WIDE_STAGGER hold the read-only host configuration)
    lib.gn_optim_blocks.restype = _i
    lib.gn_optim_blocks.argtypes = [_i64]
    lib.gn_csr_ws_bytes.restype = _i64
    lib.gn_csr_ws_bytes.argtypes = [_i64, _i64]
    lib.gn_index_gpu_ws_bytes.restype = _i64
    lib.gn_index_gpu_ws_bytes.argtypes = [_i, _i64, _i]
    lib.gn_pack_weight_split_bytes.restype = _i64
    lib.gn_pack_weight_split_bytes.argtypes = [_i, _i]
    lib.gn_error_string.restype = ctypes.c_char_p
    lib.gn_error_string.argtypes = [_i]
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _i
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().gn_error_string(code)
        raise RuntimeError(f"{what} failed: hip error {code} ({msg.decode() if msg else '?'})")


def stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    if TRACE is not None:
        TRACE.touch(t)
    return _vp(t.data_ptr())


def addr(t):
    """Raw device address of a tensor (for argument blocks packed by hand); reported to the recorder like `ptr`."""
    if TRACE is not None:
        TRACE.touch(t)
    return t.data_ptr()


def note(reads=(), writes=()):
    """Operands of the next launch that live in a device-resident table (invisible in the argument list)."""
    if TRACE is not None:
        TRACE.note(reads, writes)


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "gemnet_pytorch_amd ops run only on a HIP device (MI355X); got a CPU tensor. "
                "There is no CPU fallback — use the oracle in tests if you need a CPU result.")
