"""ctypes binding of the C-ABI declared in include/tmgcn.h.

The HIP library is the product path: there is no CPU fallback.  Loading fails loudly
(``TmgcnLibraryError``) if ``libtmgcn_hip.so`` has not been built
(``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C tm-gcn_amd/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtmgcn_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "tmgcn.h")

ACT_IDS = {None: 0, "none": 0, "relu": 1, "leaky": 2, "selu": 3}
DW_ALGOS = {None: 0, "auto": 0, "f32mfma": 1}   # TMGCN_DW_AUTO / TMGCN_DW_F32MFMA
GEMM_ALGOS = {None: 0, "auto": 0, "f32mfma": 1}  # TMGCN_GEMM_AUTO / TMGCN_GEMM_F32MFMA
ABI_VERSION = 5
SYNC_INTS = 272          # include/tmgcn.h: TMGCN_SYNC_INTS (a hand-off block of the last-block reductions)


class TmgcnLibraryError(RuntimeError):
    pass


_p = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64

# name -> (restype, argtypes); mirrors include/tmgcn.h one to one
SIGNATURES = {
    "tmgcn_abi_version": (C.c_int, []),
    "tmgcn_last_error": (C.c_char_p, []),
    "tmgcn_pool_stats": (C.c_int, [_p, _i32]),
    "tmgcn_mtransform_f32": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _i64, _i32, _i32, _p]),
    "tmgcn_mtransform_ld_f32": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _i64, _p, _i64, _i64, _i32, _i32, _p]),
    "tmgcn_spmm_csr_batched_f32": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    "tmgcn_spmm_csr_batched_f32_hint": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _i32, C.c_float, _p]),
    "tmgcn_spmm_giant_workspace_bytes": (_i64, [_i32, _i32]),
    "tmgcn_spmm_csr_batched_f32_plan": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _i32, C.c_float, _p, _p, _i32, _i32, _p, _i64, _p]),
    "tmgcn_spmm_gemm_f32_plan": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _i32, _p, _i32, _i32, _i64, _i64, _i32, _p, _p, _p, _i32, C.c_float,
                                           _p, _p, _i32, _i32, _p, _i64, _p]),
    "tmgcn_spmm_gemm_supported": (C.c_int, [_i32, _i32]),
    "tmgcn_spmm_gemm_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _i32, _p, _i32, _i32, _i64, _i64, _i32, _p, _p, _p, _i32, _p]),
    "tmgcn_spmm_gemm_f32_hint": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _i32, _p, _i32, _i32, _i64, _i64, _i32, _p, _p, _p, _i32, C.c_float, _p]),
    "tmgcn_gemm_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _i32, _i32, _i64, _i64, _i32, _i32, _p]),
    "tmgcn_gemm_bf16w_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _i32, _i32, _i64, _i64, _i32, _i32, _p]),
    "tmgcn_gemm_dw_workspace_bytes": (_i64, [_i64, _i32, _i32, _i64]),
    "tmgcn_gemm_dw_f32": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _i64, _i32, _p, _i64, _p]),
    "tmgcn_gemm_dw_act_supported": (C.c_int, [_i32, _i32]),
    "tmgcn_gemm_dw_act_f32": (C.c_int, [_p, _p, _p, _i32, _p, _i64, _i32, _i32, _i64, _p, _i64, _p]),
    "tmgcn_layer12_supported": (C.c_int, [_i32, _i32, _i32]),
    "tmgcn_layer12_fwd_f32": (C.c_int, [_p, _p, _p, _p, _p, _i32, _p, _i32, _i64, _i32, _i32, _i32, _i32, _p, _p, _p, C.c_float, _p, _i32, _p]),
    "tmgcn_layer12_fwd_pays": (C.c_int, [_i64, _i32, _i32, C.c_float]),
    "tmgcn_layer12_bwd_workspace_bytes": (_i64, [_i32, _i32, _i32, _i64, _i32]),
    "tmgcn_layer12_bwd_forms_dw2": (C.c_int, [_i64, _i32, _i32, _i32, C.c_float, _i32]),
    "tmgcn_layer12_bwd_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i32, _p, _i32, _i64, _i32, _i32, _i32, _i32, _p, _p, _p, C.c_float, _p, _i32, _p, _i64, _p]),
    "tmgcn_edge_head_supported": (C.c_int, [_i32, _i32]),
    "tmgcn_edge_head_fwd_f32": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    "tmgcn_edge_head_bwd_workspace_bytes": (_i64, [_i64, _i32, _i32]),
    "tmgcn_edge_head_bwd_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _i64, _p]),
    "tmgcn_edge_head_fwd_i32_f32": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    "tmgcn_edge_head_bwd_i32_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _i64, _p]),
    "tmgcn_wce_workspace_bytes": (_i64, [_i64]),
    "tmgcn_wce_fwd_f32": (C.c_int, [_p, _p, _p, _i64, _i32, _i64, _p, _p, _p, _i64, _p]),
    "tmgcn_wce_bwd_f32": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _i64, _p, _p]),
    "tmgcn_head_loss_supported": (C.c_int, [_i32, _i32, _i32]),
    "tmgcn_head_loss_combine_f32": (C.c_int, [_p, _i32, _p, _i64, _i32, _p]),
    "tmgcn_head_loss_lanes": (C.c_int, [_i64, _i64]),
    "tmgcn_head_loss_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "tmgcn_head_loss_sgd_f32": (C.c_int, [_p, _p, _i32, _p, _p, _p, _i64, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _p, _p, _p, _p, _i64, _p, _p]),
    "tmgcn_head_loss_f32": (C.c_int, [_p, _p, _i32, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _p, _p, _p, _p, _p, _i64, _p, _p]),
    "tmgcn_sgd_step": (C.c_int, [_p, _p, _p, _p, _i32, _i32, C.c_float, C.c_float, C.c_float, C.c_float, _i32, _i32, _i32, _p]),
    "tmgcn_cast_multi": (C.c_int, [_p, _p, _p, _i32, _i32, _p]),
    "tmgcn_scale2_f32": (C.c_int, [_p, _p, _p, _i64, _p, _p, _i64, _p]),
    "tmgcn_adj_make_keys": (C.c_int, [_p, _p, _p, _i64, _i64, _p, _p]),
    "tmgcn_coo_sort_reduce_workspace_bytes": (_i64, [_i64]),
    "tmgcn_coo_sort_reduce": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _i64, _p]),
    "tmgcn_adj_symmetrise": (C.c_int, [_p, _p, _i64, _i64, _p, _p, _p]),
    "tmgcn_adj_edge_life": (C.c_int, [_p, _p, _i64, _i64, _i32, _i32, _p, _p, _p]),
    "tmgcn_adj_identity": (C.c_int, [_i64, _i64, _p, _p, _p]),
    "tmgcn_adj_normalise": (C.c_int, [_p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "tmgcn_adj_mproduct_expand": (C.c_int, [_p, _p, _i64, _i64, _i32, _p, _i32, _i32, _i32, _p, _p, _p]),
    "tmgcn_adj_mproduct_merge_count": (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _i32, _i32, _i32, _p, _p]),
    "tmgcn_adj_mproduct_merge_fill": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _p, _i32, _i32, _i32, _p, _p, _p, _p]),
    "tmgcn_adj_keys_to_csr": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _p]),
    "tmgcn_adj_transpose_keys": (C.c_int, [_p, _p, _i64, _i64, _p, _p]),
    "tmgcn_act_fwd_f32": (C.c_int, [_p, _p, _i64, _i32, _p]),
    "tmgcn_act_bwd_f32": (C.c_int, [_p, _p, _p, _i64, _i32, _p]),
}

_lib = None


def load():
    """Load libtmgcn_hip.so (once) and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TmgcnLibraryError(
            f"{LIB_PATH} not found: build the HIP library first "
            "(make -C tm-gcn_amd/csrc, or __graft_entry__.build()). "
            "There is no CPU fallback for the TM-GCN layer."
        )
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing libamdhip64 etc.
        raise TmgcnLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise TmgcnLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


TORCH_LIB_PATH = os.path.join(_HERE, "libtmgcn_torch.so")
_torch_ops = None


def load_torch_ops():
    """Load the TORCH_LIBRARY layer (libtmgcn_torch.so, built from csrc/torch_ops.cpp) once and
    return ``torch.ops.tmgcn``.  No fallback: a missing library raises."""
    global _torch_ops
    if _torch_ops is not None:
        return _torch_ops
    load()                                            # the C-ABI library it links against
    if not os.path.exists(TORCH_LIB_PATH):
        raise TmgcnLibraryError(
            f"{TORCH_LIB_PATH} not found: build the torch extension layer first "
            "(make -C tm-gcn_amd/csrc torch, or __graft_entry__.build()).")
    import torch
    try:
        torch.ops.load_library(TORCH_LIB_PATH)
    except OSError as e:
        raise TmgcnLibraryError(f"cannot load {TORCH_LIB_PATH}: {e}") from e
    ops = torch.ops.tmgcn
    if int(ops.abi_version()) != ABI_VERSION:
        raise TmgcnLibraryError(f"{TORCH_LIB_PATH} was built against ABI {int(ops.abi_version())}, expected {ABI_VERSION}")
    _torch_ops = ops
    return ops


def check(rc, what):
    if rc != 0:
        msg = load().tmgcn_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (status {rc}): {msg}")
