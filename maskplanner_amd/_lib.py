"""ctypes binding of libmaskplanner_hip.so (C ABI: include/maskplanner_hip.h).

The library is the product: there is NO CPU or eager-PyTorch fallback.  If the shared object is missing or
does not export the ABI this module raises, and every op in `maskplanner_amd.ops` refuses non-HIP tensors.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MASKPLANNER_HIP_LIB") or os.path.join(_HERE, "lib", "libmaskplanner_hip.so")  # env: diagnostic builds
ABI_VERSION = 1

MP_OK = 0
MP_EINVAL = -1
MP_EUNSUPPORTED = -2
MP_EWORKSPACE = -3
MP_ELAUNCH = -4
MASK_CAP = 64

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_dbl = ctypes.c_double
_sz = ctypes.c_size_t

_fp = ctypes.POINTER(ctypes.c_float)


class MlpLayer(ctypes.Structure):
    """mp_mlp_layer_t"""
    _fields_ = [("weight", _vp), ("bias", _vp), ("gamma", _vp), ("beta", _vp), ("running_mean", _vp),
                ("running_var", _vp), ("c_in", _i64), ("c_out", _i64), ("z", _vp), ("mean", _vp), ("rstd", _vp),
                ("scale", _vp), ("shift", _vp), ("bn_state", _vp)]


class HeadBlock(ctypes.Structure):
    """mp_head_block_t"""
    _fields_ = [("x", _vp), ("weight", _vp), ("bias", _vp), ("O", _i64), ("bn", _int), ("training", _int), ("momentum", _dbl), ("eps", _dbl),
                ("gamma", _vp), ("beta", _vp), ("running_mean", _vp), ("running_var", _vp), ("z", _vp), ("y", _vp), ("save_mean", _vp),
                ("save_rstd", _vp), ("drop_p", _dbl), ("rng", _vp), ("layer", _int), ("grad_y", _vp), ("dz", _vp), ("grad_gamma", _vp),
                ("grad_beta", _vp), ("grad_x", _vp)]


class MlpGrads(ctypes.Structure):
    """mp_mlp_grads_t"""
    _fields_ = [("d_weight", _vp), ("d_bias", _vp), ("d_gamma", _vp), ("d_beta", _vp)]


# mp_allreduce_f64_fn / mp_syncbn_t: the caller-supplied collective of the SyncBN option (mp_sa_mlp_{fwd,bwd}_ex)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)


class Gather(ctypes.Structure):
    """mp_gather_t"""
    _fields_ = [("feats", _vp), ("xyz", _vp), ("new_xyz", _vp), ("idx", _vp), ("N", _i64), ("S", _i64), ("CF", _i64), ("rows", _vp)]


class SyncBN(ctypes.Structure):
    """mp_syncbn_t"""
    _fields_ = [("allreduce", ALLREDUCE_FN), ("user", _vp), ("world", _i64), ("exchange", _vp)]


# name -> (restype, argtypes).  One entry per symbol declared in include/maskplanner_hip.h.
SIGNATURES = {
    "mp_abi_version": (_int, []),
    "mp_error_string": (ctypes.c_char_p, [_int]),
    "mp_fps_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "mp_fps_floor_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "mp_ball_query_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _dbl, _i64, _vp, _vp]),
    "mp_ball_query_multi_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, ctypes.POINTER(_dbl), ctypes.POINTER(_i64), ctypes.POINTER(_vp), _vp]),
    "mp_square_distance_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "mp_index_points_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "mp_index_points_bwd_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _vp]),
    "mp_group_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _i64, _vp, _vp]),
    "mp_group_bwd_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _i64, _vp, _int, _vp]),
    "mp_three_nn_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "mp_three_interpolate_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "mp_three_interpolate_bwd_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _vp]),
    "mp_lsap_f32": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "mp_cdist_batch_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "mp_chamfer_reduce_f32": (_int, [_vp, _vp, _i64, _i64, _int, _int, _dbl, _dbl, _vp, _vp, _vp, _vp]),
    "mp_chamfer_reduce1_f32": (_int, [_vp, _vp, _i64, _i64, _int, _int, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp]),
    "mp_chamfer_reduce_bwd_f32": (_int, [_vp, _vp, _i64, _i64, _int, _int, _dbl, _dbl, _vp, _vp]),
    "mp_permute_cols_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "mp_permute_cols_multi_f32": (_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mp_pose_output_f32": (_int, [_vp, _vp, _i64, _dbl, _vp, _vp]),
    "mp_pose_output_bwd_f32": (_int, [_vp, _vp, _i64, _dbl, _vp, _vp, _vp]),
    "mp_mask_loss_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _dbl, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mp_mask_loss_bwd_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _dbl, _dbl, _dbl, _vp, _vp, _vp]),
    "mp_bn_relu_rows_f32": (_int, [_vp, _i64, _i64, _int, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mp_bn_relu_rows_bwd_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mp_bn_relu_drop_rows_f32": (_int, [_vp, _i64, _i64, _int, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _vp, _int, _vp]),
    "mp_bn_relu_drop_rows_bwd_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _vp]),
    "mp_knn_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "mp_knn_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "mp_knn1_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "mp_knn1_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _int, _vp, _sz, _vp]),
    "mp_knn_bwd_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int, _vp]),
    "mp_knn1_prepare_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _sz, _vp]),
    "mp_knn1_prepared_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "mp_knn_bwd_reduced_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _dbl, _dbl, _i64, _i64, _i64, _i64, _vp, _vp, _int, _vp]),
    "mp_padded_lengths_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "mp_mask_match_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mp_adam_lowrank_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _dbl, _dbl, _dbl, _dbl, _dbl, _i64, _vp, _vp]),
    "mp_linear_dx_skinny_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "mp_linear_dx_skinny_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "mp_csr_rows_i64": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "mp_dw_gemm_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "mp_head_blocks_fwd_f32": (_int, [_int, _vp, _i64, _i64, _vp]),
    "mp_head_blocks_bwd_f32": (_int, [_int, _vp, _i64, _i64, _vp]),
    "mp_head_block_supported": (_int, [_i64, _i64, _i64]),
    "mp_head_block_fwd_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _int, _int, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _vp, _int, _vp]),
    "mp_head_linear2_fwd_f32": (_int, [_vp, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp]),
    "mp_linear_dx_mfma2_f32": (_int, [_vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "mp_head_block_bwd_slices": (_int, [_i64]),
    "mp_head_block_bwd_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _int, _vp, _vp, _vp, _dbl, _vp, _vp, _vp, _vp, _vp]),
    "mp_linear_dx_mfma_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "mp_linear_dw_outer_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "mp_zero_arena_arm": (_int, [_vp, _sz, _vp]),
    "mp_zero_arena_arm_ticks": (_int, [_vp, _sz, _int, _vp, _int, _vp, _vp]),
    "mp_zero_arena_disarm": (_int, []),
    "mp_zero_arena_disarm_stream": (_int, [_vp]),
    "mp_profiler_enable": (_int, [_int]),
    "mp_profiler_collect": (_int, [ctypes.c_char_p, _sz]),
    "mp_profiler_mark": (_int, [ctypes.c_char_p]),
    "mp_profiler_read_marks": (_int, [ctypes.c_char_p, _sz, _int]),
    "mp_sa_mlp_workspace_bytes": (_sz, [_i64, _i64, _int, ctypes.POINTER(_i64), _int]),
    "mp_sa_mlp_recompute_first": (_int, [_int, ctypes.POINTER(_i64), _i64]),
    "mp_sa_mlp_bf16_storage": (_int, [_int, ctypes.POINTER(_i64), _i64, _int]),
    "mp_adam_multi_f32": (_int, [_i64, _vp, _vp, _vp, _vp, ctypes.POINTER(_i64), ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                 ctypes.c_double, ctypes.c_double, _i64, _vp, _vp]),
    "mp_colsum_multi_f32": (_int, [_i64, _vp, _vp, ctypes.POINTER(_i64), _i64, _vp]),
    "mp_pad_ragged_f32": (_int, [_vp, _vp, _i64, _i64, _i64, ctypes.c_float, _vp, _vp]),
    "mp_lambda_segments_f32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "mp_sa_mlp_fwd_gather_f32": (_int, [ctypes.POINTER(Gather), _i64, _i64, _int, ctypes.POINTER(MlpLayer), _int, _dbl, _dbl, _vp, _vp, _vp, _vp,
                                        _sz, _vp]),
    "mp_sa_mlp_fwd_gather_bf16": (_int, [ctypes.POINTER(Gather), _i64, _i64, _int, ctypes.POINTER(MlpLayer), _int, _dbl, _dbl, _vp, _vp, _vp, _vp,
                                        _sz, _vp]),
    "mp_sa_mlp_bwd_gather_f32": (_int, [ctypes.POINTER(Gather), _i64, _i64, _int, ctypes.POINTER(MlpLayer), _int, _vp, _vp, _vp, _vp,
                                        ctypes.POINTER(MlpGrads), _vp, _i64, _vp, _sz, _vp]),
    "mp_sa_mlp_bwd_gather_bf16": (_int, [ctypes.POINTER(Gather), _i64, _i64, _int, ctypes.POINTER(MlpLayer), _int, _vp, _vp, _vp, _vp,
                                        ctypes.POINTER(MlpGrads), _vp, _i64, _vp, _sz, _vp]),
    "mp_sa_mlp_fwd_f32": (_int, [_vp, _i64, _i64, _int, ctypes.POINTER(MlpLayer), _int, _dbl, _dbl, _vp, _vp, _vp, _vp,
                                 _sz, _vp]),
    "mp_sa_mlp_bwd_f32": (_int, [_vp, _i64, _i64, _int, ctypes.POINTER(MlpLayer), _int, _vp, _vp, _vp, _vp,
                                 ctypes.POINTER(MlpGrads), _vp, _i64, _vp, _sz, _vp]),
    "mp_sa_mlp_fwd_gather_ex": (_int, [ctypes.POINTER(Gather), _i64, _i64, _int, ctypes.POINTER(MlpLayer), _int, _dbl, _dbl, _vp, _vp, _vp, _vp,
                                       _sz, _int, ctypes.POINTER(SyncBN), _vp]),
    "mp_sa_mlp_bwd_gather_ex": (_int, [ctypes.POINTER(Gather), _i64, _i64, _int, ctypes.POINTER(MlpLayer), _int, _vp, _vp, _vp, _vp,
                                       ctypes.POINTER(MlpGrads), _vp, _i64, _vp, _sz, _int, ctypes.POINTER(SyncBN), _vp]),
    "mp_sa_mlp_fwd_ex": (_int, [_vp, _i64, _i64, _int, ctypes.POINTER(MlpLayer), _int, _dbl, _dbl, _vp, _vp, _vp, _vp,
                                _sz, _int, ctypes.POINTER(SyncBN), _vp]),
    "mp_sa_mlp_bwd_ex": (_int, [_vp, _i64, _i64, _int, ctypes.POINTER(MlpLayer), _int, _vp, _vp, _vp, _vp,
                                ctypes.POINTER(MlpGrads), _vp, _i64, _vp, _sz, _int, ctypes.POINTER(SyncBN), _vp]),
    "mp_sa_mlp_fwd_bf16": (_int, [_vp, _i64, _i64, _int, ctypes.POINTER(MlpLayer), _int, _dbl, _dbl, _vp, _vp, _vp, _vp,
                                 _sz, _vp]),
    "mp_sa_mlp_bwd_bf16": (_int, [_vp, _i64, _i64, _int, ctypes.POINTER(MlpLayer), _int, _vp, _vp, _vp, _vp,
                                 ctypes.POINTER(MlpGrads), _vp, _i64, _vp, _sz, _vp]),
}

_lib = None


class MaskPlannerHipError(RuntimeError):
    pass


def _promote_hip_runtime():
    """libmaskplanner_hip.so is linked without a NEEDED entry for the HIP runtime (csrc/Makefile): it binds to
    the runtime already in the process.  PyTorch-ROCm ships its own libamdhip64.so and loads it with local
    visibility; re-opening it RTLD_GLOBAL makes its symbols the ones our library resolves against, so
    kernels launch on the very runtime instance that owns torch's streams and allocations."""
    import torch  # noqa: F401  (loads torch's HIP runtime)
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    else:  # a torch build that uses the system ROCm
        ctypes.CDLL("libamdhip64.so", mode=ctypes.RTLD_GLOBAL)


def load():
    """Load the library, bind every ABI symbol, check the ABI version.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MaskPlannerHipError(
            f"{LIB_PATH} not found: build it with `make -C maskplanner_amd/csrc` (or __graft_entry__.build()). "
            "There is no CPU / PyTorch fallback for the MaskPlanner hot path.")
    _promote_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MaskPlannerHipError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.mp_abi_version()
    if got != ABI_VERSION:
        raise MaskPlannerHipError(f"ABI mismatch: library {got}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, op):
    if rc == MP_OK:
        return
    msg = load().mp_error_string(rc).decode()
    if rc == MP_EINVAL:
        raise ValueError(f"{op}: {msg}")
    raise MaskPlannerHipError(f"{op}: {msg} (code {rc})")
