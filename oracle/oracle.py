"""ctypes/numpy front-end of oracle/mp_oracle.c -- TEST INFRASTRUCTURE ONLY.

Each wrapper takes/returns numpy arrays in the reference's dtypes (f32 values, i64 indices)
and documents the reference lines it restates (see mp_oracle.c for the arithmetic).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmp_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64 = ctypes.c_int64


def build(force=False):
    """Compile mp_oracle.c with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "mp_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64a(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _check(rc, name):
    if rc != 0:
        raise RuntimeError(f"oracle {name} failed with code {rc}")


def fps(xyz, npoint, start_idx):
    """models/pointnet2_utils.py:65-86 with an explicit start index per cloud."""
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    start = _i64a(start_idx)
    out = np.empty((B, npoint), dtype=np.int64)
    _check(lib().mpo_fps_f32(_p(xyz, _f32p), _i64(B), _i64(N), _i64(npoint), _p(start, _i64p), _p(out, _i64p)), "fps")
    return out


def square_distance(src, dst):
    """models/pointnet2_utils.py:21-42 (expanded form, reference rounding)."""
    src, dst = _f32(src), _f32(dst)
    B, S, _ = src.shape
    N = dst.shape[1]
    out = np.empty((B, S, N), dtype=np.float32)
    _check(lib().mpo_square_distance_f32(_p(src, _f32p), _p(dst, _f32p), _i64(B), _i64(S), _i64(N), _p(out, _f32p)), "square_distance")
    return out


def ball_query(radius, nsample, xyz, new_xyz, return_counts=False):
    """models/pointnet2_utils.py:89-109."""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = np.empty((B, S, nsample), dtype=np.int64)
    cnt = np.empty((B, S), dtype=np.int32)
    _check(lib().mpo_ball_query_f32(_p(xyz, _f32p), _p(new_xyz, _f32p), _i64(B), _i64(N), _i64(S),
                                    ctypes.c_double(float(radius)), _i64(nsample), _p(out, _i64p), _p(cnt, _i32p)), "ball_query")
    return (out, cnt) if return_counts else out


def index_points(points, idx):
    """models/pointnet2_utils.py:45-62."""
    points = _f32(points)
    idx = _i64a(idx)
    B, N, C = points.shape
    M = int(np.prod(idx.shape[1:]))
    out = np.empty((B, M, C), dtype=np.float32)
    _check(lib().mpo_index_points_f32(_p(points, _f32p), _p(idx, _i64p), _i64(B), _i64(N), _i64(C), _i64(M), _p(out, _f32p)), "index_points")
    return out.reshape(*idx.shape, C)


def index_points_bwd(grad_out, idx, N):
    grad_out = _f32(grad_out)
    idx = _i64a(idx)
    B = idx.shape[0]
    C = grad_out.shape[-1]
    M = int(np.prod(idx.shape[1:]))
    out = np.empty((B, N, C), dtype=np.float32)
    _check(lib().mpo_index_points_bwd_f32(_p(grad_out, _f32p), _p(idx, _i64p), _i64(B), _i64(N), _i64(C), _i64(M), _p(out, _f32p)), "index_points_bwd")
    return out


def three_nn(xyz1, xyz2):
    """models/pointnet2_utils.py:310-316: 3 nearest xyz2 points of every xyz1 point (expanded-form distances,
    lowest index first on ties) and the normalised inverse-distance weights.  -> (dist, idx, weight), each [B,N,3]."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    dist = np.empty((B, N, 3), dtype=np.float32)
    idx = np.empty((B, N, 3), dtype=np.int64)
    w = np.empty((B, N, 3), dtype=np.float32)
    _check(lib().mpo_three_nn_f32(_p(xyz1, _f32p), _p(xyz2, _f32p), _i64(B), _i64(N), _i64(S), _p(dist, _f32p), _p(idx, _i64p),
                                  _p(w, _f32p)), "three_nn")
    return dist, idx, w


def three_interpolate(points2, idx, weight):
    """models/pointnet2_utils.py:317: sum_k points2[b, idx[b,n,k], :] * weight[b,n,k]."""
    points2, weight = _f32(points2), _f32(weight)
    idx = _i64a(idx)
    B, S, D = points2.shape
    N = idx.shape[1]
    out = np.empty((B, N, D), dtype=np.float32)
    _check(lib().mpo_three_interpolate_f32(_p(points2, _f32p), _p(idx, _i64p), _p(weight, _f32p), _i64(B), _i64(N), _i64(S), _i64(D),
                                           _p(out, _f32p)), "three_interpolate")
    return out


def three_interpolate_bwd(grad_out, idx, weight, S):
    grad_out, weight = _f32(grad_out), _f32(weight)
    idx = _i64a(idx)
    B, N, D = grad_out.shape
    out = np.empty((B, S, D), dtype=np.float32)
    _check(lib().mpo_three_interpolate_bwd_f32(_p(grad_out, _f32p), _p(idx, _i64p), _p(weight, _f32p), _i64(B), _i64(N), _i64(S),
                                               _i64(D), _p(out, _f32p)), "three_interpolate_bwd")
    return out


def group(xyz, feats, new_xyz, idx):
    """models/pointnet2_utils.py:133-143: cat([xyz[idx]-new_xyz, feats[idx]], -1)."""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    idx = _i64a(idx)
    B, N, _ = xyz.shape
    _, S, K = idx.shape
    D = 0 if feats is None else feats.shape[-1]
    feats = None if feats is None else _f32(feats)
    out = np.empty((B, S, K, 3 + D), dtype=np.float32)
    _check(lib().mpo_group_f32(_p(xyz, _f32p), _p(feats, _f32p), _p(new_xyz, _f32p), _p(idx, _i64p),
                               _i64(B), _i64(N), _i64(S), _i64(K), _i64(D), _p(out, _f32p)), "group")
    return out


def knn_points(p1, p2, lengths1=None, lengths2=None, K=1):
    """pytorch3d.ops.knn.knn_points contract (see mp_oracle.c header: parity unpinned)."""
    p1, p2 = _f32(p1), _f32(p2)
    B, P1, D = p1.shape
    P2 = p2.shape[1]
    l1 = None if lengths1 is None else _i64a(lengths1)
    l2 = None if lengths2 is None else _i64a(lengths2)
    dists = np.empty((B, P1, K), dtype=np.float32)
    idx = np.empty((B, P1, K), dtype=np.int64)
    _check(lib().mpo_knn_f32(_p(p1, _f32p), _p(p2, _f32p), _p(l1, _i64p), _p(l2, _i64p), _i64(B), _i64(P1), _i64(P2),
                             _i64(D), _i64(K), _p(dists, _f32p), _p(idx, _i64p)), "knn")
    return dists, idx


def knn_points_bwd(p1, p2, lengths1, lengths2, idx, grad_dists):
    p1, p2 = _f32(p1), _f32(p2)
    B, P1, D = p1.shape
    P2 = p2.shape[1]
    idx = _i64a(idx)
    K = idx.shape[-1]
    g = _f32(grad_dists)
    l1 = None if lengths1 is None else _i64a(lengths1)
    l2 = None if lengths2 is None else _i64a(lengths2)
    g1 = np.empty_like(p1)
    g2 = np.empty_like(p2)
    _check(lib().mpo_knn_bwd_f32(_p(p1, _f32p), _p(p2, _f32p), _p(l1, _i64p), _p(l2, _i64p), _p(idx, _i64p), _p(g, _f32p),
                                 _i64(B), _i64(P1), _i64(P2), _i64(D), _i64(K), _p(g1, _f32p), _p(g2, _f32p)), "knn_bwd")
    return g1, g2


def padded_lengths(y):
    """pytorch3d_chamfer.py:138-149."""
    y = _f32(y)
    B, P2, D = y.shape
    out = np.empty((B,), dtype=np.int64)
    _check(lib().mpo_padded_lengths_f32(_p(y, _f32p), _i64(B), _i64(P2), _i64(D), _p(out, _i64p)), "padded_lengths")
    return out


def mask_bce_cost(pred, tgt):
    """loss_handler.py:804-813,863-873: cost[m,k] = sum_s BCEWithLogits(pred[m,s], tgt[k,s])."""
    pred, tgt = _f32(pred), _f32(tgt)
    M, S = pred.shape
    Kt = tgt.shape[0]
    out = np.empty((M, Kt), dtype=np.float32)
    _check(lib().mpo_mask_bce_cost_f32(_p(pred, _f32p), _p(tgt, _f32p), _i64(M), _i64(Kt), _i64(S), _p(out, _f32p)), "mask_bce_cost")
    return out


def stroke_ids_to_masks(ids):
    """loss_handler.py:938-967 (binary masks, sorted unique ids, -1 skipped)."""
    ids = _f32(ids)
    S = ids.shape[0]
    masks = np.empty((S, S), dtype=np.float32)
    uniq = np.empty((S,), dtype=np.float32)
    l = lib()
    l.mpo_stroke_ids_to_masks_f32.restype = ctypes.c_int64
    nk = l.mpo_stroke_ids_to_masks_f32(_p(ids, _f32p), _i64(S), _p(masks, _f32p), _p(uniq, _f32p), _i64(S))
    if nk < 0:
        raise RuntimeError(f"oracle stroke_ids_to_masks failed with code {nk}")
    return masks[:nk].copy(), uniq[:nk].copy()


def linear_sum_assignment(cost):
    """scipy.optimize.linear_sum_assignment restated (rows ascending)."""
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    n = min(nr, nc)
    a = np.empty((n,), dtype=np.int64)
    b = np.empty((n,), dtype=np.int64)
    _check(lib().mpo_lsap_f64(_p(cost, _f64p), _i64(nr), _i64(nc), _p(a, _i64p), _p(b, _i64p)), "lsap")
    return a, b


def cdist(x, y):
    """models/hungarianMatcher.py:51 per-sample block of torch.cdist(p=2)."""
    x, y = _f32(x), _f32(y)
    P, D = x.shape
    R = y.shape[0]
    out = np.empty((P, R), dtype=np.float32)
    _check(lib().mpo_cdist_f32(_p(x, _f32p), _p(y, _f32p), _i64(P), _i64(R), _i64(D), _p(out, _f32p)), "cdist")
    return out


def num_threads():
    return int(lib().mpo_num_threads())
