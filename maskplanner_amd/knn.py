"""`pytorch3d.ops.knn` look-alike backed by the gfx950 kNN kernel.

The reference's chamfer wrapper does `from pytorch3d.ops.knn import knn_gather, knn_points`
(pytorch3d_chamfer.py:12); registering this module under that name (see maskplanner_amd.dropin) lets the
reference file run unmodified on MI355X, where pytorch3d's CUDA extension does not exist.
Contract restated from the public pytorch3d API (0.7.x): squared L2, per-cloud lengths, K smallest ascending,
first index on ties, rows beyond lengths1 / slots beyond lengths2 zero-filled, autograd through `dists`.
"""
from collections import namedtuple

import torch

from . import ops

_KNN = namedtuple("KNN", "dists idx knn")


def knn_points(p1, p2, lengths1=None, lengths2=None, norm=2, K=1, version=-1, return_nn=False, return_sorted=True):
    if p1.shape[0] != p2.shape[0]:
        raise ValueError("pts1 and pts2 must have the same batch dimension.")
    if p1.shape[2] != p2.shape[2]:
        raise ValueError("pts1 and pts2 must have the same point dimension.")
    if norm != 2:
        raise NotImplementedError("only the squared-L2 form is on the MaskPlanner path")
    dists, idx = ops.knn(p1, p2, lengths1, lengths2, K)
    nn = knn_gather(p2, idx, lengths2) if return_nn else None
    return _KNN(dists=dists, idx=idx, knn=nn)


def knn_gather(x, idx, lengths=None):
    """x [B,M,U], idx [B,L,K] -> [B,L,K,U]; neighbours beyond lengths are zero-filled."""
    B, M, U = x.shape
    _, L, K = idx.shape
    out = ops.index_points(x, idx)
    if lengths is not None:
        valid = torch.arange(K, device=x.device)[None, None, :] < lengths.to(x.device)[:, None, None]
        out = out * valid[..., None].to(out.dtype)
    return out
