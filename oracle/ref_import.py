"""Import pieces of the upstream MaskPlanner checkout on a GPU-less box -- TEST INFRASTRUCTURE ONLY.

Used only by oracle/gen_golden.py (and optional local cross-checks) in the build container, where
the reference lives at /root/reference.  Nothing here runs on the GPU box: the reference does not
travel, only the fixtures it produced (tests/golden/*.npz) do.

The reference modules are loaded *by file path*; missing third-party packages are replaced by
inert stub modules, and hard-coded `.cuda()` moves are neutralised (this box has no GPU).  No
reference source is copied.
"""
import importlib.util
import os
import sys
import types
from collections import namedtuple

import numpy as np
import torch

REF_ROOT = os.environ.get("MASKPLANNER_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def neutralise_cuda():
    """`.cuda()`, `.to('cuda')`, `.get_device()` become no-ops on this CPU-only box."""
    if getattr(torch.Tensor, "_mp_patched", False):
        return
    orig_to = torch.Tensor.to

    def to(self, *args, **kwargs):
        args = tuple(a for a in args if not (isinstance(a, str) and a.startswith("cuda")))
        if isinstance(kwargs.get("device", None), str) and kwargs["device"].startswith("cuda"):
            kwargs.pop("device")
        if args and isinstance(args[0], int):  # .to(get_device()) with our -1/0 device id
            args = args[1:]
        if not args and not kwargs:
            return self
        return orig_to(self, *args, **kwargs)

    torch.Tensor.to = to
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.get_device = lambda self: 0
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor._mp_patched = True


def pointnet2_utils():
    """models/pointnet2_utils.py -- imports natively (torch + numpy only)."""
    if "mp_ref_models.pointnet2_utils" in sys.modules:
        return sys.modules["mp_ref_models.pointnet2_utils"]
    pkg = types.ModuleType("mp_ref_models")
    pkg.__path__ = [os.path.join(REF_ROOT, "models")]
    sys.modules["mp_ref_models"] = pkg
    return _load("mp_ref_models.pointnet2_utils", "models/pointnet2_utils.py")


def pointnet2_cls_ssg():
    pointnet2_utils()
    if "mp_ref_models.pointnet2_cls_ssg" in sys.modules:
        return sys.modules["mp_ref_models.pointnet2_cls_ssg"]
    return _load("mp_ref_models.pointnet2_cls_ssg", "models/pointnet2_cls_ssg.py")


def pointnet2_seg():
    """models/pointnet2_seg.py: imports `models.pointnet2_utils` absolutely -> alias it to the loaded reference file."""
    if "mp_ref_models.pointnet2_seg" in sys.modules:
        return sys.modules["mp_ref_models.pointnet2_seg"]
    pu = pointnet2_utils()
    if "models" not in sys.modules:
        pkg = _stub("models")
        pkg.__path__ = []
    sys.modules["models.pointnet2_utils"] = pu
    return _load("mp_ref_models.pointnet2_seg", "models/pointnet2_seg.py")


def hungarian_matcher():
    if "mp_ref_models.hungarianMatcher" in sys.modules:
        return sys.modules["mp_ref_models.hungarianMatcher"]
    pointnet2_utils()
    return _load("mp_ref_models.hungarianMatcher", "models/hungarianMatcher.py")


# ---------------------------------------------------------------------------------------------
# pytorch3d stand-in: the arithmetic is the ORACLE's knn (contract-derived, see mp_oracle.c);
# what this pins is the reference WRAPPER logic in pytorch3d_chamfer.py around it.
# ---------------------------------------------------------------------------------------------
_KNN = namedtuple("KNN", "dists idx knn")


class _OracleKnn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p1, p2, lengths1, lengths2, K):
        from oracle import oracle as O
        d, i = O.knn_points(p1.detach().numpy(), p2.detach().numpy(), lengths1.numpy(), lengths2.numpy(), K)
        d, i = torch.from_numpy(d), torch.from_numpy(i)
        ctx.save_for_backward(p1, p2, lengths1, lengths2, i)
        ctx.mark_non_differentiable(i)
        return d, i

    @staticmethod
    def backward(ctx, gd, gi):
        from oracle import oracle as O
        p1, p2, l1, l2, i = ctx.saved_tensors
        g1, g2 = O.knn_points_bwd(p1.detach().numpy(), p2.detach().numpy(), l1.numpy(), l2.numpy(), i.numpy(),
                                  gd.contiguous().numpy())
        return torch.from_numpy(g1), torch.from_numpy(g2), None, None, None


def _knn_points(p1, p2, lengths1=None, lengths2=None, K=1, **kw):
    B, P1, _ = p1.shape
    P2 = p2.shape[1]
    if lengths1 is None:
        lengths1 = torch.full((B,), P1, dtype=torch.int64)
    if lengths2 is None:
        lengths2 = torch.full((B,), P2, dtype=torch.int64)
    d, i = _OracleKnn.apply(p1.contiguous(), p2.contiguous(), lengths1.to(torch.int64), lengths2.to(torch.int64), K)
    return _KNN(d, i, None)


def _knn_gather(x, idx, lengths=None):
    B, M, U = x.shape
    _, L, K = idx.shape
    return x[:, :, None].expand(-1, -1, K, -1).gather(1, idx[:, :, :, None].expand(-1, -1, -1, U))


def install_pytorch3d_stub():
    class Pointclouds:  # isinstance() target only
        pass

    _stub("pytorch3d")
    _stub("pytorch3d.ops")
    _stub("pytorch3d.ops.knn", knn_points=_knn_points, knn_gather=_knn_gather)
    _stub("pytorch3d.structures")
    _stub("pytorch3d.structures.pointclouds", Pointclouds=Pointclouds)


def chamfer_module():
    """pytorch3d_chamfer.py with the pytorch3d stand-in."""
    if "pytorch3d_chamfer" in sys.modules:
        return sys.modules["pytorch3d_chamfer"]
    install_pytorch3d_stub()
    neutralise_cuda()
    return _load("pytorch3d_chamfer", "pytorch3d_chamfer.py")


def loss_handler_module():
    """loss_handler.py with inert stand-ins for the unrelated model/util imports it drags in."""
    if "mp_ref_loss_handler" in sys.modules:
        return sys.modules["mp_ref_loss_handler"]
    chamfer_module()
    pc = pointnet2_cls_ssg()
    hm = hungarian_matcher()

    class _Unused:  # discriminators / baselines: never instantiated on the maskplanner path
        def __init__(self, *a, **k):
            raise RuntimeError("out-of-scope reference component")

    def orient_in(extra_data):
        for v in ("orientquat", "orientrotvec", "orientnorm"):
            if v in extra_data:
                return True, v
        return False, None

    def get_dim_traj_points(extra_data):
        # value table of utils/pointcloud.py:478-491 (behavioural fact, not code)
        if len(extra_data) == 0:
            return 3
        table = {"vel": 6, "orientquat": 7, "orientrotvec": 6, "orientnorm": 6}
        if len(extra_data) == 1 and extra_data[0] in table:
            return table[extra_data[0]]
        raise ValueError("unsupported extra_data")

    models_pkg = _stub("models")
    models_pkg.__path__ = []
    _stub("models.dgcnn", DGCNNDiscriminator=_Unused)
    _stub("models.pointnet2_cls_ssg", PointNet2Regressor=pc.PointNet2Regressor)
    _stub("models.pointnet", PointNetRegressor=_Unused)
    _stub("models.mlp", MLP=_Unused)
    _stub("models.gradient_penalty", GradientPenalty=_Unused)
    _stub("models.hungarianMatcher", HungarianMatcher=hm.HungarianMatcher)
    utils_pkg = _stub("utils", orient_in=orient_in)
    utils_pkg.__path__ = []
    _stub("utils.pointcloud", get_dim_traj_points=get_dim_traj_points, mean_knn_distance=None)
    return _load("mp_ref_loss_handler", "loss_handler.py")


class AttrDict(dict):
    """dict with attribute access: stands in for the OmegaConf config object."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def maskplanner_loss_config(**over):
    """Merged values of configs/maskplanner/{default,asymm_chamfer_v9}.yaml that the loss path reads."""
    cfg = AttrDict(
        extra_data=["orientnorm"], lambda_points=4, overlapping=1, weight_orient=0.25,
        weight_asymm_segment_chamfer=1.0, weight_reverse_asymm_point_chamfer=100, weight_reverse_asymm_segment_chamfer=0.01,
        explicit_weight_stroke_masks=1.0, explicit_no_stroke_weight=1.0, explicit_weight_stroke_masks_confidence=100.0,
        weight_asymm_v6_chamfer_with_stroke_masks=1.0, per_segment_confidence=False, smooth_target_stroke_masks=False,
        explicit_weight_segments_confidence=10.0, min_centroids=False, stroke_pred=False, knn_repulsion=1,
    )
    cfg.update(over)
    return cfg
