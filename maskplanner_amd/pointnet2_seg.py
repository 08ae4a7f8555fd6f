"""PointNet++ segmenters on the MI355X set-abstraction stack: mirrors `models/pointnet2_seg.py` of the reference.

The two instantiable classes of the reference -- PointNet2Segmenter_v1 (:14-96) and PointNet2Segmenter_PaintNet_v1
(:258-339); v2/v3/v4 raise NotImplementedError upstream -- share one structure: the SSG encoder (sa1..sa3 -> a
[B,1024] global feature), then a per-point Conv1d head over cat(global feature repeated N times, input).  Same
constructor arguments, forward outputs and state_dict keys/shapes/order as the reference.

The encoder runs the HIP kernels.  In the head, `conv1` is a 1x1 convolution over 1024+D channels of which the first
1024 are CONSTANT per cloud: the reference materialises [B,1024+D,N] and multiplies all of it (86 GMAC at B=32,
N=5120); here the global part is one [B,1024]x[1024,512] product added to the per-point D-channel part -- the same
sum, 1/340th of the work.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .pointnet2_utils import PointNetSetAbstraction


class _SegBase(nn.Module):
    def zero_grad(self, set_to_none=True):
        """nn.Module.zero_grad through the flattened parameter list (graphed.fast_zero_grad): the reference's loop calls it twice per iteration
        (train_maskplanner.py:183, 226), 0.1 ms of host time each with the device idle."""
        from . import graphed
        graphed.fast_zero_grad(self, set_to_none)

    def _build(self, in_channel):
        self.sa1 = PointNetSetAbstraction(npoint=512, radius=0.2, nsample=32, in_channel=in_channel, mlp=[64, 64, 128],
                                          group_all=False)
        self.sa2 = PointNetSetAbstraction(npoint=128, radius=0.4, nsample=64, in_channel=128 + 3, mlp=[128, 128, 256],
                                          group_all=False)
        self.sa3 = PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=256 + 3, mlp=[256, 512, 1024],
                                          group_all=True)
        self.conv1 = nn.Conv1d(1024 + in_channel, 512, 1)
        self.conv2 = nn.Conv1d(512, 256, 1)
        self.conv3 = nn.Conv1d(256, 128, 1)

    def _trunk(self, xyz, full_points, input_set):
        """encoder + the three shared head layers -> [B,128,N]."""
        B = input_set.shape[0]
        if xyz.shape[1] != 3:
            raise NotImplementedError("FPS / ball query kernels are 3-D: use ball_in_xyz_space or 3-D inputs")
        l1_xyz, l1_points = self.sa1(xyz, None, full_points=full_points)
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points)
        _, l3_points = self.sa3(l2_xyz, l2_points)
        g = l3_points.reshape(B, 1024)
        w = self.conv1.weight[:, :, 0]                                  # [512, 1024 + D]
        x = F.linear(g, w[:, :1024], self.conv1.bias)[:, :, None] + F.conv1d(input_set, w[:, 1024:, None])
        x = F.relu(self.bn1(x))
        x = F.relu(self.bn2(self.conv2(x)))
        return F.relu(self.bn3(self.conv3(x)))


class PointNet2Segmenter_v1(_SegBase):
    """models/pointnet2_seg.py:14-96.  forward(input_set [B,D,N]) -> [B,N,outdim]."""

    def __init__(self, outdim=2, input_orient_dim=0, lambda_points=1, ball_in_xyz_space=False):
        super().__init__()
        self.ball_in_xyz_space = ball_in_xyz_space
        self.lambda_points = lambda_points
        self.input_orient_dim = input_orient_dim
        self.outdim = outdim
        self.in_channel = (3 + input_orient_dim) * lambda_points
        self._build(self.in_channel)
        self.conv4 = nn.Conv1d(128, outdim, 1)
        self.bn1 = nn.BatchNorm1d(512)
        self.bn2 = nn.BatchNorm1d(256)
        self.bn3 = nn.BatchNorm1d(128)

    def forward(self, input_set, **kwargs):
        B, D, N = input_set.shape
        if self.ball_in_xyz_space:
            # sampling / neighbourhoods on the segment centroids in R^3, full segments as features (:58-63)
            poses = input_set.unsqueeze(-1).reshape(B, N, self.lambda_points, self.in_channel // self.lambda_points)
            xyz = poses[:, :, :, :3].mean(dim=-2).permute(0, 2, 1)
            full_points = input_set
        else:
            xyz, full_points = input_set, None
        x = self.conv4(self._trunk(xyz, full_points, input_set))
        return x.permute(0, 2, 1)


class PointNet2Segmenter_PaintNet_v1(_SegBase):
    """models/pointnet2_seg.py:258-339.  forward(input_set [B,3,N]) -> [B,N,lambda*(outdim_trasl+outdim_orient)]."""

    def __init__(self, inputdim=3, outdim_trasl=3, outdim_orient=3, weight_orient=1., lambda_points=1):
        super().__init__()
        self.lambda_points = lambda_points
        self.outdim_trasl = outdim_trasl
        self.outdim_orient = outdim_orient
        self.weight_orient = weight_orient
        self.in_channel = inputdim
        self._build(inputdim)
        self.conv4_trasl = nn.Conv1d(128, outdim_trasl * lambda_points, 1)
        if outdim_orient > 0:
            self.conv4_orient = nn.Conv1d(128, outdim_orient * lambda_points, 1)
            self.tanh = nn.Tanh()
        self.bn1 = nn.BatchNorm1d(512)
        self.bn2 = nn.BatchNorm1d(256)
        self.bn3 = nn.BatchNorm1d(128)

    def forward(self, input_set, **kwargs):
        B, D, N = input_set.shape
        last = self._trunk(input_set, None, input_set)
        if self.outdim_orient <= 0:
            raise NotImplementedError()
        x = self.conv4_trasl(last).permute(0, 2, 1).reshape(B, N, self.lambda_points, -1)
        normals = torch.tanh(self.conv4_orient(last)).permute(0, 2, 1).reshape(B, N, self.lambda_points, -1)
        normals = F.normalize(normals, dim=-1) * self.weight_orient
        return torch.cat((x, normals), dim=-1).reshape(B, N, -1)


def _not_implemented(name, line, msg):
    """The reference's constructors of these variants raise before building anything (models/pointnet2_seg.py:120,193,252)."""
    def __init__(self, *args, **kwargs):
        nn.Module.__init__(self)
        raise NotImplementedError(msg)
    return type(name, (nn.Module,), {"__init__": __init__, "__doc__": f"models/pointnet2_seg.py:{line}: raises NotImplementedError upstream as well."})


PointNet2Segmenter_v2 = _not_implemented("PointNet2Segmenter_v2", 99, "TODO: SetAbstraction with sample_all_as_centroids=True flag")
PointNet2Segmenter_v3 = _not_implemented("PointNet2Segmenter_v3", 181, "TODO")
PointNet2Segmenter_v4 = _not_implemented("PointNet2Segmenter_v4", 239, "TODO")
