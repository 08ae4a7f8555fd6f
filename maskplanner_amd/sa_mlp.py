"""Shared MLP + BatchNorm + ReLU + max-pool of a set-abstraction level (models/pointnet2_utils.py:208-214).

Input is the grouped tensor in positions-major layout [B,S,K,C] (what ops.group produces); a 1x1 Conv2d over
[B,C,K,S] is a GEMM over the last axis and BatchNorm2d statistics are statistics over all B*S*K rows, so the
module parameters (Conv2d weight [Co,Ci,1,1], BatchNorm2d) are used as they are.
"""
import torch
import torch.nn.functional as F


def _batch_norm_rows(bn, z):
    """BatchNorm2d semantics on a [rows, C] matrix, including the module's running-stat bookkeeping."""
    if bn.training and bn.track_running_stats:
        bn.num_batches_tracked.add_(1)
        momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
    else:
        momentum = 0.0 if bn.momentum is None else bn.momentum
    use_batch_stats = bn.training or (bn.running_mean is None)
    return F.batch_norm(z, bn.running_mean if (not bn.training or bn.track_running_stats) else None,
                        bn.running_var if (not bn.training or bn.track_running_stats) else None,
                        bn.weight, bn.bias, use_batch_stats, momentum, bn.eps)


def shared_mlp_max(grouped, convs, bns):
    """grouped [B,S,K,Cin] -> [B,S,Cout] = max_K relu(bn(conv(.))) chained over the layers."""
    B, S, K, C = grouped.shape
    x = grouped.reshape(B * S * K, C)
    for conv, bn in zip(convs, bns):
        w = conv.weight.view(conv.out_channels, conv.in_channels)
        x = F.relu(_batch_norm_rows(bn, F.linear(x, w, conv.bias)))
    return x.view(B, S, K, -1).max(dim=2)[0]
