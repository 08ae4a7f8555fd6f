"""Drop-in for `models/hungarianMatcher.py` (HungarianMatcher.forward, :31-63): one-to-one matching of predicted
and GT segments by Euclidean cost, used by the `emd` / `hungarian_SoPs` losses (loss_handler.py:172-174, 1000).

The reference builds ONE [B*S, sum(Sgt)] cdist matrix (every prediction against the GT of every sample, O(B^2)
wasted work, 3.7 GB at B=32) and slices the diagonal blocks; here each sample's [S, Sgt_b] block is computed on
its own.  The assignment itself (999 x ~900 per sample) is solved with scipy's rectangular LSAP on the host, as
in the reference (:61): the large device LAP is the next item of the scope table, the small
stroke-mask LAP that sits on the training path already runs on the GPU (csrc/mask_match.hip).
"""
import torch
from scipy.optimize import linear_sum_assignment
from torch import nn


class HungarianMatcher(nn.Module):
    def __init__(self):
        super().__init__()

    @torch.no_grad()
    def forward(self, outputs, targets):
        """outputs [B,S,D]; targets: list of B tensors [Sgt_b, D].  Returns a list of (index_i, index_j) int64 CPU
        tensors with len == min(S, Sgt_b), rows ascending (scipy convention)."""
        costs = [torch.cdist(outputs[b], t.to(outputs.device), p=2,
                             compute_mode='use_mm_for_euclid_dist_if_necessary') for b, t in enumerate(targets)]
        costs = [c.cpu() for c in costs]  # one sync for the whole batch: copies are queued back to back
        indices = [linear_sum_assignment(c) for c in costs]
        return [(torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)) for i, j in indices]
