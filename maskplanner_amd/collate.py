"""Drop-in for `Paintnet_ODv1_CollateBatch` (utils/dataset/paintnet_ODv1.py:713-847), MaskPlanner keys, collating ON the device.

The reference pads every sample's ragged GT on the host -- per sample and key one numpy concatenate with -100 / -1 fake rows
(add_fake_vectors_v2 :887-904, add_fake_values_v2 :907-925), one torch.as_tensor, then torch.stack -- and the training loop
copies the stacked tensors to the GPU afterwards (train_maskplanner.py:207-215).  Here each key travels as ONE flat buffer plus
an offsets vector (two host-to-device copies per key) and csrc/collate.hip lays the padded [B, R, D] tensor out on the GPU, so
the batch the step consumes is born resident.  Returned dict: same keys and dtypes as the reference (everything float32,
`stroke_ids` included); the keys of the autoregressive / prototype variants, which MaskPlanner does not load, are None as they
are in the reference when their extras are off, and asking for them raises NotImplementedError.
"""
import numpy as np
import torch

from . import _lib, ops

_SUPPORTED_EXTRAS = ("stroke_masks",)
_UNSUPPORTED_EXTRAS = ("stroke_prototypes", "segments_per_stroke", "history_of_segments_per_stroke_v1",
                       "history_of_segments_per_stroke_v2")


def pad_ragged(arrays, fill, device="cuda", total_needed=None):
    """arrays: B numpy arrays [n_b, D] (or [n_b]) -> float32 device tensor [B, R, D] (or [B, R]), R = total_needed or max n_b,
    rows >= n_b filled with `fill` (add_fake_vectors_v2 / add_fake_values_v2 of the reference, batched)."""
    one_d = np.asarray(arrays[0]).ndim == 1
    arrays = [np.asarray(a, dtype=np.float32) for a in arrays]
    mats = [a.reshape(a.shape[0], int(np.prod(a.shape[1:]))) for a in arrays]   # (an empty sample keeps its row width)
    D = mats[0].shape[1]
    if any(m.shape[1] != D for m in mats):
        raise ValueError("some vectors have different dimensionality than others.")
    lens = [m.shape[0] for m in mats]
    R = int(total_needed) if total_needed is not None else int(max(lens))
    if R < max(lens):
        raise ValueError("total_needed is smaller than the longest sequence")   # (the reference would fail in torch.stack)
    B = len(mats)
    dev = torch.device(device)
    flat = torch.from_numpy(np.concatenate(mats, axis=0)).to(dev, non_blocking=True)
    offsets = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).to(dev, non_blocking=True)
    out = torch.empty((B, R, D), dtype=torch.float32, device=dev)
    ops._run("pad_ragged", out, _lib.load().mp_pad_ragged_f32, flat.data_ptr() if flat.numel() else None, offsets.data_ptr(), B, R, D,
             float(fill), out.data_ptr())
    return out[..., 0] if one_d else out


def lambda_segments(poses, stroke_ids, lmbda, overlapping=0, device="cuda"):
    """get_sequences_of_lambda_points (utils/pointcloud.py:294-413, padding=True) for a whole batch, on the device.

    poses: B numpy arrays [n_b, D] (one pose per row, strokes back to back); stroke_ids: B arrays [n_b] (ascending 0, 0, .., 1, ..).
    -> (traj f32 [B, R, lmbda*D] padded with -100, ids f32 [B, R] padded with -1, status i32 [B]: 0, or non-zero for a sample
    the kernel refuses -- more than 1024 strokes or ids that are not ascending from 0 -- whose rows are then all padding; callers
    that can afford a host sync check it, see Paintnet_ODv1_CollateBatch) with R = the largest per-sample row count the
    reference pads to ((n - lmbda) // (lmbda - overlapping) + 1, resp. n // lmbda), i.e. the tensors its dataset + collate
    function produce together (utils/dataset/paintnet_ODv1.py:294, 738-748).  One flat host-to-device copy per key."""
    poses = [np.asarray(p, dtype=np.float32) for p in poses]
    ids = [np.asarray(i, dtype=np.float32).reshape(-1) for i in stroke_ids]
    if not poses or any(p.ndim != 2 for p in poses) or any(p.shape[1] != poses[0].shape[1] for p in poses) \
            or [i.shape[0] for i in ids] != [p.shape[0] for p in poses]:
        raise ValueError("poses must be [n_b, D] arrays with one stroke id per pose")
    D = poses[0].shape[1]
    lens = [p.shape[0] for p in poses]
    rows = [((n - lmbda) // (lmbda - overlapping) + 1 if overlapping else n // lmbda) if n >= lmbda else 0 for n in lens]
    B, R = len(poses), max(rows + [0])
    dev = torch.device(device)
    flat = torch.from_numpy(np.concatenate(poses, axis=0)).to(dev, non_blocking=True)
    fid = torch.from_numpy(np.concatenate(ids, axis=0)).to(dev, non_blocking=True)
    offsets = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).to(dev, non_blocking=True)
    traj = torch.empty((B, R, lmbda * D), dtype=torch.float32, device=dev)
    out_ids = torch.empty((B, R), dtype=torch.float32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    ops._run("lambda_segments", traj, _lib.load().mp_lambda_segments_f32, flat.data_ptr() if flat.numel() else None,
             fid.data_ptr() if fid.numel() else None, offsets.data_ptr(), B, D, int(lmbda), int(overlapping), R, traj.data_ptr(),
             out_ids.data_ptr(), status.data_ptr())
    return traj, out_ids, status


class Paintnet_ODv1_CollateBatch:
    """collate_fn for torch.utils.data.DataLoader: a list of dataset items -> the batch dict of the reference, on `device`."""

    def __init__(self, config, device="cuda"):
        self.config = config
        self.load_extra_data = config["load_extra_data"]
        self.device = torch.device(device)
        bad = [e for e in self.load_extra_data if e in _UNSUPPORTED_EXTRAS]
        if bad:
            raise NotImplementedError(f"extra data {bad} belongs to the autoregressive / prototype variants, outside the MaskPlanner hot path")

    def __call__(self, data):
        dev = self.device
        point_cloud = torch.from_numpy(np.stack([np.asarray(d["point_cloud"], dtype=np.float32) for d in data])).to(dev, non_blocking=True)
        # with traj_with_equally_spaced_points the samples are ragged (:738-748); otherwise all lengths agree and the same
        # kernel degenerates to a stack (:750-754)
        traj = pad_ragged([d["traj"] for d in data], -100.0, dev)
        traj_as_pc = pad_ragged([d["traj_as_pc"] for d in data], -100.0, dev)
        stroke_ids = pad_ragged([d["stroke_ids"] for d in data], -1.0, dev)
        stroke_ids_as_pc = pad_ragged([d["stroke_ids_as_pc"] for d in data], -1.0, dev)
        stroke_masks = None
        if "stroke_masks" in self.load_extra_data:
            stroke_masks = [torch.as_tensor(d["stroke_masks"], dtype=torch.int64).to(dev, non_blocking=True) for d in data]
        return {
            "point_cloud": point_cloud, "traj": traj, "traj_as_pc": traj_as_pc,
            "stacked_segments_per_stroke": None, "stacked_points_per_stroke": None, "unstacked_segments_per_stroke": None,
            "stacked_segments_per_substroke": None, "stacked_segments_per_init_substroke": None,
            "strokewise_history_batch": None, "strokewise_target_batch": None, "strokewise_stroke_ids_batch": None,
            "strokewise_sample_ids_batch": None, "strokewise_end_of_path_batch": None, "max_num_segments": None,
            "stroke_ids": stroke_ids, "stroke_ids_as_pc": stroke_ids_as_pc, "stroke_masks": stroke_masks,
            "stroke_prototypes": None, "dirname": [d["dirname"] for d in data], "n_strokes": [d["n_strokes"] for d in data],
        }
