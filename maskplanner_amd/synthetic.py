"""Synthetic collated batches with the reference's tensor contract.

The reference's dataset is not public (README.md:33-35,53), so bench/tests/harness use generators that
reproduce the *collated batch* the training loop consumes (utils/dataset/paintnet_ODv1.py:824-845):

    point_cloud [B,N,3] f32            object point cloud (unit-scale normalised)
    traj        [B,Sgt,lambda*6] f32   GT segments (lambda=4 poses x [x,y,z,nx,ny,nz]), -100 padded
    traj_as_pc  [B,Pgt,6] f32          GT poses, -100 padded
    stroke_ids  [B,Sgt] f32            stroke id per GT segment, -1 padded  (f32: paintnet_ODv1.py:746)

Segments: per stroke of L poses, (L - lambda)//(lambda - overlapping) + 1 windows of `lambda`
consecutive poses with stride lambda-overlapping (utils/pointcloud.py:294-413 behaviour).
"""
import dataclasses

import numpy as np
import torch


@dataclasses.dataclass(frozen=True)
class Category:
    """Per-category numbers from configs/maskplanner/<cat>_v2.yaml."""
    name: str
    out_vectors: int       # S = (n_pred_traj_points - lambda)//(lambda-overlapping) + 1
    max_n_strokes: int     # M
    strokes_lo: int
    strokes_hi: int
    points_lo: int         # GT poses per sample
    points_hi: int


CATEGORIES = {
    "cuboids": Category("cuboids", 999, 6, 6, 6, 1800, 2959),
    "windows": Category("windows", 449, 22, 8, 22, 600, 1262),
    "shelves": Category("shelves", 1266, 41, 15, 41, 2000, 3448),
    "containers": Category("containers", 1333, 33, 10, 33, 2000, 3534),
}

LAMBDA = 4
OVERLAP = 1
OUTDIM = 6


def point_cloud(rng, B, N, dist="cuboid"):
    """[B,N,3] f32.  'ucube': U[-1,1]^3 (sparse balls, full scans);  'cuboid': uniform on the faces
    of a box with half-extents U[.2,.6] (saturating balls, the reference's unit-scale regime)."""
    if dist == "ucube":
        return rng.uniform(-1.0, 1.0, size=(B, N, 3)).astype(np.float32)
    if dist != "cuboid":
        raise ValueError(f"unknown point distribution {dist!r}")
    out = np.empty((B, N, 3), dtype=np.float32)
    for b in range(B):
        h = rng.uniform(0.2, 0.6, size=3)
        area = np.array([h[1] * h[2], h[0] * h[2], h[0] * h[1]])
        axis = rng.choice(3, size=N, p=area / area.sum())
        p = rng.uniform(-1.0, 1.0, size=(N, 3)) * h
        sign = rng.choice([-1.0, 1.0], size=N)
        p[np.arange(N), axis] = sign * h[axis]
        out[b] = p.astype(np.float32)
    return out


def _stroke_split(rng, total, n):
    """Split `total` poses over n strokes, each >= 2*LAMBDA poses."""
    base = 2 * LAMBDA
    w = rng.dirichlet(np.ones(n) * 4.0)
    extra = np.floor(w * (total - base * n)).astype(np.int64)
    return base + extra


def ground_truth(rng, B, cat):
    """Padded GT (traj, traj_as_pc, stroke_ids) + per-sample (n_segments, n_points)."""
    c = CATEGORIES[cat] if isinstance(cat, str) else cat
    stride = LAMBDA - OVERLAP
    segs, pts, ids = [], [], []
    for _ in range(B):
        n_str = int(rng.integers(c.strokes_lo, c.strokes_hi + 1))
        total = int(rng.integers(c.points_lo, c.points_hi + 1))
        lens = _stroke_split(rng, total, n_str)
        s_list, p_list, i_list = [], [], []
        for sid, L in enumerate(lens):
            # a smooth-ish stroke: random walk on positions, unit normals scaled by weight_orient=0.25
            start = rng.uniform(-0.8, 0.8, size=3)
            steps = rng.normal(scale=0.01, size=(L, 3)).cumsum(0)
            pos = np.clip(start + steps, -1.0, 1.0)
            nrm = rng.normal(size=(L, 3))
            nrm = 0.25 * nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
            poses = np.concatenate([pos, nrm], axis=1).astype(np.float32)  # [L,6]
            nseg = (L - LAMBDA) // stride + 1
            win = np.stack([poses[i * stride:i * stride + LAMBDA].reshape(-1) for i in range(nseg)])
            s_list.append(win)
            p_list.append(poses)
            i_list.append(np.full((nseg,), float(sid), dtype=np.float32))
        segs.append(np.concatenate(s_list))
        pts.append(np.concatenate(p_list))
        ids.append(np.concatenate(i_list))
    Sgt = max(s.shape[0] for s in segs)
    Pgt = max(p.shape[0] for p in pts)
    traj = np.full((B, Sgt, LAMBDA * OUTDIM), -100.0, dtype=np.float32)
    traj_as_pc = np.full((B, Pgt, OUTDIM), -100.0, dtype=np.float32)
    stroke_ids = np.full((B, Sgt), -1.0, dtype=np.float32)
    for b in range(B):
        traj[b, :segs[b].shape[0]] = segs[b]
        traj_as_pc[b, :pts[b].shape[0]] = pts[b]
        stroke_ids[b, :ids[b].shape[0]] = ids[b]
    n_seg = np.array([s.shape[0] for s in segs], dtype=np.int64)
    n_pts = np.array([p.shape[0] for p in pts], dtype=np.int64)
    return traj, traj_as_pc, stroke_ids, n_seg, n_pts


def make_samples(seed, B, N, cat="cuboids", dist="cuboid"):
    """B dataset items as the reference's dataset yields them before collation (utils/dataset/paintnet_ODv1.py:470-484): ragged
    numpy arrays per sample -- point_cloud [N,3], traj [n_seg, lambda*6], traj_as_pc [n_pts, 6], stroke_ids [n_seg],
    stroke_ids_as_pc [n_pts] -- for the collate / streaming paths."""
    b = make_batch(seed, B, N, cat, dist)
    out = []
    for i in range(B):
        ns, npts = int(b["n_segments"][i]), int(b["n_points"][i])
        ids = b["stroke_ids"][i, :ns].numpy()
        # pose-level stroke ids: the segments of stroke s cover its (len - LAMBDA) // stride * stride + LAMBDA first poses
        out.append({"point_cloud": b["point_cloud"][i].numpy(), "traj": b["traj"][i, :ns].numpy(),
                    "traj_as_pc": b["traj_as_pc"][i, :npts].numpy(), "stroke_ids": ids,
                    "stroke_ids_as_pc": np.zeros((npts,), dtype=np.float32), "dirname": f"synthetic_{seed}_{i}",
                    "n_strokes": int(ids.max()) + 1 if ns else 0})
    return out


def make_batch(seed, B, N, cat="cuboids", dist="cuboid", device="cpu"):
    """One collated batch as torch tensors (reference dict keys)."""
    rng = np.random.default_rng(seed)
    pc = point_cloud(rng, B, N, dist)
    traj, traj_as_pc, stroke_ids, n_seg, n_pts = ground_truth(rng, B, cat)
    t = lambda a: torch.from_numpy(a).to(device)
    return {
        "point_cloud": t(pc),
        "traj": t(traj),
        "traj_as_pc": t(traj_as_pc),
        "stroke_ids": t(stroke_ids),
        "n_segments": torch.from_numpy(n_seg),
        "n_points": torch.from_numpy(n_pts),
        "fps_start": [torch.from_numpy(rng.integers(0, N, size=B)).to(device),
                      torch.from_numpy(rng.integers(0, 512, size=B)).to(device)],
    }
