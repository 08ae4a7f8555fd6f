"""Drop-in for the chamfer-based evaluation metrics of `metrics_handler.py` (MetricsHandler, :24-133; get_pcd :226-262,
get_chamfer_original :265-282, get_stroke_chamfer :445-496): the three metrics whose arithmetic is `chamfer_distance`, i.e. the
kNN hot path, run on the HIP kernels through `maskplanner_amd.pytorch3d_chamfer`.  The remaining reference metrics
(clustering scores via scikit-learn, start-of-path and stroke-count statistics) are host-side numpy post-processing of a few
hundred integers per sample; they are outside SURVEY 8 and raise NotImplementedError here -- except `stroke_masks_metrics`
(:285-308), the second default evaluation metric of the maskplanner configs (configs/maskplanner/default.yaml:14-16), which
is a confidence filter + arg-max over the predicted masks and is evaluated on the device.

Same constructor, metric names, output names and compute() / get_eval_metric() / pprint() behaviour as the reference class.
"""
import numpy as np
import torch

from .pytorch3d_chamfer import chamfer_distance


def get_dim_traj_points(extra_data):
    """Dimensionality of one output pose (utils/pointcloud.py:478-491)."""
    if len(extra_data) == 0:
        return 3
    if len(extra_data) == 1 and ("vel" in extra_data or "orientrotvec" in extra_data or "orientnorm" in extra_data):
        return 6
    if "orientquat" in extra_data and len(extra_data) == 1:
        return 7
    raise ValueError("Other combinations of extra_data are not supported yet.")


def to_numpy(tensor):
    return tensor.detach().cpu().numpy() if torch.is_tensor(tensor) else tensor


class MetricsHandler:
    # host-side numpy / scikit-learn statistics outside SURVEY 8 (clustering scores, start-of-path counts)
    HOST_ONLY = ("clustering_metrics", "sop_metrics", "sop_metrics_v2", "strokewise_num_of_strokes_metrics")
    _N_STROKES = ("perc_correct_n_strokes", "avg_num_of_pred_strokes", "avg_num_of_gt_strokes", "mean_absolute_error_NoP")

    def __init__(self, config, metrics=[], renormalize_output_config={}):
        self.metrics = metrics
        # the reference's registry, in its order (:42-100)
        self.metrics_names = ["pcd", "chamfer_original", "stroke_chamfer", "clustering_metrics", "sop_metrics", "sop_metrics_v2",
                              "stroke_masks_metrics", "strokewise_num_of_strokes_metrics"]
        self.output_metrics_names = [
            ("point-wise chamfer distance",), ("chamfer original",), ("stroke chamfer distance",),
            ("v_measure", "adjusted_rand_score", "avg_num_of_outliers"),
            ("avg_num_of_pred_sops", "avg_num_of_gt_sops", "avg_ratio_pred_over_gt_sops", "avg_num_of_pred_sops_if_higher_threshold",
             "avg_num_of_pred_sops_if_lower_threshold", "avg_ratio_pred_over_gt_sops_if_higher_threshold",
             "avg_ratio_pred_over_gt_sops_if_lower_threshold"),
            self._N_STROKES + ("avg_num_of_pred_strokes_if_higher_threshold",
                                                         "avg_num_of_pred_strokes_if_lower_threshold",
                                                         "mean_absolute_error_NoP_if_higher_threshold",
                                                         "mean_absolute_error_NoP_if_lower_threshold"),
            self._N_STROKES, self._N_STROKES]
        self.metric_functions = [self.get_pcd, self.get_chamfer_original, self.get_stroke_chamfer, None, None, None,
                                 self.stroke_masks_metrics, None]
        self.metric_index = {m: i for i, m in enumerate(self.metrics_names)}
        self.config = config
        self.renormalize_output_config = renormalize_output_config
        self.renormalize_output = False
        if "active" in self.renormalize_output_config and self.renormalize_output_config["active"]:   # :112-115
            assert self.config["normalization"] == "per-dataset"
            self.renormalize_output = True

    # ---- bookkeeping (:118-196) -------------------------------------------------------------------------------
    def get_eval_metric(self, metric, **kwargs):
        assert metric in self.metrics_names, f"metric {metric} is not valid"
        if metric in self.HOST_ONLY:
            raise NotImplementedError(f"metric {metric!r} is host-side post-processing outside the MaskPlanner hot path of this build")
        return self.metric_functions[self.metric_index[metric]](**kwargs)

    def compute(self, **kwargs):
        """All metrics of self.metrics as one numpy array (0 when there are none), like the reference."""
        if len(self.metrics) == 0:
            return 0
        out = []
        for metric in self.metrics:
            out += self._as_list(self.get_eval_metric(metric=metric, **kwargs))
        return np.array(out)

    def pprint(self, metric_values, prefix=""):
        if len(self.metrics) == 0:
            return
        assert self.tot_num_of_metrics() == len(metric_values)
        print(prefix)
        k = 0
        for name in self.metrics:
            for out_name in self.output_metrics_names[self.metric_index[name]]:
                print(f"\t{out_name}: {round(float(metric_values[k]), 5)}")
                k += 1

    def _as_list(self, item):
        return [to_numpy(item)] if not isinstance(item, list) else to_numpy(item)

    def tot_num_of_metrics(self):
        return sum(self.num_of_metrics(name) for name in self.metrics)

    def num_of_metrics(self, name):
        return len(self.output_metrics_names[self.metric_index[name]])

    def renormalize_traj(self, traj):
        """Rescale the positions of real (non -100) poses from one data_scale_factor to another, in place (:199-217)."""
        if not self.renormalize_output:
            return traj
        assert traj.shape[-1] == 6, "point-wise format and orientnorm is assumed."
        real = ~torch.all(traj == -100, dim=-1, keepdim=True)
        scale = float(self.renormalize_output_config["from"])
        to = float(self.renormalize_output_config["to"])
        traj[..., :3] = torch.where(real, traj[..., :3] * scale, traj[..., :3])
        traj[..., :3] = torch.where(real, traj[..., :3] / to, traj[..., :3])
        return traj

    # ---- metrics ------------------------------------------------------------------------------------------------
    def get_pcd(self, y_pred, y, traj_as_pc=None, **kwargs):
        """Pose-wise chamfer distance (x 1e4) between predictions and the -100 padded GT poses (:226-262)."""
        B = y_pred.shape[0]
        outdim = get_dim_traj_points(self.config["extra_data"])
        if self.config["lambda_points"] > 1:
            y_pred = y_pred.reshape(B, -1, outdim)
            if traj_as_pc is None:
                raise ValueError("DEPRECATED: Going from GT traj as segments to points is not ideal. Use traj_as_pc instead.")
        pred = y_pred.clone().detach()
        dev = pred.device if pred.is_cuda else torch.device("cuda")
        traj_as_pc = traj_as_pc.to(dev, dtype=torch.float)
        pred = pred.to(dev, dtype=torch.float)
        with torch.no_grad():
            if self.renormalize_output:
                pred, traj_as_pc = self.renormalize_traj(pred), self.renormalize_traj(traj_as_pc)
            return (10 ** 4) * chamfer_distance(pred, traj_as_pc, padded=True)[0]

    def stroke_masks_metrics(self, n_strokes, pred_stroke_masks, mask_scores, confidence_threshold=0.5, **kwargs):
        """Stroke-count statistics of the predicted masks (:285-308 with utils/postprocessing.py:92-152): masks whose
        confidence sigmoid is below the threshold get probability 0, every segment goes to its arg-max mask (first mask on
        ties -- all-zero columns included), the number of distinct masks chosen is the predicted stroke count."""
        with torch.no_grad():
            keep = torch.sigmoid(mask_scores.detach().float()) >= confidence_threshold                          # [B,M]
            prob = torch.sigmoid(pred_stroke_masks.detach().float()) * keep[:, :, None]
            chosen = prob.argmax(dim=1)                                                                          # [B,S]
            used = torch.zeros(prob.shape[:2], dtype=torch.bool, device=prob.device).scatter_(1, chosen, True)
            n_pred = used.sum(1).cpu().numpy().astype(int)
        n_gt = np.array(to_numpy(n_strokes)).astype(int)
        return [np.mean((n_gt == n_pred).astype(int)), np.mean(n_pred), np.mean(n_gt), np.mean(np.abs(n_pred - n_gt))]

    def get_chamfer_original(self, y_pred, y, traj_pc, **kwargs):
        """Chamfer (x 1e4) against the full, untrimmed ground-truth cloud (:265-282)."""
        B = y_pred.shape[0]
        outdim = get_dim_traj_points(self.config["extra_data"])
        if self.config["lambda_points"] > 1:
            y_pred = y_pred.reshape(B, -1, outdim)
        pred = torch.as_tensor(y_pred).detach()
        dev = pred.device if pred.is_cuda else torch.device("cuda")
        with torch.no_grad():
            return (10 ** 4) * chamfer_distance(pred.to(dev, dtype=torch.float), torch.as_tensor(traj_pc).to(dev, dtype=torch.float))[0]

    def get_stroke_chamfer(self, y_pred, y, traj_pc, stroke_ids, **kwargs):
        """Mean over predicted vectors of the smallest asymmetric chamfer (x 1e4) to any GT stroke (:445-496).  The reference
        issues one chamfer call per (sample, predicted vector, GT stroke); here a sample's predicted vectors are one batch
        against each GT stroke."""
        B = y_pred.shape[0]
        outdim = get_dim_traj_points(self.config["extra_data"])
        pred = torch.as_tensor(y_pred).detach()
        dev = pred.device if pred.is_cuda else torch.device("cuda")
        pred = pred.to(dev, dtype=torch.float)
        traj_pc = torch.as_tensor(traj_pc).to(dev, dtype=torch.float)
        ids = to_numpy(stroke_ids)
        out = []
        with torch.no_grad():
            for b in range(B):
                n_pred = pred.shape[1]
                n_gt = int(ids[b, -1]) + 1
                assert len(np.unique(ids[b])) == n_gt
                vecs = pred[b].reshape(n_pred, -1, outdim)                        # every predicted vector as a small cloud
                best = torch.full((n_pred,), float("inf"), device=dev)
                for i_gt in range(n_gt):
                    gt = traj_pc[b, torch.as_tensor(ids[b] == i_gt, device=dev)]
                    d = chamfer_distance(vecs, gt[None].expand(n_pred, -1, -1).contiguous(), asymmetric=True,
                                         batch_reduction=None)[0]
                    best = torch.minimum(best, (10 ** 4) * d)
                out.append(float(best.sum()) / n_pred)
        return np.array(out).mean()
