"""Make the UNCHANGED reference entry points (train_maskplanner.py / test_maskplanner.py) run on the MI355X path.

    import maskplanner_amd.dropin as dropin; dropin.install()      # before the reference's own imports

registers this package's modules under the names the reference imports:
    models.pointnet2_utils   <- maskplanner_amd.pointnet2_utils     (models/pointnet2_cls_ssg.py:9, pointnet2_seg.py:12)
    pytorch3d.ops.knn        <- maskplanner_amd.knn                 (pytorch3d_chamfer.py:12)
    pytorch3d_chamfer        <- maskplanner_amd.pytorch3d_chamfer   (loss_handler.py:19, metrics_handler.py:8)
    models.hungarianMatcher  <- maskplanner_amd.hungarianMatcher    (loss_handler.py:173)
    models.pointnet2_cls_ssg <- maskplanner_amd.pointnet2_cls_ssg   (models/__init__.py:20: all six regressor classes)
    models.pointnet2_seg     <- maskplanner_amd.pointnet2_seg       (models/__init__.py:21)
    loss_handler             <- maskplanner_amd.loss_handler        (train_maskplanner.py:65, test_maskplanner.py:32)
    metrics_handler          <- maskplanner_amd.metrics_handler     (train_maskplanner.py:66, test_maskplanner.py:33)
`models` itself stays the reference's package: only the listed submodules are replaced, so get_model(), the
config system and the training loop are untouched.  With every alias installed a training step of the unchanged
train_maskplanner.py performs no device->host synchronisation besides the ones the loop itself asks for
(`loss_handler.compute` returning the numpy list, `loss.item()`).

tests/test_dropin_reference.py imports the reference's `models`, `loss_handler` and `train_maskplanner` under these aliases
in the build container and builds the model and loss from the merged maskplanner config.
"""
import importlib
import sys
import types

_ALIASES = {
    "models.pointnet2_utils": "maskplanner_amd.pointnet2_utils",
    "pytorch3d.ops.knn": "maskplanner_amd.knn",
    "pytorch3d_chamfer": "maskplanner_amd.pytorch3d_chamfer",
    "models.hungarianMatcher": "maskplanner_amd.hungarianMatcher",
    "models.pointnet2_cls_ssg": "maskplanner_amd.pointnet2_cls_ssg",
    "models.pointnet2_seg": "maskplanner_amd.pointnet2_seg",
    "loss_handler": "maskplanner_amd.loss_handler",
    "metrics_handler": "maskplanner_amd.metrics_handler",
}
MINIMAL = ("models.pointnet2_utils", "pytorch3d.ops.knn")   # kernels only: the reference's own wrappers, models and losses on top


def install(names=None):
    """Install the aliases (all, or the given subset of reference module names).  Returns the list installed."""
    done = []
    for ref_name, ours in _ALIASES.items():
        if names is not None and ref_name not in names:
            continue
        mod = importlib.import_module(ours)
        if ref_name.startswith("pytorch3d"):
            # pytorch3d is absent on ROCm: create the package skeleton the import statement walks through
            for pkg in ("pytorch3d", "pytorch3d.ops", "pytorch3d.structures", "pytorch3d.structures.pointclouds"):
                if pkg not in sys.modules:
                    m = types.ModuleType(pkg)
                    m.__path__ = []
                    sys.modules[pkg] = m
            if not hasattr(sys.modules["pytorch3d.structures.pointclouds"], "Pointclouds"):
                sys.modules["pytorch3d.structures.pointclouds"].Pointclouds = type("Pointclouds", (), {})
            sys.modules["pytorch3d.ops"].knn = mod
        sys.modules[ref_name] = mod
        done.append(ref_name)
    return done
