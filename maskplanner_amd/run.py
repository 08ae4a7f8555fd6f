"""Zero-edit launcher for the reference's UNCHANGED entry points on the MI355X path:

    python -m maskplanner_amd.run train_maskplanner.py config=[maskplanner,cuboids_v2,longx_v2] seed=42
    python -m maskplanner_amd.run test_maskplanner.py --run <run dir>

is `python train_maskplanner.py ...` (README.md:115, train_maskplanner.py:5-38) with `dropin.install()` executed first: the script's
own imports (`models.pointnet2_utils`, `pytorch3d.ops.knn`, `pytorch3d_chamfer`, `loss_handler`, `metrics_handler`, `models.hungarianMatcher`,
the model classes -- dropin._ALIASES) then resolve to this package, and nothing in the reference checkout is edited.  The script runs
as `__main__` with its own directory first on `sys.path` and `sys.argv` = [script, its arguments], exactly as the interpreter would
start it (runpy.run_path); the working directory is left alone (the reference reads `configs/maskplanner` relative to it:
train_maskplanner.py:69).

Options (before the script name):
    --minimal        alias the kernel modules only (models.pointnet2_utils, pytorch3d.ops.knn): the reference's own model classes, chamfer
                     wrapper and losses on top of the HIP kernels (dropin.MINIMAL)
    --hip-required   (default) fail at start-up if libmaskplanner_hip.so cannot be loaded; --no-hip-check skips the check (import-only dry
                     runs in a container without the library)
    --no-graphs      `model(...)` and `loss_handler.compute(...)` launched op by op (default: after three calls per shape they replay graphs
                     recorded from their own eager code, maskplanner_amd/graphed.py; the same as MASKPLANNER_DROPIN_GRAPH=0)
    --fused-adam     opt-in: `torch.optim.Adam(...)` calls of the script default to `fused=True` (train_maskplanner.py:159 passes no such
                     argument; same update rule, torch's single-kernel implementation: the unchanged loop is bound by host time, and the
                     foreach Adam over 143 MB of dense gradients is 0.85 ms of it -- 4.4 -> 4.0 ms per step)
"""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    minimal, check, fused_adam = False, True, False
    while argv and argv[0].startswith("--"):
        opt = argv.pop(0)
        if opt == "--minimal":
            minimal = True
        elif opt == "--no-hip-check":
            check = False
        elif opt == "--hip-required":
            check = True
        elif opt == "--fused-adam":
            fused_adam = True
        elif opt == "--no-graphs":
            os.environ["MASKPLANNER_DROPIN_GRAPH"] = "0"      # (read when maskplanner_amd.graphed is imported, below)
        else:
            raise SystemExit(f"maskplanner_amd.run: unknown option {opt} (options go before the script name)")
    if not argv:
        raise SystemExit(__doc__)
    script = argv[0]
    if not os.path.isfile(script):
        raise SystemExit(f"maskplanner_amd.run: no such script: {script}")
    from . import dropin
    if check:
        from . import _lib
        _lib.load()          # the product path has no CPU fallback: fail here, not at the first kernel call
    dropin.install(dropin.MINIMAL if minimal else None)
    if fused_adam:
        import torch
        plain = torch.optim.Adam.__init__

        def init(self, params, *a, **kw):
            params = list(params)
            first = params[0]["params"][0] if (params and isinstance(params[0], dict)) else (params[0] if params else None)
            if "fused" not in kw and "foreach" not in kw and first is not None and getattr(first, "is_cuda", False):
                kw["fused"] = True
            plain(self, params, *a, **kw)
        torch.optim.Adam.__init__ = init
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")      # (puts the script's directory first on sys.path, like `python script.py`)


if __name__ == "__main__":
    main()
