"""The switchable kernel variants against the default route, each in its own process (the library reads its switches once):
the role-split fused backward of the 256-output layer (default) against the eight-wave kernel, its loads-straight-into-LDS forms,
and the planes route of the few-row levels.  Same mathematics, other summation orders: outputs to fp32 rounding, gradients inside
the max-pool routing bound."""
import os
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(shape, **env):
    out = os.path.join(tempfile.mkdtemp(prefix="mp_var_"), "r.pt")
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_worker.py"), shape, out], env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return torch.load(out)


def _same(a, b, what):
    for k in a:
        ref = b[k]
        tol = 2e-5 if k in ("y",) or k.startswith("rm") else 2e-3        # gradients: dW atomics + max-pool routing of near-ties
        err = float((a[k] - ref).abs().max())
        assert err <= tol * max(float(ref.abs().max()), 1e-6), (what, k, err, float(ref.abs().max()))


@pytest.mark.parametrize("env", [dict(MP_BF_ROLES=0), dict(MP_BF_ROLES=3), dict(MP_BF_ROLES_LDS=2), dict(MP_BF_ROLES_LDS=3), dict(MP_LEAN_LAST=1)])
def test_fused_backward_variants_of_the_second_level(env):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    _same(_run("sa2", **env), _run("sa2"), env)


def test_interior_256_output_layer_on_both_fused_backward_kernels():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    _same(_run("mid256", MP_BF_ROLES=0), _run("mid256"), "interior 256")


def test_planes_route_of_the_group_all_level():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    _same(_run("sa3", MP_PLANES=1), _run("sa3"), "planes")
