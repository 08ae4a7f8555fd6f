"""The two contraction families of the set-abstraction MLP against each other, each in its own process (the library reads its switches
once): the default split-plane kernels (fp32 operands as three bf16 planes on the bf16 matrix cores) and the fp32-MFMA kernels
(MP_SA_SPLIT=0), on the second level's shape, an INTERIOR 128 -> 256 layer (the non-pooled form of the role-split backward) and the
group_all level (tiled GEMMs).  Same mathematics, other summation orders: outputs to fp32 rounding, gradients inside the max-pool
routing bound."""
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
        ref = b[k].double()
        if k == "y" or k.startswith("rm"):       # forward values: fp32 rounding of two summation orders
            err = float((a[k].double() - ref).abs().max())
            assert err <= 2e-5 * max(float(ref.abs().max()), 1e-6), (what, k, err, float(ref.abs().max()))
        else:                                    # gradients: the max-pool routes near-ties differently (relative L2, as the full-size tests)
            err = float((a[k].double() - ref).norm() / ref.norm().clamp_min(1e-12))
            assert err <= 3e-2, (what, k, err)


@pytest.mark.parametrize("shape", ["sa2", "mid256", "sa3"])
def test_split_plane_kernels_against_the_fp32_mfma_kernels(shape):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    _same(_run(shape, MP_SA_SPLIT=0), _run(shape), shape)
