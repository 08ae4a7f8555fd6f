"""The fp32 contractions of the grouped MLP on the bf16 matrix cores (sa_mlp.hip: split3 / MP_SA_SPLIT, the default): every
operand is staged as three bf16 planes h + m + l (24 significant bits together) and a product is the six plane products of
order <= 2^-16, accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  What is dropped is of the size of one fp32 rounding, so the
claim is: the split kernels are as close to the exact (fp64) result as the fp32-MFMA kernels they replace, and the two agree
with each other within the contract's 1e-5.

The library reads MP_SA_SPLIT once per process, so both settings run in child processes (tools/split_check.py --json): the three
set-abstraction levels of BASELINE configs[1] at full size, forward + backward, against an fp64 torch evaluation on the device."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(split):
    env = dict(os.environ, MP_SA_SPLIT=str(split))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "split_check.py"), "--json"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("[")][-1])


@pytest.fixture(scope="module")
def both():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return _run(0), _run(1)


def test_split_planes_are_as_accurate_as_the_fp32_mfma(both):
    f32, split = both
    assert [r["c0"] for r in split] == [3, 131, 259]
    for a, b in zip(f32, split):
        # forward: max error over max |y| against fp64 -- both at fp32 rounding level, the split one no worse than twice the other
        assert a["fwd_err"] < 2e-6 and b["fwd_err"] < 2e-6, (a["c0"], a["fwd_err"], b["fwd_err"])
        assert b["fwd_err"] < 2.0 * a["fwd_err"] + 2e-7, (a["c0"], a["fwd_err"], b["fwd_err"])
        # parameter gradients against fp64: dominated by the max-pool routing (a 1e-7 forward difference re-routes the gradient
        # of groups whose two largest members are that close), the same for both kernels -- never worse than 3x + fp32 noise
        assert b["grad_err"] < 3.0 * a["grad_err"] + 2e-5, (a["c0"], a["grad_err"], b["grad_err"])
        if a["xgrad_err"] is not None:
            assert b["xgrad_err"] < 3.0 * a["xgrad_err"] + 2e-5, (a["c0"], a["xgrad_err"], b["xgrad_err"])


def test_split_and_fp32_kernels_agree_within_the_contract(both):
    f32, split = both
    for a, b in zip(f32, split):
        ya, yb = np.array(a["out"]), np.array(b["out"])
        assert np.allclose(ya, yb, rtol=1e-5, atol=1e-5 * float(np.abs(ya).max())), (a["c0"], float(np.abs(ya - yb).max()))
