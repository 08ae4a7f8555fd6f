"""The one-launch head blocks (csrc/head_linear.hip; models/pointnet2_cls_ssg.py:309-327 `dropout(relu(bn1(fc1(x))))`) against torch's
own nn.Linear + nn.BatchNorm1d + ReLU in float64 (outputs, every gradient, running statistics), and their dropout mask against the
rows kernel's (same counter-based hash)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from maskplanner_amd import ops as O
    return O


def _modules(I, O, seed):
    torch.manual_seed(seed)
    lin = torch.nn.Linear(I, O).cuda()
    bn = torch.nn.BatchNorm1d(O).cuda()
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 1.5, O))
        bn.bias.copy_(torch.linspace(-0.3, 0.3, O))
        bn.running_mean.copy_(torch.linspace(-0.2, 0.2, O))
        bn.running_var.copy_(torch.linspace(0.5, 1.5, O))
    return lin, bn


def _rel(a, b):
    b = b.detach().double()
    a = a.detach()
    return float((a.double() - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("store", [True, False])
@pytest.mark.parametrize("B,I,O", [(32, 1024, 1024), (32, 1024, 512), (8, 128, 128), (5, 256, 200), (32, 512, 1000), (2, 2048, 48), (4, 128, 50),
                                   (64, 1024, 1024), (64, 1024, 512), (48, 256, 200), (33, 128, 50), (64, 2048, 48)])   # [r5] B > 32: four row tiles
def test_head_block_matches_linear_batchnorm_relu(ops, train, store, B, I, O):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from maskplanner_amd import factor_heads as fh
    lin, bn = _modules(I, O, B * I + O)
    lin64, bn64 = torch.nn.Linear(I, O).cuda().double(), torch.nn.BatchNorm1d(O).cuda().double()
    lin64.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
    bn64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
    bn.train(train)
    bn64.train(train)
    x = torch.randn(B, I, device="cuda") * 0.7 + 0.1
    xa, xb = x.clone().requires_grad_(True), x.double().requires_grad_(True)
    assert fh.head_block_ok(xa, lin, bn)
    st = {fh.BIAS_QUEUE: []} if store else None
    ya = fh.head_block(xa, lin, bn, st, "w")
    yb = F.relu(bn64(lin64(xb)))
    assert _rel(ya, yb) <= (2e-6 if B >= 8 else 2e-5)      # (two to five rows: the normalisation divides by a difference of neighbours)
    # the ReLU decisions of the two precisions agree except at pre-activations within rounding of zero
    flips = ((ya.detach() > 0) != (yb.detach() > 0))
    assert int(flips.sum()) <= max(2, ya.numel() // 20000)
    g = torch.randn(B, O, device="cuda")
    g[flips] = 0.0
    ya.backward(g)
    yb.backward(g.double())
    tol = 2e-5 if train else 5e-6       # (training: the BatchNorm backward subtracts two nearly equal sums over <= 32 rows)
    assert _rel(xa.grad, xb.grad) <= tol
    assert _rel(bn.weight.grad, bn64.weight.grad) <= tol
    assert _rel(bn.bias.grad, bn64.bias.grad) <= tol
    if store:
        fx, fg = st["w"]
        assert _rel(fg.t() @ fx, lin64.weight.grad) <= tol
        fh.flush_bias_grads(st)
    else:
        assert _rel(lin.weight.grad, lin64.weight.grad) <= tol
    # (training: the column sums of dz vanish identically -- compare on the scale of dz, not of their rounding residue)
    scale = float(lin64.weight.grad.abs().max()) if train else float(lin64.bias.grad.abs().max())
    assert float((lin.bias.grad.double() - lin64.bias.grad).abs().max()) <= 1e-4 * max(scale, 1e-6)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), bn64.running_mean.float().cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), bn64.running_var.float().cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_head_block_dropout_is_the_rows_kernels_mask(ops):
    """dropout = (p, rng, layer) inside the block: the kept elements are the ones ops.bn_relu_rows keeps for the same (seed, step, layer),
    scaled by 1 / (1 - p); the backward passes gradient through exactly those."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from maskplanner_amd import factor_heads as fh
    B, I, O, p = 32, 1024, 1024, 0.3
    lin, bn = _modules(I, O, 7)
    rng = torch.tensor([1234567, 5], dtype=torch.int64, device="cuda")
    x = torch.randn(B, I, device="cuda")
    bn_rows = torch.nn.BatchNorm1d(O).cuda()
    bn_rows.load_state_dict(bn.state_dict())
    base = fh.head_block(x, lin, bn, None, "w").detach()
    bn.load_state_dict(bn_rows.state_dict())
    xa = x.clone().requires_grad_(True)
    y = fh.head_block(xa, lin, bn, None, "w", dropout=(p, rng, 2))
    rows = ops.bn_relu_rows(lin(x).detach(), bn_rows, dropout=(p, rng, 2))
    kept = y != 0
    sure = base > 1e-4                      # (away from the ReLU edge, where the two Linears may round differently)
    assert torch.equal(kept[sure], (rows != 0)[sure])
    np.testing.assert_allclose(y[kept].detach().cpu().numpy(), (base[kept] / (1 - p)).cpu().numpy(), rtol=1e-6)
    assert 0.25 <= 1.0 - float(kept[sure].float().mean()) <= 0.35
    y.backward(torch.ones_like(y))
    # gradient reaches x only through kept, active units: with every unit dropped it would be zero -- compare against autograd of
    # the same mask applied outside
    xb = x.clone().requires_grad_(True)
    bn.load_state_dict(bn_rows.state_dict())
    lin.zero_grad()
    yb = fh.head_block(xb, lin, bn, None, "w") * kept.float() / (1 - p)
    yb.backward(torch.ones_like(yb))
    assert _rel(xa.grad, xb.grad) <= 1e-5


def test_plain_head_linear_forward(ops):
    """bn == 0: the wide heads' nn.Linear through the same kernel."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from maskplanner_amd import _lib
    lib = _lib.load()
    for B, I, O in [(32, 1024, 11988), (32, 1024, 5994), (3, 256, 77)]:
        torch.manual_seed(O)
        lin = torch.nn.Linear(I, O).cuda()
        x = torch.randn(B, I, device="cuda")
        y = torch.empty(B, O, device="cuda")
        rc = lib.mp_head_block_fwd_f32(x.data_ptr(), lin.weight.data_ptr(), lin.bias.data_ptr(), B, I, O, 0, 0, 0.0, 0.0, None, None, None, None,
                                       None, y.data_ptr(), None, None, 0.0, None, 0, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        ref = F.linear(x.double(), lin.weight.double(), lin.bias.double())
        assert _rel(y, ref) <= 1e-6


def test_two_linears_on_one_input(ops):
    """factor_linear2 (fc3 / fc_normals): outputs, grad_x = g1 W1 + g2 W2 and the stored factors against float64."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from maskplanner_amd import factor_heads as fh
    for B, I, O1, O2 in [(32, 1024, 11988, 11988), (7, 256, 600, 50), (32, 128, 5994, 333), (32, 512, 40, 4000), (64, 1024, 11988, 11988), (37, 256, 600, 50)]:
        torch.manual_seed(O1 + O2)
        l1, l2 = torch.nn.Linear(I, O1).cuda(), torch.nn.Linear(I, O2).cuda()
        x = torch.randn(B, I, device="cuda")
        xa, xb = x.clone().requires_grad_(True), x.double().requires_grad_(True)
        st = {fh.BIAS_QUEUE: []}
        y1, y2 = fh.factor_linear2(xa, l1, l2, st, "a", "b")
        r1 = F.linear(xb, l1.weight.detach().double(), l1.bias.detach().double())
        r2 = F.linear(xb, l2.weight.detach().double(), l2.bias.detach().double())
        assert _rel(y1.detach(), r1.detach()) <= 1e-6 and _rel(y2.detach(), r2.detach()) <= 1e-6
        g1, g2 = torch.randn(B, O1, device="cuda"), torch.randn(B, O2, device="cuda")
        torch.autograd.backward([y1, y2], [g1, g2])
        torch.autograd.backward([r1, r2], [g1.double(), g2.double()])
        assert _rel(xa.grad, xb.grad) <= 2e-6
        assert torch.equal(st["a"][1], g1) and torch.equal(st["b"][1], g2) and st["a"][0].data_ptr() == st["b"][0].data_ptr()
        fh.flush_bias_grads(st)
        assert _rel(l1.bias.grad, g1.double().sum(0)) <= 1e-6 and _rel(l2.bias.grad, g2.double().sum(0)) <= 1e-6
        # one of the two outputs unused: its gradient arrives as None
        xa2 = x.clone().requires_grad_(True)
        st2 = {fh.BIAS_QUEUE: []}
        y1, _ = fh.factor_linear2(xa2, l1, l2, st2, "a", "b")
        y1.backward(g1)
        assert _rel(xa2.grad, g1.double() @ l1.weight.detach().double()) <= 2e-6


def test_two_linears_second_one_dense(ops):
    """factor_linear2 with key None for the second layer (sm_fc3 + the 6-wide mask-confidence layer): its weight gradient is dense."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from maskplanner_amd import factor_heads as fh
    for B in (32, 64):
        _second_dense(fh, B)


def _second_dense(fh, B):
    I, O1, O2 = 1024, 5994, 6
    torch.manual_seed(3)
    l1, l2 = torch.nn.Linear(I, O1).cuda(), torch.nn.Linear(I, O2).cuda()
    x = torch.randn(B, I, device="cuda")
    xa, xb = x.clone().requires_grad_(True), x.double().requires_grad_(True)
    st = {fh.BIAS_QUEUE: []}
    y1, y2 = fh.factor_linear2(xa, l1, l2, st, "a", None)
    r1 = F.linear(xb, l1.weight.detach().double(), l1.bias.detach().double())
    w2 = l2.weight.detach().double().requires_grad_(True)
    r2 = F.linear(xb, w2, l2.bias.detach().double())
    assert _rel(y1, r1) <= 1e-6 and _rel(y2, r2) <= 1e-6
    g1, g2 = torch.randn(B, O1, device="cuda"), torch.randn(B, O2, device="cuda")
    torch.autograd.backward([y1, y2], [g1, g2])
    torch.autograd.backward([r1, r2], [g1.double(), g2.double()])
    assert _rel(xa.grad, xb.grad) <= 2e-6
    assert "a" in st and l1.weight.grad is None
    assert _rel(l2.weight.grad, w2.grad) <= 1e-6
    fh.flush_bias_grads(st)
    assert _rel(l2.bias.grad, g2.double().sum(0)) <= 1e-6


@pytest.mark.parametrize("shared", [True, False])
@pytest.mark.parametrize("B,I,O", [(32, 1024, 1024), (6, 128, 512), (64, 1024, 1024), (40, 128, 512)])
def test_two_blocks_in_one_launch(ops, shared, B, I, O):
    """factor_heads.head_blocks2 (fc1 / sm_fc1 on one input, fc2 / sm_fc2 on two) against the two single-block calls: identical outputs
    and statistics (the same kernel code per column tile), gradients to fp32 summation order, ONE grad_x for a shared input."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from maskplanner_amd import factor_heads as fh
    la, bna = _modules(I, O, 11)
    lb, bnb = _modules(I, O, 12)
    la2, bna2 = _modules(I, O, 11)
    lb2, bnb2 = _modules(I, O, 12)
    rng = torch.tensor([99, 3], dtype=torch.int64, device="cuda")
    xa = torch.randn(B, I, device="cuda")
    xb = None if shared else torch.randn(B, I, device="cuda")
    xa1, xa2 = xa.clone().requires_grad_(True), xa.clone().requires_grad_(True)
    xb1 = None if shared else xb.clone().requires_grad_(True)
    xb2 = None if shared else xb.clone().requires_grad_(True)
    assert fh.head_blocks2_ok(xa1, xb1, la, bna, lb, bnb)
    st1, st2 = {fh.BIAS_QUEUE: []}, {fh.BIAS_QUEUE: []}
    ya, yb = fh.head_blocks2(xa1, xb1, la, bna, lb, bnb, st1, "a", "b", (0.3, rng), (0, 2))
    ra = fh.head_block(xa2, la2, bna2, st2, "a", dropout=(0.3, rng, 0))
    rb = fh.head_block(xa2 if shared else xb2, lb2, bnb2, st2, "b", dropout=(0.3, rng, 2))
    assert torch.equal(ya, ra) and torch.equal(yb, rb)
    assert torch.equal(bna.running_var, bna2.running_var) and torch.equal(bnb.running_mean, bnb2.running_mean)
    ga, gb = torch.randn(B, O, device="cuda"), torch.randn(B, O, device="cuda")
    torch.autograd.backward([ya, yb], [ga, gb])
    torch.autograd.backward([ra, rb], [ga, gb])
    assert _rel(xa1.grad, xa2.grad) <= 2e-6
    if not shared:
        assert _rel(xb1.grad, xb2.grad) <= 2e-6
    for k in ("a", "b"):
        assert torch.equal(st1[k][1], st2[k][1])            # dz: the same arithmetic per column
    assert torch.equal(bna.weight.grad, bna2.weight.grad) and torch.equal(bnb.bias.grad, bnb2.bias.grad)


@pytest.mark.parametrize("seg_conf", [False, True])
def test_dropin_heads_draw_dropout_masks_in_the_reference_order(ops, seg_conf):
    """[r5, ADVICE r4] A training model WITHOUT `fused_dropout` (the drop-in model inside the reference's unchanged loop) applies torch's own
    nn.Dropout: with the same torch seed its masks are the ones the reference's statements draw -- fc1, fc2, [seg_conf x 2], sm_fc1, sm_fc2
    (models/pointnet2_cls_ssg.py:309-324) -- not the paired launches' order.  Checked against the same statements in plain torch."""
    from maskplanner_amd.pointnet2_cls_ssg import PointNet2Regressor_StrokeMasks
    torch.manual_seed(4)
    m = PointNet2Regressor_StrokeMasks(outdim=12, outdim_orient=12, weight_orient=0.25, out_vectors=40, hidden_size=(256, 256), pred_stroke_masks=True,
                                       n_stroke_masks=6, mask_confidence_scores=True, segment_confidence_scores=seg_conf).cuda().train()
    B = 8
    feat = torch.randn(B, 1024, device="cuda")
    torch.manual_seed(77)
    out, sm_out, mask_conf, sc = m.heads(feat)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    torch.manual_seed(77)
    drop = lambda t: F.dropout(t, 0.3, True)
    bn = lambda t, n: F.batch_norm(t, None, None, sd[n + ".weight"], sd[n + ".bias"], True, 0.1, 1e-5)
    lin = lambda t, n: F.linear(t, sd[n + ".weight"], sd[n + ".bias"])
    x = drop(F.relu(bn(lin(feat, "fc1"), "bn1")))
    final = drop(F.relu(bn(lin(x, "fc2"), "bn2")))
    x3 = lin(final, "fc3")
    if seg_conf:
        s = drop(F.relu(lin(feat, "seg_conf_fc1")))
        s = drop(F.relu(lin(s, "seg_conf_fc2")))
        want_sc = torch.sigmoid(lin(s, "seg_conf_out"))
    s1 = drop(F.relu(bn(lin(feat, "sm_fc1"), "sm_bn1")))
    s2 = drop(F.relu(bn(lin(s1, "sm_fc2"), "sm_bn2")))
    want_sm = lin(s2, "sm_fc3").view(B, 6, -1)
    want_conf = lin(s2, "mask_conf_out")
    normals = F.normalize(torch.tanh(lin(final, "fc_normals")).view(B, -1, 3), dim=-1) * 0.25
    want_out = torch.cat((x3.view(B, -1, 3), normals), dim=-1).view(B, 40, -1)
    # the SAME masks: a different draw order would change a third of the activations outright
    assert _rel(sm_out, want_sm) <= 2e-5 and _rel(mask_conf, want_conf) <= 2e-5 and _rel(out, want_out) <= 2e-5
    if seg_conf:
        assert _rel(sc, want_sc) <= 2e-5
