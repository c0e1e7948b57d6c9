"""GPU parity of the native scalar losses (fdgan_hip/losses.py, csrc/losses.hip) against
torch.nn.functional on the CPU (fp64), value and gradient: F.l1_loss, F.mse_loss, F.binary_cross_entropy (incl. the
-100 log clamp and constant targets) and the perceptual MSE over Vgg16's four taps, which must agree with the composition
`sum(F.mse_loss(a, b) for a, b in zip(vgg(x), vgg(t)))` of the module's NCHW outputs."""
import pytest
import torch
import torch.nn.functional as F

from hiputil import seeded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def LS():
    from fdgan_hip import lib, losses
    lib.load()
    return losses


@pytest.mark.parametrize("shape", [(16, 3, 256, 256), (2, 3, 37, 41), (1, 1, 1, 3)])
def test_l1_and_mse_value_and_gradient(LS, shape):
    x, t = seeded(shape, 1, -1, 1), seeded(shape, 2, 0, 1)
    x.view(-1)[0] = t.view(-1)[0]                      # sign(0) = 0
    for fn, ref in ((LS.l1_loss, F.l1_loss), (LS.mse_loss, F.mse_loss)):
        xr = x.double().requires_grad_(True)
        lr = ref(xr, t.double())
        lr.backward()
        xd = x.to(DEV).requires_grad_(True)
        l = fn(xd, t.to(DEV))
        (3.0 * l).backward()
        assert l.shape == () and l.dtype == torch.float32
        assert abs(float(l) - float(lr)) < 2e-6 * max(1.0, abs(float(lr)))
        assert (xd.grad.cpu().double() - 3.0 * xr.grad).abs().max() < 1e-6 * float(xr.grad.abs().max()) + 1e-12
        with torch.no_grad():
            assert float(fn(xd, t.to(DEV))) == float(l)       # ordered reduction: bit-reproducible


def test_bce_matches_torch_including_the_log_clamp(LS):
    p = seeded((16, 1, 126, 126), 3, 0.0, 1.0)
    p.view(-1)[:4] = torch.tensor([0.0, 1.0, 1e-30, 1.0 - 1e-7])      # saturated sigmoid outputs
    for target in (1.0, 0.0, seeded((16, 1, 126, 126), 4, 0.0, 1.0)):
        tt = target if isinstance(target, torch.Tensor) else torch.full_like(p, target)
        pr = p.clone().requires_grad_(True)
        lr = F.binary_cross_entropy(pr, tt)
        lr.backward()
        pd = p.to(DEV).requires_grad_(True)
        l = LS.bce_loss(pd, target.to(DEV) if isinstance(target, torch.Tensor) else target)
        l.backward()
        assert abs(float(l) - float(lr)) < 1e-5 * max(1.0, abs(float(lr)))
        g, gr = pd.grad.cpu(), pr.grad
        assert ((g - gr).abs() / (gr.abs() + 1e-6)).max() < 1e-4


def test_losses_reject_cpu_tensors_and_target_gradients(LS):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        LS.l1_loss(torch.zeros(4), torch.zeros(4))
    x = torch.zeros(4, device=DEV, requires_grad=True)
    with pytest.raises(NotImplementedError):
        LS.mse_loss(torch.zeros(4, device=DEV), x)


def test_vgg_perceptual_equals_the_module_composition(LS):
    from myutils.vgg16 import Vgg16
    torch.manual_seed(0)
    vgg = Vgg16().to(DEV)
    for p in vgg.parameters():
        p.requires_grad_(False)
    x, t = seeded((2, 3, 64, 48), 5, 0, 1).to(DEV), seeded((2, 3, 64, 48), 6, 0, 1).to(DEV)
    # reference composition on the module's own NCHW fp32 outputs (the round-1 path)
    with torch.no_grad():
        ft = vgg(t)
    xa = x.clone().requires_grad_(True)
    ref = sum(F.mse_loss(a, b) for a, b in zip(vgg(xa), ft))
    ref.backward()
    xb = x.clone().requires_grad_(True)
    l = LS.vgg_perceptual(vgg, xb, t)
    (2.0 * l).backward()
    assert abs(float(l) - float(ref)) < 1e-5 * abs(float(ref))
    rel = float((xb.grad - 2.0 * xa.grad).norm() / (2.0 * xa.grad).norm())
    assert rel < 2e-2, rel            # same kernels; the seed gradient is rounded to bf16 once here, once there
    with torch.no_grad():
        assert abs(float(LS.vgg_perceptual(vgg, x, t)) - float(ref)) < 1e-5 * abs(float(ref))
    assert float(LS.vgg_perceptual(vgg, t, t)) == 0.0


def test_vgg_backward_detects_overwritten_activations(LS):
    from myutils.vgg16 import Vgg16
    vgg = Vgg16().to(DEV)
    for p in vgg.parameters():
        p.requires_grad_(False)
    x = seeded((1, 3, 32, 32), 7, 0, 1).to(DEV).requires_grad_(True)
    feats = vgg(x)
    with torch.no_grad():
        vgg(seeded((1, 3, 32, 32), 8, 0, 1).to(DEV))       # same shape: overwrites the plan's activations
    with pytest.raises(RuntimeError, match="overwritten"):
        sum(f.mean() for f in feats).backward()
    l = LS.vgg_perceptual(vgg, x, x.detach() * 0.5)
    LS.vgg_perceptual(vgg, x.detach(), x.detach())            # runs slot 0 and 1 again
    with pytest.raises(RuntimeError, match="overwritten"):
        l.backward()


def test_contextual_loss_matches_the_oracle_restatement():
    """loss.ContextualLoss (fused row kernel) vs oracle/contextual_ref.py (the bytecode's torch formulation, fp32 CPU): value,
    gradient w.r.t. the image features, the unfused method chain, and non-default sigma / b."""
    import loss as hl
    from oracle import contextual_ref as cr
    for (shape, kw) in (((2, 64, 12, 10), {}), ((1, 512, 32, 32), {}), ((3, 16, 7, 9), dict(sigma=0.25, b=0.5))):
        I, T = seeded(shape, 11, -1, 1), seeded(shape, 12, -1, 1)
        I = 0.6 * I + 0.4 * T                                     # correlated features: a non-trivial best match per row
        Ir = I.clone().requires_grad_(True)
        lr = cr.ContextualLoss(**kw)(Ir, T)
        lr.backward()
        Id = I.to(DEV).requires_grad_(True)
        mod = hl.ContextualLoss(**kw)
        l = mod(Id, T.to(DEV))
        l.backward()
        assert abs(float(l) - float(lr)) < 2e-4 * max(1.0, abs(float(lr))), (float(l), float(lr))
        rel = float((Id.grad.cpu() - Ir.grad).norm() / Ir.grad.norm())
        assert rel < 5e-3, rel
        with torch.no_grad():                                     # the reference's own method chain on the same matrix
            d = mod.cos_similarity(Id, T.to(DEV))
            cx = mod.weighted_average_distances(mod.relative_distances(d))
            m_ref = cx.permute(0, 2, 1).max(dim=1)[0]
            m = hl._CxRowsFn.apply(d, float(mod.sigma), float(mod.e))
            assert ((m - m_ref).abs() / m_ref).max() < 1e-4
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hl.ContextualLoss()(torch.rand(1, 4, 3, 3), torch.rand(1, 4, 3, 3))
