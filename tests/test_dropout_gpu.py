"""GPU: dropout on the HIP path.  torch's CPU Philox stream cannot be reproduced on the device, so these
are (1) statistical checks of the mask, (2) exact checks of the arithmetic *given* the mask (extracted from
the kernels' own outputs) against an fp32 autograd reference, (3) state/site determinism."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from mmfusion import ops  # noqa: E402
from mmfusion.lib import EPI_BIAS, EPI_RELU, GEMM_NT  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def rel(a, b, floor=1e-3):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), floor))


def test_elementwise_dropout_statistics_and_determinism():
    p, n = 0.1, 1 << 20
    ops.seed_dropout(123)
    ops.begin_training_forward()
    x = torch.ones(n, device=DEV, requires_grad=True)
    y = ops.dropout(x, p, True)                                   # site 1
    kept = float((y != 0).float().mean())
    assert abs(kept - (1 - p)) < 4 * math.sqrt(p * (1 - p) / n)
    assert torch.allclose(y[y != 0], torch.full_like(y[y != 0], 1 / (1 - p)), rtol=1e-6)
    y.sum().backward()
    assert torch.equal(x.grad, y.detach())                        # backward regenerates the same mask
    ops._site = 0
    y2 = ops.dropout(torch.ones(n, device=DEV), p, True)          # same state, same site -> same mask
    assert torch.equal(y2, y.detach())
    y3 = ops.dropout(torch.ones(n, device=DEV), p, True)          # next site -> different mask
    assert not torch.equal(y3, y.detach())
    ops.begin_training_forward()                                  # next step -> different mask
    y4 = ops.dropout(torch.ones(n, device=DEV), p, True)
    assert not torch.equal(y4, y.detach())
    xb = torch.ones(4096, device=DEV, dtype=torch.bfloat16)
    yb = ops.dropout(xb, 0.5, True)
    assert set(yb.float().unique().tolist()) <= {0.0, 2.0}
    assert ops.dropout(xb, 0.5, False) is xb and ops.dropout(xb, 0.0, True) is xb


def test_ffn_dropout_matches_reference_given_the_mask():
    """x + W2 dropout(relu(W1 x + b1)) + b2: mask extracted from the FFN1 epilogue itself"""
    from mmfusion import arena as arena_mod
    p, M, d = 0.25, 200, 64
    torch.manual_seed(0)
    mod = torch.nn.Sequential(torch.nn.Linear(d, 4 * d), torch.nn.Linear(4 * d, d)).cuda()
    arena_mod.ensure(mod)
    x = rnd(M, d, seed=1).to(DEV).bfloat16()
    ops.seed_dropout(7)
    ops.begin_training_forward()
    # the same epilogue with and without dropout at site 1 -> the mask
    h0 = torch.empty((M, 4 * d), dtype=torch.bfloat16, device=DEV)
    h1 = torch.empty_like(h0)
    ops.gemm_group(GEMM_NT, [(x, ops.shadow(mod[0].weight), h0, mod[0].bias.detach(), None)], EPI_BIAS | EPI_RELU)
    ops.gemm_group(GEMM_NT, [(x, ops.shadow(mod[0].weight), h1, mod[0].bias.detach(), None)], EPI_BIAS | EPI_RELU,
                   dropout=(p, 1))
    mask = ((h1 != 0) | (h0 == 0)).float()
    kept = float(mask[h0 != 0].mean())
    assert abs(kept - (1 - p)) < 0.02
    assert rel(h1.float(), h0.float() * mask / (1 - p)) < 2 ** -7
    # fused FFN with dropout (site 1 again) vs fp32 autograd with that mask
    ops._site = 0
    xin = x.clone().requires_grad_(True)
    y = ops.ffn_residual_group([(xin, mod[0], mod[1])], dropout_p=p)[0]
    gy = rnd(M, d, seed=2).to(DEV).bfloat16()
    y.backward(gy)
    torch.cuda.synchronize()
    W1, b1 = ops.shadow(mod[0].weight).float().cpu().requires_grad_(True), mod[0].bias.detach().cpu().clone().requires_grad_(True)
    W2, b2 = ops.shadow(mod[1].weight).float().cpu().requires_grad_(True), mod[1].bias.detach().cpu().clone().requires_grad_(True)
    xr = x.float().cpu().requires_grad_(True)
    hr = torch.relu(xr @ W1.t() + b1) * mask.cpu() / (1 - p)
    yr = xr + hr.to(torch.bfloat16).float() @ W2.t() + b2
    yr.backward(gy.float().cpu())
    assert rel(y, yr) < 2 ** -6
    assert rel(xin.grad, xr.grad) < 3e-2
    assert rel(mod[0].weight.grad, W1.grad) < 3e-2 and rel(mod[1].weight.grad, W2.grad) < 3e-2
    assert rel(mod[0].bias.grad, b1.grad) < 3e-2 and rel(mod[1].bias.grad, b2.grad) < 3e-2


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("f32_in", [True, False])
@pytest.mark.parametrize("p", [0.0, 0.3])
def test_row_linear_folds_cast_dropout_and_relu_gradient(relu, f32_in, p):
    """mmfusion.ops._RowLinear (round 4; mmf_skinny_linear_*_ex): a group of two (B, d)-row linears with an f32 or bf16 input,
    ReLU or not, dropout or not, both output dtypes from one launch.  Forward against fp32 arithmetic on the bf16-rounded
    operands with the mask read off the output (a dropped element is exactly 0); backward against fp32 autograd WITH that mask:
    the dgrad kernel regenerates it (dropout without ReLU) or gates by the saved output (ReLU), and leaves the gated gradient
    behind for the deferred weight-gradient launch."""
    from mmfusion import arena as arena_mod
    from mmfusion.ops import LinearSpec, W
    M, K, N1, N2 = 16, 1536, 520, 264
    torch.manual_seed(0)
    mod = torch.nn.ModuleList([torch.nn.Linear(K, N1), torch.nn.Linear(K, N2)]).cuda()
    arena = arena_mod.ensure(mod)
    arena.zero_grad()
    xs = [rnd(M, K, seed=10 + i).to(DEV) for i in range(2)]
    xin = [(x if f32_in else x.bfloat16()).clone().requires_grad_(True) for x in xs]
    ops.seed_dropout(11)
    ops.begin_training_forward()
    outs = ops.linear_group([(x, LinearSpec(W(m.weight), W(m.bias), relu), None) for x, m in zip(xin, mod)], out_f32=True,
                            dropout_p=p, dual=True)
    gys = [rnd(M, n, seed=20 + i).to(DEV) for i, n in enumerate((N1, N2))]
    # gradient into the f32 copy of problem 0 and into the bf16 copy of problem 1 (as the next linear would send it)
    torch.autograd.backward([outs[0][0], outs[1][1]], [gys[0], gys[1].bfloat16()])
    torch.cuda.synchronize()
    for i, (m, x, (y, y16), gy) in enumerate(zip(mod, xs, outs, gys)):
        assert y.dtype == torch.float32 and y16.dtype == torch.bfloat16 and torch.equal(y16, y.bfloat16())
        W16 = ops.shadow(m.weight).float().cpu().requires_grad_(True)
        b = m.bias.detach().cpu().clone().requires_grad_(True)
        xr = x.bfloat16().float().cpu().requires_grad_(True)
        z = xr @ W16.t() + b
        z = torch.relu(z) if relu else z
        mask = (y.detach().cpu() != 0).float() if p > 0 else torch.ones_like(z)
        if p > 0:
            live = (z.detach() != 0)
            kept = float(mask[live].mean())
            assert abs(kept - (1 - p)) < 0.03, kept
        yr = z * mask / (1 - p)
        assert rel(y, yr) < 1e-5
        g = gy.cpu() if i == 0 else gy.bfloat16().float().cpu()
        yr.backward(g)
        assert rel(xin[i].grad, xr.grad) < 2e-2, (i, rel(xin[i].grad, xr.grad))
        assert xin[i].grad.dtype == (torch.float32 if f32_in else torch.bfloat16)
        assert rel(m.weight.grad, W16.grad) < 2e-2 and rel(m.bias.grad, b.grad) < 2e-2


@pytest.mark.parametrize("B,H,dh,Tq,Tk", [(2, 2, 96, 70, 40), (1, 2, 64, 33, 150), (2, 1, 96, 130, 96),
                                          (1, 2, 96, 30, 150), (2, 1, 96, 130, 30), (1, 2, 64, 20, 100)])   # one narrow side
def test_attention_dropout_matches_reference_given_the_mask(B, H, dh, Tq, Tk):
    """P_dropped is read out by making V a shifted identity; fwd and bwd are then checked with that mask"""
    p, d = 0.2, H * dh
    q = rnd(B, Tq, d, seed=1).to(DEV).bfloat16().reshape(B * Tq, d)
    k = rnd(B, Tk, d, seed=2).to(DEV).bfloat16()
    ops.seed_dropout(99)
    ops.begin_training_forward()
    spec = [ops.AttnSpec(B, Tq, Tk, q=(0, 0), k=(1, 0), v=(1, d))]
    qf = q.float().cpu().view(B, Tq, H, dh).transpose(1, 2)
    kf = k.float().cpu().view(B, Tk, H, dh).transpose(1, 2)
    P = torch.softmax(qf @ kf.transpose(-1, -2) / math.sqrt(dh), -1)            # (B,H,Tq,Tk) reference, undropped
    Pd = torch.zeros_like(P)
    for off in range(0, Tk, dh):                                                 # V[key][c] = (key == off + c)
        v = torch.zeros(B, Tk, H, dh)
        for c in range(min(dh, Tk - off)):
            v[:, off + c, :, c] = 1.0
        kv = torch.cat([k, v.view(B, Tk, d).to(DEV).bfloat16()], -1).reshape(B * Tk, 2 * d)
        ops._site = 0
        o = ops.attention_group(spec, H, dh, [q, kv], dropout_p=p)[0]
        ow = o.float().cpu().view(B, Tq, H, dh).transpose(1, 2)
        Pd[..., off:off + dh] = ow[..., :min(dh, Tk - off)]
    mask = (Pd != 0).float()
    big = P > 1e-2                                        # bf16 of a tiny probability may round to 0
    assert abs(float(mask[big].mean()) - (1 - p)) < 0.03
    assert rel(Pd[big], (P * mask / (1 - p))[big]) < 2 ** -6
    # backward with a general V, same state and site -> same mask
    vgen = rnd(B, Tk, d, seed=3)
    kv = torch.cat([k, vgen.to(DEV).bfloat16()], -1).reshape(B * Tk, 2 * d).requires_grad_(True)
    qin = q.clone().requires_grad_(True)
    ops._site = 0
    o = ops.attention_group(spec, H, dh, [qin, kv], dropout_p=p)[0]
    go = rnd(B * Tq, d, seed=4).to(DEV).bfloat16()
    o.backward(go)
    qr = q.float().cpu().view(B, Tq, d).requires_grad_(True)
    kvr = kv.detach().float().cpu().view(B, Tk, 2 * d).requires_grad_(True)
    qh = qr.view(B, Tq, H, dh).transpose(1, 2)
    kh = kvr[..., :d].reshape(B, Tk, H, dh).transpose(1, 2)
    vh = kvr[..., d:].reshape(B, Tk, H, dh).transpose(1, 2)
    Pr = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(dh), -1) * mask / (1 - p)
    orf = (Pr @ vh).transpose(1, 2).reshape(B * Tq, d)
    orf.backward(go.float().cpu())
    assert rel(o, orf) < 2 ** -6
    assert rel(qin.grad.view(B, Tq, d), qr.grad) < 3e-2
    assert rel(kv.grad.view(B, Tk, 2 * d), kvr.grad) < 3e-2


def test_modules_apply_dropout_only_in_training():
    import config as cfgmod
    from models import fusion_layers as fl
    cfg = cfgmod.ModelConfig()
    cfg.fusion_hidden_size, cfg.fusion_num_heads, cfg.fusion_dropout = 192, 2, 0.3
    torch.manual_seed(0)
    m = fl.MultimodalTransformer(cfg).cuda()
    xs = [rnd(2, T, 192, seed=T).to(DEV) for T in (40, 70, 6)]
    m.eval()
    a, b = m(*xs)["fused_features"], m(*xs)["fused_features"]
    assert torch.equal(a, b)
    m.train()
    c, d_ = m(*xs)["fused_features"], m(*xs)["fused_features"]
    assert not torch.equal(c, d_) and not torch.equal(c, a)
    frac_zero = float((c == 0).float().mean())
    assert frac_zero > 0.25                       # ReLU zeros plus p = 0.3 dropout on the output
    c.sum().backward()                            # backward runs with masks regenerated
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
