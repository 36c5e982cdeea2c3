"""GPU parity: the HIP-backed modules (models/fusion_layers.py), loaded with the same parameters,
against (1) the committed golden vectors captured from the reference's own classes and (2) the
CPU oracle elementwise, forward AND backward.

Tolerance (BASELINE.json north_star: 1e-2 for bf16): outputs within 1e-2 absolute of the fp32
reference; gradients — which cross several bf16 intermediates — within 3e-2 of the tensor's
max-magnitude (`rel_err`), and their probe-dot / norm checksums from the fixtures within 3e-2."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_cases import CASES  # noqa: E402
from helpers import (cfg_from_meta, fixture_inputs, fixture_params, load_fixture, oracle_fwd_bwd,
                     rel_err)  # noqa: E402
from mmfusion import synth  # noqa: E402

OUT_ATOL = 1e-2
GRAD_RTOL = 3e-2


def build_module(meta):
    from models import fusion_layers as fl
    from models import encoders as enc
    cfg = cfg_from_meta(meta)
    if meta["cls"] == "AdapterLayer":
        m = enc.AdapterLayer(*meta["ctor"])
    else:
        m = getattr(fl, meta["cls"])(cfg)
    m.load_state_dict(fixture_params(meta), strict=True)      # reference state_dict keys/shapes
    return m.cuda().train()                                   # dropout p = 0 in every fixture


def run_hip(meta):
    m = build_module(meta)
    xs = [t.cuda().requires_grad_(True) for t in fixture_inputs(meta)]
    out = m(*xs, **meta.get("kwargs", {}))
    synth.probe_loss(out).backward()
    torch.cuda.synchronize()
    flat = {k: v.detach().float().cpu() for k, v in synth.flatten_outputs(out).items()}
    gp = {k: p.grad.detach().float().cpu() for k, p in m.named_parameters()}
    return flat, [x.grad.float().cpu() for x in xs], gp


@pytest.mark.parametrize("name", sorted(CASES))
def test_module_parity(name):
    fx = load_fixture(name)
    meta = fx.meta
    out, gin, gp = run_hip(meta)
    o_out, o_gin, o_gp = oracle_fwd_bwd(meta)
    assert set(out) == set(fx.out)
    for k, ref in fx.out.items():
        assert out[k].shape == ref.shape, k
        err = float((out[k] - ref).abs().max())
        assert err <= OUT_ATOL * max(1.0, float(ref.abs().max())), f"{name}: output {k} abs err {err:.3e}"
        assert rel_err(out[k], o_out[k]) <= GRAD_RTOL, f"{name}: output {k} vs oracle"
    for i, ref in enumerate(fx.gin):
        assert rel_err(gin[i], ref) <= GRAD_RTOL, f"{name}: input grad {i} rel {rel_err(gin[i], ref):.3e}"
    for k, ref in o_gp.items():
        scale = float(ref.abs().max())
        if scale == 0.0:
            assert float(gp[k].abs().max()) == 0.0, f"{name}: {k} should have zero grad"
            continue
        assert rel_err(gp[k], ref) <= GRAD_RTOL, f"{name}: param grad {k} rel {rel_err(gp[k], ref):.3e}"
    for k, (norm, dot) in meta["grad_checks"].items():
        g = gp[k]
        assert abs(float(g.norm()) - norm) <= GRAD_RTOL * max(norm, 1e-6), f"{name}: grad norm {k}"
