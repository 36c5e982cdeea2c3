"""GPU parity: the HIP-backed modules (models/fusion_layers.py), loaded with the same parameters,
against (1) the committed golden vectors captured from the reference's own classes and (2) the
CPU oracle elementwise, forward AND backward.

Tolerance (BASELINE.json north_star: 1e-2 for bf16): outputs within 1e-2 (absolute, scaled by the
tensor's magnitude when that exceeds 1) of the fp32 reference, every element.  Gradients cross
several bf16 intermediates and ReLU kinks; their tolerances are the constants below."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_cases import CASES  # noqa: E402
from helpers import (cfg_from_meta, fixture_inputs, fixture_params, l2_rel, load_fixture, oracle_fwd_bwd,
                     rel_err)  # noqa: E402
from mmfusion import synth  # noqa: E402

OUT_ATOL = 1e-2          # forward outputs: |hip - reference| <= 1e-2 * max(1, max|reference|), every element
GIN_L2, GIN_MAX = 1.2e-1, 2.5e-1   # input gradients: relative L2 error, and max error relative to max|ref|
GP_L2 = 1.5e-1                     # parameter gradients: relative L2 error ...
GP_L2_RELU = 3.5e-1                # ... except weights/biases that feed a ReLU directly (see below)
GP_NORM = 1.5e-1                   # every parameter gradient's norm vs the reference fixture
# Measured on MI355X (tools/parity_report.py, profiles/parity_r01.txt): input-grad L2 1-8e-2, parameter
# L2 0.2-10e-2 (ReLU-fed up to 15e-2), norm ratios 0.99-1.03 except 3-element / near-cancelling tensors.
# A structural error (transpose, missing term, double accumulation) shows up as an error of O(1).

# Why gradients are looser than outputs: with bf16 activations a pre-activation within ~2e-3 sigma of
# zero lands on the other side of the ReLU kink than in the fp32 reference (~0.2-0.4 % of the units).
# A flipped unit removes/adds its whole term: the unit's own weight row / bias moves by O(max|grad|),
# and everything upstream sees sqrt(f / 0.5) ~ 5-9 % relative L2 noise.  This is a property of bf16
# storage, not of the kernels: test_kernels_gpu.py pins each kernel tightly (1e-5 for f32-output
# GEMMs, 2^-8 for bf16 outputs, 2e-2 for attention grads) with masks taken from the same activations.
RELU_FED = ("ffn.0.", "fusion_layers.0.", "fusion_layers.3.", "final_fusion.0.", "_projector.0.",
            "fusion_layer.0.", "weight_predictor.0.", "meta_fusion.0.", "down_project.", "gcn_layers.",
            "node_type_embedding")


def build_module(meta):
    from models import fusion_layers as fl
    from models import encoders as enc
    cfg = cfg_from_meta(meta)
    if meta["cls"] == "AdapterLayer":
        m = enc.AdapterLayer(*meta["ctor"])
    else:
        m = getattr(fl, meta["cls"])(cfg)
    m.load_state_dict(fixture_params(meta), strict=True)      # reference state_dict keys/shapes
    return m.cuda().eval()      # as captured: eval mode (fixtures have p = 0; AdapterLayer's own 0.1 must be off)


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
        assert rel_err(out[k], o_out[k]) <= OUT_ATOL, f"{name}: output {k} vs oracle"
    for i, ref in enumerate(fx.gin):
        assert l2_rel(gin[i], ref) <= GIN_L2, f"{name}: input grad {i} L2 {l2_rel(gin[i], ref):.3e}"
        assert rel_err(gin[i], ref) <= GIN_MAX, f"{name}: input grad {i} max {rel_err(gin[i], ref):.3e}"
    for k, ref in o_gp.items():
        if float(ref.abs().max()) == 0.0:
            assert float(gp[k].abs().max()) == 0.0, f"{name}: {k} should have zero grad"
            continue
        tol = GP_L2_RELU if any(t in k for t in RELU_FED) else GP_L2
        assert l2_rel(gp[k], ref) <= tol, f"{name}: param grad {k} L2 {l2_rel(gp[k], ref):.3e} > {tol}"
    for k, (norm, dot) in meta["grad_checks"].items():
        assert abs(float(gp[k].norm()) - norm) <= GP_NORM * max(norm, 1e-6), f"{name}: grad norm {k}"
