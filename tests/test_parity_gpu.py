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
from helpers import (cfg_from_meta, fixture_inputs, fixture_params, l2_rel, load_fixture, oracle_fwd_bwd, relu_agreement,
                     rel_err)  # noqa: E402
from mmfusion import synth  # noqa: E402

OUT_ATOL = 1e-2          # forward outputs: |hip - reference| <= 1e-2 * max(1, max|reference|), every element
GIN_L2, GIN_MAX = 1.2e-1, 2.5e-1   # input gradients: relative L2 error, and max error relative to max|ref|
GP_L2 = 1.5e-1                     # parameter gradients: relative L2 error ...
GP_L2_RELU = 3.5e-1                # ... except weights/biases that feed a ReLU directly (see below)
GP_NORM = 1.5e-1                   # every parameter gradient's norm vs the reference fixture
# Against the oracle in bf16-storage mode (oracle/ref_cpu.py: it rounds where the HIP path stores bf16 and models the
# attention kernels' online softmax), the ReLU masks coincide and what is left is fp32 summation order plus the rare
# rounding-boundary flip.  Measured on MI355X (profiles/parity_r02.txt): input gradients <= 1.2e-2, parameters <= 1.5e-2.
OUT_BF16 = 5e-3                    # forward outputs: max |hip - oracle_bf16| / max|oracle_bf16| (one bf16 ulp of the largest value)
GIN_L2_BF16 = 2e-2                 # input gradients, relative L2
GP_L2_BF16 = 2e-2                  # every parameter gradient, relative L2 (ReLU-fed ones included)
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


def run_hip(meta, precision=None):
    m = build_module(meta)
    if precision is not None:
        m.precision = precision            # "fp32": the parity mode of mmfusion.ops_f32 (f32 storage, exact f32 MFMA)
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
    # the tight check: the same arithmetic with bf16 storage
    b_out, b_gin, b_gp = oracle_fwd_bwd(meta, storage="bf16")
    for k in b_out:
        assert rel_err(out[k], b_out[k]) <= OUT_BF16 * 2, f"{name}: output {k} vs bf16-storage oracle {rel_err(out[k], b_out[k]):.3e}"
    for i, ref in enumerate(b_gin):
        assert l2_rel(gin[i], ref) <= GIN_L2_BF16, f"{name}: input grad {i} vs bf16-storage oracle L2 {l2_rel(gin[i], ref):.3e}"
    for k, ref in b_gp.items():
        if float(ref.abs().max()) == 0.0:
            continue
        # GAT att_dst: softmax_j is almost invariant to the per-target term s_dst[i] (only the leaky-relu kink breaks
        # the invariance), so this gradient is a small difference of large cancelling sums — measured 2.9e-2
        tol = 6e-2 if k.endswith("att_dst") else GP_L2_BF16
        assert l2_rel(gp[k], ref) <= tol, f"{name}: param grad {k} vs bf16-storage oracle L2 {l2_rel(gp[k], ref):.3e}"


FP32_TOL = 1e-4        # BASELINE.json north_star asks "within 1e-3 fp32"; measured on MI355X <= 1e-6 outputs, <= 1.2e-5 grads


@pytest.mark.parametrize("name", sorted(CASES))
def test_module_parity_fp32_storage(name):
    """The fp32-storage mode (module.precision = "fp32": f32 activations and weights in HBM, exact f32 MFMA GEMMs,
    explicit-scores attention, f32 LayerNorm — mmfusion/ops_f32.py, csrc/gemmf32.hip) against the reference's own
    vectors and the fp32 oracle, ten times inside the north_star's fp32 tolerance: every output element within
    1e-4 * max(1, |ref|max), every input and parameter gradient within 1e-4 relative L2.  No ReLU-mask excuse here."""
    fx = load_fixture(name)
    meta = fx.meta
    out, gin, gp = run_hip(meta, precision="fp32")
    o_out, o_gin, o_gp = oracle_fwd_bwd(meta)
    worst = [0.0, 0.0, 0.0]
    for k, ref in fx.out.items():
        err = float((out[k] - ref).abs().max()) / max(1.0, float(ref.abs().max()))
        worst[0] = max(worst[0], err)
        assert err <= FP32_TOL, f"{name}: output {k} scaled abs err {err:.3e}"
    for i, ref in enumerate(fx.gin):
        worst[1] = max(worst[1], l2_rel(gin[i], ref))
        assert l2_rel(gin[i], ref) <= FP32_TOL, f"{name}: input grad {i} L2 {l2_rel(gin[i], ref):.3e}"
    scale = max(float(v.norm()) for v in o_gp.values())
    for k, ref in o_gp.items():
        if float(ref.norm()) <= 1e-7 * scale:                    # a gradient that is zero up to cancellation noise
            assert float(gp[k].norm()) <= 1e-5 * scale, f"{name}: {k} should have (near-)zero grad"
            continue
        worst[2] = max(worst[2], l2_rel(gp[k], ref))
        assert l2_rel(gp[k], ref) <= FP32_TOL, f"{name}: param grad {k} L2 {l2_rel(gp[k], ref):.3e}"
    print(f"fp32 mode {name}: outputs {worst[0]:.2e}, input grads {worst[1]:.2e}, param grads {worst[2]:.2e}")


def test_mult_at_the_bench_configuration_matches_oracle():
    """BASELINE.json configs[1] at its full size — B=16, T=512/400/30, d=768, H=8 (the workload bench.py times):
    every MulT output within 1e-2 * max(1, |ref|max) of the fp32 CPU oracle, input gradients by relative L2, and
    two runs of the HIP path bit-identical (no atomics in the attention backward, fixed reduction orders).  The
    oracle is pinned against the reference's own classes at the small golden shapes (tests/test_oracle_golden.py);
    at this size it is the reference-equivalent arithmetic (same results to 1e-5, same time)."""
    import config as cfgmod
    from models import fusion_layers as fl
    from oracle import ref_cpu
    S = synth.C2_SHAPES
    cfg = cfgmod.ModelConfig()
    cfg.fusion_hidden_size, cfg.fusion_num_heads, cfg.fusion_dropout = S["d"], S["heads"], 0.0
    torch.manual_seed(synth.WEIGHT_SEED)
    m = fl.MultimodalTransformer(cfg)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    xs = synth.make_features(S["B"], (S["T_text"], S["T_audio"], S["T_frames"]), S["d"])
    xr = [x.clone().requires_grad_(True) for x in xs]
    torch.set_num_threads(16)
    ref = ref_cpu.multimodal_transformer(P, "", *xr, S["heads"])
    synth.probe_loss(ref).backward()
    # The tight check further down: the same arithmetic with bf16 storage (the oracle rounds where the HIP path stores bf16).
    Pb = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    xb = [x.clone().requires_grad_(True) for x in xs]
    m = m.cuda().eval()

    def run():
        xg = [x.cuda().requires_grad_(True) for x in xs]
        for p in m.parameters():
            if p.grad is not None:
                p.grad.zero_()
        out = m(*xg)
        synth.probe_loss(out).backward()
        torch.cuda.synchronize()
        return out, xg
    out, xg = run()
    for k, want in ref.items():
        got, want = out[k].detach().float().cpu(), want.detach()
        err = float((got - want).abs().max())
        assert err <= OUT_ATOL * max(1.0, float(want.abs().max())), f"{k}: abs err {err:.3e}"
    for g, r in zip(xg, xr):
        assert l2_rel(g.grad, r.grad) <= GIN_L2, f"input grad rel L2 {l2_rel(g.grad, r.grad):.3e}"
    for name in ("final_fusion.0.weight", "text_to_audio.attention.in_proj_weight", "audio_to_text.ffn.3.weight"):
        got = dict(m.named_parameters())[name].grad.float().cpu()
        tol = GP_L2_RELU if any(t in name for t in RELU_FED) else GP_L2
        assert l2_rel(got, P[name].grad) <= tol, f"param grad {name}: {l2_rel(got, P[name].grad):.3e}"
    # ---- against the bf16-storage oracle, flip-aware: the units of fused_features whose ReLU state
    # differs between the two sides are counted (few) and left out of the probe loss on BOTH sides — fused_features feeds
    # nothing else, so masking it in the loss switches those units off completely — and then EVERY gradient, final_fusion's
    # included, must agree to 2e-2.  (Round 2 left fused_features out of this comparison altogether.)
    with ref_cpu.bf16_storage(), torch.no_grad():
        ref_b0 = ref_cpu.multimodal_transformer({k: v.detach() for k, v in P.items()}, "", *xs, S["heads"])
    for k, want in ref_b0.items():                       # forward against the bf16-storage oracle
        assert l2_rel(out[k], want) <= OUT_BF16, f"{k} vs bf16-storage oracle rel L2 {l2_rel(out[k], want):.3e}"
    agree, nflip, bound = relu_agreement(out["fused_features"], ref_b0["fused_features"], "fused_features")
    print(f"bench config: {nflip} of {agree.numel()} fused_features units differ in ReLU state from the bf16-storage oracle "
          f"(bound from the oracle's pre-activations near zero: {bound})")

    def masked(o, mask):
        o = dict(o)
        o["fused_features"] = o["fused_features"] * mask
        return o
    with ref_cpu.bf16_storage():
        ref_b = ref_cpu.multimodal_transformer(Pb, "", *xb, S["heads"])
        synth.probe_loss(masked(ref_b, agree)).backward()
    xg3 = [x.cuda().requires_grad_(True) for x in xs]
    for p in m.parameters():
        p.grad.zero_()
    synth.probe_loss(masked(m(*xg3), agree.cuda())).backward()
    torch.cuda.synchronize()
    worst = 0.0
    for g, rb in zip(xg3, xb):
        assert l2_rel(g.grad, rb.grad) <= GIN_L2_BF16, f"input grad vs bf16-storage oracle rel L2 {l2_rel(g.grad, rb.grad):.3e}"
    for name, p in m.named_parameters():                 # EVERY parameter gradient
        e = l2_rel(p.grad.float().cpu(), Pb[name].grad)
        worst = max(worst, e)
        assert e <= GP_L2_BF16, f"param grad {name} vs bf16-storage oracle: {e:.3e}"
    print(f"bench config: worst parameter-gradient rel L2 vs bf16-storage oracle {worst:.3e}; input grads "
          f"{[round(l2_rel(g.grad, rb.grad), 5) for g, rb in zip(xg3, xb)]}")
    # and the fp32-storage mode at this size against the fp32 oracle, full probe loss, 1e-4 on everything
    m.precision = "fp32"
    out32, xg32 = run()
    for k, want in ref.items():
        err = float((out32[k].detach().float().cpu() - want.detach()).abs().max()) / max(1.0, float(want.abs().max()))
        assert err <= FP32_TOL, f"fp32 mode: {k} scaled abs err {err:.3e}"
    for g, r in zip(xg32, xr):
        assert l2_rel(g.grad, r.grad) <= FP32_TOL, f"fp32 mode: input grad rel L2 {l2_rel(g.grad, r.grad):.3e}"
    w32, wname = max((l2_rel(p.grad.float().cpu(), P[n].grad), n) for n, p in m.named_parameters())
    # parameters: the north_star's 1e-3 (measured 7.6e-4 on the worst tensor at this size — 8192-row fp32 reductions in
    # a different order than the CPU's blocked sums; outputs and input gradients above hold 1e-4)
    assert w32 <= 1e-3, f"fp32 mode: worst parameter gradient {wname} rel L2 {w32:.3e}"
    print(f"bench config, fp32 mode: worst parameter gradient rel L2 vs the fp32 oracle {w32:.3e} ({wname})")
    m.precision = None
    out, xg = run()
    out2, xg2 = run()
    assert all(torch.equal(out[k], out2[k]) for k in out), "forward is not bit-reproducible"
    assert all(torch.equal(a.grad, b.grad) for a, b in zip(xg, xg2)), "backward is not bit-reproducible"
