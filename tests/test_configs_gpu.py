"""GPU parity of the BASELINE.json configurations that round 1 benchmarked but never compared to anything
(VERDICT r1 "configs_untested"):

  * configs[2] / [3]  *hier-seq*: ``HierarchicalFusion`` on (B, T, d) sequences — MulT on the sequences, the other four
    branches on the mean over T (the reference cannot take 3-D inputs, fact 3) — against
    ``oracle.ref_cpu.hierarchical_fusion(..., mult_inputs=...)`` at the full config-3 size and at a small size;
  * configs[3]        one full training step (``mmfusion.train.FusionTrainStep``: CE(ls=0.1) + 0.1 * contrastive,
    clip_grad_norm_(1.0), AdamW(wd=1e-5) at the OneCycle learning rate — reference advanced_trainer.py:139-182)
    against oracle forward/backward + ``torch.nn.utils.clip_grad_norm_`` + ``torch.optim.AdamW`` + ``OneCycleLR``;
  * configs[4]        the seven missing-modality scenarios of advanced_trainer.py:611-619 through
    ``MultimodalEmotionModel.forward(missing_modalities=...)`` (multimodal_model.py:77-86) and the statistics of
    ``ModalityDropout`` (encoders.py:289-321).

Tolerances: forward 1e-2 * max(1, max|ref|) (north_star bf16), gradients by relative L2 as in test_parity_gpu.py.
GAT arithmetic inside the graph branch stays parity-unpinned; everything else in the oracle is pinned
by the golden fixtures."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import l2_rel, relu_agreement  # noqa: E402
from mmfusion import synth  # noqa: E402
from oracle import ref_cpu  # noqa: E402
from test_parity_gpu import GIN_L2, GP_L2, GP_L2_RELU, OUT_ATOL, RELU_FED  # noqa: E402


def _hier_cfg(d, H, G, L):
    import config as cfgmod
    cfg = cfgmod.ModelConfig()
    cfg.fusion_hidden_size, cfg.fusion_num_heads = d, H
    cfg.graph_hidden_size, cfg.graph_num_layers = G, L
    cfg.fusion_dropout = cfg.graph_dropout = 0.0
    return cfg


def _oracle_hier_seq(cfg, P, xs, unit_masks=None):
    # (bf16-storage mode: the HIP path pools the bf16 rows and stores the means as bf16; no-ops in fp32 mode)
    pooled = [ref_cpu._st(ref_cpu._st(x).mean(dim=1)) for x in xs]
    return ref_cpu.hierarchical_fusion(P, "", *pooled, num_heads=cfg.fusion_num_heads,
                                       graph_num_layers=cfg.graph_num_layers, temperature=cfg.contrastive_temperature,
                                       compute_contrastive_loss=True, mult_inputs=tuple(xs), unit_masks=unit_masks)


HIER_KEYS = ["fused_features", "early_features", "mult_features", "graph_features", "contrastive_features",
             "adaptive_features", "attention_weights", "adaptive_weights"]


@pytest.mark.parametrize("name,B,Ts,d,H", [("small", 3, (24, 20, 6), 128, 2),
                                           ("config3", 16, (512, 400, 30), 768, 8)])
def test_hier_seq_matches_oracle(name, B, Ts, d, H):
    """BASELINE configs[2]: all nine output keys (incl. the three contrastive losses and the returned (B,3,3)
    attention weights), input gradients (which sum the MulT path and the pooled path) and one parameter gradient per
    branch."""
    from helpers import masked_hierarchical_fusion
    cfg = _hier_cfg(d, H, d, 3)
    torch.manual_seed(synth.WEIGHT_SEED)
    m = masked_hierarchical_fusion(cfg)          # the product module + the test-side unit-mask instrument (tests/helpers.py)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    xs = synth.make_features(B, Ts, d)
    xr = [x.clone().requires_grad_(True) for x in xs]
    torch.set_num_threads(16)
    ref = _oracle_hier_seq(cfg, P, xr)
    synth.probe_loss(ref).backward()
    Pb = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    xb = [x.clone().requires_grad_(True) for x in xs]
    cap_ref, cap_hip = {}, {}
    with ref_cpu.bf16_storage():                            # the same arithmetic with bf16 storage
        ref_b = _oracle_hier_seq(cfg, Pb, xb, unit_masks={"capture": cap_ref})
        synth.probe_loss(ref_b).backward()

    m = m.cuda().eval()
    xg = [x.cuda().requires_grad_(True) for x in xs]
    m.unit_masks = {"capture": cap_hip}
    out = m(*xg, compute_contrastive_loss=True)
    m.unit_masks = None
    synth.probe_loss(out).backward()
    torch.cuda.synchronize()
    assert set(out) == set(ref)
    for k in HIER_KEYS:
        got, want = out[k].detach().float().cpu(), ref[k].detach()
        assert got.shape == want.shape, k
        err = float((got - want).abs().max())
        assert err <= OUT_ATOL * max(1.0, float(want.abs().max())), f"{name}: {k} abs err {err:.3e}"
    assert set(out["contrastive_losses"]) == {"text_audio", "text_video", "audio_video"}
    for k, want in ref["contrastive_losses"].items():
        got, want = float(out["contrastive_losses"][k]), float(want)
        assert abs(got - want) <= OUT_ATOL * max(1.0, abs(want)), f"{name}: contrastive loss {k}: {got} vs {want}"
    for i, (g, r) in enumerate(zip(xg, xr)):
        assert l2_rel(g.grad, r.grad) <= GIN_L2, f"{name}: input grad {i} rel L2 {l2_rel(g.grad, r.grad):.3e}"
    params = dict(m.named_parameters())
    for pn in ("meta_fusion.3.weight", "meta_fusion.0.weight", "mult_fusion.text_to_audio.attention.in_proj_weight",
               "mult_fusion.video_to_text.ffn.3.weight", "mult_fusion.final_fusion.0.weight",
               "early_fusion.fusion_layers.3.weight", "contrastive_fusion.text_projector.2.weight",
               "adaptive_fusion.attention.in_proj_weight", "adaptive_fusion.weight_predictor.2.weight",
               "graph_fusion.output_projection.weight", "graph_fusion.gcn_layers.0.lin.weight"):
        want = P[pn].grad
        got = params[pn].grad.detach().float().cpu()
        tol = GP_L2_RELU if any(t in pn for t in RELU_FED) else GP_L2
        assert l2_rel(got, want) <= tol, f"{name}: param grad {pn} rel L2 {l2_rel(got, want):.3e} > {tol}"
    # ---- against the bf16-storage oracle, flip-aware.  Every branch output sits behind a ReLU over only
    # B x d units; at the config-3 size MulT's pooled features differ from the oracle's by ~1e-3 (one-ulp flips of bf16
    # roundings averaged over T = 512), so a handful of top-level units whose pre-activation is within that of zero are on
    # in one and off in the other, and ONE such unit moves every upstream gradient by ~1.3e-2.  Round 2 halved the fp32
    # bound for this; now the flips are COUNTED (they must be few) and those units are switched off on both sides
    # (tests/helpers.py masked_hierarchical_fusion / hierarchical_fusion(unit_masks=...): the four ReLU-terminated branch outputs and the
    # meta MLP's hidden layer), after which the gradients must agree to the same 2e-2 as at the small size.
    from test_parity_gpu import GIN_L2_BF16, GP_L2_BF16, OUT_BF16
    for k in HIER_KEYS:
        if k == "attention_weights":
            continue
        want = ref_b[k].detach()
        assert l2_rel(out[k], want) <= OUT_BF16, f"{name}: {k} vs bf16-storage oracle rel L2 {l2_rel(out[k], want):.3e}"
    relu_out = ["early_features", "mult_features", "contrastive_features", "adaptive_features"]
    agree, nflip, bound = {}, 0, 0
    for k in relu_out + ["meta_hidden"]:
        g_, w_ = (cap_hip[k], cap_ref[k]) if k == "meta_hidden" else (out[k], ref_b[k])
        agree[k], n_, b_ = relu_agreement(g_, w_, f"{name}: {k}")      # per tensor: flips bounded by that tensor's oracle units near zero
        nflip, bound = nflip + n_, bound + b_
    nunit = sum(a.numel() for a in agree.values())
    print(f"hier-seq {name}: {nflip} of {nunit} top-level ReLU units differ in state from the bf16-storage oracle (bound {bound})")
    Pm = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    xm = [x.clone().requires_grad_(True) for x in xs]
    with ref_cpu.bf16_storage():
        ref_m = _oracle_hier_seq(cfg, Pm, xm, unit_masks=agree)
        synth.probe_loss(ref_m).backward()
    m.unit_masks = {k: a.cuda() for k, a in agree.items()}
    for p_ in m.parameters():
        p_.grad.zero_()
    xg2 = [x.cuda().requires_grad_(True) for x in xs]
    synth.probe_loss(m(*xg2, compute_contrastive_loss=True)).backward()
    torch.cuda.synchronize()
    m.unit_masks = None
    for i, (g, r) in enumerate(zip(xg2, xm)):
        assert l2_rel(g.grad, r.grad) <= GIN_L2_BF16, f"{name}: input grad {i} vs bf16-storage oracle (flipped units off) {l2_rel(g.grad, r.grad):.3e}"
    scale = max(float(v.grad.norm()) for v in Pm.values() if v.grad is not None)
    worst = ("", 0.0)
    for pn, p_ in params.items():
        want = Pm[pn].grad
        if want is None or float(want.norm()) <= 1e-6 * scale:      # e.g. the attention vectors of a saturated GAT softmax
            assert float(p_.grad.norm()) <= 1e-4 * scale, f"{name}: {pn} should have a (near-)zero gradient"
            continue
        e = l2_rel(p_.grad.detach().float().cpu(), want)
        worst = max(worst, (pn, e), key=lambda t: t[1])
        tol = 6e-2 if pn.endswith("att_dst") or pn.endswith("att_src") else GP_L2_BF16
        assert e <= tol, f"{name}: param grad {pn} vs bf16-storage oracle (flipped units off) {e:.3e}"
    print(f"hier-seq {name}: worst parameter gradient vs bf16-storage oracle {worst[1]:.3e} ({worst[0]})")
    # ---- the fp32-storage mode against the fp32 oracle at this size: the excuse-free check of the composition
    # (north_star: 1e-3 fp32; asserted 1e-4 on outputs and input gradients, 1e-3 on parameter gradients).  GAT, the
    # adaptive combine, residual sums and pooling are f32 torch glue in this mode (fusion_layers.py), the GEMMs, softmax
    # and LayerNorm HIP.
    m.precision = "fp32"
    for p_ in m.parameters():
        p_.grad.zero_()
    x32 = [x.cuda().requires_grad_(True) for x in xs]
    out32 = m(*x32, compute_contrastive_loss=True)
    synth.probe_loss(out32).backward()
    torch.cuda.synchronize()
    m.precision = None
    for k in HIER_KEYS:
        want = ref[k].detach()
        err = float((out32[k].detach().float().cpu() - want).abs().max()) / max(1.0, float(want.abs().max()))
        assert err <= 1e-4, f"{name}: fp32 mode: {k} scaled abs err {err:.3e}"
    for k, want in ref["contrastive_losses"].items():
        assert abs(float(out32["contrastive_losses"][k]) - float(want)) <= 1e-4 * max(1.0, abs(float(want))), k
    for i, (g, r) in enumerate(zip(x32, xr)):
        assert l2_rel(g.grad, r.grad) <= 1e-4, f"{name}: fp32 mode: input grad {i} rel L2 {l2_rel(g.grad, r.grad):.3e}"
    scale32 = max(float(v.grad.norm()) for v in P.values() if v.grad is not None)
    w32 = ("", 0.0)
    for pn, p_ in params.items():
        want = P[pn].grad
        if want is None or float(want.norm()) <= 1e-6 * scale32:
            continue
        e = l2_rel(p_.grad.detach().float().cpu(), want)
        w32 = max(w32, (pn, e), key=lambda t: t[1])
        assert e <= 1e-3, f"{name}: fp32 mode: param grad {pn} rel L2 {e:.3e}"
    print(f"hier-seq {name}: fp32 mode worst parameter gradient vs the fp32 oracle {w32[1]:.3e} ({w32[0]})")


# ------------------------------------------------------------------------------------------------------------
# one full training step
# ------------------------------------------------------------------------------------------------------------
def _train_setup(d=128, H=2, B=8, Ts=(12, 10, 5)):
    from models import fusion_layers as fl
    from models.multimodal_model import EmotionClassifier
    cfg = _hier_cfg(d, H, d, 2)
    torch.manual_seed(3)
    fusion, head = fl.HierarchicalFusion(cfg), EmotionClassifier(cfg)

    class FusionWithHead(fl._FusionBase):           # one arena over fusion + classifier head (as bench.py --workload train)
        def __init__(self):
            super().__init__()
            self.fusion_layer, self.classifier = fusion, head

        def forward(self, t, a, v, compute_contrastive_loss=False):
            return self.fusion_layer(t, a, v, compute_contrastive_loss=compute_contrastive_loss)
    model = FusionWithHead()
    xs = synth.make_features(B, Ts, d, seed=21)
    labels = torch.randint(0, 7, (B,), generator=torch.Generator().manual_seed(5))
    return cfg, model, xs, labels


def _oracle_train_loss(cfg, P, xs, labels):
    fo = ref_cpu.hierarchical_fusion({k[len("fusion_layer."):]: v for k, v in P.items() if k.startswith("fusion_layer.")},
                                     "", *[x.mean(1) for x in xs], num_heads=cfg.fusion_num_heads,
                                     graph_num_layers=cfg.graph_num_layers, temperature=cfg.contrastive_temperature,
                                     compute_contrastive_loss=True, mult_inputs=tuple(xs))
    logits = ref_cpu.emotion_classifier(P, "classifier.", fo["fused_features"])
    # reference advanced_trainer.py:53,139-166: CE(label_smoothing=0.1) + 0.1 * sum(contrastive)
    loss = torch.nn.functional.cross_entropy(logits, labels, label_smoothing=0.1)
    return loss + 0.1 * sum(fo["contrastive_losses"].values())


def test_training_step_matches_oracle_clip_adamw_onecycle():
    """Two consecutive ``FusionTrainStep`` calls, no host sync in between, against the reference recipe on the CPU:
    oracle loss -> backward -> clip_grad_norm_(1.0) -> AdamW(lr=OneCycle(step), wd=1e-5).

    (1) loss and global gradient norm vs the oracle;  (2) the optimiser arithmetic end to end: torch's clip + AdamW +
    OneCycleLR fed with THIS path's gradients must reproduce this path's parameters to 1e-5 after both steps (schedule,
    bias correction, clip coefficient, weight decay, device-side step counter);  (3) the parameters after each step vs
    the all-oracle run: the Adam update is ~ lr * sign(g) in the first steps, so agreement is judged on the update
    direction (cosine) and on the element-wise bound |dp| <= ~lr per step."""
    from mmfusion import arena as arena_mod
    from mmfusion.train import FusionTrainStep, one_cycle_lr
    cfg, model, xs, labels = _train_setup()
    max_lr, total = 1e-3, 20
    names = [n for n, _ in model.named_parameters()]
    P0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

    # all-oracle run: parameters after step 1 and 2, losses, pre-clip norms
    Pr = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
    ropt = torch.optim.AdamW([Pr[n] for n in names], lr=max_lr, weight_decay=1e-5)
    rsch = torch.optim.lr_scheduler.OneCycleLR(ropt, max_lr=max_lr, total_steps=total, pct_start=0.1, anneal_strategy="cos")
    ref_loss, ref_norm, ref_params = [], [], []
    torch.set_num_threads(16)
    for step in range(2):
        ropt.zero_grad()
        loss = _oracle_train_loss(cfg, Pr, xs, labels)
        loss.backward()
        for n in names:
            if Pr[n].grad is None:
                Pr[n].grad = torch.zeros_like(Pr[n])
        ref_norm.append(float(torch.nn.utils.clip_grad_norm_([Pr[n] for n in names], 1.0)))
        assert abs(ropt.param_groups[0]["lr"] - one_cycle_lr(step, total, max_lr)) < 1e-12
        ropt.step()
        rsch.step()
        ref_loss.append(float(loss))
        ref_params.append({n: Pr[n].detach().clone() for n in names})

    model = model.cuda().train()
    ar = arena_mod.ensure(model)
    ts = FusionTrainStep(model, model.classifier, ar, lr=max_lr, weight_decay=1e-5, max_grad_norm=1.0,
                         total_steps=total, contrastive=True)
    xg = [x.cuda() for x in xs]
    # shadow run: torch's clip + AdamW + OneCycleLR on CPU copies, fed with the HIP path's gradients
    Ps = {n: P0[n].clone().requires_grad_(True) for n in names}
    sopt = torch.optim.AdamW([Ps[n] for n in names], lr=max_lr, weight_decay=1e-5)
    ssch = torch.optim.lr_scheduler.OneCycleLR(sopt, max_lr=max_lr, total_steps=total, pct_start=0.1, anneal_strategy="cos")
    losses, grads_dev, params_dev = [], [], []
    for step in range(2):                                   # NO sync between the steps: everything is cloned on the device
        losses.append(ts(*xg, labels.cuda()).detach().clone())
        grads_dev.append(ar.grads.clone())
        params_dev.append(ar.master.clone())
    torch.cuda.synchronize()
    plist = dict(model.named_parameters())
    offs = {id(p): o for p, o in zip(ar.params, ar.offsets)}

    def views(flat):
        return {n: flat[offs[id(plist[n])]:offs[id(plist[n])] + plist[n].numel()].view(plist[n].shape).cpu() for n in names}
    prev = P0
    for step in range(2):
        g = views(grads_dev[step])
        got = views(params_dev[step])
        # (1)
        assert abs(float(losses[step]) - ref_loss[step]) <= 1e-2 * max(1.0, abs(ref_loss[step])), \
            f"step {step}: loss {float(losses[step])} vs oracle {ref_loss[step]}"
        norm = math.sqrt(sum(float(v.double().pow(2).sum()) for v in g.values()))
        assert abs(norm - ref_norm[step]) <= 0.05 * ref_norm[step], f"step {step}: grad norm {norm} vs oracle {ref_norm[step]}"
        # (2)
        for n in names:
            Ps[n].grad = g[n].clone()
        torch.nn.utils.clip_grad_norm_([Ps[n] for n in names], 1.0)
        sopt.step()
        ssch.step()
        for n in names:
            dev = float((got[n] - Ps[n].detach()).abs().max())
            assert dev <= 1e-5 * max(1.0, float(Ps[n].detach().abs().max())) + 2e-7, \
                f"step {step}: {n} deviates from torch clip+AdamW+OneCycle on the same gradients by {dev:.3e}"
        # (3)
        lr = one_cycle_lr(step, total, max_lr)
        before_ref = P0 if step == 0 else ref_params[0]
        num = den_a = den_b = 0.0
        for n in names:
            da, db = (got[n] - prev[n]).double(), (ref_params[step][n] - before_ref[n]).double()
            if step == 0:            # first Adam step: |dp| = lr |g| / (|g| + eps') + lr wd |p| <= lr (1 + wd |p|)
                assert float(da.abs().max()) <= lr * (1.0 + 1e-5 * float(prev[n].abs().max())) * 1.001 + 1e-9, n
            num, den_a, den_b = num + float((da * db).sum()), den_a + float((da * da).sum()), den_b + float((db * db).sum())
        cos = num / math.sqrt(den_a * den_b)
        print(f"train step {step}: loss {float(losses[step]):.5f} (oracle {ref_loss[step]:.5f}) |g| {norm:.4f} "
              f"(oracle {ref_norm[step]:.4f}) update cosine {cos:.4f}")
        assert cos >= 0.8, f"step {step}: update direction cosine vs the all-oracle run {cos:.3f}"
        prev = got
    assert ts.opt.t == 2 and int(ts.opt.step_dev.item()) == 2


def test_training_step_at_the_bench_size_matches_oracle_loss_and_gradient_norm():
    """``FusionTrainStep`` at the size ``bench.py --workload train`` times (BASELINE configs[3]: hier-seq, B = 16,
    T = 512/400/30, d = 768, H = 8, G = 768, L = 3 + classifier head): loss within 1e-2 and the global pre-clip gradient
    norm within 5 % of the oracle's (reference recipe advanced_trainer.py:139-182), and the clipped update moves no
    parameter by more than lr (1 + wd |p|) in the first Adam step."""
    from mmfusion import arena as arena_mod
    from mmfusion.train import FusionTrainStep, one_cycle_lr
    S = synth.C2_SHAPES
    from models import fusion_layers as fl
    from models.multimodal_model import EmotionClassifier
    cfg = _hier_cfg(S["d"], S["heads"], S["d"], 3)
    torch.manual_seed(synth.WEIGHT_SEED)
    fusion, head = fl.HierarchicalFusion(cfg), EmotionClassifier(cfg)

    class FusionWithHead(fl._FusionBase):
        def __init__(self):
            super().__init__()
            self.fusion_layer, self.classifier = fusion, head

        def forward(self, t, a, v, compute_contrastive_loss=False):
            return self.fusion_layer(t, a, v, compute_contrastive_loss=compute_contrastive_loss)
    model = FusionWithHead()
    xs = synth.make_features(S["B"], (S["T_text"], S["T_audio"], S["T_frames"]), S["d"])
    labels = torch.randint(0, 7, (S["B"],), generator=torch.Generator().manual_seed(99))
    names = [n for n, _ in model.named_parameters()]
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    torch.set_num_threads(16)
    ref_loss = _oracle_train_loss(cfg, P, xs, labels)
    ref_loss.backward()
    ref_norm = math.sqrt(sum(float(P[n].grad.double().pow(2).sum()) for n in names if P[n].grad is not None))

    model = model.cuda().train()
    ar = arena_mod.ensure(model)
    max_lr, total = 1e-4, 1000
    ts = FusionTrainStep(model, model.classifier, ar, lr=max_lr, weight_decay=1e-5, max_grad_norm=1.0, total_steps=total)
    before = ar.master.clone()
    loss = ts(*[x.cuda() for x in xs], labels.cuda())
    grads = ar.grads.clone()
    torch.cuda.synchronize()
    norm = float(grads.double().norm())
    print(f"train step at the bench size: loss {float(loss):.5f} (oracle {float(ref_loss):.5f}), |g| {norm:.4f} (oracle {ref_norm:.4f})")
    assert abs(float(loss) - float(ref_loss)) <= 1e-2 * max(1.0, abs(float(ref_loss)))
    assert abs(norm - ref_norm) <= 0.05 * ref_norm
    lr0 = one_cycle_lr(0, total, max_lr)
    moved = (ar.master - before).abs()
    # (2 f32 ulps of the parameter: `moved` is a difference of rounded f32 values)
    assert float((moved - lr0 * (1.0 + 1e-5 * before.abs()) * 1.001 - 2.4e-7 * before.abs()).max()) <= 1e-9, \
        "an update larger than the first Adam step allows"
    assert float(moved.max()) > 0.5 * lr0, "the optimiser did not move the parameters"
    assert int(ts.opt.step_dev.item()) == 1


def test_optimizer_step_counter_survives_graph_replay_and_checkpoint(tmp_path):
    """``FusedAdamW.advance()`` captured in a hipGraph moves the DEVICE step counter on every replay but the
    host mirror only once; ``state_dict()`` must report the device's count, and a run resumed from the checkpoint must
    continue exactly like the uninterrupted one (bias corrections, OneCycle position)."""
    from mmfusion import arena as arena_mod
    from mmfusion.train import FusedAdamW, load_checkpoint, save_checkpoint

    def make():
        torch.manual_seed(11)
        lin = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.Linear(128, 64)).cuda()
        ar = arena_mod.ensure(lin)
        opt = FusedAdamW(ar, lr=1e-3, weight_decay=1e-5, max_grad_norm=1.0)
        opt.set_schedule(1e-3, 50)
        return lin, ar, opt

    def fill(ar, k):
        g = torch.Generator(device="cuda").manual_seed(100 + k)
        ar.grads.copy_(torch.randn(ar.grads.shape, device="cuda", generator=g))

    def captured(opt):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            opt.advance(); opt.launch()                       # one eager warm-up step (step 1)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            opt.advance(); opt.launch()
        return g

    # uninterrupted: the eager warm-up step (step 1; a capture executes nothing) + 6 replays = 7 steps
    lin_a, ar_a, opt_a = make()
    fill(ar_a, 0)
    ga = captured(opt_a)
    for k in range(6):
        fill(ar_a, k + 1)
        ga.replay()
    torch.cuda.synchronize()
    assert int(opt_a.step_dev.item()) == 7 and opt_a.t == 2            # the host mirror only saw the two Python calls
    sd = opt_a.state_dict(lin_a.parameters())
    assert float(sd["state"][0]["step"]) == 7.0, "state_dict reports the host mirror, not the device counter"

    # interrupted after 3 replays (step 4): checkpoint, reload into a fresh optimiser, 3 more eager device-side steps
    lin_b, ar_b, opt_b = make()
    fill(ar_b, 0)
    gb = captured(opt_b)
    for k in range(3):
        fill(ar_b, k + 1)
        gb.replay()
    path = str(tmp_path / "ck.pth")
    save_checkpoint(path, lin_b, opt_b, epoch=1)
    lin_c, ar_c, opt_c = make()
    load_checkpoint(path, lin_c, opt_c)
    assert opt_c.t == 4 and int(opt_c.step_dev.item()) == 4
    for k in range(3, 6):
        fill(ar_c, k + 1)
        opt_c.advance(); opt_c.launch()
    torch.cuda.synchronize()
    assert int(opt_c.step_dev.item()) == 7
    dev = float((ar_c.master - ar_a.master).abs().max())
    assert dev <= 1e-7, f"resumed run deviates from the uninterrupted one by {dev:.3e}"
    # and the host-side path after device-side steps: set_hparams must continue from the device's count
    opt_a.set_hparams(lr=1e-3)
    assert opt_a.t == 8 and int(opt_a.step_dev.item()) == 8


# ------------------------------------------------------------------------------------------------------------
# config 5: missing-modality robustness scenarios, ModalityDropout
# ------------------------------------------------------------------------------------------------------------
SCENARIOS = [[], ["text"], ["audio"], ["video"], ["text", "audio"], ["text", "video"], ["audio", "video"]]


@pytest.mark.parametrize("missing", SCENARIOS, ids=lambda m: "all" if not m else "_".join(m) + "_missing")
def test_missing_modality_scenarios_match_oracle(missing):
    """advanced_trainer.py:611-619 (eval mode, no_grad) through MultimodalEmotionModel.forward(missing_modalities=...):
    the model zeroes the RAW inputs of a missing modality before the encoders (multimodal_model.py:77-86) — in
    feature mode the raw inputs are the backbone features — so the oracle gets zeroed features through the same
    encoder tails (biases, the LSTM and the MHA heads still act on the zeros)."""
    from test_model_gpu import _build, _inputs, _oracle
    cfg, model = _build("hierarchical")
    text, mask, audio, video = _inputs()
    zt = torch.zeros_like(text) if "text" in missing else text
    za = torch.zeros_like(audio) if "audio" in missing else audio
    zv = torch.zeros_like(video) if "video" in missing else video
    with torch.no_grad():
        _, ref = _oracle(cfg, model, zt, mask, za, zv, "hierarchical")
        model = model.cuda().eval()
        out = model({"input_ids": text.cuda(), "attention_mask": mask.cuda()}, audio.cuda(), video.cuda(),
                    compute_contrastive_loss=True, missing_modalities=list(missing))
    torch.cuda.synchronize()
    for k in ["emotion_logits", "emotion_probs", "valence", "arousal", "uncertainty", "text_features", "audio_features",
              "video_features", "early_features", "mult_features", "graph_features", "contrastive_features",
              "adaptive_features", "adaptive_weights"]:
        got, want = out[k].detach().float().cpu(), ref[k].detach()
        assert got.shape == want.shape, k
        err = float((got - want).abs().max())
        assert err <= OUT_ATOL * max(1.0, float(want.abs().max())), f"{missing}: {k} abs err {err:.3e}"
    # what the scenario reports: argmax predictions (advanced_trainer.py:641).  A prediction may only differ where
    # the oracle's top-2 logits are closer than the tolerance
    lg, lr_ = out["emotion_logits"].float().cpu(), ref["emotion_logits"]
    top2 = lr_.topk(2, dim=-1).values
    decided = (top2[:, 0] - top2[:, 1]) > 2 * OUT_ATOL * max(1.0, float(lr_.abs().max()))
    assert torch.equal(lg.argmax(-1)[decided], lr_.argmax(-1)[decided])


def test_missing_modalities_ignored_names_and_input_untouched():
    """The caller's tensors are not modified (zeros_like copies, multimodal_model.py:80-86) and an empty list is the
    full model."""
    from test_model_gpu import _build, _inputs
    cfg, model = _build("late")
    text, mask, audio, video = (t.cuda() for t in _inputs())
    model = model.cuda().eval()
    keep = audio.clone()
    with torch.no_grad():
        a = model({"input_ids": text, "attention_mask": mask}, audio, video, missing_modalities=[])
        b = model({"input_ids": text, "attention_mask": mask}, audio, video)
        c = model({"input_ids": text, "attention_mask": mask}, audio, video, missing_modalities=["audio"])
    assert torch.equal(audio, keep)
    assert torch.equal(a["emotion_logits"], b["emotion_logits"])
    assert not torch.equal(a["emotion_logits"], c["emotion_logits"])


def test_modality_dropout_statistics():
    """encoders.py:289-321: per-sample Bernoulli keep-masks at rate 1 - p, NO 1/(1-p) rescale (kept rows are bit-equal
    to the input, dropped rows exactly zero), never all three modalities dropped (one is given back uniformly at
    random), identity when not training."""
    from models.encoders import ModalityDropout
    B, d = 20000, 16
    g = torch.Generator().manual_seed(0)
    t, a, v = (torch.randn(B, d, generator=g).cuda() + 3.0 for _ in range(3))      # no exact zeros in the inputs
    md = ModalityDropout(dropout_rate=0.1)
    torch.manual_seed(1)
    yt, ya, yv = md(t, a, v, training=True)
    kept = []
    for x, y in ((t, yt), (a, ya), (v, yv)):
        row_kept = (y != 0).all(dim=1)
        row_zero = (y == 0).all(dim=1)
        assert bool((row_kept | row_zero).all()), "a row is neither kept nor dropped as a whole"
        assert torch.equal(y[row_kept], x[row_kept]), "kept rows must not be rescaled"
        kept.append(row_kept)
        rate = float(row_kept.float().mean())
        # P(keep) = 0.9 + P(all dropped) / 3 = 0.9003; 5 sigma of a B-sample mean
        assert abs(rate - 0.9003) <= 5 * math.sqrt(0.9 * 0.1 / B), rate
    assert bool((kept[0] | kept[1] | kept[2]).all()), "a sample lost all three modalities"
    both = float((kept[0] & kept[1]).float().mean())                 # independence of the masks
    assert abs(both - 0.81) <= 5 * math.sqrt(0.81 * 0.19 / B) + 1e-3
    # the rescue branch: at p = 0.95 almost every sample loses all three draws
    md2 = ModalityDropout(dropout_rate=0.95)
    yt, ya, yv = md2(t, a, v, training=True)
    k = torch.stack([(y != 0).all(dim=1) for y in (yt, ya, yv)], dim=1)
    assert bool(k.any(dim=1).all())
    only_one = k.sum(dim=1) == 1
    share = k[only_one].float().mean(dim=0)                          # rescued samples: uniform over the modalities
    assert float((share - 1 / 3).abs().max()) < 0.03
    # eval: identity, same objects
    o = md(t, a, v, training=False)
    assert o[0] is t and o[1] is a and o[2] is v


# ------------------------------------------------------------------------------------------------------------
# edge cases of the boundary: batch of one, sequence of one, empty batch, unsupported widths
# ------------------------------------------------------------------------------------------------------------
def _mult(d=192, H=2):
    import config as cfgmod
    from models import fusion_layers as fl
    cfg = cfgmod.ModelConfig()
    cfg.fusion_hidden_size, cfg.fusion_num_heads, cfg.fusion_dropout = d, H, 0.0
    torch.manual_seed(1)
    return cfg, fl.MultimodalTransformer(cfg)


@pytest.mark.parametrize("B,Ts", [(1, (1, 1, 1)), (1, (65, 3, 1)), (2, (1, 130, 7))])
def test_mult_smallest_shapes_match_oracle(B, Ts):
    """Batch of one, sequences of one token (softmax over one key), a modality far longer than the others."""
    cfg, m = _mult()
    P = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    xs = synth.make_features(B, Ts, 192, seed=5)
    xr = [x.clone().requires_grad_(True) for x in xs]
    with ref_cpu.bf16_storage():
        ref = ref_cpu.multimodal_transformer(P, "", *xr, 2)
        synth.probe_loss(ref).backward()
    m = m.cuda().eval()
    xg = [x.cuda().requires_grad_(True) for x in xs]
    out = m(*xg)
    synth.probe_loss(out).backward()
    torch.cuda.synchronize()
    for k, want in ref.items():
        assert l2_rel(out[k], want.detach()) <= 5e-3, f"{k}: {l2_rel(out[k], want.detach()):.3e}"
    for g, r in zip(xg, xr):
        assert l2_rel(g.grad, r.grad) <= 2e-2, f"input grad {l2_rel(g.grad, r.grad):.3e}"


def test_empty_batch_and_bad_widths_raise_cleanly():
    """An empty batch or a feature width the kernels cannot take must come back as a Python exception from the
    C ABI's validation (no launch with an empty grid, no fault), and the module must stay usable afterwards."""
    cfg, m = _mult()
    m = m.cuda().eval()
    with pytest.raises((RuntimeError, ValueError)):
        m(torch.zeros(0, 4, 192, device="cuda"), torch.zeros(0, 3, 192, device="cuda"), torch.zeros(0, 2, 192, device="cuda"))
    with pytest.raises((RuntimeError, ValueError)):                 # feature width differs from the module's
        m(torch.zeros(2, 4, 190, device="cuda"), torch.zeros(2, 3, 190, device="cuda"), torch.zeros(2, 2, 190, device="cuda"))
    with pytest.raises(RuntimeError):                               # CPU tensors: there is no fallback
        m(torch.zeros(2, 4, 192), torch.zeros(2, 3, 192), torch.zeros(2, 2, 192))
    out = m(torch.randn(2, 4, 192, device="cuda"), torch.randn(2, 3, 192, device="cuda"), torch.randn(2, 2, 192, device="cuda"))
    torch.cuda.synchronize()
    assert out["fused_features"].shape == (2, 192) and bool(torch.isfinite(out["fused_features"]).all())
