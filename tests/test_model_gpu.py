"""GPU parity of the whole model glue (BASELINE configs[4], MELD-shaped):
``MultimodalEmotionModel`` in feature mode (backbone features in, because the HF backbones cannot be fetched
offline) against the CPU oracle's composition of the same arithmetic — encoder projection tails, hier-ref
hierarchical fusion at d = 512 / 8 heads (head_dim 64) / G = 512, classifier and auxiliary heads.

The reference model cannot be instantiated in the build container (its constructor downloads the backbones by
name), so this level is oracle-vs-HIP: the oracle's pieces are pinned one by one against the
reference's own classes by the golden fixtures (tests/test_oracle_golden.py), the composition follows
models/multimodal_model.py:104-181.  The video BiLSTM runs on torch.nn.LSTM on both sides with the same weights.
Tolerance: outputs 1e-2 * max(1, |ref|max) (north_star bf16 tolerance); gradients by relative L2."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import l2_rel  # noqa: E402
from oracle import ref_cpu  # noqa: E402

OUT_ATOL = 1e-2


def _build(fusion_type="hierarchical"):
    import config as cfgmod
    from models.multimodal_model import MultimodalEmotionModel
    cfg = cfgmod.ModelConfig()
    cfg.feature_inputs = True
    cfg.fusion_type = fusion_type
    cfg.fusion_hidden_size, cfg.fusion_num_heads = 512, 8            # MELD-shaped: encoder dim 768 -> d = 512
    cfg.graph_hidden_size, cfg.graph_num_layers = 512, 3
    cfg.fusion_dropout = cfg.graph_dropout = 0.0
    torch.manual_seed(5)
    return cfg, MultimodalEmotionModel(cfg)


def _inputs(B=16):
    g = torch.Generator().manual_seed(1234)
    text = torch.randn(B, 9, 768, generator=g)
    audio = torch.randn(B, 21, 768, generator=g)
    video = torch.randn(B, 6, 768, generator=g)
    mask = torch.ones(B, 9, dtype=torch.long)
    return text, mask, audio, video


def _oracle(cfg, model, text, mask, audio, video, fusion_type):
    P = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    tf = ref_cpu.text_projection_tail(P, "text_encoder.", text, mask, cls_pool=True)
    af, _ = ref_cpu.seq_projection_tail(P, "audio_encoder.", audio, "temporal_attention")
    lstm = torch.nn.LSTM(768, 384, num_layers=2, batch_first=True, bidirectional=True)
    lstm.load_state_dict({k[len("video_encoder.temporal_lstm."):]: v.detach().cpu() for k, v in model.state_dict().items()
                          if k.startswith("video_encoder.temporal_lstm.")})
    with torch.no_grad():
        lstm_out, _ = lstm(video)
    vf, _ = ref_cpu.seq_projection_tail(P, "video_encoder.", lstm_out, "facial_attention")
    if fusion_type == "hierarchical":
        fo = ref_cpu.hierarchical_fusion(P, "fusion_layer.", tf, af, vf, num_heads=cfg.fusion_num_heads,
                                         graph_num_layers=cfg.graph_num_layers, temperature=cfg.contrastive_temperature,
                                         compute_contrastive_loss=True)
        fused = fo["fused_features"]
        out = ref_cpu.model_heads(P, fused)
        out.update({k: v for k, v in fo.items() if k != "fused_features"})
    else:
        fo = ref_cpu.late_fusion(P, "fusion_layer.", tf, af, vf)
        head_in = (tf + af + vf) / 3
        out = {"emotion_logits": fo["fused_logits"], "emotion_probs": ref_cpu.softmax_lastdim(fo["fused_logits"]),
               "valence": ref_cpu.linear(head_in, P["valence_regressor.weight"], P["valence_regressor.bias"]),
               "arousal": ref_cpu.linear(head_in, P["arousal_regressor.weight"], P["arousal_regressor.bias"]),
               "uncertainty": ref_cpu.softmax_lastdim(ref_cpu.linear(head_in, P["uncertainty_head.weight"],
                                                                      P["uncertainty_head.bias"])),
               "fusion_weights": fo["fusion_weights"]}
    out.update({"text_features": tf, "audio_features": af, "video_features": vf})
    return P, out


def _loss(out, labels):
    loss = torch.nn.functional.cross_entropy(out["emotion_logits"].float(), labels, label_smoothing=0.1)
    loss = loss + out["valence"].float().sum() * 0.01 + out["arousal"].float().sum() * 0.01
    if isinstance(out.get("contrastive_losses"), dict) and out["contrastive_losses"]:
        loss = loss + 0.1 * sum(out["contrastive_losses"].values())
    return loss


@pytest.mark.parametrize("fusion_type", ["hierarchical", "late"])
def test_meld_shaped_model_matches_oracle(fusion_type):
    cfg, model = _build(fusion_type)
    text, mask, audio, video = _inputs()
    labels = torch.randint(0, 7, (text.shape[0],), generator=torch.Generator().manual_seed(7))
    P, ref = _oracle(cfg, model, text, mask, audio, video, fusion_type)
    _loss(ref, labels).backward()

    # training mode with every dropout probability at zero (MIOpen's LSTM backward needs training mode; the
    # ModalityDropout masks are all-ones at rate 0), so the arithmetic is the oracle's
    model = model.cuda().train()
    model.modality_dropout.dropout_rate = 0.0
    kw = dict(compute_contrastive_loss=True) if fusion_type == "hierarchical" else {}
    out = model({"input_ids": text.cuda(), "attention_mask": mask.cuda()}, audio.cuda(), video.cuda(), **kw)
    _loss(out, labels.cuda()).backward()
    torch.cuda.synchronize()

    keys = ["emotion_logits", "emotion_probs", "valence", "arousal", "uncertainty", "text_features", "audio_features",
            "video_features"]
    if fusion_type == "hierarchical":
        keys += ["early_features", "mult_features", "graph_features", "contrastive_features", "adaptive_features",
                 "adaptive_weights"]
    for k in keys:
        got, want = out[k].detach().float().cpu(), ref[k].detach()
        assert got.shape == want.shape, k
        err = float((got - want).abs().max())
        assert err <= OUT_ATOL * max(1.0, float(want.abs().max())), f"{fusion_type}: {k} abs err {err:.3e}"
    if fusion_type == "hierarchical":
        for k, want in ref["contrastive_losses"].items():
            got = float(out["contrastive_losses"][k].detach())
            assert abs(got - float(want.detach())) <= 2e-2 * max(1.0, abs(float(want.detach()))), k
    # gradients of the head and projection parameters (the glue this test is about); the fusion layers'
    # gradients are covered by tests/test_parity_gpu.py
    names = ["text_encoder.projection.weight", "audio_encoder.projection.weight", "video_encoder.projection.weight",
             "valence_regressor.weight", "arousal_regressor.weight"]
    names += ["classifier.classifier.3.weight", "classifier.classifier.0.weight"] if fusion_type == "hierarchical" else \
        ["fusion_layer.text_classifier.weight", "fusion_layer.audio_classifier.bias"]
    params = dict(model.named_parameters())
    for n in names:
        want = P[n].grad
        assert want is not None, n
        got = params[n].grad.detach().float().cpu()
        assert l2_rel(got, want) <= 0.2, f"{fusion_type}: grad {n} rel L2 {l2_rel(got, want):.3e}"
    # ... and EVERY other parameter the oracle differentiates ("a handful of parameters"): the fusion module's, the heads'
    # and the encoder tails', with the module-level tolerances of tests/test_parity_gpu.py (ReLU-fed tensors looser: a unit on the other
    # side of zero moves its whole row); tensors whose oracle gradient is zero up to cancellation noise must be (near-)zero here too
    from test_parity_gpu import GP_L2, GP_L2_RELU, RELU_FED
    relu_fed = RELU_FED + ("classifier.classifier.0.", "projection.")
    scale = max(float(v.grad.norm()) for v in P.values() if v.grad is not None)
    worst, checked = (0.0, ""), 0
    for n, prm in params.items():
        want = P[n].grad if n in P else None
        if want is None or prm.grad is None or "temporal_lstm" in n:
            continue
        got = prm.grad.detach().float().cpu()
        if float(want.norm()) <= 1e-6 * scale:
            assert float(got.norm()) <= 1e-3 * scale, f"{fusion_type}: {n} should have a (near-)zero gradient"
            continue
        e = l2_rel(got, want)
        checked += 1
        if e > worst[0]:
            worst = (e, n)
        tol = GP_L2_RELU if any(s in n for s in relu_fed) else GP_L2
        assert e <= tol, f"{fusion_type}: grad {n} rel L2 {e:.3e} > {tol}"
    print(f"model level ({fusion_type}): {checked} parameter gradients checked, worst rel L2 {worst[0]:.3e} ({worst[1]})")
    assert checked >= (60 if fusion_type == "hierarchical" else 10)


@pytest.mark.parametrize("T,use_adapter", [(499, False), (499, True), (1, False)])
def test_audio_temporal_head_long_sequence(T, use_adapter):
    """the audio encoder's temporal head (encoders.py:126-131,151-161) on wav2vec2-length
    sequences — self-MHA(768, 8 heads, head_dim 96) over T = 499 frames (not a multiple of any tile size: the
    attention and GEMM kernels' ragged edges), mean over T, Linear(768, d); optionally the AdapterLayer
    (:271-277) first.  T = 1 is the degenerate sequence (softmax over one key).  Forward and the gradients of
    the head's parameters and of the input features against the CPU oracle."""
    import config as cfgmod
    from models.encoders import AudioEncoder
    cfg = cfgmod.ModelConfig()
    cfg.feature_inputs = True
    cfg.fusion_hidden_size, cfg.fusion_dropout = 512, 0.0
    torch.manual_seed(11)
    enc = AudioEncoder(cfg)
    B = 16 if T > 1 else 4
    seq = torch.randn(B, T, 768, generator=torch.Generator().manual_seed(99))
    w = torch.randn(B, 512, generator=torch.Generator().manual_seed(3))

    P = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in enc.state_dict().items()}
    x_ref = seq.clone().requires_grad_(True)
    h = ref_cpu.adapter_layer(P, "adapter.", x_ref) if use_adapter else x_ref
    feat_ref, att_ref = ref_cpu.seq_projection_tail(P, "", h, "temporal_attention")
    ((feat_ref * w).sum() + 1e-3 * att_ref.sum()).backward()

    enc = enc.cuda().eval()          # the AdapterLayer's own Dropout(0.1) (:262) is off, as in the oracle
    x = seq.cuda().requires_grad_(True)
    out = enc(x, use_adapter=use_adapter)
    ((out["features"].float() * w.cuda()).sum() + 1e-3 * out["sequence_output"].float().sum()).backward()
    torch.cuda.synchronize()

    for name, got, want in (("features", out["features"], feat_ref), ("sequence_output", out["sequence_output"], att_ref)):
        got, want = got.detach().float().cpu(), want.detach()
        assert got.shape == want.shape, name
        err = float((got - want).abs().max())
        assert err <= OUT_ATOL * max(1.0, float(want.abs().max())), f"T={T}: {name} abs err {err:.3e}"
    assert l2_rel(x.grad.float().cpu(), x_ref.grad) <= 5e-2, f"T={T}: input grad rel L2 {l2_rel(x.grad.float().cpu(), x_ref.grad):.3e}"
    names = ["projection.weight", "temporal_attention.in_proj_weight", "temporal_attention.out_proj.weight",
             "temporal_attention.in_proj_bias"]
    if use_adapter:
        names += ["adapter.down_project.weight", "adapter.up_project.weight"]
    params = dict(enc.named_parameters())
    for n in names:
        want = P[n].grad
        assert want is not None, n
        got = params[n].grad.detach().float().cpu()
        assert l2_rel(got, want) <= 5e-2, f"T={T}: grad {n} rel L2 {l2_rel(got, want):.3e}"
