"""GPU: the text encoder's prompt-tuning path and masked-mean pooling (reference models/encoders.py:49-71, :90-94) with a
stub backbone — the HF models cannot be fetched here, and what is under test is everything AROUND the
backbone: prompt embeddings prepended to the word embeddings, the attention mask extended by ones, CLS pooling for
'bert'-typed models and the masked mean otherwise, then the projection on the HIP kernels.
The expected values are the reference's lines restated with torch on the CPU plus oracle.ref_cpu.text_projection_tail."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu  # noqa: E402


class _StubBackbone(torch.nn.Module):
    """``.embeddings.word_embeddings``, ``.config.hidden_size`` / ``.model_type`` and a forward that takes either
    ``input_ids`` or ``inputs_embeds`` and returns ``.last_hidden_state`` — the surface encoders.py:41-71 uses.  The body
    is a masked cumulative mix, so the output depends on the prompt rows and on the attention mask."""

    def __init__(self, model_type: str, hidden: int = 768, vocab: int = 50):
        super().__init__()
        self.config = types.SimpleNamespace(hidden_size=hidden, model_type=model_type)
        self.embeddings = torch.nn.Module()
        self.embeddings.word_embeddings = torch.nn.Embedding(vocab, hidden)
        self.mix = torch.nn.Linear(hidden, hidden)

    def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None):
        x = inputs_embeds if inputs_embeds is not None else self.embeddings.word_embeddings(input_ids)
        m = attention_mask.unsqueeze(-1).to(x.dtype)
        ctx = (x * m).cumsum(dim=1) / m.cumsum(dim=1).clamp_min(1.0)
        return types.SimpleNamespace(last_hidden_state=torch.tanh(self.mix(x + ctx)))


def _expected(enc_state, backbone, prompt, input_ids, mask, use_prompt, cls_pool):
    with torch.no_grad():
        if use_prompt:
            B = input_ids.shape[0]
            emb = backbone.embeddings.word_embeddings(input_ids)
            emb = torch.cat([prompt.unsqueeze(0).expand(B, -1, -1), emb], dim=1)                  # reference :52-60
            mask = torch.cat([torch.ones(B, prompt.shape[0], dtype=mask.dtype), mask], dim=1)      # :63-66
            seq = backbone(inputs_embeds=emb, attention_mask=mask).last_hidden_state
        else:
            seq = backbone(input_ids=input_ids, attention_mask=mask).last_hidden_state
        P = {k: v.detach().float().clone() for k, v in enc_state.items()}
        return ref_cpu.text_projection_tail(P, "", seq, mask, cls_pool=cls_pool), seq, mask


@pytest.mark.parametrize("model_type,use_prompt", [("deberta-v2", False), ("deberta-v2", True), ("roformer", False), ("roformer", True)])
def test_text_encoder_prompt_and_pooling_paths(model_type, use_prompt):
    import config as cfgmod
    from models.encoders import TextEncoder
    cfg = cfgmod.ModelConfig()
    cfg.fusion_hidden_size, cfg.fusion_dropout = 256, 0.0
    torch.manual_seed(4)
    cpu_backbone = _StubBackbone(model_type)
    enc = TextEncoder(cfg, backbone=cpu_backbone)
    assert enc.prompt_embeddings.shape == (cfg.prompt_length, 768)
    g = torch.Generator().manual_seed(9)
    B, T = 5, 12
    ids = torch.randint(0, 50, (B, T), generator=g)
    mask = (torch.arange(T)[None, :] < torch.tensor([12, 7, 3, 12, 1])[:, None]).long()
    cls_pool = "bert" in model_type                                                               # reference :87
    want, want_seq, want_mask = _expected({k: v for k, v in enc.state_dict().items() if k.startswith("projection")},
                                          cpu_backbone, enc.prompt_embeddings.detach(), ids, mask, use_prompt, cls_pool)
    enc = enc.cuda().eval()
    with torch.no_grad():
        out = enc(ids.cuda(), mask.cuda(), use_prompt=use_prompt)
    assert out["sequence_output"].shape == want_seq.shape                                          # T + prompt_length with prompts
    assert torch.equal(out["attention_mask"].cpu(), want_mask)
    assert float((out["sequence_output"].cpu() - want_seq).abs().max()) <= 1e-5                    # torch on both sides
    err = float((out["features"].cpu() - want).abs().max())
    assert err <= 1e-2 * max(1.0, float(want.abs().max())), f"{model_type} prompt={use_prompt}: features abs err {err:.3e}"
    # masked mean must ignore the padded positions: changing a padded token must not change the features
    if not cls_pool and not use_prompt:
        ids2 = ids.clone()
        ids2[2, 5] = (ids2[2, 5] + 1) % 50                    # row 2 has 3 valid tokens: position 5 is padding
        with torch.no_grad():
            out2 = enc(ids2.cuda(), mask.cuda())
        assert torch.equal(out2["features"], out["features"])


def test_text_encoder_gradients_reach_prompt_and_projection():
    import config as cfgmod
    from models.encoders import TextEncoder
    cfg = cfgmod.ModelConfig()
    cfg.fusion_hidden_size, cfg.fusion_dropout = 256, 0.0
    torch.manual_seed(6)
    enc = TextEncoder(cfg, backbone=_StubBackbone("roformer")).cuda().train()
    ids = torch.randint(0, 50, (4, 9), generator=torch.Generator().manual_seed(1)).cuda()
    mask = torch.ones(4, 9, dtype=torch.long).cuda()
    out = enc(ids, mask, use_prompt=True)
    out["features"].float().pow(2).sum().backward()
    torch.cuda.synchronize()
    assert enc.prompt_embeddings.grad is not None and float(enc.prompt_embeddings.grad.abs().max()) > 0
    assert float(enc.projection.weight.grad.abs().max()) > 0
