"""Per-modality encoder projection tails — drop-in names for the reference's ``models/encoders.py``.

In scope: everything *after* the pretrained backbone —
pool -> optional ``AdapterLayer`` -> ``projection`` Linear(hidden -> fusion_hidden_size) -> dropout
(reference encoders.py:86-98, :151-161, :232-245), plus ``AdapterLayer`` (:254-277) and
``ModalityDropout`` (:280-321).  The projections, the adapter GEMMs and the temporal / facial
self-attention heads run on the HIP kernels (``mmfusion.ops``).

Out of scope: the HuggingFace backbones themselves (DeBERTa-v3 / Wav2Vec2 / ViT, reference :20,116,179).
They are third-party pretrained models fetched by name; there is no network here.  The encoders
therefore take a ``backbone`` argument:
  * ``backbone=None`` (default) reproduces the reference: ``from_pretrained(config.*_model_name)``;
  * any ``nn.Module`` returning an object with ``.last_hidden_state`` is used as is;
  * ``config.feature_inputs = True`` (dynamic attribute) builds no backbone at all: the ``forward``
    inputs are then precomputed backbone features ``(B, T, hidden)`` — the synthetic-feature route
    of BASELINE.json's configs.
The video BiLSTM (reference :183-190,233) runs on the HIP path too (``mmfusion.lstm_ops``: grouped MFMA GEMMs for the
input projections, one persistent launch per layer for the 30 sequential steps of both directions); the
``torch.nn.LSTM`` module is only the parameter container (state_dict keys ``temporal_lstm.weight_ih_l0`` ...).
The audio / video heads return the head-averaged ``attention_weights`` (B, T, T) like the reference
(:152-154,236-238; no gradient); ``config.encoder_attention_weights = False`` (dynamic attribute) skips that kernel.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from mmfusion import ops
from mmfusion.ops import AttnSpec, W
from .fusion_layers import _FusionBase, _MHAParams, _as_rows, _p, _wb


def _feature_mode(config) -> bool:
    return bool(getattr(config, "feature_inputs", False))


def _load_backbone(kind: str, name: str):
    """Reference behaviour (encoders.py:20,116,179).  Needs the HF weights on disk or a network."""
    from transformers import AutoModel, ViTModel, Wav2Vec2Model
    cls = {"text": AutoModel, "audio": Wav2Vec2Model, "video": ViTModel}[kind]
    return cls.from_pretrained(name)


class AdapterLayer(_FusionBase):
    """x + up(relu(down(x)))  (reference :254-277; N(0, 0.02) weights, zero biases).
    Both GEMMs run on the MFMA kernel; the residual add is the up-projection's epilogue."""

    def __init__(self, hidden_size: int, adapter_size: int):
        super().__init__()
        self.down_project = nn.Linear(hidden_size, adapter_size)
        self.up_project = nn.Linear(adapter_size, hidden_size)
        self.activation = nn.ReLU()
        self.dropout = nn.Dropout(0.1)
        nn.init.normal_(self.down_project.weight, std=0.02)
        nn.init.normal_(self.up_project.weight, std=0.02)
        nn.init.zeros_(self.down_project.bias)
        nn.init.zeros_(self.up_project.bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        rows = _as_rows(x)
        h = ops.dropout(ops.linear(rows, *_wb(self.down_project), relu=True), _p(self, self.dropout.p), True)
        y = ops.linear(h, *_wb(self.up_project), residual=rows)
        return ops.to_f32(y).reshape(x.shape)


def _mha_mean_project(mha: _MHAParams, projection: nn.Linear, seq: torch.Tensor, p: float = 0.0,
                      want_weights: bool = True) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """self-MHA over T -> mean(T) -> Linear  (reference :152-160 / :236-244).
    Returns (projected (B, d_fusion) fp32, attended (B, T, hidden) fp32, head-averaged weights (B, T, T) fp32 | None)."""
    B, T, hdim = seq.shape
    rows = _as_rows(seq)
    qkv = ops.linear(rows, mha.qkv_spec().w, mha.qkv_spec().b)
    weights = None
    if want_weights and ops.fp32_mode():
        # fp32 parity mode: qkv holds f32 rows, which the bf16 kernel below must not read — the head-averaged
        # softmax(q k^T / sqrt(dh)) as f32 torch arithmetic on the GPU (no gradient, like the kernel)
        with torch.no_grad():
            H, dh = mha.num_heads, mha.head_dim
            q = qkv[:, :hdim].reshape(B, T, H, dh).transpose(1, 2)
            k = qkv[:, hdim:2 * hdim].reshape(B, T, H, dh).transpose(1, 2)
            weights = torch.softmax((q * (dh ** -0.5)) @ k.transpose(-1, -2), dim=-1).mean(dim=1)
    elif want_weights and T <= 2048 and mha.head_dim % 8 == 0:
        from mmfusion import lib as _lib
        weights = torch.empty((B, T, T), dtype=torch.float32, device=qkv.device)
        _lib.check(_lib.load().mmf_attn_weights_mean(qkv.data_ptr(), weights.data_ptr(), B, T, mha.num_heads, mha.head_dim,
                                                     _lib.stream_ptr()))
    att = ops.attention_group([AttnSpec(B, T, T, q=(0, 0), k=(0, hdim), v=(0, 2 * hdim))],
                              mha.num_heads, mha.head_dim, [qkv], dropout_p=p)[0]
    attended = ops.linear(att, *_wb(mha.out_proj))
    pooled = ops.meanpool_cat([attended.view(B, T, hdim)])
    projected = ops.linear(pooled, *_wb(projection), out_f32=True, dropout_p=p)              # reference :161 / :245
    return projected, ops.to_f32(attended).view(B, T, hdim), weights


class TextEncoder(_FusionBase):
    def __init__(self, config, backbone: Optional[nn.Module] = None):
        super().__init__()
        self.config = config
        if _feature_mode(config):
            self.model, self.hidden_size = None, config.text_hidden_size
        else:
            self.model = backbone if backbone is not None else _load_backbone("text", config.text_model_name)
            self.hidden_size = self.model.config.hidden_size
        self.adapter = AdapterLayer(self.hidden_size, config.adapter_size) if hasattr(config, "adapter_size") else None
        self.prompt_embeddings = nn.Parameter(torch.randn(config.prompt_length, self.hidden_size)) \
            if hasattr(config, "prompt_length") else None
        self.projection = nn.Linear(self.hidden_size, config.fusion_hidden_size)
        self.dropout = nn.Dropout(config.fusion_dropout)

    def forward(self, input_ids, attention_mask, use_adapter: bool = False, use_prompt: bool = False
                ) -> Dict[str, torch.Tensor]:
        cls_pool = True
        if self.model is None:                       # feature mode: input_ids holds (B, T, hidden)
            sequence_output = input_ids
        else:
            B = input_ids.size(0)
            if use_prompt and self.prompt_embeddings is not None:        # reference :49-71
                pe = self.prompt_embeddings.unsqueeze(0).expand(B, -1, -1)
                emb = self.model.embeddings.word_embeddings(input_ids)
                pm = torch.ones(B, self.config.prompt_length, device=attention_mask.device, dtype=attention_mask.dtype)
                attention_mask = torch.cat([pm, attention_mask], dim=1)
                outputs = self.model(inputs_embeds=torch.cat([pe, emb], dim=1), attention_mask=attention_mask)
            else:
                outputs = self.model(input_ids=input_ids, attention_mask=attention_mask)
            sequence_output = outputs.last_hidden_state
            cls_pool = "bert" in getattr(self.model.config, "model_type", "")     # reference :87
        if use_adapter and self.adapter is not None:
            sequence_output = self.adapter(sequence_output)
        if cls_pool:
            pooled = sequence_output[:, 0]
        else:                                                              # reference :90-94
            m = attention_mask.unsqueeze(-1).to(sequence_output.dtype)
            pooled = (sequence_output * m).sum(1) / m.sum(1).clamp_min(1e-9)
        projected = ops.linear(_as_rows(pooled), *_wb(self.projection), out_f32=True,
                               dropout_p=_p(self, self.config.fusion_dropout))               # reference :97-98
        return {"features": projected, "sequence_output": sequence_output, "attention_mask": attention_mask}


class AudioEncoder(_FusionBase):
    def __init__(self, config, backbone: Optional[nn.Module] = None):
        super().__init__()
        self.config = config
        if _feature_mode(config):
            self.model, self.hidden_size = None, config.audio_hidden_size
        else:
            self.model = backbone if backbone is not None else _load_backbone("audio", config.audio_model_name)
            self.hidden_size = self.model.config.hidden_size
        self.adapter = AdapterLayer(self.hidden_size, config.adapter_size) if hasattr(config, "adapter_size") else None
        self.temporal_attention = _MHAParams(self.hidden_size, 8)
        self.projection = nn.Linear(self.hidden_size, config.fusion_hidden_size)
        self.dropout = nn.Dropout(config.fusion_dropout)

    def forward(self, waveform, use_adapter: bool = False) -> Dict[str, torch.Tensor]:
        seq = waveform if self.model is None else self.model(waveform).last_hidden_state
        if use_adapter and self.adapter is not None:
            seq = self.adapter(seq)
        projected, attended, weights = _mha_mean_project(self.temporal_attention, self.projection, seq,
                                                         _p(self, self.config.fusion_dropout),
                                                         getattr(self.config, "encoder_attention_weights", True))
        return {"features": projected, "sequence_output": attended, "attention_weights": weights}


class VideoEncoder(_FusionBase):
    def __init__(self, config, backbone: Optional[nn.Module] = None):
        super().__init__()
        self.config = config
        if _feature_mode(config):
            self.vit, self.hidden_size = None, config.video_hidden_size
        else:
            self.vit = backbone if backbone is not None else _load_backbone("video", config.video_model_name)
            self.hidden_size = self.vit.config.hidden_size
        self.temporal_lstm = nn.LSTM(self.hidden_size, self.hidden_size // 2, num_layers=2, batch_first=True,
                                     bidirectional=True, dropout=config.fusion_dropout)
        self.facial_attention = _MHAParams(self.hidden_size, 8)
        self.adapter = AdapterLayer(self.hidden_size, config.adapter_size) if hasattr(config, "adapter_size") else None
        self.projection = nn.Linear(self.hidden_size, config.fusion_hidden_size)
        self.dropout = nn.Dropout(config.fusion_dropout)

    def forward(self, video_frames, use_adapter: bool = False) -> Dict[str, torch.Tensor]:
        if self.vit is None:                         # feature mode: (B, frames, hidden) CLS features
            frame_features = video_frames
        else:
            B, n, c, h, w = video_frames.shape
            cls = self.vit(pixel_values=video_frames.view(-1, c, h, w)).last_hidden_state[:, 0]
            frame_features = cls.view(B, n, -1)
        if use_adapter and self.adapter is not None:
            frame_features = self.adapter(frame_features)
        from mmfusion import lstm_ops
        lstm_out = lstm_ops.bilstm(self.temporal_lstm, frame_features, _p(self, self.config.fusion_dropout))   # :233
        projected, attended, weights = _mha_mean_project(self.facial_attention, self.projection, lstm_out,
                                                         _p(self, self.config.fusion_dropout),
                                                         getattr(self.config, "encoder_attention_weights", True))
        return {"features": projected, "sequence_output": attended, "attention_weights": weights}


class ModalityDropout(nn.Module):
    """Per-sample Bernoulli keep-masks, p = dropout_rate, **no** 1/(1-p) rescale, at least one
    modality kept (reference :289-321).  On the GPU (2-D features) the masks come from the build's counter-based RNG and are
    drawn and applied by one kernel; the torch formulation serves CPU tensors, 3-D features and the fp32 parity mode."""

    def __init__(self, dropout_rate: float = 0.1):
        super().__init__()
        self.dropout_rate = dropout_rate

    def forward(self, text_features, audio_features, video_features, training: bool = True):
        if not training:
            return text_features, audio_features, video_features
        B, dev = text_features.size(0), text_features.device
        if (text_features.is_cuda and text_features.dim() == 2 and text_features.shape[1] % 4 == 0
                and text_features.shape == audio_features.shape == video_features.shape and not ops.fp32_mode()):
            # masks drawn and applied by ONE kernel from the build's counter-based RNG (csrc/loss.hip modality_dropout_kernel:
            # Bernoulli(1 - p) per (sample, modality), a sample with nothing left gets one modality back) — the torch formulation
            # below is ~16 launches forward and 6 backward for 3 x B numbers
            from mmfusion import small_ops as sops
            t, a, v, _ = sops.modality_dropout(text_features, audio_features, video_features, self.dropout_rate)
            return t, a, v
        keep = torch.rand(B, 3, device=dev) > self.dropout_rate
        # samples that lost all three modalities get one back at random (:308-314) — written without the
        # reference's data-dependent branch so that nothing synchronises with the host (hipGraph-capturable)
        choice = torch.nn.functional.one_hot(torch.randint(0, 3, (B,), device=dev), 3).bool()
        keep = torch.where(keep.any(dim=1, keepdim=True), keep, choice)
        k = keep.float()
        return text_features * k[:, 0:1], audio_features * k[:, 1:2], video_features * k[:, 2:3]
