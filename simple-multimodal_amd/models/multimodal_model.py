"""``MultimodalEmotionModel`` — drop-in for the reference's ``models/multimodal_model.py:12-219``.

Kept verbatim: constructor ``(config)``, ``forward(text_input, audio_input, video_input,
use_adapter, use_prompt, compute_contrastive_loss, missing_modalities) -> Dict`` and its output
keys (reference :159-181), ``fusion_type`` dispatch incl. the ``ValueError`` (:29-46),
``EmotionClassifier`` (:186-219), ``create_model`` / ``load_pretrained_model`` (:453-485) and all
``state_dict`` names.  The fusion layer, the encoder tails and the heads are the MI355X HIP path: the
classifier's d -> d/2 layer on the skinny MFMA kernel, the 7-class / 1-unit output layers (classifier, valence,
arousal, uncertainty) on the narrow-linear kernel (f32 masters); the softmaxes over 7 logits are torch glue.

Research wrappers (``KnowledgeDistillationModel``, ``FewShotModel``, ``RobustMultimodalModel``,
reference :222-450) are outside the hot path: the names exist so that
``train_advanced.py:21-25`` imports, constructing them raises ``NotImplementedError``.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from mmfusion import ops
from mmfusion import small_ops as sops
from mmfusion.ops import W

from .encoders import AudioEncoder, ModalityDropout, TextEncoder, VideoEncoder
from .fusion_layers import (AdaptiveFusion, ContrastiveFusion, EarlyFusion, GraphFusion,
                            HierarchicalFusion, LateFusion, MultimodalTransformer, _FusionBase)

_FUSIONS = {"early": EarlyFusion, "late": LateFusion, "mult": MultimodalTransformer, "graph": GraphFusion,
            "contrastive": ContrastiveFusion, "adaptive": AdaptiveFusion, "hierarchical": HierarchicalFusion}


class EmotionClassifier(nn.Module):
    """d -> d/2 -> num_emotions main head; the three hierarchical heads exist for state_dict
    compatibility (the reference computes and discards them, :204-219)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        d = config.fusion_hidden_size
        self.classifier = nn.Sequential(nn.Linear(d, d // 2), nn.ReLU(), nn.Dropout(config.fusion_dropout),
                                        nn.Linear(d // 2, config.num_emotions))
        self.sentiment_classifier = nn.Linear(d, 3)
        self.positive_classifier = nn.Linear(d, 2)
        self.negative_classifier = nn.Linear(d, 4)

    def forward(self, features: torch.Tensor) -> torch.Tensor:
        """d -> d/2 (ReLU, dropout) on the skinny MFMA kernel, d/2 -> num_emotions on the narrow-linear kernel."""
        if not features.is_cuda:
            raise RuntimeError("mmfusion: the classifier head runs on the GPU only (no CPU fallback)")
        from mmfusion import arena as _arena_mod
        if getattr(self.classifier[0].weight, "_mmf_arena", None) is None:
            _arena_mod.ensure(self)                # stand-alone use; inside MultimodalEmotionModel the root did it
        p = float(self.config.fusion_dropout) if self.training else 0.0
        x = features.float().contiguous()                  # f32 rows: narrowed by the linear itself (mmfusion.ops._RowLinear)
        if ops.fp32_mode():
            x = ops.to_bf16(x)
        l0 = self.classifier[0]
        h = ops.linear(x, W(l0.weight), W(l0.bias), relu=True, out_f32=True, dropout_p=p)
        return sops.narrow_linear(h, self.classifier[3])


class MultimodalEmotionModel(_FusionBase):
    def __init__(self, config, backbones: Optional[Dict[str, nn.Module]] = None):
        super().__init__()
        self.config = config
        bb = backbones or {}
        self.text_encoder = TextEncoder(config, bb.get("text"))
        self.audio_encoder = AudioEncoder(config, bb.get("audio"))
        self.video_encoder = VideoEncoder(config, bb.get("video"))
        self.modality_dropout = ModalityDropout(dropout_rate=0.1)
        self.fusion_type = getattr(config, "fusion_type", "hierarchical")
        if self.fusion_type not in _FUSIONS:
            raise ValueError(f"Unknown fusion type: {self.fusion_type}")
        self.fusion_layer = _FUSIONS[self.fusion_type](config)
        self.classifier = None if self.fusion_type == "late" else EmotionClassifier(config)
        d = config.fusion_hidden_size
        self.valence_regressor = nn.Linear(d, 1)
        self.arousal_regressor = nn.Linear(d, 1)
        self.uncertainty_head = nn.Linear(d, config.num_emotions)

    def forward(self, text_input: Dict[str, torch.Tensor], audio_input: torch.Tensor, video_input: torch.Tensor,
                use_adapter: bool = False, use_prompt: bool = False, compute_contrastive_loss: bool = False,
                missing_modalities: Optional[List[str]] = None) -> Dict[str, torch.Tensor]:
        if missing_modalities:                                                   # reference :77-86
            if "text" in missing_modalities:
                text_input = {"input_ids": torch.zeros_like(text_input["input_ids"]),
                              "attention_mask": torch.zeros_like(text_input["attention_mask"])}
            if "audio" in missing_modalities:
                audio_input = torch.zeros_like(audio_input)
            if "video" in missing_modalities:
                video_input = torch.zeros_like(video_input)
        tf = self.text_encoder(text_input["input_ids"], text_input["attention_mask"],
                               use_adapter=use_adapter, use_prompt=use_prompt)["features"]
        af = self.audio_encoder(audio_input, use_adapter=use_adapter)["features"]
        vf = self.video_encoder(video_input, use_adapter=use_adapter)["features"]
        if self.training:                                                        # reference :104-107
            tf, af, vf = self.modality_dropout(tf, af, vf, training=True)

        individual_logits = fusion_weights = None
        if self.fusion_type == "late":
            fo = self.fusion_layer(tf, af, vf)
            emotion_logits = fo["fused_logits"]
            individual_logits = {"text": fo["text_logits"], "audio": fo["audio_logits"], "video": fo["video_logits"]}
            fusion_weights = fo["fusion_weights"]
            head_in = (tf + af + vf) / 3                                         # reference :153
        else:
            if self.fusion_type in ("contrastive", "hierarchical"):
                fo = self.fusion_layer(tf, af, vf, compute_contrastive_loss=compute_contrastive_loss)
            else:
                fo = self.fusion_layer(tf, af, vf)
            head_in = fo["fused_features"] if isinstance(fo, dict) else fo
            emotion_logits = self.classifier(head_in)
        out = {"emotion_logits": emotion_logits, "emotion_probs": F.softmax(emotion_logits, dim=-1),
               "valence": sops.narrow_linear(head_in, self.valence_regressor),
               "arousal": sops.narrow_linear(head_in, self.arousal_regressor),
               "uncertainty": F.softmax(sops.narrow_linear(head_in, self.uncertainty_head), dim=-1),
               "text_features": tf, "audio_features": af, "video_features": vf}
        if self.fusion_type == "late":
            out.update({"individual_logits": individual_logits, "fusion_weights": fusion_weights})
        if isinstance(fo, dict):                                                 # reference :177-181
            for k, v in fo.items():
                if k != "fused_features":
                    out[k] = v
        return out


def _out_of_scope(name: str):
    class _Stub(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            raise NotImplementedError(
                f"{name} is a research wrapper around the fusion path (reference models/multimodal_model.py) "
                f"and is outside the MI355X hot-path scope (SURVEY.md section 2 row 5).")
    _Stub.__name__ = _Stub.__qualname__ = name
    return _Stub


KnowledgeDistillationModel = _out_of_scope("KnowledgeDistillationModel")
FewShotModel = _out_of_scope("FewShotModel")
RobustMultimodalModel = _out_of_scope("RobustMultimodalModel")


def create_model(config, model_type: str = "standard") -> nn.Module:
    if model_type == "standard":
        return MultimodalEmotionModel(config)
    if model_type in ("few_shot", "robust", "distillation"):
        raise NotImplementedError(f"model_type '{model_type}' wraps the fusion path and is out of scope here")
    raise ValueError(f"Unknown model type: {model_type}")


def load_checkpoint_file(checkpoint_path: str) -> Dict:
    """Read a reference-format checkpoint (advanced_trainer.py:396-411: epoch, model_state_dict,
    optimizer_state_dict, scheduler_state_dict, metrics, config) without executing anything from the file:
    ``weights_only=True``, with this package's own ``config`` dataclasses as the only extra classes the unpickler
    may construct — the reference pickles its ``ExperimentConfig`` instance under the same module path
    (``config.ExperimentConfig``), which is why a bare ``weights_only=True`` refuses its files.
    The reference's ``metrics`` entry holds sklearn results (advanced_trainer.py:245-261: ``f1_score`` /
    ``accuracy_score`` return ``numpy.float64``), which pickle as ``numpy._core.multiarray.scalar`` + ``numpy.dtype``:
    those reconstructors (data only, no code) and the float / int dtype classes are allowed too.
    A file that needs any other class is refused with torch's error."""
    import config as _cfg
    import numpy as _np
    allow = [getattr(_cfg, n) for n in ("ModelConfig", "DataConfig", "ExperimentConfig") if hasattr(_cfg, n)]
    allow.append(_np.dtype)
    for modname in ("numpy._core.multiarray", "numpy.core.multiarray"):      # numpy >= 2 / numpy 1.x pickles
        try:
            mod = __import__(modname, fromlist=["scalar"])
            # torch matches allowed globals by the (module, name) the pickle names: register under both spellings
            allow.append((mod.scalar, f"{modname}.scalar"))
        except (ImportError, AttributeError):
            pass
    for tname in ("float64", "float32", "int64", "int32", "bool_"):
        allow.append(type(_np.dtype(getattr(_np, tname))))
    with torch.serialization.safe_globals(allow):
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    return ckpt if isinstance(ckpt, dict) and "model_state_dict" in ckpt else {"model_state_dict": ckpt}


def load_pretrained_model(checkpoint_path: str, config) -> MultimodalEmotionModel:
    """Reference ``load_pretrained_model`` (multimodal_model.py:472-485): {'model_state_dict': ...} or a bare
    state_dict, loaded through ``load_checkpoint_file`` (nothing in the file is executed)."""
    model = MultimodalEmotionModel(config)
    model.load_state_dict(load_checkpoint_file(checkpoint_path)["model_state_dict"])
    return model
