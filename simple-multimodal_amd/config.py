"""Configuration surface of the fusion path (drop-in for the reference's ``config.py``).

Field names and defaults follow reference ``config.py:6-140`` one for one, because
``train_advanced.py:21-25`` and every ``models.*`` constructor read them by name.  Only the
fields listed in SURVEY.md section 5 are read by the MI355X fusion path
(``fusion_hidden_size``, ``fusion_dropout``, ``fusion_num_heads``, ``num_emotions``,
``graph_*``, ``contrastive_temperature``, ``adapter_size``, ``prompt_length``); the rest are
carried so that callers and saved-config JSON keep working.

Differences from the reference, on purpose:
  * directory creation (reference ``config.py:77-79``) can be switched off with
    ``MMFUSION_CONFIG_MKDIRS=0`` (tests do) and never raises;
  * attributes the callers bolt on dynamically (``fusion_type`` ``train_advanced.py:118``,
    ``use_wandb`` ``:433``, ``patience``) are still accepted because these are plain dataclasses.
"""
import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

_EMOTIONS = ("happy", "sad", "angry", "fear", "surprise", "disgust", "neutral")


def _mkdirs_enabled() -> bool:
    return os.environ.get("MMFUSION_CONFIG_MKDIRS", "1") not in ("0", "false", "no")


@dataclass
class ModelConfig:
    # --- encoders (backbone names are kept; the fusion path never fetches them) ---
    text_model_name: str = "microsoft/deberta-v3-base"
    text_hidden_size: int = 768
    text_max_length: int = 512
    audio_model_name: str = "facebook/wav2vec2-base-960h"
    audio_hidden_size: int = 768
    audio_sample_rate: int = 16000
    audio_max_length: int = 160000
    video_model_name: str = "google/vit-base-patch16-224"
    video_hidden_size: int = 768
    video_frame_size: Tuple[int, int] = (224, 224)
    video_max_frames: int = 30
    # --- fusion ---
    fusion_hidden_size: int = 512
    fusion_dropout: float = 0.1
    fusion_num_heads: int = 8
    fusion_num_layers: int = 4
    # --- classification ---
    num_emotions: int = 7
    emotion_labels: List[str] = None
    # --- graph fusion ---
    graph_hidden_size: int = 256
    graph_num_layers: int = 3
    graph_dropout: float = 0.1
    # --- contrastive ---
    contrastive_temperature: float = 0.07
    contrastive_margin: float = 0.5
    # --- few-shot ---
    adapter_size: int = 64
    prompt_length: int = 10
    # --- distillation ---
    distill_temperature: float = 4.0
    distill_alpha: float = 0.7
    # --- optimisation ---
    batch_size: int = 8
    learning_rate: float = 1e-4
    weight_decay: float = 1e-5
    num_epochs: int = 100
    warmup_steps: int = 1000
    gradient_clip_norm: float = 1.0
    # --- paths ---
    data_path: str = "./data"
    save_path: str = "./checkpoints"
    log_path: str = "./logs"
    # --- device ---
    device: str = "auto"
    mixed_precision: bool = True

    def __post_init__(self):
        if self.emotion_labels is None:
            self.emotion_labels = list(_EMOTIONS)
        if _mkdirs_enabled():
            for p in (self.data_path, self.save_path, self.log_path):
                try:
                    os.makedirs(p, exist_ok=True)
                except OSError:
                    pass


@dataclass
class DataConfig:
    primary_dataset: str = "sample"
    supplementary_datasets: List[str] = None
    normalize_audio: bool = True
    augment_data: bool = True
    balance_classes: bool = True
    k_folds: int = 5
    test_split: float = 0.2
    val_split: float = 0.1
    num_workers: int = 0
    pin_memory: bool = True

    def __post_init__(self):
        if self.supplementary_datasets is None:
            self.supplementary_datasets = ["meld"]


@dataclass
class ExperimentConfig:
    enable_early_fusion: bool = True
    enable_late_fusion: bool = True
    enable_mult_fusion: bool = True
    enable_graph_fusion: bool = True
    enable_contrastive_learning: bool = True
    enable_prompt_tuning: bool = True
    enable_adapter_tuning: bool = True
    few_shot_samples: List[int] = None
    test_missing_modalities: bool = True
    missing_modality_rates: List[float] = None
    enable_knowledge_distillation: bool = True
    teacher_model_path: Optional[str] = None

    def __post_init__(self):
        if self.few_shot_samples is None:
            self.few_shot_samples = [1, 5, 10, 20, 50]
        if self.missing_modality_rates is None:
            self.missing_modality_rates = [0.1, 0.3, 0.5, 0.7]


# module-level singletons, as the reference exposes them (config.py:144-146)
model_config = ModelConfig()
data_config = DataConfig()
experiment_config = ExperimentConfig()
