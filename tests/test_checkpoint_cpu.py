"""CPU: checkpoint interchange.

The reference writes {'epoch', 'model_state_dict', 'optimizer_state_dict', 'scheduler_state_dict', 'metrics',
'config'} with its ExperimentConfig instance pickled inside (training/advanced_trainer.py:396-411) and reads it back
with a plain ``torch.load`` (models/multimodal_model.py:472-485).  Here such files are read with ``weights_only=True``
and an allowlist of this package's own config dataclasses; a file written by the reference cannot be produced in
this container (its trainer needs wandb / seaborn and its model downloads the backbones), so the file under test is
written in exactly that layout with ``torch.save`` from the build's own modules — whose ``state_dict`` keys and
shapes are pinned against the reference's classes by tests/test_oracle_golden.py."""
import pickle

import pytest
import torch

import config as cfgmod
from models.multimodal_model import MultimodalEmotionModel, load_checkpoint_file, load_pretrained_model


def _cfg():
    cfg = cfgmod.ModelConfig()
    cfg.feature_inputs = True
    cfg.fusion_type = "hierarchical"
    cfg.fusion_hidden_size, cfg.fusion_num_heads = 64, 8
    cfg.graph_hidden_size, cfg.graph_num_layers = 64, 2
    return cfg


def _reference_layout_checkpoint(path, model):
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-5)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=100, pct_start=0.1)
    torch.save({"epoch": 3, "model_state_dict": model.state_dict(), "optimizer_state_dict": opt.state_dict(),
                "scheduler_state_dict": sch.state_dict(), "metrics": {"accuracy": 0.5, "f1_weighted": 0.4},
                "config": cfgmod.ExperimentConfig()}, path)


def test_reference_layout_checkpoint_loads_with_safe_loader(tmp_path):
    torch.manual_seed(0)
    src = MultimodalEmotionModel(_cfg())
    path = str(tmp_path / "best_model.pth")
    _reference_layout_checkpoint(path, src)
    with pytest.raises(pickle.UnpicklingError):                 # the pickled config: a bare safe load refuses the file
        torch.load(path, map_location="cpu", weights_only=True)
    ckpt = load_checkpoint_file(path)
    assert set(ckpt) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "metrics", "config"}
    assert isinstance(ckpt["config"], cfgmod.ExperimentConfig) and ckpt["epoch"] == 3
    torch.manual_seed(1)
    dst = load_pretrained_model(path, _cfg())
    a, b = src.state_dict(), dst.state_dict()
    assert list(a) == list(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_bare_state_dict_checkpoint(tmp_path):
    torch.manual_seed(0)
    src = MultimodalEmotionModel(_cfg())
    path = str(tmp_path / "weights.pth")
    torch.save(src.state_dict(), path)
    dst = load_pretrained_model(path, _cfg())
    for k, v in src.state_dict().items():
        assert torch.equal(v, dst.state_dict()[k]), k


class _NotAllowed:
    def __init__(self):
        self.x = 1


def test_foreign_class_in_checkpoint_is_refused(tmp_path):
    path = str(tmp_path / "bad.pth")
    torch.save({"model_state_dict": {}, "config": _NotAllowed()}, path)
    with pytest.raises(pickle.UnpicklingError):
        load_checkpoint_file(path)
