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


def test_numpy_scalar_metrics_load_and_other_classes_stay_refused(tmp_path):
    """The reference stores sklearn metrics in ``metrics`` (advanced_trainer.py:245-261,400-407): f1_score /
    accuracy_score return numpy.float64, pickled as numpy's scalar reconstructor + dtype.  The safe loader admits
    exactly those (data, no code) and still refuses a file that needs any other class."""
    import numpy as np
    torch.manual_seed(0)
    src = MultimodalEmotionModel(_cfg())
    path = str(tmp_path / "best_model.pth")
    torch.save({"epoch": 1, "model_state_dict": src.state_dict(), "optimizer_state_dict": {}, "scheduler_state_dict": {},
                "metrics": {"val_f1_macro": np.float64(0.5), "val_f1_weighted": np.float64(0.25), "val_accuracy": np.float64(0.75),
                            "n": np.int64(7)},
                "config": cfgmod.ExperimentConfig()}, path)
    ckpt = load_checkpoint_file(path)
    assert float(ckpt["metrics"]["val_f1_macro"]) == 0.5 and int(ckpt["metrics"]["n"]) == 7
    dst = load_pretrained_model(path, _cfg())
    assert all(torch.equal(v, dst.state_dict()[k]) for k, v in src.state_dict().items())

    import fractions
    bad = str(tmp_path / "bad.pth")
    torch.save({"model_state_dict": src.state_dict(), "metrics": {"f": fractions.Fraction(1, 2)}}, bad)
    with pytest.raises(pickle.UnpicklingError):
        load_checkpoint_file(bad)


def test_pyg_23_gat_keys_are_remapped_on_load():
    """torch-geometric 2.3 / 2.4 ``GATConv`` checkpoints carry ``lin_src.weight`` and ``lin_dst.weight`` (one shared
    tensor under two names), >= 2.5 a single ``lin.weight``: both spellings load into the build's
    ``gcn_layers.N.lin.weight`` and nothing else is tolerated as missing or unexpected."""
    from models import fusion_layers as fl
    cfg = _cfg()
    torch.manual_seed(0)
    src = fl.GraphFusion(cfg)
    new_sd = src.state_dict()
    old_sd = {}
    for k, v in new_sd.items():
        if k.endswith(".lin.weight"):
            old_sd[k.replace(".lin.weight", ".lin_src.weight")] = v.clone()
            old_sd[k.replace(".lin.weight", ".lin_dst.weight")] = v.clone()
        else:
            old_sd[k] = v.clone()
    assert any("lin_src" in k for k in old_sd) and not any(k.endswith(".lin.weight") for k in old_sd)
    torch.manual_seed(1)
    dst = fl.GraphFusion(cfg)
    res = dst.load_state_dict(old_sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in new_sd.items():
        assert torch.equal(v, dst.state_dict()[k]), k
    with pytest.raises(RuntimeError):
        bad = dict(old_sd)
        bad["gcn_layers.0.lin_other.weight"] = torch.zeros(1)
        fl.GraphFusion(cfg).load_state_dict(bad, strict=True)
