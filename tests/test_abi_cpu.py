"""CPU: the C-ABI library builds/loads and exports every symbol include/mmfusion.h declares; the
host-side mirror keeps the reference's state_dict surface; the product path refuses CPU tensors."""
import os
import re

import pytest
import torch

import config as cfgmod
from mmfusion import lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_match_loader_table():
    hdr = open(os.path.join(REPO, "include", "mmfusion.h")).read()
    declared = set(re.findall(r"\b(mmf_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)


def test_library_loads_and_exports_every_symbol():
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = lib.load()                      # getattr()s every symbol; no compute call without a GPU
    assert L.mmf_version() == 1


def test_struct_sizes_match_c_layout():
    import ctypes
    assert ctypes.sizeof(lib.GemmProblem) == 72
    assert ctypes.sizeof(lib.AttnProblem) == 112
    assert ctypes.sizeof(lib.LnProblem) == 88


def test_state_dict_surface_matches_reference_fixtures():
    """every fixture lists the reference module's own state_dict keys and shapes"""
    from golden_cases import CASES
    from helpers import cfg_from_meta, load_fixture
    from models import fusion_layers as fl, encoders as enc
    for name in CASES:
        meta = load_fixture(name).meta
        m = enc.AdapterLayer(*meta["ctor"]) if meta["cls"] == "AdapterLayer" else getattr(fl, meta["cls"])(cfg_from_meta(meta))
        mine = {k: list(v.shape) for k, v in m.state_dict().items()}
        assert mine == {k: s for k, s in meta["shapes"]}, name


def test_default_config_surface():
    c = cfgmod.ModelConfig()
    assert (c.fusion_hidden_size, c.fusion_dropout, c.fusion_num_heads, c.num_emotions) == (512, 0.1, 8, 7)
    assert (c.graph_hidden_size, c.graph_num_layers, c.contrastive_temperature) == (256, 3, 0.07)
    c.fusion_type = "mult"              # dynamic attribute, as train_advanced.py:118 does
    assert c.emotion_labels[0] == "happy" and len(c.emotion_labels) == 7


def test_product_path_refuses_cpu_tensors():
    from models import fusion_layers as fl
    c = cfgmod.ModelConfig()
    c.fusion_hidden_size, c.fusion_dropout = 64, 0.0
    m = fl.EarlyFusion(c)
    with pytest.raises(RuntimeError, match="GPU only"):
        m(torch.randn(2, 64), torch.randn(2, 64), torch.randn(2, 64))


def test_unknown_fusion_type_raises_value_error():
    from models import multimodal_model as mm
    c = cfgmod.ModelConfig()
    c.feature_inputs, c.fusion_type = True, "nope"
    with pytest.raises(ValueError, match="Unknown fusion type"):
        mm.MultimodalEmotionModel(c)
    with pytest.raises(ValueError, match="Unknown model type"):
        mm.create_model(c, "nope")
