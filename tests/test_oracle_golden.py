"""CPU: oracle/ref_cpu.py against the committed golden vectors captured from the reference's
own classes (tools/capture_golden.py).  This is what pins the oracle."""
import pytest
import torch

from golden_cases import CASES
from helpers import load_fixture, oracle_fwd_bwd, rel_err
from mmfusion import synth

TOL = 2e-5


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference_vectors(name):
    fx = load_fixture(name)
    out, gin, gp = oracle_fwd_bwd(fx.meta)
    assert set(out) == set(fx.out)
    for k, v in fx.out.items():
        assert out[k].shape == v.shape, k
        assert rel_err(out[k], v) <= TOL, f"output {k}"
    for i, g in enumerate(fx.gin):
        assert rel_err(gin[i], g) <= TOL, f"input grad {i}"
    for k, g in fx.gsmall.items():
        assert rel_err(gp[k], g) <= TOL, f"param grad {k}"
    for k, (norm, dot) in fx.meta["grad_checks"].items():
        g = gp[k]
        assert abs(float(g.norm()) - norm) <= TOL * max(1.0, norm), f"grad norm {k}"
        d = float((g.flatten() * synth.probe_vector("g:" + k, g.numel())).sum())
        assert abs(d - dot) <= 1e-4 * max(1.0, norm), f"grad probe {k}"
    for k in fx.meta["no_grad_params"]:
        assert float(gp[k].abs().max()) == 0.0


def test_gat_cases_are_flagged_unpinned():
    assert load_fixture("graph").meta["gat_unpinned"] and load_fixture("hier_ref").meta["gat_unpinned"]
    assert not load_fixture("mult_seq_dh96").meta["gat_unpinned"]


def test_softmax_rows_sum_to_one_and_t1_attention_is_query_independent():
    # with T=1 the attention output does not depend on the query
    fx = load_fixture("mult_2d")
    from helpers import fixture_params, fixture_inputs, run_oracle
    P, x = fixture_params(fx.meta), fixture_inputs(fx.meta)
    a = run_oracle(fx.meta, P, x)["fused_features"]
    assert torch.isfinite(a).all()
    from oracle import ref_cpu
    q1, q2 = torch.randn(2, 1, 192), torch.randn(2, 1, 192)
    kv = torch.randn(2, 1, 192)
    o1, w1 = ref_cpu.mha(P, "text_self_attn.", q1, kv, 2)
    o2, _ = ref_cpu.mha(P, "text_self_attn.", q2, kv, 2)
    assert torch.allclose(o1, o2) and torch.allclose(w1, torch.ones_like(w1))
