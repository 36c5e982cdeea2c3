"""GPU: the stream-K form of the 256 x 256 GEMM (csrc/gemm4.hip, round 3) against an fp32 torch product of the same bf16
operands, on the MulT launch groups whose tile counts do not fill their CU rounds (out-projection 177 tiles, FFN1 708, the
in-projection group, the dgrads) and on ragged shapes (M, N, K not multiples of the tile), with every epilogue the NT / NN
launches use.  bf16 outputs: relative error 2^-7 of the output's scale; the partial tiles meet in ascending order, so two
runs must agree bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from mmfusion import lib, ops  # noqa: E402
from mmfusion.lib import EPI_ADD_AUX, EPI_BIAS, EPI_MASK_AUX, EPI_RELU, GEMM_NN, GEMM_NT  # noqa: E402

CASES = [
    ("out-proj NT", GEMM_NT, [(8192, 768, 768), (6400, 768, 768), (480, 768, 768)], EPI_BIAS | EPI_ADD_AUX),
    ("ffn1 NT", GEMM_NT, [(8192, 3072, 768), (6400, 3072, 768), (480, 3072, 768)], EPI_BIAS | EPI_RELU),
    ("ffn2 NT", GEMM_NT, [(8192, 768, 3072), (8192, 768, 3072)], EPI_BIAS),
    ("in-proj NT", GEMM_NT, [(8192, 768, 768), (480, 1536, 768), (6400, 768, 768), (8192, 1536, 768), (480, 768, 768),
                             (6400, 1536, 768)], EPI_BIAS),
    ("dH NN", GEMM_NN, [(8192, 3072, 768), (6400, 3072, 768), (480, 3072, 768)], EPI_MASK_AUX),
    ("dX NN", GEMM_NN, [(8192, 768, 3072), (6400, 768, 3072)], EPI_ADD_AUX),
    ("ragged NT", GEMM_NT, [(3000, 1032, 520), (2500, 520, 200), (1700, 264, 72)], EPI_BIAS),
    ("ragged NN", GEMM_NN, [(2900, 264, 520), (3300, 1032, 136)], 0),
    ("one tile, long K", GEMM_NT, [(256, 256, 4096)], 0),
]


def _run(layout, shapes, epi, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    probs, refs = [], []
    for (M, N, K) in shapes:
        A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        Bm = (torch.randn(N, K, device="cuda", generator=g) if layout == GEMM_NT
              else torch.randn(K, N, device="cuda", generator=g)).bfloat16()
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        bias = torch.randn(N, device="cuda", generator=g) if epi & EPI_BIAS else None
        aux = torch.randn(M, N, device="cuda", generator=g).bfloat16() if epi & (EPI_ADD_AUX | EPI_MASK_AUX) else None
        probs.append((A, Bm, C, bias, aux))
        r = A.float() @ (Bm.float().t() if layout == GEMM_NT else Bm.float())
        if bias is not None:
            r = r + bias
        if epi & EPI_RELU:
            r = r.relu()
        if epi & EPI_MASK_AUX:
            r = r * (aux.float() > 0)
        if epi & EPI_ADD_AUX:
            r = r + aux.float()
        refs.append(r)
    return probs, refs


@pytest.mark.parametrize("name,layout,shapes,epi", CASES, ids=[c[0] for c in CASES])
def test_streamk_matches_fp32_product_and_is_reproducible(name, layout, shapes, epi):
    probs, refs = _run(layout, shapes, epi, seed=5)
    lib.check(lib.load().mmf_gemm_select_impl(4))      # (the automatic rule sends launches of < CUs / 4 tiles elsewhere)
    try:
        _check(name, layout, probs, refs, epi)
    finally:
        lib.check(lib.load().mmf_gemm_select_impl(0))


def _check(name, layout, probs, refs, epi):
    ops.gemm_group(layout, probs, epi)
    torch.cuda.synchronize()
    for (A, Bm, C, _, _), r in zip(probs, refs):
        err = float((C.float() - r).abs().max()) / max(1e-6, float(r.abs().max()))
        assert err <= 2 ** -7, f"{name}: max abs err / max |ref| = {err:.3e}"
    first = [p[2].clone() for p in probs]
    for _ in range(3):                       # the arrival order of the partial tiles changes, the sums must not
        ops.gemm_group(layout, probs, epi)
    torch.cuda.synchronize()
    assert all(torch.equal(a, p[2]) for a, p in zip(first, probs)), f"{name}: not bit-reproducible"


def test_streamk_equals_data_parallel_up_to_f32_order(monkeypatch):
    """The same launch with and without the workspace: identical up to the f32 summation order of the split tiles."""
    layout, shapes, epi = GEMM_NT, [(8192, 768, 768), (6400, 768, 768), (480, 768, 768)], EPI_BIAS | EPI_ADD_AUX
    probs, _ = _run(layout, shapes, epi, seed=9)
    ops.gemm_group(layout, probs, epi)
    sk = [p[2].clone() for p in probs]
    monkeypatch.setenv("MMF_GEMM_STREAMK", "0")
    ops.gemm_group(layout, probs, epi)
    torch.cuda.synchronize()
    for a, p in zip(sk, probs):
        d = float((a.float() - p[2].float()).abs().max()) / float(p[2].float().abs().max())
        assert d <= 2 ** -7
