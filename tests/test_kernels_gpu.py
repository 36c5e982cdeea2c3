"""GPU: each HIP kernel of libmmfusion.so, called through the C ABI (ctypes), against an fp32
restatement of the same op on the bf16-rounded inputs.  Tolerances are written per test:
bf16 outputs carry a relative rounding error of 2^-9 per element; gradients that pass through a
bf16 intermediate are compared relative to the tensor's scale."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from mmfusion import lib, ops  # noqa: E402
from mmfusion.lib import (EPI_ACCUM, EPI_ADD_AUX, EPI_BIAS, EPI_COLSUM_A, EPI_MASK_AUX, EPI_RELU, GEMM_NN,
                          GEMM_NT, GEMM_TN)  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def bf(x):
    return x.to(torch.bfloat16).to(DEV)


def rel(a, b, floor=1e-3):
    """max |a-b| relative to the reference's scale (floored: an exactly-zero reference, e.g. dQ of a
    one-key softmax, is compared absolutely)."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), floor))


# ---------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 72, 40), (16, 768, 2304), (480, 1536, 768),
                                   (1000, 264, 3072), (3, 8, 8)])
def test_gemm_nt_epilogues(M, N, K):
    A, B_, bias, aux = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    a16, b16, aux16 = bf(A), bf(B_), bf(aux)
    ref = a16.float().cpu() @ b16.float().cpu().t()
    # plain, f32 out
    C = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(GEMM_NT, a16, b16, C)
    assert rel(C, ref) < 1e-5
    # bias + relu, bf16 out
    C16 = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    ops.gemm(GEMM_NT, a16, b16, C16, bias=bias.to(DEV), epilogue=EPI_BIAS | EPI_RELU)
    assert rel(C16, torch.relu(ref + bias)) < 2 ** -8
    # bias + residual
    ops.gemm(GEMM_NT, a16, b16, C16, bias=bias.to(DEV), aux=aux16, epilogue=EPI_BIAS | EPI_ADD_AUX)
    assert rel(C16, ref + bias + aux16.float().cpu()) < 2 ** -8
    # mask by aux > 0
    ops.gemm(GEMM_NT, a16, b16, C, aux=aux16, epilogue=EPI_MASK_AUX)
    assert rel(C, ref * (aux16.float().cpu() > 0)) < 1e-5
    # accumulate
    C.fill_(1.0)
    ops.gemm(GEMM_NT, a16, b16, C, epilogue=EPI_ACCUM)
    assert rel(C, ref + 1.0) < 1e-5


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 72, 40), (16, 2304, 768), (8192 // 8, 768, 3072),
                                   (30, 96, 192)])
def test_gemm_nn_dgrad(M, N, K):
    A, B_ = rnd(M, K, seed=5), rnd(K, N, seed=6, scale=K ** -0.5)
    a16, b16 = bf(A), bf(B_)
    C = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(GEMM_NN, a16, b16, C)
    assert rel(C, a16.float().cpu() @ b16.float().cpu()) < 1e-5


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (72, 200, 100), (768, 768, 1000), (3072, 96, 16),
                                   (64, 2304, 48), (8, 8, 3)])
def test_gemm_tn_wgrad(M, N, K):
    A, B_ = rnd(K, M, seed=7), rnd(K, N, seed=8, scale=K ** -0.5)
    a16, b16 = bf(A), bf(B_)
    C = torch.zeros((M, N), device=DEV)
    ops.gemm(GEMM_TN, a16, b16, C, epilogue=EPI_ACCUM)
    ops.gemm(GEMM_TN, a16, b16, C, epilogue=EPI_ACCUM)
    assert rel(C, 2 * (a16.float().cpu().t() @ b16.float().cpu())) < 1e-5


@pytest.mark.parametrize("M,N,K", [(768, 768, 1000), (200, 72, 100), (3072, 768, 30), (8, 2304, 515)])
def test_gemm_tn_fused_bias_grad(M, N, K):
    """wgrad + bias gradient in one launch: bias[m] += sum_k A[k][m]"""
    A, B_ = rnd(K, M, seed=7), rnd(K, N, seed=8, scale=K ** -0.5)
    a16, b16 = bf(A), bf(B_)
    C = torch.zeros((M, N), device=DEV)
    db = torch.full((M,), 2.0, device=DEV)
    ops.gemm(GEMM_TN, a16, b16, C, bias=db, epilogue=EPI_ACCUM | EPI_COLSUM_A)
    assert rel(C, a16.float().cpu().t() @ b16.float().cpu()) < 1e-5
    assert rel(db, a16.float().cpu().sum(0) + 2.0) < 1e-5


def test_gemm_identity_asymmetric():
    """A = I with an asymmetric B catches a transposed C write (guide section 3)."""
    n = 128
    eye = torch.eye(n)
    Bm = torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 100.0     # exact in bf16? values <= 151
    C = torch.empty((n, n), device=DEV)
    ops.gemm(GEMM_NT, bf(eye), bf(Bm), C)          # C = I B^T = B^T
    assert torch.equal(C.cpu(), Bm.t().contiguous())
    ops.gemm(GEMM_NN, bf(eye), bf(Bm), C)          # C = I B = B
    assert torch.equal(C.cpu(), Bm)
    ops.gemm(GEMM_TN, bf(eye), bf(Bm), C)          # C = I^T B = B
    assert torch.equal(C.cpu(), Bm)
    ops.gemm(GEMM_TN, bf(Bm), bf(eye), C)          # C = B^T I
    assert torch.equal(C.cpu(), Bm.t().contiguous())


def test_gemm_grouped_and_strided():
    """several problems in one launch, operands that are column slices of wider buffers"""
    big = bf(rnd(300, 512, seed=9))
    w1, w2 = bf(rnd(64, 128, seed=10, scale=0.1)), bf(rnd(200, 256, seed=11, scale=0.1))
    x1, x2 = big[:, 128:256], big[:100, 256:512]
    out = torch.zeros((300, 512), dtype=torch.bfloat16, device=DEV)
    y1, y2 = out[:, 0:64], out[:100, 64:264]
    ops.gemm_group(GEMM_NT, [(x1, w1, y1, None, None), (x2, w2, y2, None, None)], 0)
    assert rel(y1, x1.float().cpu() @ w1.float().cpu().t()) < 2 ** -8
    assert rel(y2, x2.float().cpu() @ w2.float().cpu().t()) < 2 ** -8
    assert float(out[:, 264:].float().abs().max()) == 0.0 and float(out[100:, 64:264].float().abs().max()) == 0.0


@pytest.mark.parametrize("layout,impl", [(GEMM_TN, 2), (GEMM_TN, 6), (GEMM_NT, 2), (GEMM_NT, 6), (GEMM_NN, 2), (GEMM_NN, 6)])
def test_gemm_grouped_many_tiles_mixed_k(layout, impl):
    """A grouped launch big enough for the granule tile -> XCD map (mmf_xcd_tile: >= 8 x 32 tiles handed out in
    32-tile granules, the remainder as one range per XCD) with the maximum problem count and K from 16 to 2048, as in
    the deferred wgrad flush: every output is pre-filled with NaN (overwrite mode), so a tile the map skipped, or
    handed out twice with different neighbours, shows up; checked against an fp32 matmul of the same bf16 operands."""
    L = lib.load()
    shapes, ks = [(768, 768), (256, 384), (1536, 768), (520, 200)], [1000, 480, 16, 2048, 64]
    if impl == 6 and layout != GEMM_TN:      # the one-wave-per-SIMD kernel takes NT / NN reductions in whole 32-column stages
        ks = [1024, 480, 32, 2048, 64]
    probs, refs = [], []
    for i in range(lib.GEMM_MAX_PROBLEMS):
        (M, N), K = shapes[i % len(shapes)], ks[i % len(ks)]
        a = rnd(*((K, M) if layout == GEMM_TN else (M, K)), seed=1000 + i)
        b = rnd(*((N, K) if layout == GEMM_NT else (K, N)), seed=2000 + i, scale=K ** -0.5)
        a16, b16 = bf(a), bf(b)
        C = torch.full((M, N), float("nan"), device=DEV)
        probs.append((a16, b16, C, None, None))
        af, bfl = a16.float().cpu(), b16.float().cpu()
        refs.append((af.t() if layout == GEMM_TN else af) @ (bfl.t() if layout == GEMM_NT else bfl))
    lib.check(L.mmf_gemm_select_impl(impl))
    try:
        ops.gemm_group(layout, probs, 0)
        torch.cuda.synchronize()
    finally:
        lib.check(L.mmf_gemm_select_impl(0))
    for i, ((_, _, C, _, _), ref) in enumerate(zip(probs, refs)):
        assert not bool(torch.isnan(C).any()), f"problem {i}: unwritten output tile"
        assert rel(C, ref) < 1e-5, f"problem {i}: {rel(C, ref):.3e}"


    if impl == 6:
        assert L.mmf_gemm_last_impl() == 6


def test_gemm6_epilogues_and_edge_tiles():
    """The one-wave-per-SIMD kernel (gemm6.hip) pinned explicitly: every bf16 epilogue form on ragged NT / NN problems
    (tiles that hang over M and N), and f32 accumulate."""
    L = lib.load()
    lib.check(L.mmf_gemm_select_impl(6))
    try:
        for (M, N, K) in [(1000, 264, 3072), (300, 520, 96), (257, 8, 32)]:
            A, B_, bias, aux = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
            a16, b16, aux16 = bf(A), bf(B_), bf(aux)
            ref = a16.float().cpu() @ b16.float().cpu().t()
            C16 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            ops.gemm(GEMM_NT, a16, b16, C16, bias=bias.to(DEV), epilogue=EPI_BIAS | EPI_RELU)
            assert L.mmf_gemm_last_impl() == 6
            assert rel(C16, torch.relu(ref + bias)) < 2 ** -8
            ops.gemm(GEMM_NT, a16, b16, C16, bias=bias.to(DEV), aux=aux16, epilogue=EPI_BIAS | EPI_ADD_AUX)
            assert rel(C16, ref + bias + aux16.float().cpu()) < 2 ** -8
            ops.gemm(GEMM_NT, a16, b16, C16, aux=aux16, epilogue=EPI_MASK_AUX)
            assert rel(C16, ref * (aux16.float().cpu() > 0)) < 2 ** -8
            C = torch.full((M, N), 1.0, device=DEV)
            ops.gemm(GEMM_NT, a16, b16, C, epilogue=EPI_ACCUM)
            assert L.mmf_gemm_last_impl() == 6 and rel(C, ref + 1.0) < 1e-5
            bn = bf(rnd(K, N, seed=5, scale=K ** -0.5))                       # NN: dx = dy . W
            ops.gemm(GEMM_NN, a16, bn, C16, aux=aux16, epilogue=EPI_ADD_AUX)
            assert L.mmf_gemm_last_impl() == 6
            assert rel(C16, a16.float().cpu() @ bn.float().cpu() + aux16.float().cpu()) < 2 ** -8
    finally:
        lib.check(L.mmf_gemm_select_impl(0))


def _persistent_case(layout, shapes_k, epi, wgs, seed0=0, alpha=1.0, dropout=None):
    """one grouped launch on generation 6 and on generation 7 (persistent, `wgs` workgroups): -> (outputs6, outputs7, refs);
    refs are without dropout (the caller compares the kept elements)"""
    L = lib.load()
    probs6, probs7, refs = [], [], []
    for i, (M, N, K, pad) in enumerate(shapes_k):
        # operands as column slices of wider buffers (pad extra columns): leading dimensions differ between the problems
        abuf = bf(rnd(M, K + pad, seed=seed0 + 10 * i + 1))
        a16 = abuf[:, pad:]
        if layout == GEMM_NT:
            bbuf = bf(rnd(N, K + 2 * pad, seed=seed0 + 10 * i + 2, scale=K ** -0.5))
            b16 = bbuf[:, 2 * pad:]
        else:
            bbuf = bf(rnd(K, N + 2 * pad, seed=seed0 + 10 * i + 2, scale=K ** -0.5))
            b16 = bbuf[:, :N]
        bias = rnd(N, seed=seed0 + 10 * i + 3).to(DEV) if epi & EPI_BIAS else None
        aux16 = bf(rnd(M, N, seed=seed0 + 10 * i + 4)) if epi & (EPI_ADD_AUX | EPI_MASK_AUX) else None
        ref = a16.float().cpu() @ (b16.float().cpu().t() if layout == GEMM_NT else b16.float().cpu())
        if epi & EPI_BIAS:
            ref = ref + bias.cpu()
        if epi & EPI_RELU:
            ref = torch.relu(ref)
        if epi & EPI_MASK_AUX:
            ref = ref * (aux16.float().cpu() > 0) * alpha
        if epi & EPI_ADD_AUX:
            ref = ref + aux16.float().cpu()
        refs.append(ref)
        c6 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        c7 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        probs6.append((a16, b16, c6, bias, aux16))
        probs7.append((a16, b16, c7, bias, aux16))
    try:
        lib.check(L.mmf_gemm_select_impl(6))
        ops.gemm_group(layout, probs6, epi, alpha=alpha, dropout=dropout)
        assert L.mmf_gemm_last_impl() == 6
        lib.check(L.mmf_gemm_select_impl(7))
        lib.check(L.mmf_gemm_set_persistent_workgroups(wgs))
        ops.gemm_group(layout, probs7, epi, alpha=alpha, dropout=dropout)
        assert L.mmf_gemm_last_impl() == 7
        torch.cuda.synchronize()
    finally:
        lib.check(L.mmf_gemm_select_impl(0))
        lib.check(L.mmf_gemm_set_persistent_workgroups(0))
    return [p[2] for p in probs6], [p[2] for p in probs7], refs


@pytest.mark.parametrize("layout", [GEMM_NT, GEMM_NN])
@pytest.mark.parametrize("wgs", [1, 3, 8, 0])
def test_gemm7_persistent_walks_tiles_bit_identical_to_gemm6(layout, wgs):
    """The persistent kernel (gemm7.hip): a grouped launch of ragged problems with different K (from the minimum of five stages up)
    and different leading dimensions, walked by 1 / 3 / 8 / CU-count workgroups — so a workgroup crosses tile AND problem
    boundaries with its ring running (descriptor and per-lane offset switch, the ring running dry only at its last tile) — must
    write every output element (NaN pre-fill), match the f32 matmul and equal generation 6 BIT FOR BIT (same MFMA order per tile;
    with a bias: up to one bf16 ulp on a few elements, see below)."""
    shapes = [(520, 264, 160, 8), (300, 520, 768, 0), (257, 8, 192, 16), (1000, 392, 512, 8), (256, 256, 2048, 0), (40, 1032, 224, 24)]
    # the (layout, flag set) pairs generation 7 is instantiated for — the fusion step's: everything else stays on generation 6
    epis = (0, EPI_BIAS, EPI_BIAS | EPI_RELU, EPI_BIAS | EPI_ADD_AUX) if layout == GEMM_NT else (0, EPI_MASK_AUX, EPI_ADD_AUX)
    for epi in epis:
        c6, c7, refs = _persistent_case(layout, shapes, epi, wgs, seed0=100 * epi)
        for i, (a, b, r) in enumerate(zip(c6, c7, refs)):
            assert not bool(torch.isnan(b.float()).any()), f"epi {epi} problem {i}: unwritten output"
            assert rel(b, r) < 2 ** -8, f"epi {epi} problem {i}: {rel(b, r):.3e}"
            if epi & EPI_BIAS:
                # generation 7 STARTS its accumulators from the bias (sixteen MFMAs per tile) instead of adding it at the end: the same
                # real number rounded along a different f32 path — a bf16 output may differ by one ulp on a small fraction of elements
                d = (a.float() - b.float()).abs()
                assert float(d.max()) <= 2 ** -7 * max(1.0, float(a.float().abs().max())), f"epi {epi} problem {i}"
                assert float((d > 0).float().mean()) < 0.02, f"epi {epi} problem {i}: {float((d > 0).float().mean()):.4f} of the elements differ"
            else:
                assert torch.equal(a, b), f"epi {epi} problem {i}: generation 7 differs from generation 6"


@pytest.mark.parametrize("wgs", [3, 0])
def test_gemm7_training_epilogues_match_gemm6(wgs):
    """The two epilogues a training step adds (round 4): dropout on the FFN hidden layer (NT, bias + ReLU + dropout: the mask is the
    stateless hash of (state, site, caller's problem index, element) — the persistent kernel SORTS its problems by K, so the caller's
    index must survive) and the dropout backward riding on the ReLU mask (NN, mask x alpha = 1 / (1 - p)).  Same masks and, without a
    bias, the same bits as generation 6; kept elements equal the undropped reference x 1 / (1 - p)."""
    shapes = [(520, 264, 160, 8), (300, 520, 768, 0), (1000, 392, 512, 8), (256, 256, 2048, 0)]
    p = 0.25
    ops.seed_dropout(1234)
    c6, c7, refs = _persistent_case(GEMM_NT, shapes, EPI_BIAS | EPI_RELU, wgs, seed0=7, dropout=(p, 5))
    for i, (a, b, r) in enumerate(zip(c6, c7, refs)):
        assert not bool(torch.isnan(b.float()).any())
        assert torch.equal(a == 0, b == 0), f"problem {i}: the two generations drew different masks"
        kept = (b.float().cpu() != 0)
        live = r > 0
        frac = float(kept[live].float().mean())
        assert abs(frac - (1 - p)) < 0.02, (i, frac)
        assert rel(b.float().cpu()[kept], (r / (1 - p))[kept]) < 2 ** -7
        d = (a.float() - b.float()).abs()                        # (bias: one bf16 ulp on a few elements, as in the test above)
        assert float((d > 0).float().mean()) < 0.02
    c6, c7, refs = _persistent_case(GEMM_NN, shapes, EPI_MASK_AUX, wgs, seed0=9, alpha=1.0 / (1 - p))
    for i, (a, b, r) in enumerate(zip(c6, c7, refs)):
        assert torch.equal(a, b), f"problem {i}: generation 7 differs from generation 6"
        assert rel(b, r) < 2 ** -8


def test_gemm7_is_the_automatic_choice_where_it_has_the_flag_set():
    """automatic rule: an NT / NN launch gemm6 would take (K a multiple of 32, at least a quarter of the CUs get a 256 x 256 tile) goes to the
    persistent kernel when it has the launch's flag set; f32 output and flag sets it lacks stay on generation 6; smaller launches and
    other K go to the 256 x 128 ring (generation 2); the generations that left the build are refused by name"""
    L = lib.load()
    cus = L.mmf_device_cu_count()
    rows = 256 * (cus // 3 + 1)                                   # x 3 column tiles > CUs
    a, w = bf(rnd(rows, 512, seed=1)), bf(rnd(768, 512, seed=2, scale=0.05))
    c = torch.empty((rows, 768), dtype=torch.bfloat16, device=DEV)
    ops.gemm(GEMM_NT, a, w, c)
    assert L.mmf_gemm_last_impl() == 7
    assert rel(c, a.float().cpu() @ w.float().cpu().t()) < 2 ** -8
    half = rows // 2 // 256 * 256 + 256
    ops.gemm(GEMM_NT, a[:half], w, c[:half])                       # one round, more than half of the CUs
    assert L.mmf_gemm_last_impl() == 7
    cf = torch.empty((rows, 768), device=DEV)
    ops.gemm(GEMM_NT, a, w, cf)                                    # f32 output
    assert L.mmf_gemm_last_impl() == 6
    aux = bf(rnd(rows, 768, seed=3))
    ops.gemm(GEMM_NT, a, w, c, aux=aux, epilogue=EPI_MASK_AUX)     # a flag set only the NN form has
    assert L.mmf_gemm_last_impl() == 6
    few = 256 * max(cus // 16, 1)                                  # x 3 column tiles < a quarter of the CUs
    ops.gemm(GEMM_NT, a[:few], w, c[:few])
    assert L.mmf_gemm_last_impl() == 2
    a2, w2 = bf(rnd(rows, 520, seed=4)), bf(rnd(768, 520, seed=5, scale=0.05))    # K not a multiple of 32
    ops.gemm(GEMM_NT, a2, w2, c)
    assert L.mmf_gemm_last_impl() == 2 and rel(c, a2.float().cpu() @ w2.float().cpu().t()) < 2 ** -8
    for gone in (1, 3, 4, 5):
        assert L.mmf_gemm_select_impl(gone) != 0


def test_gemm6_wgrad_reads_nothing_it_should_not_use():
    """wgrad operands as the LAST column blocks of packed buffers whose other columns are NaN, with M and N that leave
    edge tiles: the tile is fetched without a column predicate, so NaNs from beyond the valid columns sit in LDS — they
    must reach neither the weight gradient nor (through the selector rows of the bias-gradient MFMA) the bias gradient,
    and nothing past the last valid element may be read (the buffer ranges end there)."""
    L = lib.load()
    K, M, N = 515, 200, 328
    pa = torch.full((K, 3 * M), float("nan"), dtype=torch.bfloat16, device=DEV)
    pb = torch.full((K, 2 * N), float("nan"), dtype=torch.bfloat16, device=DEV)
    A, B_ = rnd(K, M, seed=7), rnd(K, N, seed=8, scale=K ** -0.5)
    pa[:, 2 * M:] = bf(A)
    pb[:, N:] = bf(B_)
    a16, b16 = pa[:, 2 * M:], pb[:, N:]
    C = torch.full((M, N), float("nan"), device=DEV)
    db = torch.full((M,), 2.0, device=DEV)
    lib.check(L.mmf_gemm_select_impl(6))            # (two tiles: the automatic rule would take the 256 x 128 ring)
    try:
        ops.gemm(GEMM_TN, a16, b16, C, bias=db, epilogue=EPI_COLSUM_A)
        assert L.mmf_gemm_last_impl() == 6
    finally:
        lib.check(L.mmf_gemm_select_impl(0))
    assert rel(C, a16.float().cpu().t() @ b16.float().cpu()) < 1e-5
    assert rel(db, a16.float().cpu().sum(0) + 2.0) < 1e-5


def test_linear_group_on_split3_returns_one_gradient_buffer():
    """Three linears on the column thirds of one pooled (B, 3d) tensor (MulT's pooled out-projections, :171): the input
    gradients come back as the thirds of one buffer and split3's backward returns that buffer (no concatenation
    kernel); an f32-output ReLU linear fed with an f32 tensor (final_fusion, :172) masks and narrows its gradient in one
    kernel and writes an f32 input gradient.  Checked against torch autograd on the same bf16-rounded operands."""
    from mmfusion import arena as arena_mod, small_ops as sops
    B, d = 16, 64
    torch.manual_seed(3)
    lins = torch.nn.ModuleList([torch.nn.Linear(d, d) for _ in range(3)] + [torch.nn.Linear(3 * d, d)]).cuda()
    arena_mod.ensure(lins)
    c = bf(rnd(B, 3 * d, seed=21)).requires_grad_(True)
    wsum = rnd(B, d, seed=22).to(DEV)
    hits0 = sops.split3_nocopy_hits
    pf = ops.linear_group([(x, ops.LinearSpec(ops.W(l.weight), ops.W(l.bias), False), None)
                           for x, l in zip(sops.split3(c), lins[:3])], out_f32=True, cat=True)
    y = ops.linear(pf, ops.W(lins[3].weight), ops.W(lins[3].bias), relu=True, out_f32=True)
    assert pf.dtype == torch.float32 and y.dtype == torch.float32
    (y * wsum).sum().backward()
    torch.cuda.synchronize()
    assert sops.split3_nocopy_hits == hits0 + 1, "split3 backward concatenated although its gradients were one buffer"
    # reference: same bf16-rounded weights / inputs, f32 math
    cr = c.detach().float().cpu().requires_grad_(True)
    ws = [l.weight.detach().to(torch.bfloat16).float().cpu() for l in lins]
    bs = [l.bias.detach().float().cpu() for l in lins]
    pr = torch.cat([cr[:, i * d:(i + 1) * d] @ ws[i].t() + bs[i] for i in range(3)], dim=1)
    yr = torch.relu(pr.to(torch.bfloat16).float() @ ws[3].t() + bs[3])
    (yr * wsum.cpu()).sum().backward()
    assert rel(y, yr) < 2e-2 and rel(pf, pr) < 1e-2
    assert rel(c.grad, cr.grad) < 3e-2


def test_add3_group_matches_three_sums():
    """MulT's three residual sums (:156-158) as one launch; ragged element counts (one not a multiple of 8), gradients
    are the incoming gradient for all three operands."""
    shapes = [(8192, 768), (480, 768), (7, 13)]
    trips = [tuple(bf(rnd(*sh, seed=300 + 10 * i + j)).requires_grad_(True) for j in range(3)) for i, sh in enumerate(shapes)]
    outs = ops.add3_group(trips)
    for (a, b, c), y in zip(trips, outs):
        ref = (a.float() + b.float() + c.float()).to(torch.bfloat16)
        assert torch.equal(y, ref)
    sum(float(i + 1) * o.float().sum() for i, o in enumerate(outs)).backward()
    for i, (a, b, c) in enumerate(trips):
        for t in (a, b, c):
            assert torch.equal(t.grad.float(), torch.full_like(t, float(i + 1)).float())


def test_gemm_rejects_bad_shapes():
    a, b = bf(rnd(16, 12)), bf(rnd(8, 12))
    with pytest.raises(RuntimeError, match="granularity"):
        ops.gemm(GEMM_NT, a, b, torch.empty((16, 8), device=DEV))


# ---------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("rows,d", [(7, 64), (1000, 768), (130, 192), (33, 1536), (5, 2048)])
def test_layernorm_fwd_bwd(rows, d):
    x, gam, bet, dy = rnd(rows, d, seed=1) * 2 + 0.5, rnd(d, seed=2) * 0.1 + 1, rnd(d, seed=3) * 0.1, rnd(rows, d, seed=4)
    x16, dy16 = bf(x), bf(dy)
    g = torch.nn.Parameter(gam.to(DEV))
    b = torch.nn.Parameter(bet.to(DEV))
    g.grad, b.grad = torch.zeros_like(g), torch.zeros_like(b)
    xin = x16.clone().requires_grad_(True)
    y = ops.layernorm_group([(xin, g, b)])[0]
    y.backward(dy16)
    xr = x16.float().cpu().requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (d,), gr, br, 1e-5)
    yr.backward(dy16.float().cpu())
    assert rel(y, yr.detach()) < 2 ** -7
    assert rel(xin.grad, xr.grad) < 2 ** -7
    assert rel(g.grad, gr.grad) < 1e-3
    assert rel(b.grad, br.grad) < 1e-3


# ---------------------------------------------------------------------------------------- attention
def attn_ref(q, k, v, H):
    """fp32 reference on (B,T,H*dh) tensors; returns output (B,Tq,d)."""
    B, Tq, d = q.shape
    dh = d // H
    qh = q.view(B, Tq, H, dh).transpose(1, 2)
    kh = k.view(B, -1, H, dh).transpose(1, 2)
    vh = v.view(B, -1, H, dh).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(dh), dim=-1)
    return (p @ vh).transpose(1, 2).reshape(B, Tq, d)


@pytest.mark.parametrize("B,H,dh,Tq,Tk", [(2, 2, 96, 70, 40), (1, 3, 96, 128, 64), (2, 2, 64, 33, 150),
                                          (3, 2, 96, 1, 1), (2, 8, 96, 30, 400), (1, 2, 64, 200, 3),
                                          (1, 1, 96, 257, 129)])
def test_attention_fwd_bwd(B, H, dh, Tq, Tk):
    d = H * dh
    q, kv, do = rnd(B, Tq, d, seed=1), rnd(B, Tk, 2 * d, seed=2), rnd(B, Tq, d, seed=3)
    q16 = bf(q).reshape(B * Tq, d).requires_grad_(True)
    kv16 = bf(kv).reshape(B * Tk, 2 * d).requires_grad_(True)
    o = ops.attention_group([ops.AttnSpec(B, Tq, Tk, q=(0, 0), k=(1, 0), v=(1, d))], H, dh, [q16, kv16])[0]
    o.backward(bf(do).reshape(B * Tq, d))
    qr = q16.detach().float().cpu().view(B, Tq, d).requires_grad_(True)
    kvr = kv16.detach().float().cpu().view(B, Tk, 2 * d).requires_grad_(True)
    orf = attn_ref(qr, kvr[..., :d], kvr[..., d:], H)
    orf.backward(bf(do).float().cpu())
    assert rel(o.view(B, Tq, d), orf.detach()) < 2 ** -7
    assert rel(q16.grad.view(B, Tq, d), qr.grad) < 2e-2
    assert rel(kv16.grad.view(B, Tk, 2 * d)[..., :d], kvr.grad[..., :d]) < 2e-2
    assert rel(kv16.grad.view(B, Tk, 2 * d)[..., d:], kvr.grad[..., d:]) < 2e-2


def test_attention_grouped_self_packed():
    """two problems in one launch; q, k, v as column ranges of one packed (rows, 3d) buffer"""
    H, dh = 2, 96
    d = H * dh
    specs, srcs, refs = [], [], []
    for i, (B, T) in enumerate([(2, 50), (1, 130)]):
        qkv = bf(rnd(B * T, 3 * d, seed=20 + i)).requires_grad_(True)
        srcs.append(qkv)
        specs.append(ops.AttnSpec(B, T, T, q=(i, 0), k=(i, d), v=(i, 2 * d)))
    outs = ops.attention_group(specs, H, dh, srcs)
    (outs[0].float().sum() + 2 * outs[1].float().sum()).backward()
    for i, (B, T) in enumerate([(2, 50), (1, 130)]):
        r = srcs[i].detach().float().cpu().view(B, T, 3 * d).requires_grad_(True)
        o = attn_ref(r[..., :d], r[..., d:2 * d], r[..., 2 * d:], H)
        (o.sum() * (i + 1)).backward()
        assert rel(outs[i].view(B, T, d), o.detach()) < 2 ** -7
        assert rel(srcs[i].grad.view(B, T, 3 * d), r.grad) < 2e-2


@pytest.mark.parametrize("dh", [96, 64])
def test_attention_group_mixes_narrow_and_wide_problems(dh):
    """One grouped call of problems with a <= 32-row side on either end beside wide ones (MulT's 30-frame stream against 512 / 400
    positions): every output and gradient against the fp32 reference, and the grouped results bit-identical to the same problems
    launched one by one (a problem's arithmetic must not depend on its company in the launch)."""
    B, H = 2, 2
    d = H * dh
    shapes = [(30, 200), (200, 30), (30, 30), (100, 100), (17, 333), (333, 9)]
    def run(idxs):
        srcs, specs = [], []
        for n, i in enumerate(idxs):
            Tq, Tk = shapes[i]
            srcs += [bf(rnd(B * Tq, d, seed=50 + i)).requires_grad_(True), bf(rnd(B * Tk, 2 * d, seed=70 + i)).requires_grad_(True)]
            specs.append(ops.AttnSpec(B, Tq, Tk, q=(2 * n, 0), k=(2 * n + 1, 0), v=(2 * n + 1, d)))
        outs = ops.attention_group(specs, H, dh, srcs)
        torch.autograd.backward(outs, [bf(rnd(B * shapes[i][0], d, seed=90 + i)) for i in idxs])
        return outs, srcs
    outs, srcs = run(range(len(shapes)))
    for i, (Tq, Tk) in enumerate(shapes):
        q16, kv16 = srcs[2 * i], srcs[2 * i + 1]
        qr = q16.detach().float().cpu().view(B, Tq, d).requires_grad_(True)
        kvr = kv16.detach().float().cpu().view(B, Tk, 2 * d).requires_grad_(True)
        orf = attn_ref(qr, kvr[..., :d], kvr[..., d:], H)
        orf.backward(bf(rnd(B * Tq, d, seed=90 + i)).float().cpu().view(B, Tq, d))
        assert rel(outs[i].view(B, Tq, d), orf.detach()) < 2 ** -7, (Tq, Tk)
        assert rel(q16.grad.view(B, Tq, d), qr.grad) < 2e-2, (Tq, Tk)
        assert rel(kv16.grad.view(B, Tk, 2 * d), kvr.grad) < 2e-2, (Tq, Tk)
        o1, s1 = run([i])
        assert torch.equal(o1[0], outs[i]) and torch.equal(s1[0].grad, q16.grad) and torch.equal(s1[1].grad, kv16.grad), (Tq, Tk)


def test_attention_online_softmax_rescale_branch():
    """Force the running max to jump at a later KV tile (guide rule 26): one key far above the rest."""
    B, H, dh, Tq, Tk = 1, 1, 96, 32, 200
    q, k, v = rnd(B, Tq, dh, seed=1), rnd(B, Tk, dh, seed=2), rnd(B, Tk, dh, seed=3)
    k[0, 170] = q[0, 5] * 4.0                       # spikes the score of query 5 in the third tile
    q16 = bf(q).reshape(Tq, dh)
    kv16 = bf(torch.cat([k, v], -1)).reshape(Tk, 2 * dh)
    o = ops.attention_group([ops.AttnSpec(B, Tq, Tk, q=(0, 0), k=(1, 0), v=(1, dh))], H, dh, [q16, kv16])[0]
    ref = attn_ref(q16.float().cpu().view(1, Tq, dh), kv16.float().cpu()[:, :dh].reshape(1, Tk, dh),
                   kv16.float().cpu()[:, dh:].reshape(1, Tk, dh), 1)
    assert rel(o.view(1, Tq, dh), ref) < 2 ** -7


# ---------------------------------------------------------------------------------------- streaming
def test_cast_add_pool_colsum_relu():
    x = rnd(1000, 771, seed=1)
    xg = x.to(DEV)
    y = ops.cast_to_bf16(xg)
    assert torch.equal(y.cpu(), x.to(torch.bfloat16))
    assert torch.equal(ops.cast_to_f32(y).cpu(), x.to(torch.bfloat16).float())
    a, b, c = bf(rnd(3, 50, 64, seed=2)), bf(rnd(3, 50, 64, seed=3)), bf(rnd(3, 50, 64, seed=4))
    s = ops.add3(a, b, c)
    assert rel(s, a.float() + b.float() + c.float()) < 2 ** -8
    assert rel(ops.add3(a, b), a.float() + b.float()) < 2 ** -8
    xs = [bf(rnd(4, T, 192, seed=10 + T)).requires_grad_(True) for T in (70, 40, 6)]
    p = ops.meanpool_cat(xs)
    ref = torch.cat([t.float().mean(1) for t in xs], -1)
    assert rel(p, ref) < 2 ** -8
    g = bf(rnd(4, 3 * 192, seed=5))
    p.backward(g)
    for i, t in enumerate(xs):
        exp = (g.float()[:, i * 192:(i + 1) * 192] / t.shape[1]).unsqueeze(1).expand_as(t)
        assert rel(t.grad, exp) < 2 ** -8
    m = bf(rnd(777, 264, seed=6))
    out = torch.ones(264, device=DEV)
    L = lib.load()
    lib.check(L.mmf_colsum_bf16(m.data_ptr(), out.data_ptr(), 777, 264, 264, lib.stream_ptr()))
    assert rel(out, m.float().sum(0) + 1) < 1e-4
    dy, yv = bf(rnd(999, seed=7)), bf(rnd(999, seed=8))
    dx = torch.empty_like(dy)
    lib.check(L.mmf_relu_bwd_bf16(dy.data_ptr(), yv.data_ptr(), dx.data_ptr(), 999, lib.stream_ptr()))
    assert torch.equal(dx, torch.where(yv.float() > 0, dy, torch.zeros_like(dy)))


# ---------------------------------------------------------------------------------------- optimiser
def test_fused_adamw_matches_torch():
    """3 steps of clip_grad_norm_(1.0) + AdamW(wd=1e-5) on a small module vs the fused arena kernels"""
    from mmfusion import arena as arena_mod
    from mmfusion.train import FusedAdamW
    torch.manual_seed(0)
    mod = torch.nn.Sequential(torch.nn.Linear(40, 72), torch.nn.Linear(72, 8)).cuda()
    ref = torch.nn.Sequential(torch.nn.Linear(40, 72), torch.nn.Linear(72, 8))
    ref.load_state_dict({k: v.cpu() for k, v in mod.state_dict().items()})
    ar = arena_mod.ensure(mod)
    opt = FusedAdamW(ar, lr=1e-2, weight_decay=1e-5, max_grad_norm=1.0)
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=1e-5)
    for step in range(3):
        grads = [rnd(*p.shape, seed=100 * step + i) * 3 for i, p in enumerate(ref.parameters())]
        for p, rp, g in zip(mod.parameters(), ref.parameters(), grads):
            p.grad.copy_(g.to(DEV))
            rp.grad = g.clone()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        ropt.step()
        opt.step()
        for p, rp in zip(mod.parameters(), ref.parameters()):
            assert rel(p.detach(), rp.detach(), floor=1e-6) < 1e-5
            assert rel(ops.shadow(p), rp.detach().to(torch.bfloat16), floor=1e-6) < 2 ** -7


def test_checkpoint_resume_and_torch_adamw_interchange(tmp_path):
    """the fused optimiser's state leaves and re-enters in torch.optim.AdamW's state_dict layout
    inside a reference-layout checkpoint (advanced_trainer.py:396-411).  (1) save after 2 steps, restore into a
    differently initialised copy, take a third step on both: same parameters and bf16 shadows; (2) the saved
    optimizer_state_dict loads into a real torch.optim.AdamW over the same parameters, whose third step agrees."""
    from mmfusion import arena as arena_mod
    from mmfusion.train import FusedAdamW, load_checkpoint, save_checkpoint

    def make(seed):
        torch.manual_seed(seed)
        return torch.nn.Sequential(torch.nn.Linear(40, 72), torch.nn.LayerNorm(72), torch.nn.Linear(72, 8))

    def grads_for(step, params):
        return [rnd(*p.shape, seed=100 * step + i) * 3 for i, p in enumerate(params)]

    mod = make(0).cuda()
    ar = arena_mod.ensure(mod)
    opt = FusedAdamW(ar, lr=1e-2, weight_decay=1e-5, max_grad_norm=1.0)
    for step in range(2):
        for p, g in zip(mod.parameters(), grads_for(step, list(mod.parameters()))):
            p.grad.copy_(g.to(DEV))
        opt.step()
    path = str(tmp_path / "ckpt.pth")
    save_checkpoint(path, mod, opt, epoch=7, metrics={"accuracy": 0.25}, config=None)

    mod2 = make(123).cuda()                                         # different weights until the checkpoint is loaded
    ar2 = arena_mod.ensure(mod2)
    opt2 = FusedAdamW(ar2, lr=1e-2, weight_decay=1e-5, max_grad_norm=1.0)
    ck = load_checkpoint(path, mod2, opt2)
    assert ck["epoch"] == 7 and ck["metrics"] == {"accuracy": 0.25} and opt2.t == 2
    for p, q in zip(mod.parameters(), mod2.parameters()):
        assert torch.equal(p.detach(), q.detach())
        assert torch.equal(ops.shadow(p), ops.shadow(q))            # the shadow was re-cast from the loaded masters

    ref = make(5)
    ref.load_state_dict(ck["model_state_dict"])
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=1e-5)
    ropt.load_state_dict(ck["optimizer_state_dict"])                # torch validates groups / parameter counts

    g3 = grads_for(2, list(ref.parameters()))
    for p, q, rp, g in zip(mod.parameters(), mod2.parameters(), ref.parameters(), g3):
        p.grad.copy_(g.to(DEV))
        q.grad.copy_(g.to(DEV))
        rp.grad = g.clone()
    torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
    ropt.step()
    opt.step()
    opt2.step()
    for p, q, rp in zip(mod.parameters(), mod2.parameters(), ref.parameters()):
        # resumed run == uninterrupted run (up to the summation order of the global-norm reduction's f32 atomics)
        assert rel(p.detach(), q.detach(), floor=1e-6) < 1e-6
        assert rel(q.detach(), rp.detach(), floor=1e-6) < 1e-5      # == torch.optim.AdamW resumed from the same file

    # and the other direction: moments written by torch.optim.AdamW (two parameter groups, as the reference builds them)
    params = list(ref.parameters())
    ropt2 = torch.optim.AdamW([{"params": params[:2], "lr": 1e-3}, {"params": params[2:], "lr": 1e-2}], weight_decay=1e-5)
    for rp, g in zip(params, g3):
        rp.grad = g.clone()
    ropt2.step()
    opt2.load_state_dict(ropt2.state_dict(), mod2.parameters())
    assert opt2.t == 1
    sd = opt2.state_dict(mod2.parameters())
    for j, rp in enumerate(params):
        assert rel(sd["state"][j]["exp_avg"], ropt2.state[rp]["exp_avg"], floor=1e-9) < 1e-6
        assert rel(sd["state"][j]["exp_avg_sq"], ropt2.state[rp]["exp_avg_sq"], floor=1e-12) < 1e-6


# ---------------------------------------------------------------------------------------- skinny-M linear
@pytest.mark.parametrize("M,N,K", [(16, 768, 2304), (4, 256, 768), (48, 3072, 768), (64, 384, 768), (1, 8, 8),
                                   (16, 1536, 3840), (3, 72, 40)])
def test_skinny_fwd_and_dgrad(M, N, K):
    x, w, b = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=K ** -0.5)), rnd(N, seed=3).to(DEV)
    ref = x.float().cpu() @ w.float().cpu().t() + b.cpu()
    y = torch.full((M, N), float("nan"), device=DEV)
    p = lib.SkinnyProblem(x.data_ptr(), w.data_ptr(), y.data_ptr(), b.data_ptr(), None, M, N, K, K, K, N, 0)
    lib.skinny_fwd([p], EPI_BIAS, True)
    assert rel(y, ref) < 1e-5
    y16 = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    p = lib.SkinnyProblem(x.data_ptr(), w.data_ptr(), y16.data_ptr(), b.data_ptr(), None, M, N, K, K, K, N, 0)
    lib.skinny_fwd([p], EPI_BIAS | EPI_RELU, False)
    assert rel(y16, torch.relu(ref)) < 2 ** -8
    # dgrad: dx = (dy W) * (aux > 0) * alpha
    dy, aux = bf(rnd(M, N, seed=4)), bf(rnd(M, K, seed=5))
    dref = dy.float().cpu() @ w.float().cpu()
    dx = torch.full((M, K), float("nan"), device=DEV)
    p = lib.SkinnyProblem(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), None, None, M, N, K, N, K, K, 0)
    lib.skinny_dgrad([p], 0, 1.0, True)
    assert rel(dx, dref) < 1e-5
    dx16 = torch.empty((M, K), dtype=torch.bfloat16, device=DEV)
    p = lib.SkinnyProblem(dy.data_ptr(), w.data_ptr(), dx16.data_ptr(), None, aux.data_ptr(), M, N, K, N, K, K, K)
    lib.skinny_dgrad([p], EPI_MASK_AUX, 2.0, False)
    assert rel(dx16, dref * (aux.float().cpu() > 0) * 2.0) < 2 ** -8


def test_skinny_grouped_strided():
    big = bf(rnd(16, 2304, seed=9))
    ws = [bf(rnd(768, 768, seed=10 + i, scale=0.05)) for i in range(3)]
    ys = [torch.empty((16, 768), dtype=torch.bfloat16, device=DEV) for _ in range(3)]
    probs = [lib.SkinnyProblem(big.data_ptr() + 2 * 768 * i, ws[i].data_ptr(), ys[i].data_ptr(), None, None,
                               16, 768, 768, 2304, 768, 768, 0) for i in range(3)]
    lib.skinny_fwd(probs, 0, False)
    for i in range(3):
        assert rel(ys[i], big[:, 768 * i:768 * (i + 1)].float().cpu() @ ws[i].float().cpu().t()) < 2 ** -8


def test_lazy_zero_grad_is_bit_identical_and_zeroes_untouched_matrices():
    """ParamArena.zero_grad(lazy=True): the first wgrad GEMM of a step overwrites instead of accumulating
    onto a memset; gradients must be bit-identical to the eager-zero step, and a wgrad-managed matrix that
    receives no gradient in a later step (contrastive projectors when the loss is off) must read zero
    after finalize_grads()."""
    import config as cfgmod
    from mmfusion import arena as arena_mod
    from models import fusion_layers as fl
    cfg = cfgmod.ModelConfig()
    cfg.fusion_hidden_size, cfg.fusion_num_heads, cfg.fusion_dropout = 128, 2, 0.0
    torch.manual_seed(3)
    m = fl.ContrastiveFusion(cfg).cuda().train()
    ar = arena_mod.ensure(m)
    xs = [torch.randn(16, 128, device="cuda") for _ in range(3)]

    def step(lazy, contrastive):
        ar.zero_grad(lazy=lazy)
        out = m(*xs, compute_contrastive_loss=contrastive)
        loss = out["fused_features"].sum()
        if contrastive:
            loss = loss + sum(out["contrastive_losses"].values())
        loss.backward()
        ar.finalize_grads()
        torch.cuda.synchronize()
        return ar.grads.clone()

    g_eager = step(False, True)                 # also teaches the arena which matrices the wgrad GEMMs own
    assert ar._managed, "no wgrad-managed parameters recorded"
    ar.grads.fill_(7.0)                         # stale values a lazy zero must not let through
    g_lazy = step(True, True)
    assert torch.equal(g_eager, g_lazy)
    g_eager2 = step(False, False)               # projector weights get no gradient now
    ar.grads.fill_(7.0)
    g_lazy2 = step(True, False)
    assert torch.equal(g_eager2, g_lazy2)
    proj = [p for n, p in m.named_parameters() if "projector" in n and p.dim() == 2]
    assert proj and all(float(p.grad.abs().max()) == 0.0 for p in proj)


# ---- small fused branch kernels (csrc/small.hip) against plain torch fp32 references -----------------------
def _small_param(*shape):
    p = torch.nn.Parameter(torch.randn(*shape, device="cuda") * 0.3)
    p.grad = torch.zeros_like(p)
    return p


def test_gat3_dense_matches_torch_dense_gat():
    from mmfusion import small_ops as sops
    torch.manual_seed(11)
    B, H, C = 5, 4, 136
    h = torch.randn(B * 3, H * C, device="cuda", requires_grad=True)
    att_src, att_dst, bias = _small_param(1, H, C), _small_param(1, H, C), _small_param(C)
    y, pooled = sops.gat3(h, att_src, att_dst, bias, B, H, relu=True, pool=True)
    gy = torch.randn(B * 3, C, device="cuda").bfloat16()
    gp = torch.randn(B, C, device="cuda").bfloat16()
    torch.autograd.backward([y, pooled], [gy, gp])

    h2 = h.detach().clone().requires_grad_(True)
    ps = [p.detach().clone().requires_grad_(True) for p in (att_src, att_dst, bias)]
    hv = h2.view(B, 3, H, C)
    s_src, s_dst = (hv * ps[0]).sum(-1), (hv * ps[1]).sum(-1)
    e = torch.nn.functional.leaky_relu(s_dst.unsqueeze(2) + s_src.unsqueeze(1), 0.2)
    ex = torch.exp(e - e.max(dim=2, keepdim=True).values)
    alpha = ex / (ex.sum(dim=2, keepdim=True) + 1e-16)
    out = torch.relu(torch.einsum("bijh,bjhc->bihc", alpha, hv).mean(dim=2) + ps[2])
    # the kernel applies the ReLU mask of its bf16-rounded output; do the same here
    torch.autograd.backward([out, out.mean(dim=1)], [gy.float().view(B, 3, C), gp.float()])
    out = out.detach()
    assert (y.float().view(B, 3, C) - out).abs().max() <= 2 ** -7 * max(1.0, float(out.abs().max()))
    assert (pooled.float() - out.mean(dim=1)).abs().max() <= 2 ** -7 * max(1.0, float(out.abs().max()))
    for got, ref in ((h.grad, h2.grad), (att_src.grad, ps[0].grad), (att_dst.grad, ps[1].grad), (bias.grad, ps[2].grad)):
        assert (got - ref).abs().max() <= 2e-3 * max(1.0, float(ref.abs().max())), float((got - ref).abs().max())


def test_gat3_alpha_dropout_is_consistent_between_forward_and_backward():
    """With dropout on alpha the mask is regenerated in the backward: d<y, g>/dh by finite differences of the
    kernel itself (same step state => same mask) must match the analytic dh."""
    from mmfusion import ops, small_ops as sops
    torch.manual_seed(12)
    B, H, C = 3, 4, 64
    att_src, att_dst, bias = _small_param(1, H, C), _small_param(1, H, C), _small_param(C)
    h = torch.randn(B * 3, H * C, device="cuda", requires_grad=True)
    g = torch.randn(B * 3, C, device="cuda")

    def run(hh):
        ops._site = 0
        return sops.gat3(hh, att_src, att_dst, bias, B, H, relu=False, pool=False, dropout_p=0.4)
    y = run(h)
    y.backward(g.bfloat16())
    d = torch.randn_like(h) * 1e-2
    with torch.no_grad():
        fd = ((run(h + d).float() - run(h - d).float()) * g).sum() / 2
    an = (h.grad * d).sum()
    assert abs(float(fd - an)) <= 0.05 * max(1.0, abs(float(an))), (float(fd), float(an))


def test_normalize_infonce_matches_torch():
    import torch.nn.functional as Fn
    from mmfusion import small_ops as sops
    torch.manual_seed(13)
    B, D, T = 16, 384, 0.07
    zs = [torch.randn(B, D, device="cuda", requires_grad=True) for _ in range(3)]
    ns, ls = sops.normalize_infonce(zs, T, True)
    wl = torch.tensor([0.3, 1.1, 0.7], device="cuda")
    gn = [torch.randn(B, D, device="cuda") * 0.01 for _ in range(3)]
    (sum(w * l for w, l in zip(wl, ls)) + sum((n * g).sum() for n, g in zip(ns, gn))).backward()

    z2 = [z.detach().clone().requires_grad_(True) for z in zs]
    n2 = [Fn.normalize(z, dim=-1) for z in z2]
    lab = torch.arange(B, device="cuda")

    def nce(a, b):
        sim = a @ b.t() / T
        return (Fn.cross_entropy(sim, lab) + Fn.cross_entropy(sim.t(), lab)) / 2
    l2 = [nce(n2[0], n2[1]), nce(n2[0], n2[2]), nce(n2[1], n2[2])]
    (sum(w * l for w, l in zip(wl, l2)) + sum((n * g).sum() for n, g in zip(n2, gn))).backward()
    for a, b in zip(ns, n2):
        assert (a - b).abs().max() <= 1e-6
    for a, b in zip(ls, l2):
        assert abs(float(a - b)) <= 1e-4 * max(1.0, abs(float(b)))
    for a, b in zip(zs, z2):
        assert (a.grad - b.grad).abs().max() <= 1e-4 * max(1.0, float(b.grad.abs().max()))


def test_adaptive_combine_and_narrow_linear_match_torch():
    import torch.nn.functional as Fn
    from mmfusion import small_ops as sops
    torch.manual_seed(14)
    B, d = 16, 200
    hp = torch.randn(B, d, device="cuda", requires_grad=True)
    att = torch.randn(B, 3, d, device="cuda", requires_grad=True)
    w2, b2 = _small_param(3, d), _small_param(3)
    weighted, aw = sops.adaptive_combine(hp, att, w2, b2)
    gw, ga = torch.randn(B, d, device="cuda").bfloat16(), torch.randn(B, 3, device="cuda")
    torch.autograd.backward([weighted, aw], [gw, ga])
    hp2, att2 = hp.detach().clone().requires_grad_(True), att.detach().clone().requires_grad_(True)
    w22, b22 = w2.detach().clone().requires_grad_(True), b2.detach().clone().requires_grad_(True)
    aw2 = Fn.softmax(Fn.linear(hp2, w22, b22), dim=-1)
    wt2 = (att2 * aw2.unsqueeze(-1)).sum(dim=1)
    torch.autograd.backward([wt2, aw2], [gw.float(), ga])
    assert (aw - aw2).abs().max() <= 1e-5
    assert (weighted.float() - wt2).abs().max() <= 2 ** -7 * max(1.0, float(wt2.abs().max()))
    for got, ref in ((hp.grad, hp2.grad), (att.grad, att2.grad), (w2.grad, w22.grad), (b2.grad, b22.grad)):
        assert (got - ref).abs().max() <= 1e-4 * max(1.0, float(ref.abs().max()))

    lin = torch.nn.Linear(d, 7).cuda()
    lin.weight.grad, lin.bias.grad = torch.zeros_like(lin.weight), torch.zeros_like(lin.bias)
    x = torch.randn(B, d, device="cuda", requires_grad=True)
    y = sops.narrow_linear(x, lin)
    gy = torch.randn(B, 7, device="cuda")
    y.backward(gy)
    x2 = x.detach().clone().requires_grad_(True)
    wr, br = lin.weight.detach().clone().requires_grad_(True), lin.bias.detach().clone().requires_grad_(True)
    y2 = Fn.linear(x2, wr, br)
    y2.backward(gy)
    assert (y - y2).abs().max() <= 1e-4
    for got, ref in ((x.grad, x2.grad), (lin.weight.grad, wr.grad), (lin.bias.grad, br.grad)):
        assert (got - ref).abs().max() <= 1e-4 * max(1.0, float(ref.abs().max()))


def test_stack3_embed_and_rowmask():
    from mmfusion import small_ops as sops
    torch.manual_seed(15)
    B, d = 6, 72
    base = torch.randn(B, 3 * d, device="cuda", requires_grad=True)
    t, a, v = base[:, :d], base[:, d:2 * d], base[:, 2 * d:]
    assert sops.cat3(t, a, v) is base                          # thirds of one buffer: no copy
    emb = _small_param(3, d)
    x = sops.stack3_embed(sops.cat3(t, a, v), emb)
    ref = (torch.stack([t, a, v], dim=1) + emb).reshape(B * 3, d)
    assert (x.float() - ref).abs().max() <= 2 ** -7 * max(1.0, float(ref.abs().max()))
    g = torch.randn(B * 3, d, device="cuda").bfloat16()
    x.backward(g)
    assert torch.equal(base.grad.view(B, 3, d), g.float().view(B, 3, d))
    assert (emb.grad - g.float().view(B, 3, d).sum(0)).abs().max() <= 1e-5
    xm = torch.randn(B, d, device="cuda", requires_grad=True)
    m = (torch.rand(B, device="cuda") > 0.5).float()
    ym = sops.rowmask(xm, m)
    ym.backward(torch.ones_like(ym))
    assert torch.equal(ym, xm.detach() * m[:, None]) and torch.equal(xm.grad, m[:, None].expand(B, d))


@pytest.mark.parametrize("dh", [96, 64])
@pytest.mark.parametrize("Tq,Tk", [(512, 400), (400, 512), (30, 512), (512, 30), (300, 97), (257, 65), (129, 200), (600, 1000),
                                   (1, 1), (33, 31), (64, 64)])
def test_attention_at_mult_shapes(dh, Tq, Tk):
    """The attention kernels (attention2.hip; one generation since round 3) on the MulT sequence shapes and around the
    tile edges: several 128-row query chunks, ragged last tiles whose second 32-key block is pure padding (400 = 6 tiles +
    16 keys), one-block problems (30, 1), 31 / 33 / 64 rows, and the rescale of the running maximum in a late tile."""
    from mmfusion import lib
    B, H = 2, 8
    d = H * dh
    q, kv, do = rnd(B, Tq, d, seed=31), rnd(B, Tk, 2 * d, seed=32), rnd(B, Tq, d, seed=33)
    kv[0, max(Tk - 3, 0), :dh] = q[0, min(Tq - 1, 7), :dh] * 3.0   # a late key far above the rest for one (b, h, query)
    q16 = bf(q).reshape(B * Tq, d).requires_grad_(True)
    kv16 = bf(kv).reshape(B * Tk, 2 * d).requires_grad_(True)
    assert lib.load().mmf_attn_select_impl(1) != 0 and lib.load().mmf_attn_select_impl(2) == 0    # superseded generations are gone
    o = ops.attention_group([ops.AttnSpec(B, Tq, Tk, q=(0, 0), k=(1, 0), v=(1, d))], H, dh, [q16, kv16])[0]
    o.backward(bf(do).reshape(B * Tq, d))
    torch.cuda.synchronize()
    qr = q16.detach().float().cpu().view(B, Tq, d).requires_grad_(True)
    kvr = kv16.detach().float().cpu().view(B, Tk, 2 * d).requires_grad_(True)
    orf = attn_ref(qr, kvr[..., :d], kvr[..., d:], H)
    orf.backward(bf(do).float().cpu())
    assert rel(o.view(B, Tq, d), orf.detach()) < 2 ** -7
    assert rel(q16.grad.view(B, Tq, d), qr.grad) < 2e-2
    assert rel(kv16.grad.view(B, Tk, 2 * d)[..., :d], kvr.grad[..., :d]) < 2e-2
    assert rel(kv16.grad.view(B, Tk, 2 * d)[..., d:], kvr.grad[..., d:]) < 2e-2


# ---- round-2 additions: ADVICE r1 findings ---------------------------------------------------------------------
def test_fused_adamw_device_schedule_many_steps_without_host_sync():
    """the per-step hyper-parameters must not race with a host that runs ahead.  40 optimiser
    steps are enqueued back to back with NO synchronisation — once through the device-side schedule (``advance()``:
    step counter, bias corrections and OneCycle LR computed by a one-thread kernel) and once through the host-side
    ``set_hparams`` (ring of event-guarded pinned buffers, longer than the ring) — against torch.optim.AdamW +
    OneCycleLR (torch's defaults, as the reference: beta1 is cycled with the learning rate) + clip_grad_norm_
    stepping on the CPU.  Early steps are where a late-read buffer would show: the bias correction 1 - beta2^t
    changes 2x-3x per step there."""
    from mmfusion import arena as arena_mod
    from mmfusion.train import FusedAdamW, one_cycle, one_cycle_lr
    steps, total, max_lr = 40, 50, 3e-3
    for mode in ("device", "host"):
        torch.manual_seed(0)
        mod = torch.nn.Sequential(torch.nn.Linear(40, 72), torch.nn.Linear(72, 8)).cuda()
        ref = torch.nn.Sequential(torch.nn.Linear(40, 72), torch.nn.Linear(72, 8))
        ref.load_state_dict({k: v.cpu() for k, v in mod.state_dict().items()})
        ar = arena_mod.ensure(mod)
        opt = FusedAdamW(ar, lr=max_lr, weight_decay=1e-5, max_grad_norm=1.0)
        opt.set_schedule(max_lr, total)
        ropt = torch.optim.AdamW(ref.parameters(), lr=max_lr, weight_decay=1e-5)
        rsch = torch.optim.lr_scheduler.OneCycleLR(ropt, max_lr=max_lr, total_steps=total, pct_start=0.1, anneal_strategy="cos")
        gdev = [[(rnd(*p.shape, seed=1000 * s + i) * (0.2 + 3.0 * (s % 3))).to(DEV) for i, p in enumerate(mod.parameters())]
                for s in range(steps)]
        torch.cuda.synchronize()
        for s in range(steps):                                  # enqueue only
            for p, g in zip(mod.parameters(), gdev[s]):
                p.grad.copy_(g, non_blocking=True)
            if mode == "device":
                opt.advance()
            else:
                lr_s, b1_s = one_cycle(opt.t, total, max_lr)
                opt.set_hparams(lr=lr_s, beta1=b1_s)
            opt.launch()
        for s in range(steps):
            for rp, g in zip(ref.parameters(), gdev[s]):
                rp.grad = g.cpu()
            torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
            ropt.step()
            rsch.step()
        torch.cuda.synchronize()
        assert opt.t == steps and int(opt.step_dev.item()) == steps
        for p, rp in zip(mod.parameters(), ref.parameters()):
            assert rel(p.detach(), rp.detach(), floor=1e-6) < 2e-5, mode
        if mode == "device":
            assert abs(float(opt.hparams[0]) - one_cycle_lr(steps - 1, total, max_lr)) < 1e-9 + 1e-6 * max_lr


@pytest.mark.parametrize("short_before", [False, True])
@pytest.mark.parametrize("lazy", [False, True])
def test_linear_applied_twice_accumulates_both_uses(lazy, short_before):
    """a weight used twice in one forward queues two deferred wgrad problems for ONE gradient
    region; they must not share a grouped launch (non-atomic accumulate) nor run accumulate-before-overwrite under
    lazy zeroing.  Three uses (the lazy race needs three), with and without a bias, against torch autograd.
    short_before: a backward that queues only TWO problems runs first, so the early flush (ops.py, MMF_WGRAD_EARLY)
    fires after the second of the five problems on its own stream and the other three — accumulates into the same two
    regions — are issued from the end-of-backward callback: they must wait for the early launch."""
    from mmfusion import arena as arena_mod
    from mmfusion.ops import W
    torch.manual_seed(0)
    mod = torch.nn.ModuleDict({"a": torch.nn.Linear(128, 128), "b": torch.nn.Linear(128, 128, bias=False)}).cuda()
    ar = arena_mod.ensure(mod)
    x0 = bf(rnd(300, 128, seed=1))

    def run_hip():
        x = x0.clone().requires_grad_(True)
        h = ops.linear(x, W(mod["a"].weight), W(mod["a"].bias))
        h = ops.linear(h, W(mod["b"].weight))
        h = ops.linear(h, W(mod["a"].weight), W(mod["a"].bias))
        h = ops.linear(h, W(mod["b"].weight))
        h = ops.linear(h, W(mod["a"].weight), W(mod["a"].bias))
        (h.float() * 1e-2).sum().backward()
        return x.grad
    ar.zero_grad()
    run_hip()                                   # teaches the arena its wgrad-managed regions
    from mmfusion import ops as ops_mod
    saved_early = ops_mod._WGRAD_EARLY
    ops_mod._WGRAD_EARLY = short_before or saved_early      # the early flush is off by default: exercise it here
    if short_before:
        xs = x0.clone().requires_grad_(True)
        (ops.linear(ops.linear(xs, W(mod["a"].weight), W(mod["a"].bias)), W(mod["b"].weight)).float() * 1e-2).sum().backward()
    if lazy:
        ar.grads.fill_(7.0)                     # stale values a lazy zero must not let through
    ar.zero_grad(lazy=lazy)
    try:
        gx = run_hip()
    finally:
        ops_mod._WGRAD_EARLY = saved_early
    ar.finalize_grads()
    torch.cuda.synchronize()
    wa, ba, wb = (t.detach().to(torch.bfloat16).float().requires_grad_(True) for t in
                  (mod["a"].weight, mod["a"].bias, mod["b"].weight))
    ba = mod["a"].bias.detach().float().clone().requires_grad_(True)
    x = x0.float().clone().requires_grad_(True)
    r = lambda t: t.to(torch.bfloat16).float()          # bf16 storage (the cast's autograd is the identity)
    h = r(x @ wa.t() + ba)
    h = r(h @ wb.t())
    h = r(h @ wa.t() + ba)
    h = r(h @ wb.t())
    h = r(h @ wa.t() + ba)
    (h * 1e-2).sum().backward()
    assert rel(mod["a"].weight.grad, wa.grad) < 2e-2
    assert rel(mod["b"].weight.grad, wb.grad) < 2e-2
    assert rel(mod["a"].bias.grad, ba.grad) < 2e-2
    assert rel(gx, x.grad) < 2e-2


def test_contrastive_fusion_large_batch_branch_matches_small_batch_kernel():
    """ContrastiveFusion with a per-rank batch above the fused InfoNCE kernel's limit (B > 64) takes the torch-op
    branch (models/fusion_layers.py); both branches are the same arithmetic (reference :338-347, 361-375): checked
    against the CPU oracle at B = 80, and the fused kernel against the torch branch at B = 64."""
    import config as cfgmod
    from models import fusion_layers as fl
    from mmfusion import small_ops as sops
    from oracle import ref_cpu
    cfg = cfgmod.ModelConfig()
    cfg.fusion_hidden_size, cfg.fusion_num_heads, cfg.fusion_dropout = 128, 2, 0.0
    torch.manual_seed(4)
    m = fl.ContrastiveFusion(cfg)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    for B in (80, 64):
        xs = [rnd(B, 128, seed=40 + i) for i in range(3)]
        xr = [x.clone().requires_grad_(True) for x in xs]
        ref = ref_cpu.contrastive_fusion(P, "", *xr, cfg.contrastive_temperature, compute_contrastive_loss=True)
        (ref["fused_features"].sum() + sum(ref["contrastive_losses"].values())).backward()
        xg = [x.cuda().requires_grad_(True) for x in xs]
        assert (B > sops.NCE_MAX_B) == (B == 80)
        out = m(*xg, compute_contrastive_loss=True)
        (out["fused_features"].float().sum() + sum(out["contrastive_losses"].values())).backward()
        torch.cuda.synchronize()
        for k in ("text_audio", "text_video", "audio_video"):
            want = float(ref["contrastive_losses"][k])
            assert abs(float(out["contrastive_losses"][k]) - want) <= 1e-2 * max(1.0, abs(want)), (B, k)
        for k in ("text_proj", "audio_proj", "video_proj", "fused_features"):
            assert rel(out[k], ref[k]) < 1e-2, (B, k)
        for g, r_ in zip(xg, xr):
            l2 = float((g.grad.float().cpu() - r_.grad).norm() / r_.grad.norm())
            assert l2 < 0.12, (B, l2)
        for p in P.values():
            p.grad = None


@pytest.mark.parametrize("n", [2, 5, 7, 11])
def test_fanout_sums_all_gradients_in_one_pass(n):
    """ops.fanout: n handles on one bf16 tensor; the backward sums the n gradients with mmf_addn_bf16 (f32 accumulate,
    one rounding) — compared with the f32 sum of the same bf16 gradients; n = 11 exercises the chunked form (> 8)."""
    x = bf(rnd(333, 200, seed=1)).requires_grad_(True)            # 66,600 elements: vector body + scalar tail
    hs = ops.fanout(x, n)
    assert len(hs) == n and all(h.data_ptr() == x.data_ptr() for h in hs)
    gs = [bf(rnd(333, 200, seed=10 + i)) for i in range(n)]
    torch.autograd.backward(hs, gs)
    want = sum(g.float() for g in gs)
    assert x.grad.dtype == torch.bfloat16
    tol = 2 ** -8 if n <= 8 else 2 ** -7                           # the chunked form rounds the running sum once more
    assert rel(x.grad, want) < tol
    if n <= 8:
        assert torch.equal(x.grad, want.to(torch.bfloat16))        # exactly one rounding of the f32 sum
    y = bf(rnd(8, 8, seed=2))                                      # no gradient wanted: plain aliases, no node
    assert all(h is y for h in ops.fanout(y, 3))


def test_fanout_group_sums_per_tensor_and_takes_odd_gradients():
    """ops.fanout_group: m tensors x n handles, one grouped launch in backward (mmf_addn_grouped); a gradient the kernel cannot
    take (a view starting at an odd element -> pointer not 16-byte aligned) is summed with stock adds instead of raising from
    inside backward."""
    xs = [bf(rnd(64, 40, seed=s)).requires_grad_(True) for s in (1, 2, 3)]
    groups = ops.fanout_group(xs, 3)
    heads, grads, want = [], [], []
    for i, hs in enumerate(groups):
        gs = [bf(rnd(64, 40, seed=20 + 3 * i + k)) for k in range(3)]
        if i == 1:                                                 # the middle tensor's first gradient: storage offset 1 element
            odd = torch.empty(64 * 40 + 1, dtype=torch.bfloat16, device=gs[0].device)[1:].view(64, 40)
            odd.copy_(gs[0])
            assert odd.data_ptr() % 16 != 0
            gs[0] = odd
        heads += hs
        grads += gs
        want.append(sum(g.float() for g in gs).to(torch.bfloat16))
    torch.autograd.backward(heads, grads)
    for x, w in zip(xs, want):
        assert torch.equal(x.grad, w)
