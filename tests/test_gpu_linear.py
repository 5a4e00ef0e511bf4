"""-m gpu: the projection GEMM with its fused epilogues (sgpt_linear = the kernels sgpt_encode launches), directly,
against a plain PyTorch fp32 product of the SAME 16-bit operands: both 256x256 LDS-DMA
kernels (16x16x32 and 32x32x16 MFMA), bf16 and f16, the ring parities K = 128 / 192 / 320 / 768 / 3072, M = 256
(single tile row) and the register-staged small-tile kernels.  Tolerance: fp32 accumulation-order noise only,
1e-3 * sqrt(K / 64) relative to the output scale for fp32 outputs; one rounding to the 16-bit output format on top
(2^-8 bf16 / 2^-11 f16 relative) for 16-bit outputs."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HALF = {"bf16": torch.bfloat16, "f16": torch.float16}
ULP = {"bf16": 2.0 ** -8, "f16": 2.0 ** -11}


@pytest.fixture(scope="module")
def ctx():
    from sgpt_amd import get_context
    return get_context("cuda:0")


@pytest.fixture(params=[0, 2], ids=["default-tiles", "forced256"])
def variant(request, ctx):
    """2: 256x256 LDS-DMA tiles even when they leave most CUs idle (single-tile cases); 0: the default tile policy."""
    old = ctx.set_tile_policy(request.param == 2)
    yield request.param
    ctx.set_tile_policy(old)


def gelu_new(u):
    return 0.5 * u * (1.0 + torch.tanh(0.7978845608028654 * (u + 0.044715 * u ** 3)))


def operands(M, N, K, dt, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn((M, K), generator=g, device="cuda").to(HALF[dt])
    w = (torch.randn((N, K), generator=g, device="cuda") * (1.0 / math.sqrt(K))).to(HALF[dt])
    bias = torch.randn((N,), generator=g, device="cuda") * 0.3
    resid = torch.randn((M, N), generator=g, device="cuda") * 2.0
    return a, w, bias, resid


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(4096, 2304, 128), (4096, 2304, 192), (2304, 4096, 320), (8192, 768, 768),
                                   (2048, 768, 3072), (256, 768, 768), (512, 256, 128), (131072 // 8, 1536, 768)])
def test_linear_epilogues_vs_torch_fp32(ctx, variant, dt, M, N, K):
    a, w, bias, resid = operands(M, N, K, dt, seed=M + N + K)
    acc = a.float() @ w.float().T                                    # fp32 reference of the same operands
    scale = float(acc.abs().max())
    tol32 = 1e-3 * math.sqrt(K / 64) * scale
    # bias + residual, in place on the residual stream (out aliases resid)
    x = resid.clone()
    ctx._chk(ctx.lib.sgpt_linear(ctx.handle, 1 if dt == "bf16" else 3, 2, 0, a.data_ptr(), w.data_ptr(), bias.data_ptr(),
                                 x.data_ptr(), x.data_ptr(), M, N, K, None), "sgpt_linear")
    want = resid + acc + bias
    assert torch.isfinite(x).all()
    assert float((x - want).abs().max()) < tol32 + 1e-6 * float(want.abs().max()), "bias + residual"
    # bias + gelu_new -> 16-bit
    h = ctx.linear(a, w, bias, epi="gelu")
    want = gelu_new(acc + bias)
    err = (h.float() - want).abs()
    assert float((err - ULP[dt] * want.abs()).max()) < tol32 + 2e-3 * (1.0 if dt == "bf16" else 0.25), "bias + gelu (fast sigmoid form)"
    # plain store (+ bias) -> 16-bit, and the transposed V^T store
    s = ctx.linear(a, w, bias, epi="store")
    want = acc + bias
    assert float(((s.float() - want).abs() - ULP[dt] * want.abs()).max()) < tol32
    if M % 128 == 0:
        vt = ctx.linear(a, w, None, epi="vt")
        assert vt.shape == (N, M)
        assert float(((vt.float() - acc.T).abs() - ULP[dt] * acc.T.abs()).max()) < tol32
        vtb = ctx.linear(a, w, bias, epi="vt")
        want = (acc + bias).T
        assert float(((vtb.float() - want).abs() - ULP[dt] * want.abs()).max()) < tol32


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_linear_layout_is_not_transposed(ctx, variant, dt):
    """A = I-like probe with an ASYMMETRIC weight matrix: a swapped row/column or k-slot map cannot pass."""
    M = N = 4096
    K = 256
    a = torch.zeros((M, K), device="cuda")
    a[torch.arange(M), torch.arange(M) % K] = 1.0
    w = (torch.arange(N, device="cuda")[:, None] * 3 + torch.arange(K, device="cuda")[None, :] * 7) % 61 - 30.0
    out = ctx.linear(a.to(HALF[dt]), w.to(HALF[dt]), None, epi="store", out_dtype=HALF[dt])
    want = w[:, torch.arange(M, device="cuda") % K].T                # out[m][n] = w[n][m % K], small integers: exact
    assert torch.equal(out.float(), want)


def test_linear_tile_policies_agree_bitwise(ctx):
    """The 256x256 LDS-DMA kernel and the register-staged 128x128 / 64x64 one feed every output element the same MFMA
    sequence: identical bits whichever the tile policy picks."""
    a, w, bias, resid = operands(1024, 768, 768, "f16", seed=5)
    outs = []
    for force in (False, True):
        old = ctx.set_tile_policy(force)
        outs.append((ctx.linear(a, w, bias, epi="resid", resid=resid), ctx.linear(a, w, bias, epi="gelu"),
                     ctx.linear(a, w, None, epi="store"), ctx.linear(a, w, None, epi="vt")))
        ctx.set_tile_policy(old)
    assert all(torch.equal(u, v) for u, v in zip(*outs))


def test_f16_range_flag_raised_by_store_epilogue(ctx):
    """|v| >= 32768 rounded to f16 raises the context's range flag; bf16 outputs of the same values do not."""
    a = torch.full((4096, 128), 16.0, device="cuda").to(torch.float16)
    w = torch.full((2304, 128), 16.0, device="cuda").to(torch.float16)           # acc = 128 * 256 = 32768
    assert not ctx.range_check()
    ctx.linear(a, w, None, epi="store")
    assert ctx.range_check()                                                     # and it resets
    assert not ctx.range_check()
    ctx.linear((a.float() * 0.5).to(torch.float16), w, None, epi="store")         # 16384: inside the guard band
    assert not ctx.range_check()
    ctx.linear(a.to(torch.bfloat16), w.to(torch.bfloat16), None, epi="store")
    assert not ctx.range_check()
    ctx.linear(a[:256], w[:256], None, epi="store")                               # the small-tile kernel tracks it too
    assert ctx.range_check()
    ctx.linear(a[:256], w[:256], None, epi="vt")
    assert ctx.range_check()


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(256, 768, 3072), (512, 768, 3072), (512, 768, 768), (512, 3072, 768), (128, 1536, 768),
                                   (3328, 768, 3072), (3328, 768, 768), (1024, 768, 2048 + 64)])
def test_k_group_launches_are_deterministic_and_match_the_unsplit_kernel(ctx, dt, M, N, K):
    """Query-sized launches (fewer 64x64 tiles than CUs, long k: >= 6 steps of 128 elements per group) run 2 k-groups per
    workgroup in the opt-in low-latency mode, each walking its own part of the k range; group 0 adds the fp32 accumulators in group order (no atomics) and runs the epilogue.  So:
    identical bits from run to run, and agreement with the un-split 256x256 kernel to accumulation-order noise, for
    every epilogue, in place on the residual stream included.  (M = 3328: too many tiles, one group -- the control.)"""
    a, w, bias, resid = operands(M, N, K, dt, seed=7 * M + N + K)
    acc = a.float() @ w.float().T
    scale = float(acc.abs().max())
    tol32 = 1e-3 * math.sqrt(K / 64) * scale
    old_kg = ctx.set_low_latency(False)                             # mode off (the default): one group
    try:
        off = ctx.linear(a, w, bias, epi="resid", resid=resid)
        ctx.set_low_latency(True)
        _k_group_checks(ctx, dt, M, N, K, a, w, bias, resid, acc, tol32, off)
    finally:
        ctx.set_low_latency(old_kg)


def _k_group_checks(ctx, dt, M, N, K, a, w, bias, resid, acc, tol32, off):
    runs = []
    for rep in range(3):
        x = resid.clone()
        ctx._chk(ctx.lib.sgpt_linear(ctx.handle, 1 if dt == "bf16" else 3, 2, 0, a.data_ptr(), w.data_ptr(), bias.data_ptr(),
                                     x.data_ptr(), x.data_ptr(), M, N, K, None), "sgpt_linear")
        h = ctx.linear(a, w, bias, epi="gelu")
        s = ctx.linear(a, w, bias, epi="store")
        vt = ctx.linear(a, w, bias, epi="vt") if M % 128 == 0 else s
        runs.append((x, h, s, vt))
    for r in runs[1:]:
        assert all(torch.equal(u, v) for u, v in zip(runs[0], r)), "k-group result changed between identical launches"
    x, h, s, vt = runs[0]
    assert float((x - (resid + acc + bias)).abs().max()) < tol32 + 1e-6 * float(resid.abs().max())
    want = gelu_new(acc + bias)
    assert float(((h.float() - want).abs() - ULP[dt] * want.abs()).max()) < tol32 + 2e-3 * (1.0 if dt == "bf16" else 0.25)
    want = acc + bias
    assert float(((s.float() - want).abs() - ULP[dt] * want.abs()).max()) < tol32
    if M % 128 == 0:
        assert float(((vt.float() - want.T).abs() - ULP[dt] * want.T.abs()).max()) < tol32
    if M % 256 == 0 and N % 256 == 0 and K % 64 == 0:
        old = ctx.set_tile_policy(True)                              # 256x256 tiles, one workgroup per tile, k ascending
        ref = ctx.linear(a, w, bias, epi="resid", resid=resid)
        ctx.set_tile_policy(old)
        assert float((x - ref).abs().max()) < 2e-5 * float(ref.abs().max())
        assert torch.equal(off, ref), "with the mode off every kernel produces the k-ascending sum, bit for bit"
        # launch rule (gemm.hip::launch): at most 512 tiles of 64x64, an even number of 128-element k-steps, >= 6 per group
        # -- K = 3072; K = 768 (3 steps per group) and K = 2112 (17 steps) stay one group, like M = 3328: the controls
        nk16 = (K + 127) // 128
        if M <= 1024 and nk16 % 2 == 0 and nk16 // 2 >= 6:
            assert not torch.equal(x, ref), "the k-group path did not run (the test would be vacuous)"
        else:
            assert torch.equal(x, ref), "one group: the k-ascending sum, bit for bit"


# ---------------------------------------------------------------- query- / mid-sized projections (csrc/qgemm.hip, round 6) -----
QSHAPES = [(32, 768, 768), (32, 768, 3072), (96, 2304, 768), (352, 768, 3072), (512, 3072, 768), (1024, 768, 768), (2304, 768, 3072),
           (2816, 2304, 768), (160, 1024, 1024), (64, 4096, 4096), (448, 2048, 8192)]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", QSHAPES)
def test_query_projections_vs_torch_fp32_and_vs_the_bulk_kernels(ctx, dt, M, N, K):
    """sgpt_linear_query (register-staged deep-prefetch tiles of 32 .. 128 rows) against the fp32 product of the same 16-bit
    operands, every epilogue -- and bit for bit against sgpt_linear (the bulk path's kernels) on the same operands: every kernel
    feeds an output element the k-ascending MFMA chain."""
    a, w, bias, resid = operands(M, N, K, dt, seed=M + N + K + 1)
    acc = a.float() @ w.float().T
    scale = float(acc.abs().max())
    tol32 = 1e-3 * math.sqrt(K / 64) * scale
    tol16 = tol32 + ULP[dt] * scale
    got = ctx.linear_query(w, a=a, bias=bias, epi="resid", resid=resid)
    assert float((got - (resid + acc + bias)).abs().max()) < tol32 + 1e-6 * float((resid + acc + bias).abs().max())
    assert torch.equal(got, ctx.linear(a, w, bias, "resid", resid))
    got = ctx.linear_query(w, a=a, bias=bias, epi="gelu")
    want = gelu_new(acc + bias)
    assert float((got.float() - want).abs().max()) < tol16 + ULP[dt] * float(want.abs().max())
    assert torch.equal(got, ctx.linear(a, w, bias, "gelu"))
    got = ctx.linear_query(w, a=a, epi="store")
    assert float((got.float() - acc).abs().max()) < tol16
    assert torch.equal(got, ctx.linear(a, w, None, "store"))
    if N % 96 == 0:                                                      # q | k | V^T in one launch (n_split = 2 N / 3)
        ns = N // 3 * 2
        qk, vt = ctx.linear_query(w, a=a, epi="qkv", n_split=ns)
        assert float((qk.float() - acc[:, :ns]).abs().max()) < tol16 and float((vt.float() - acc[:, ns:].T).abs().max()) < tol16
        assert torch.equal(qk, ctx.linear(a, w[:ns].contiguous(), None, "store"))
        assert torch.equal(vt, ctx.linear(a, w[ns:].contiguous(), None, "store").T.contiguous())


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,d", [(32, 768), (96, 768), (352, 768), (1024, 768), (2816, 768), (64, 1024), (640, 1024), (128, 512)])
def test_query_projections_with_layernorm_prologue_vs_torch(ctx, dt, M, d):
    """The LayerNorm prologue (LN1 -> QKV, LN2 -> fc1 + GELU inside the projection): against torch's fp32 LayerNorm rounded to the
    operand format followed by the fp32 product.  The kernel's LayerNorm differs from torch's in the last fp32 bits, so a few
    normalised values land on the other side of a 16-bit rounding boundary: tolerance = one operand ulp carried through the
    product (sqrt(K) random-sign terms) on top of the accumulation-order noise."""
    g = torch.Generator(device="cuda").manual_seed(M + d)
    x = torch.randn((M, d), generator=g, device="cuda") * 3.0 + 0.5
    gamma = 1.0 + 0.1 * torch.randn((d,), generator=g, device="cuda")
    beta = 0.1 * torch.randn((d,), generator=g, device="cuda")
    a = torch.nn.functional.layer_norm(x, (d,), gamma, beta, 1e-5).to(HALF[dt])
    for epi, N in (("qkv", 3 * d), ("gelu", 4 * d)):
        w = (torch.randn((N, d), generator=g, device="cuda") * (1.0 / math.sqrt(d))).to(HALF[dt])
        bias = torch.randn((N,), generator=g, device="cuda") * 0.3
        acc = a.float() @ w.float().T
        scale = float(acc.abs().max())
        tol = 1e-3 * math.sqrt(d / 64) * scale + 2 * ULP[dt] * scale + 4 * ULP[dt] * float(a.float().abs().max())
        if epi == "qkv":
            qk, vt = ctx.linear_query(w, x=x, ln=(gamma, beta, 1e-5), epi="qkv", n_split=2 * d)
            assert torch.isfinite(qk.float()).all() and torch.isfinite(vt.float()).all()
            assert float((qk.float() - acc[:, : 2 * d]).abs().max()) < tol and float((vt.float() - acc[:, 2 * d:].T).abs().max()) < tol
        else:
            got = ctx.linear_query(w, x=x, ln=(gamma, beta, 1e-5), bias=bias, epi="gelu")
            want = gelu_new(acc + bias)
            assert float((got.float() - want).abs().max()) < tol + ULP[dt] * float(want.abs().max())


def test_query_projection_refuses_shapes_it_does_not_serve(ctx):
    a, w, bias, resid = operands(48, 768, 768, "f16", seed=1)          # M % 32 != 0
    with pytest.raises(ValueError, match="not served"):                 # (SGPT_ERR_INVALID surfaces as ValueError, like every entry)
        ctx.linear_query(w, a=a, epi="store")
    a, w, bias, resid = operands(32, 768, 640, "f16", seed=2)          # K / 128 = 5
    with pytest.raises(ValueError, match="not served"):
        ctx.linear_query(w, a=a, epi="store")
