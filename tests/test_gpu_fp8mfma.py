"""-m gpu: the fp8-MFMA path (dtype='fp8mfma', BASELINE configs[4] "fp8 weights (CDNA4 fp8 MFMA)").

fp8 e4m3 has 3 mantissa bits: the path CANNOT meet the 1e-3 bar, so the tests
  * pin the kernels exactly: e4m3 x e4m3 products are exact in fp32, so sgpt_linear_fp8 must equal a PyTorch fp32
    product of the de-quantised operands up to accumulation order, and the quantising LayerNorm must produce the
    very codes torch.float8_e4m3fn produces;
  * report the model-level deviation (max |dcos|, top-10 overlap) against the reference fixtures, next to the
    fp8-storage / bf16-arithmetic mode, under stated budgets."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, build_model, load_case, maxabs
from oracle import sgpt_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from sgpt_amd import get_context
    return get_context("cuda:0")


def gelu_new(u):
    return 0.5 * u * (1.0 + torch.tanh(0.7978845608028654 * (u + 0.044715 * u ** 3)))


def dq(codes, scale):
    return codes.view(torch.float8_e4m3fn).float() * scale[:, None]


@pytest.mark.parametrize("M,N,K", [(4096, 3072, 768), (4096, 768, 3072), (256, 256, 256), (512, 2048, 256), (2048, 256, 512)])
def test_linear_fp8_equals_fp32_product_of_dequantised_operands(ctx, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn((M, K), generator=g, device="cuda") * torch.exp(torch.randn((M, 1), generator=g, device="cuda"))
    w = torch.randn((N, K), generator=g, device="cuda") / math.sqrt(K)
    bias = torch.randn((N,), generator=g, device="cuda") * 0.3
    resid = torch.randn((M, N), generator=g, device="cuda") * 2
    a8, sa = ctx.fp8_quantize_rows(a)
    w8, sw = ctx.fp8_quantize_rows(w)
    acc = dq(a8, sa) @ dq(w8, sw).T                        # exact products, fp32 accumulation
    tol = 1e-3 * math.sqrt(K / 64) * float(acc.abs().max())
    x = resid.clone()
    ctx._chk(ctx.lib.sgpt_linear_fp8(ctx.handle, 2, 0, a8.data_ptr(), sa.data_ptr(), 1.0, w8.data_ptr(), sw.data_ptr(), bias.data_ptr(),
                                     x.data_ptr(), x.data_ptr(), 1.0, M, N, K, None), "sgpt_linear_fp8")     # in place
    assert float((x - (resid + acc + bias)).abs().max()) < tol
    # per-tensor A scale instead of per-row scales (the GELU-output operand of the second MLP projection)
    a8t = (a / 0.25).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    acc_t = (a8t.view(torch.float8_e4m3fn).float() * 0.25) @ dq(w8, sw).T
    y = ctx.linear_fp8(a8t, w8, sw, bias, "resid", a_scalar=0.25, resid=resid)
    assert float((y - (resid + acc_t + bias)).abs().max()) < 1e-3 * math.sqrt(K / 64) * float(acc_t.abs().max())
    # bias + gelu -> e4m3 codes under an output scale
    want = gelu_new(acc + bias)
    s_out = 2.0 ** math.ceil(math.log2(float(want.abs().max()) * 1.5 / 448))
    h8 = ctx.linear_fp8(a8, w8, sw, bias, "gelu", a_scale=sa, out_scale=s_out)
    assert ctx.range_check() == 0
    got = h8.view(torch.float8_e4m3fn).float() * s_out
    ref8 = (want / s_out).clamp(-448, 448).to(torch.float8_e4m3fn)
    same = float((h8 == ref8.view(torch.uint8)).float().mean())
    # one e4m3 step = 2^-3 relative (2^-9 * s_out absolute for subnormals); accumulation noise may flip a rounding
    step = torch.maximum(want.abs() * 2.0 ** -3, torch.tensor(2.0 ** -9 * s_out, device="cuda"))
    assert float(((got - want).abs() - step).max()) < tol and same > 0.98, same
    # 16-bit store epilogues of the attention projections: row-major (+ bias) and transposed (V^T)
    for odt, ulp in ((torch.bfloat16, 2.0 ** -8), (torch.float16, 2.0 ** -11)):
        st = ctx.linear_fp8(a8, w8, sw, bias, "store", a_scale=sa, out_dtype=odt)
        assert float(((st.float() - (acc + bias)).abs() - ulp * (acc + bias).abs()).max()) < tol
        vt = ctx.linear_fp8(a8, w8, sw, None, "vt", a_scale=sa, out_dtype=odt)
        assert vt.shape == (N, M) and float(((vt.float() - acc.T).abs() - ulp * acc.T.abs()).max()) < tol
    # saturation raises bit 1 of the range flag
    ctx.linear_fp8(a8, w8, sw, bias + 100.0, "gelu", a_scale=sa, out_scale=2.0 ** -8)
    assert ctx.range_check() == 2 and ctx.range_check() == 0


def test_linear_fp8_layout_probe(ctx):
    """Small-integer codes with an ASYMMETRIC weight pattern: every product and sum is exact, so any swapped lane / byte /
    k-chunk map of the 16x16x128 fragments or of the epilogue fails bit-exactly."""
    M, N, K = 512, 768, 512
    ai = (torch.arange(M, device="cuda")[:, None] * 5 + torch.arange(K, device="cuda")[None, :] * 3) % 9 - 4
    wi = (torch.arange(N, device="cuda")[:, None] * 7 + torch.arange(K, device="cuda")[None, :] * 11) % 13 - 6
    a8 = ai.float().to(torch.float8_e4m3fn).view(torch.uint8)
    w8 = wi.float().to(torch.float8_e4m3fn).view(torch.uint8)
    one_n, zero_n = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    out = ctx.linear_fp8(a8, w8, one_n, zero_n, "resid", resid=torch.zeros((M, N), device="cuda"))
    assert torch.equal(out, (ai.float() @ wi.float().T))


@pytest.mark.parametrize("d", [256, 768, 4096])
def test_layernorm_fp8_codes_equal_torch(ctx, d):
    g = torch.Generator(device="cuda").manual_seed(d)
    x = torch.randn((1000, d), generator=g, device="cuda") * 3 + 0.5
    gamma = 1 + 0.1 * torch.randn((d,), generator=g, device="cuda")
    beta = 0.05 * torch.randn((d,), generator=g, device="cuda")
    codes, scale = ctx.layernorm_fp8(x, gamma, beta, 1e-5)
    y = torch.nn.functional.layer_norm(x, (d,), gamma, beta, 1e-5)
    amax = y.abs().amax(dim=1)
    assert torch.all(torch.log2(scale) == torch.round(torch.log2(scale)))                    # powers of two
    assert torch.all(amax / scale <= 448 * (1 + 1e-5)) and torch.all(amax / scale > 223.9)   # the tightest one
    dec = codes.view(torch.float8_e4m3fn).float() * scale[:, None]
    # LN arithmetic differs in the last fp32 bits between the kernel and torch: codes may differ where y / scale sits on a
    # rounding boundary; every decoded value must still be within one e4m3 step of torch's LN output
    step = torch.maximum(y.abs() * 2.0 ** -3, scale[:, None] * 2.0 ** -9)
    assert float(((dec - y).abs() - 0.5001 * step).max()) < 1e-5 * float(y.abs().max())
    ref = (y / scale[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert float((codes == ref).float().mean()) > 0.999


@pytest.mark.parametrize("tag", ["tiny_dh128", "cfg1_125m_32x64", "tiny_bloom_right"])
def test_fp8mfma_model_vs_reference(tag):
    """Whole-model deviation of the fp8-MFMA mode against the reference golden vectors, next to the fp8-STORAGE mode
    (same e4m3 weights, bf16 arithmetic).  Measured (round 2): the e4m3 WEIGHTS dominate -- fp8 storage alone gives
    max |dcos| 1.8e-2 / 2.0e-2 / 3.2e-3 on the three cases, fp8 MFMA on the MLP 1.4e-2 / 1.6e-2 / 4.4e-3; raw embeddings
    within 0.23-0.34 of O(1..3) values, row cosine >= 0.9923.  Budgets at ~1.5x: 0.5 / 0.99 / 3e-2 (reported, not the gate:
    3 mantissa bits are far outside the 1e-3 bar by construction) and fp8 MFMA at most 1.5x the storage-only deviation."""
    fx, cfg_kw, seqs, pad_left, ids, mask = load_case(tag)
    ref = fx["emb_weightedmean"]
    out = {}
    for dtype in ("fp8", "fp8mfma"):
        m = build_model(cfg_kw, int(fx["seed"]), float(fx["std"]), dtype)
        got = m.encode_ids(seqs, mode="weightedmean", pad_left=pad_left).cpu().numpy()
        rc = (got * ref).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(ref, axis=1))
        out[dtype] = (got, maxabs(got, ref), float(rc.min()), maxabs(O.cos_sim(got, got), O.cos_sim(ref, ref)))
        print(f"{tag} {dtype}: max|emb-ref| = {out[dtype][1]:.3e}, min row cosine = {out[dtype][2]:.5f}, max|dcos| = {out[dtype][3]:.3e}")
    m8 = build_model(cfg_kw, int(fx["seed"]), float(fx["std"]), "fp8mfma")
    assert m8.act_scales is not None and np.all(np.log2(m8.act_scales) == np.round(np.log2(m8.act_scales)))
    got, err, rcmin, dcos = out["fp8mfma"]
    assert np.isfinite(got).all() and err < 0.5 and rcmin > 0.99 and dcos < 3e-2
    assert dcos < 1.5 * out["fp8"][3] + 1e-3
    assert not np.array_equal(got, out["fp8"][0])          # the MLP really ran on different arithmetic


def test_fp8mfma_cfg2_ranking_report():
    """configs[1]-size ranked comparison against the reference stack: fp8 MFMA on the MLP, 1024 documents x 128 tokens."""
    fx = np.load(os.path.join(GOLDEN, "cfg2_125m_1024x128.npz"))
    from sgpt_amd import get_context
    ctx = get_context("cuda:0")
    m = build_model(dict(O.SGPT_125M), 1, 0.02, "fp8mfma")
    m.max_tokens_per_call = 1024 * 128
    d_emb = m.encode_ids(fx["doc_ids"].astype(np.int64))
    q_emb = m.encode_ids([fx["query_ids"][i, :n].tolist() for i, n in enumerate(fx["query_lens"].tolist())])
    dn, qn = ctx.l2_normalize(d_emb), ctx.l2_normalize(q_emb)
    cos = ctx.scores(ctx._operand(qn, torch.bfloat16), ctx._operand(dn, torch.bfloat16), dtype=torch.bfloat16).cpu().numpy()
    val, idx, _ = ctx.score_topk(ctx._operand(qn, torch.bfloat16), ctx._operand(dn, torch.bfloat16), 10, dtype=torch.bfloat16)
    idx = idx.cpu().numpy()
    overlap = [len(set(idx[i].tolist()) & set(fx["top10"][i].tolist())) for i in range(idx.shape[0])]
    dev = maxabs(cos, fx["cos"])
    print(f"cfg2 fp8mfma: max|cos-ref| = {dev:.3e}, top-10 overlap mean {np.mean(overlap):.2f} / 10 (min {min(overlap)}), "
          f"top-1 agreement {float(np.mean(idx[:, 0] == fx['top10'][:, 0])):.2f}")
    assert np.isfinite(cos).all() and dev < 3e-2 and np.mean(overlap) > 7.0
