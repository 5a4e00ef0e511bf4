"""-m gpu: each HIP kernel through the C ABI against the numpy oracle (small sizes) and against a
plain PyTorch fp32 reference of the same op on the GPU (large sizes)."""
import os

import numpy as np
import pytest
import torch

from oracle import sgpt_oracle as O
from helpers import GOLDEN, load_case, maxabs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from sgpt_amd import get_context
    return get_context("cuda:0")


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0").to(dtype)


# ---------------------------------------------------------------- GEMM (scores) ----------
@pytest.mark.parametrize("na,nb,d", [(16, 16, 32), (50, 37, 100), (130, 257, 768), (1, 5, 4), (300, 1000, 2048)])
def test_scores_fp32_vs_numpy(ctx, na, nb, d):
    rng = np.random.default_rng(na * 7 + nb)
    a = rng.standard_normal((na, d)).astype(np.float32)
    b = rng.standard_normal((nb, d)).astype(np.float32)
    b[:, 0] += 3.0                                       # asymmetric: catches an output transpose
    got = ctx.scores(dev(a), dev(b)).cpu().numpy()
    want = a.astype(np.float64) @ b.astype(np.float64).T
    assert got.shape == (na, nb)
    assert maxabs(got, want) < 2e-4 * np.sqrt(d / 100)  # fp32 fma chain vs fp64


@pytest.mark.parametrize("na,nb,d", [(16, 16, 64), (130, 257, 768), (77, 1001, 1024)])
def test_scores_bf16_vs_torch(ctx, na, nb, d):
    g = torch.Generator(device="cpu").manual_seed(na + nb)
    a = torch.randn(na, d, generator=g).to(torch.bfloat16).cuda()
    b = torch.randn(nb, d, generator=g).to(torch.bfloat16).cuda()
    got = ctx.scores(a, b, dtype=torch.bfloat16)
    want = a.float() @ b.float().T                       # exact products of bf16 inputs, fp32 accumulate
    assert torch.max(torch.abs(got - want)).item() < 1e-3 * np.sqrt(d / 64)


def test_scores_identity_layout(ctx):
    """A = I against an asymmetric B (cdna guide: transpose-detecting check)."""
    n = 128
    b = (torch.arange(n * 64, dtype=torch.float32).reshape(n, 64) % 97).cuda()
    eye = torch.eye(64, dtype=torch.float32).cuda()
    got = ctx.scores(eye, b)                             # [64, n] = B^T
    assert torch.equal(got, b.T.contiguous())
    gotb = ctx.scores(eye.to(torch.bfloat16), b.to(torch.bfloat16), dtype=torch.bfloat16)
    assert torch.equal(gotb, b.T.contiguous())           # small integers are exact in bf16


def test_cos_sim_dot_score_golden(ctx):
    """Reference outputs of util.cos_sim / dot_score / normalize_embeddings (scoring.npz)."""
    from sgpt_amd import util
    fx = np.load(f"{GOLDEN}/scoring.npz")
    assert maxabs(util.cos_sim(fx["a"], fx["b"]).numpy(), fx["cos"]) < 1e-5
    assert maxabs(util.dot_score(fx["a"], fx["b"]).numpy(), fx["dot"]) < 1e-4
    assert maxabs(util.normalize_embeddings(torch.from_numpy(fx["a"])).numpy(), fx["nrm"]) < 1e-6
    assert util.cos_sim(fx["a"][0], fx["b"]).shape == (1, 37)          # 1-D promoted (util.py:35-39)
    out = util.cos_sim(torch.from_numpy(fx["a"]).cuda(), torch.from_numpy(fx["b"]).cuda())
    assert out.is_cuda


def test_ref_test_util_cos_sim_vs_sklearn(ctx):
    """sentence-transformers/tests/test_util.py:21-30 on the HIP scorer."""
    from sklearn.metrics.pairwise import cosine_similarity
    from sgpt_amd import util
    rng = np.random.default_rng(1)
    a, b = rng.standard_normal((50, 100)), rng.standard_normal((50, 100))
    assert np.abs(cosine_similarity(a, b) - util.pytorch_cos_sim(a, b).numpy()).max() < 1e-3


def test_ref_test_util_pairwise_scores(ctx):
    """sentence-transformers/tests/test_util.py:69-76 (pairwise_cos_sim / pairwise_dot_score vs sklearn's paired distances) and the
    reference outputs of scoring.npz, through sgpt_pairwise_scores."""
    from sklearn.metrics.pairwise import paired_cosine_distances
    from sgpt_amd import util
    rng = np.random.default_rng(3)
    a, b = rng.standard_normal((50, 100)), rng.standard_normal((50, 100))
    assert np.allclose(1 - paired_cosine_distances(a, b), util.pairwise_cos_sim(a, b).numpy(), atol=1e-6)
    assert np.allclose((a * b).sum(-1), util.pairwise_dot_score(a, b).numpy(), atol=1e-4)
    fx = np.load(f"{GOLDEN}/scoring.npz")
    assert maxabs(util.pairwise_cos_sim(fx["a"][:37], fx["b"]).numpy(), fx["pcs"]) < 1e-6
    at, bt = torch.from_numpy(fx["a"][:37]).cuda(), torch.from_numpy(fx["b"]).cuda()
    out = util.pairwise_dot_score(at, bt)
    assert out.is_cuda and out.shape == (37,) and maxabs(out.cpu().numpy(), (fx["a"][:37] * fx["b"]).sum(-1)) < 1e-4
    z = torch.zeros(3, 768)
    assert torch.equal(util.pairwise_cos_sim(z, z), torch.zeros(3))            # zero rows: x / max(|x|, 1e-12) = 0, no NaN
    with pytest.raises(ValueError):
        util.pairwise_cos_sim(torch.zeros(3, 8), torch.zeros(4, 8))


def test_ref_test_util_normalize(ctx):
    """tests/test_util.py:9-18."""
    from sgpt_amd import util
    a = torch.tensor(np.random.default_rng(0).standard_normal((50, 100)))
    for e in util.normalize_embeddings(a):
        assert len(e) == 100 and abs(torch.norm(e).item() - 1) < 1e-4


# ---------------------------------------------------------------- pooling -----------------
@pytest.mark.parametrize("tag", ["tiny_right", "tiny_left", "tiny_dh128"])
@pytest.mark.parametrize("mode", ["weightedmean", "mean", "lasttoken"])
def test_pool_golden(ctx, tag, mode):
    """Reference Pooling.py outputs on HF hidden states (golden), through sgpt_pool."""
    fx, _, _, _, _, mask = load_case(tag)
    got = ctx.pool(dev(fx["last_hidden"]), dev(mask, torch.int32), mode).cpu().numpy()
    assert maxabs(got, fx[f"emb_{mode}"]) < 2e-6
    assert maxabs(got, O.pool(fx["last_hidden"], mask, mode)) < 2e-6


def test_pool_edge_cases(ctx):
    rng = np.random.default_rng(3)
    h = rng.standard_normal((5, 33, 772)).astype(np.float32)       # d % 256 != 0, ragged masks
    mask = np.zeros((5, 33), np.int64)
    mask[0, :33] = 1; mask[1, :1] = 1; mask[2, 10:20] = 1; mask[3, 32:] = 1       # row 4: all padding
    for mode in ("weightedmean", "mean"):
        got = ctx.pool(dev(h), dev(mask, torch.int32), mode).cpu().numpy()
        assert maxabs(got, O.pool(h, mask, mode)) < 5e-6
        assert np.all(got[4] == 0)                                   # clamp(1e-9): 0/1e-9 = 0 (Pooling.py:122)
    hb = torch.from_numpy(h).to(torch.bfloat16)
    got = ctx.pool(hb.cuda(), dev(mask, torch.int32), "weightedmean").cpu().numpy()
    assert maxabs(got, O.pool(hb.float().numpy(), mask, "weightedmean")) < 5e-6


def test_pool_large_vs_torch(ctx):
    B, S, d = 256, 128, 768
    h = torch.randn(B, S, d, device="cuda")
    lens = torch.randint(1, S + 1, (B,), device="cuda")
    mask = (torch.arange(S, device="cuda")[None, :] < lens[:, None]).to(torch.int32)
    w = mask.float() * torch.arange(1, S + 1, device="cuda").float()[None, :]
    want = (h * w[:, :, None]).sum(1) / w.sum(1, keepdim=True)
    got = ctx.pool(h, mask, "weightedmean")
    assert torch.max(torch.abs(got - want)).item() < 1e-5


def test_l2_normalize(ctx):
    x = torch.randn(1000, 768, device="cuda") * 5
    x[7] = 0
    got = ctx.l2_normalize(x)
    want = torch.nn.functional.normalize(x, p=2, dim=1)
    assert torch.max(torch.abs(got - want)).item() < 1e-6
    gb = ctx.l2_normalize(x, out_dtype=torch.bfloat16)
    assert torch.equal(gb, want.to(torch.bfloat16)) or torch.max(torch.abs(gb.float() - want)).item() < 4e-3
    assert torch.equal(ctx.to_bf16(x), x.to(torch.bfloat16))         # RNE identical to torch


# ---------------------------------------------------------------- top-k -------------------
def _check_topk(val, idx, scores, k, idx_base=0):
    n = scores.shape[1]
    kk = min(k, n)
    order = np.argsort(-scores, axis=1, kind="stable")[:, :kk]
    want_v = np.take_along_axis(scores, order, axis=1)
    assert np.array_equal(val[:, :kk], want_v)                       # values bit-exact, sorted descending
    got_scores = np.take_along_axis(scores, idx[:, :kk] - idx_base, axis=1)
    assert np.array_equal(got_scores, want_v)                        # indices point at those values
    assert np.all(val[:, kk:] == -np.inf) and np.all(idx[:, kk:] == -1)
    for r in range(scores.shape[0]):
        assert len(set(idx[r, :kk].tolist())) == kk                  # no duplicates


@pytest.mark.parametrize("nq,n,k", [(7, 5000, 10), (3, 50000, 1001), (5, 17, 10), (4, 11, 11), (2, 9, 20), (1, 1, 1),
                                    (9, 4096, 2048)])
def test_topk_vs_numpy(ctx, nq, n, k):
    scores = np.random.default_rng(n + k).standard_normal((nq, n)).astype(np.float32)
    val, idx = ctx.topk(dev(scores), k, idx_base=100)
    _check_topk(val.cpu().numpy(), idx.cpu().numpy(), scores, k, idx_base=100)


def test_topk_ties_and_nan(ctx):
    scores = np.zeros((3, 300), np.float32)                          # all tied: lowest indices win
    scores[1, 5] = np.nan                                            # NaN -> -1 (exact_search.py:99)
    scores[2, ::2] = 1.0
    val, idx = ctx.topk(dev(scores), 10)
    val, idx = val.cpu().numpy(), idx.cpu().numpy()
    assert idx[0].tolist() == list(range(10)) and np.all(val[0] == 0)
    assert 5 not in idx[1].tolist() and np.all(val[1] == 0)
    assert idx[2].tolist() == list(range(0, 20, 2)) and np.all(val[2] == 1)
    s2 = np.full((1, 50), np.nan, np.float32)
    v2, _ = ctx.topk(dev(s2), 5)
    assert np.all(v2.cpu().numpy() == -1)


def test_topk_merge_and_exclude(ctx):
    rng = np.random.default_rng(5)
    val = rng.standard_normal((6, 40)).astype(np.float32)
    idx = np.stack([rng.permutation(1000)[:40] for _ in range(6)]).astype(np.int64)
    idx[0, :5] = -1                                                  # empty slots are ignored
    ex = np.array([-1, idx[1, np.argmax(val[1])], -1, 12345, -1, idx[5, 3]], np.int64)
    ov, oi = ctx.topk_merge(dev(val), dev(idx, torch.int64), 11, exclude_idx=dev(ex, torch.int64))
    ov, oi = ov.cpu().numpy(), oi.cpu().numpy()
    for r in range(6):
        ok = (idx[r] >= 0) & (idx[r] != ex[r])
        order = np.argsort(-val[r][ok], kind="stable")[:11]
        assert np.array_equal(ov[r], val[r][ok][order])
        assert np.array_equal(oi[r], idx[r][ok][order])


def test_topk_merge_duplicate_pairs_and_nan_same_order_on_both_short_row_paths(ctx):
    """the register-key rounds of the short-row merge (indices < 2^32) and the comparator rounds (one index >= 2^32 in
    the row) rank the same lists the same way -- repeated (score, id) pairs are each emitted (a caller's lists may repeat an id),
    a negative id is an empty slot, a positive NaN ranks above +inf (torch.topk's convention), -0.0 ties with +0.0 on the id."""
    rng = np.random.default_rng(11)
    nq, m, k = 4, 60, 16
    val = rng.standard_normal((nq, m)).astype(np.float32)
    idx = np.stack([rng.permutation(500)[:m] for _ in range(nq)]).astype(np.int64)
    val[0, 7] = val[0, 3] = 9.0; idx[0, 7] = idx[0, 3] = 77             # the same (score, id) pair twice, at the top
    val[1, 10] = val[1, 20] = val[1, 30] = float(val[1].max()) + 1.0; idx[1, [10, 20, 30]] = [5, 5, 4]   # a triple tie, one id repeated
    val[2, 2] = np.nan; val[2, 9] = np.inf                              # NaN above +inf
    val[3, 4] = -0.0; val[3, 5] = 0.0; idx[3, 4], idx[3, 5] = 9, 8; val[3, 6:] = -np.abs(val[3, 6:]) - 1.0; val[3, :4] = -5.0
    idx[:, -3:] = -1                                                    # empty slots
    got = {}
    for tag, big in (("keys", False), ("comparator", True)):
        v, i = val.copy(), idx.copy()
        if big:
            v = np.concatenate([v, np.full((nq, 1), -1e30, np.float32)], axis=1)       # one far-away index: the whole row leaves the key path
            i = np.concatenate([i, np.full((nq, 1), (1 << 33) + 5, np.int64)], axis=1)
        ov, oi = ctx.topk_merge(dev(v), dev(i, torch.int64), k)
        got[tag] = (ov.cpu().numpy(), oi.cpu().numpy())
    assert np.array_equal(got["keys"][1], got["comparator"][1])
    assert np.array_equal(got["keys"][0], got["comparator"][0], equal_nan=True)
    ov, oi = got["keys"]
    assert oi[0, :2].tolist() == [77, 77] and ov[0, :2].tolist() == [9.0, 9.0]
    assert oi[1, :3].tolist() == [4, 5, 5]
    assert np.isnan(ov[2, 0]) and oi[2, 0] == idx[2, 2] and ov[2, 1] == np.inf
    assert oi[3, :2].tolist() == [8, 9] and (oi >= 0).all()


# ---------------------------------------------------------------- fused score + top-k -----
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_score_topk_vs_torch(ctx, dtype):
    nq, N, d, k = 33, 70001, 768, 11                                 # several internal chunks, ragged tail
    g = torch.Generator(device="cpu").manual_seed(0)
    q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g), dim=1).cuda()
    c = torch.nn.functional.normalize(torch.randn(N, d, generator=g), dim=1).cuda()
    if dtype != torch.float32:
        q, c = q.to(dtype), c.to(dtype)
    val, idx, n = ctx.score_topk(q, c, k, idx_base=5, dtype=dtype)
    assert n == k
    full = q.float() @ c.float().T
    wv, wi = torch.topk(full, k, dim=1)
    tol = 2e-6 if dtype == torch.float32 else 2e-5
    assert torch.max(torch.abs(val - wv)).item() < tol
    picked = torch.gather(full, 1, idx - 5)
    assert torch.max(torch.abs(picked - wv)).item() < tol            # same docs up to fp ties
    assert (idx[:, 0] - 5 == wi[:, 0]).all()


def test_score_topk_running_merge_equals_single_pass(ctx):
    nq, N, d, k = 8, 3000, 128, 16
    q = torch.randn(nq, d, device="cuda")
    c = torch.randn(N, d, device="cuda")
    v1, i1, _ = ctx.score_topk(q, c, k)
    run = None
    for s in range(0, N, 700):
        v, i, n = ctx.score_topk(q, c[s:s + 700].contiguous(), k, idx_base=s, run=run)
        run = (v, i, n)
    assert torch.equal(run[0], v1) and torch.equal(run[1], i1)
    v3, i3, n3 = ctx.score_topk(q, c[:5].contiguous(), k)           # fewer docs than k
    assert n3 == 5 and (i3[:, 5:] == -1).all() and torch.isinf(v3[:, 5:]).all()


def test_pool_learntmean_golden(ctx):
    """sgpt_pool_learnt vs the reference's WeightedMeanPooling.py output (tests/golden/extras.npz)."""
    fx = np.load(os.path.join(GOLDEN, "extras.npz"))
    h, mask, pw = torch.from_numpy(fx["lm_hidden"]), torch.from_numpy(fx["lm_mask"]), torch.from_numpy(fx["lm_pw"])
    got = ctx.pool(h, mask, "learntmean", position_weights=pw).cpu().numpy()
    assert np.max(np.abs(got - fx["lm_ref"])) < 1e-5
    assert np.max(np.abs(got - O.pool(fx["lm_hidden"], fx["lm_mask"], "learntmean", position_weights=fx["lm_pw"]))) < 1e-5
    with pytest.raises(ValueError):
        ctx.pool(h, mask, "learntmean", position_weights=pw[:10])


def test_fp8_weight_codes_bit_exact(ctx):
    """sgpt_fp8_quantize_rows / dequantize_rows: e4m3fn codes, power-of-two scales and de-quantised values are
    bit-identical to the oracle (itself bit-identical to torch.float8_e4m3fn, see make_golden.py)."""
    fx = np.load(os.path.join(GOLDEN, "extras.npz"))
    codes, scale = ctx.fp8_quantize_rows(torch.from_numpy(fx["fp8_w"]))
    assert np.array_equal(scale.cpu().numpy(), fx["fp8_scale"])
    assert np.array_equal(codes.cpu().numpy(), fx["fp8_codes"])
    deq = ctx.fp8_dequantize_rows(codes, scale).cpu().numpy()
    assert np.array_equal(deq, fx["fp8_deq"])
    deq16 = ctx.fp8_dequantize_rows(codes, scale, out_dtype=torch.bfloat16).float().cpu().numpy()
    assert np.array_equal(deq16, fx["fp8_deq"])                     # exact in bf16
    # all 256 codes decode like the oracle (NaN codes 0x7f / 0xff aside)
    allc = torch.arange(256, dtype=torch.uint8).reshape(1, 256)
    one = torch.ones(1)
    got = ctx.fp8_dequantize_rows(allc, one).cpu().numpy()[0]
    want = O.fp8_e4m3fn_decode(np.arange(256, dtype=np.uint8))
    keep = (np.arange(256) & 0x7F) != 0x7F
    assert np.array_equal(got[keep], want[keep])
    # a larger random matrix against the oracle
    rng = np.random.default_rng(3)
    w = (rng.standard_normal((300, 512)) * np.exp(rng.uniform(-6, 2, size=(300, 1)))).astype(np.float32)
    c2, s2 = ctx.fp8_quantize_rows(torch.from_numpy(w))
    oc, os_ = O.fp8_quantize_rows(w)
    assert np.array_equal(c2.cpu().numpy(), oc) and np.array_equal(s2.cpu().numpy(), os_)


def _sliced_classic(ctx, q, c, k, step, idx_base=0, run=None):
    """The same search as a sequence of calls short enough to take the materialise-and-select path."""
    for s in range(0, c.shape[0], step):
        v, i, n = ctx.score_topk(q, c[s:s + step], k, idx_base=idx_base + s, run=run)
        run = (v, i, n)
    return run


@pytest.mark.parametrize("d", [128, 192, 320])
def test_score_topk_ring_depths(ctx, d):
    """K = 128 / 192 / 320: two, three (odd: the 3-slot / 2-slot ring parities drift across tiles) and five k-steps
    per tile of the asymmetric-ring GEMM, materialised and filtered epilogues, against a plain fp32 product."""
    nq, N, k = 2048, 70_000, 10
    g = torch.Generator(device="cpu").manual_seed(d)
    q = torch.randn(nq, d, generator=g).cuda().to(torch.bfloat16)
    c = torch.randn(N, d, generator=g).cuda().to(torch.bfloat16)
    val, idx, n = ctx.score_topk(q, c, k)
    wv, wi, _ = _sliced_classic(ctx, q, c, k, 16384)
    assert torch.equal(val, wv) and torch.equal(idx, wi)
    full = q.float() @ c.float().T
    tv, ti = torch.topk(full, k, dim=1)
    assert torch.max(torch.abs(val - tv)).item() < 2e-3 * float(tv.abs().max())
    assert (torch.gather(full, 1, idx) - tv).abs().max().item() < 2e-3 * float(tv.abs().max())


@pytest.mark.parametrize("dt16", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", ["random", "ascending", "ties", "running", "drift"])
def test_score_topk_threshold_filtered_chunks_are_exact(ctx, case, dt16):
    """Long corpora take the threshold-filtered path (EPI_SCORE_FILTER: chunks after the first only append scores
    above the running k-th best; doubling chunk schedule; predicated classic fallback on candidate overflow).
    It must return exactly what the materialise-and-select chunk loop returns -- same values, same indices,
    same tie order -- including when every document beats the threshold (ascending scores -> overflow -> the
    sync-free fallback) and under massive ties."""
    nq, d, k = 2048, 128, 10                     # nq = 2048 -> 16 384-document chunks; N >= 32 768 takes the filtered path
                                                 # (d >= 128: two k-steps, the minimum of the LDS-ring kernels)
    g = torch.Generator(device="cpu").manual_seed(3)
    q = torch.randn(nq, d, generator=g)
    if case == "random":
        N = 100_003                              # ragged tail: 100 003 = 390 * 256 + 163
        c = torch.randn(N, d, generator=g)
    elif case == "ascending":
        N = 65_536                               # every query's score grows with the document index
        u = torch.randn(d, generator=g)
        q = q.abs() * u.sign()                   # <q, u> > 0 for all queries
        c = (torch.arange(1, N + 1).float() / N)[:, None] * u[None, :]
    elif case == "ties":
        N = 81_920
        c = torch.randn(1024, d, generator=g).repeat(N // 1024, 1)    # every score occurs 80 times
    elif case == "drift":
        # the score distribution jumps inside the corpus (a corpus sorted by length, SURVEY A.1 / exact_search.py:66-71):
        # the chunk that contains the jump overflows its candidate lists and is recomputed on its own; the chunks behind
        # it are filtered against the corrected thresholds again
        N = 200_037
        u = torch.randn(d, generator=g)
        q = q.abs() * u.sign()
        c = torch.randn(N, d, generator=g)
        c[40_000:] += 0.5 * u[None, :]           # documents from 40 000 on score ~0.5 |u|^2-ish higher for every query
    else:
        N = 70_000
        c = torch.randn(N, d, generator=g)
    q, c = q.cuda().to(dt16), c.cuda().to(dt16)
    run = None
    base = 0
    if case == "running":                        # incoming running best + global index base
        prev = torch.randn(5000, d, generator=torch.Generator(device="cpu").manual_seed(9)).cuda().to(dt16) * 1.5
        v0, i0, n0 = ctx.score_topk(q, prev, k, idx_base=0)
        run, base = (v0.clone(), i0.clone(), n0), 5000
        val, idx, n = ctx.score_topk(q, c, k, idx_base=base, run=(v0, i0, n0))
    else:
        val, idx, n = ctx.score_topk(q, c, k, idx_base=base)
    wv, wi, wn = _sliced_classic(ctx, q, c, k, 16384, idx_base=base, run=run)
    assert n == wn == k
    assert torch.equal(val, wv) and torch.equal(idx, wi), case
    # and both agree with a plain fp32 product of the bf16 operands
    full = q.float() @ c.float().T
    tv, tidx = torch.topk(full, k, dim=1)
    if case != "running":
        assert torch.max(torch.abs(val - tv)).item() < 1e-3 * max(1.0, float(tv.abs().max()))
    if case == "ties":
        # the best document's 80 copies tie: its ten lowest indices p, p + 1024, ..., p + 9 * 1024
        assert torch.equal(idx, idx[:, :1] + 1024 * torch.arange(k, device=idx.device)[None, :]) and (idx[:, 0] < 1024).all()
    if case == "ascending":              # (16-bit rounding maps neighbouring documents to equal scores: ties ascend by index)
        assert (idx >= N - 64).all() and (val[:, :-1] >= val[:, 1:]).all()


@pytest.mark.parametrize("nq", [2048, 16], ids=["256-row tile", "64-row tile"])
def test_score_topk_filtered_chunks_nan_and_negative_thresholds(ctx, nq):
    """The filter epilogues reject a row's tile on the raw maximum (NaNs dropped) unless the query's threshold lies below
    -1, where a NaN score -- counted as -1 (exact_search.py:99) -- can still enter the top-k: dot-product scores far below
    -1 with NaN documents spread over the corpus must come back exactly as the materialise-and-select loop returns them
    (the NaN documents rank above every real score here), and a query whose scores are all positive must not see them."""
    d, k = 128, 10
    N = 70_000 if nq > 64 else 300_000          # (nq = 16: 131 072-document chunks -> the filtered path starts at 262 144)
    g = torch.Generator(device="cpu").manual_seed(5)
    u = torch.randn(d, generator=g)
    q = -(torch.randn(nq, d, generator=g).abs() * u.sign()) * 2.0           # <q, u> < 0: every real score is very negative
    c = torch.randn(N, d, generator=g) * 0.1 + u[None, :] * 3.0
    q[1] = -q[1]                                                             # one query with positive scores throughout
    c[torch.tensor([5, 20_000, 40_001, N - 1])] = float("nan")               # NaN rows in the first chunk, later chunks, the tail
    q, c = q.cuda().to(torch.float16), c.cuda().to(torch.float16)
    val, idx, n = ctx.score_topk(q, c, k)
    wv, wi, wn = _sliced_classic(ctx, q, c, k, 16384 if nq > 64 else 100_000)
    assert n == wn == k and torch.equal(val, wv) and torch.equal(idx, wi)
    nan_docs = {5, 20_000, 40_001, N - 1}
    assert set(idx[0, :4].tolist()) == nan_docs and (val[0, :4] == -1).all() and (val[0, 4:] < -1).all()
    assert not (set(idx[1].tolist()) & nan_docs) and (val[1] > 0).all()


def test_topk_ties_take_lowest_indices_in_every_path(ctx):
    """Equal scores at the k-th place: the lowest INDICES win, whatever the position in the candidate row --
    shuffled (score, index) lists in a merge (radix path, second radix select over the indices), the
    previous-best leg of a running merge, and a row of identical scores."""
    g = torch.Generator(device="cpu").manual_seed(5)
    nq, m, k = 7, 300, 10
    idx = torch.stack([torch.randperm(5000, generator=g)[:m] for _ in range(nq)]).to(torch.int64)
    val = torch.full((nq, m), 0.5)
    val[:, :3] = 2.0                                             # three clear winners, then a 297-way tie for 7 places
    ov, oi = ctx.topk_merge(val.cuda(), idx.cuda(), k)
    for r in range(nq):
        want = sorted(idx[r, :3].tolist()) + sorted(idx[r, 3:].tolist())[:7]
        assert oi[r].cpu().tolist() == want
        assert ov[r].cpu().tolist() == [2.0] * 3 + [0.5] * 7
    # identical scores in a plain row: indices 0..k-1
    v2, i2 = ctx.topk(torch.full((3, 3000), 1.25).cuda(), 16)
    assert (i2.cpu() == torch.arange(16)).all() and (v2 == 1.25).all()
    # running merge: previous best (earlier documents) ties with every score of a later, long chunk
    d = 64
    c = torch.ones(20000, d).to(torch.bfloat16).cuda()
    q = torch.ones(5, d).to(torch.bfloat16).cuda()
    v, i, n = ctx.score_topk(q, c[:6000], 8, idx_base=0)
    v, i, n = ctx.score_topk(q, c[6000:], 8, idx_base=6000, run=(v, i, n))
    assert (i.cpu() == torch.arange(8)).all()


@pytest.mark.parametrize("k", [100, 300, 1001])
def test_score_topk_filtered_large_k(ctx, k):
    """The BEIR driver retrieves top_k = 1000 (k + 1 = 1001 kept, exact_search.py:104): the filtered path covers
    k <= 1024 (candidate capacity min(4k, 2048 - k); chunks grow by half when the capacity is below 2.5 k) and must
    equal the materialise-and-select loop exactly."""
    nq, N, d = 512, 150_000, 128
    g = torch.Generator(device="cpu").manual_seed(k)
    q = torch.randn(nq, d, generator=g).cuda().to(torch.bfloat16)
    c = torch.randn(N, d, generator=g).cuda().to(torch.bfloat16)
    val, idx, n = ctx.score_topk(q, c, k, idx_base=7)
    wv, wi, wn = _sliced_classic(ctx, q, c, k, 40_000, idx_base=7)     # 40 000 < 2 chunks: materialised path
    assert n == wn == k
    assert torch.equal(val, wv) and torch.equal(idx, wi)
    tv, _ = torch.topk(q.float() @ c.float().T, k, dim=1)
    assert torch.max(torch.abs(val - tv)).item() < 1e-3 * float(tv.abs().max())
    assert (val[:, :-1] >= val[:, 1:]).all() and (idx >= 7).all()


@pytest.mark.parametrize("case", ["ascending", "drift"])
@pytest.mark.parametrize("running", [False, True], ids=["fresh", "running"])
def test_score_topk_short_query_batches_overflow_fallback(ctx, case, running):
    """The overflow fallback on the 64-row scorer tile (nq <= 64), with and without an incoming running list: candidate
    lists that overflow (scores ascending with the index: every chunk; a drift inside the corpus: the chunk that holds the
    jump) are recomputed by the predicated materialise + select and the result is the materialise-and-select answer."""
    nq, d, k = 16, 128, 10
    g = torch.Generator(device="cpu").manual_seed(21)
    u = torch.randn(d, generator=g)
    q = torch.randn(nq, d, generator=g).abs() * u.sign()
    if case == "ascending":
        N = 300_005
        c = (torch.arange(1, N + 1).float() / N)[:, None] * u[None, :]
    else:
        N = 400_037
        c = torch.randn(N, d, generator=g)
        c[150_000:] += 0.5 * u[None, :]
    q, c = q.cuda().to(torch.float16), c.cuda().to(torch.float16)
    run, base = None, 0
    if running:
        prev = torch.randn(3000, d, generator=g).cuda().to(torch.float16)
        v0, i0, n0 = ctx.score_topk(q, prev, k, idx_base=0, dtype=torch.float16)
        run, base = (v0.clone(), i0.clone(), n0), 3000
        val, idx, n = ctx.score_topk(q, c, k, idx_base=base, run=(v0, i0, n0), dtype=torch.float16)
    else:
        val, idx, n = ctx.score_topk(q, c, k, dtype=torch.float16)
    wv, wi, wn = _sliced_classic(ctx, q, c, k, 100_000, idx_base=base, run=run)
    assert n == wn == k
    assert torch.equal(val, wv) and torch.equal(idx, wi), case
    if case == "ascending" and not running:
        assert (idx >= N - 64).all()


@pytest.mark.parametrize("dt16", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("nq", [1, 16, 33, 64])
def test_score_topk_short_query_batches_use_the_64_row_tile_exactly(ctx, nq, dt16):
    """nq <= 64 takes score64_kernel (64 query rows x 256 documents per workgroup) for the materialised first chunk and
    the threshold-filtered chunks: identical values and indices to the 256-row path's answer for the same queries
    (computed by padding the batch to 65 queries, which takes the 256-row tile), ragged tail included."""
    d, N, k = 768, 200_037, 11
    g = torch.Generator(device="cpu").manual_seed(nq)
    q = torch.nn.functional.normalize(torch.randn(65, d, generator=g), dim=1).cuda().to(dt16)
    base = torch.randn(1, d, generator=g) * 2
    c = torch.nn.functional.normalize(base + torch.randn(N, d, generator=g), dim=1).cuda().to(dt16)
    val, idx, n = ctx.score_topk(q[:nq].contiguous(), c, k, idx_base=3, dtype=dt16)
    wv, wi, wn = ctx.score_topk(q, c, k, idx_base=3, dtype=dt16)            # 65 queries -> 256-row tile
    assert n == wn == k
    assert torch.equal(val, wv[:nq]) and torch.equal(idx, wi[:nq])
    tv, ti = torch.topk(q[:nq].float() @ c.float().T, k, dim=1)
    assert torch.max(torch.abs(val - tv)).item() < 1e-3
    # running merge across calls with the small tile
    v1, i1, n1 = ctx.score_topk(q[:nq].contiguous(), c[:70_000].contiguous(), k, idx_base=3, dtype=dt16)
    v2, i2, n2 = ctx.score_topk(q[:nq].contiguous(), c[70_000:].contiguous(), k, idx_base=70_003, run=(v1, i1, n1), dtype=dt16)
    assert torch.equal(v2, val) and torch.equal(i2, idx)


@pytest.mark.parametrize("seed", range(30))
def test_score_topk_sampled_schedule_random_shapes(ctx, seed):
    """Round 4's scorer schedule (strided-sample thresholds, list-sized chunks, LDS-staged appends, arg-max merges, one-launch
    prologue) against the materialise-and-select loop on random shapes: query counts on both sides of the 64-row / 256-row tile
    switch, corpus lengths around the chunk boundaries with ragged tails, k from 1 to 64 (sampled) and beyond (doubling),
    an index base, an incoming running list, anisotropic / duplicated / drifting / plain data.  Bit-for-bit."""
    rng = np.random.default_rng(1000 + seed)
    nq = int(rng.choice([1, 17, 64, 65, 200, 1000, 2048]))
    k = int(rng.choice([1, 5, 11, 33, 64, 65, 101]))
    d = int(rng.choice([128, 192, 768]))
    chunk = 131072 if nq <= 64 else max(256, min(131072, (160 << 20) // (nq * 4) // 256 * 256))
    N = int(2 * chunk + rng.integers(0, 3 * chunk)) + int(rng.choice([0, 1, 72, 255]))      # long enough for the filtered path
    if nq > 1000 or d == 768:
        N = min(N, 400_000 if nq <= 64 else 150_000)
        N = max(N, 2 * chunk + 300)
    kind = str(rng.choice(["plain", "aniso", "dup", "drift"]))
    g = torch.Generator(device="cuda").manual_seed(seed)
    base = torch.randn(1, d, device="cuda", generator=g) * (3.0 if kind != "plain" else 0.0)
    c = base + torch.randn(N, d, device="cuda", generator=g)
    if kind == "dup":
        blk = 16384
        for s0 in range(blk, N, blk):
            e0 = min(N, s0 + blk)
            c[s0:e0] = c[: e0 - s0] + 0.02 * torch.randn(e0 - s0, d, device="cuda", generator=g)
    if kind == "drift":
        c[int(0.7 * N):] += 2.0 * base
    q = torch.nn.functional.normalize(base + torch.randn(nq, d, device="cuda", generator=g), dim=1)
    dt = torch.float16 if seed % 2 == 0 else torch.bfloat16
    c = torch.nn.functional.normalize(c, dim=1).to(dt)
    q = q.to(dt)
    idx_base = int(rng.choice([0, 123_456_789]))
    run = None
    if seed % 3 == 0:                                           # an incoming running best from earlier documents
        prev = torch.nn.functional.normalize(base + torch.randn(3000, d, device="cuda", generator=g), dim=1).to(dt)
        v0, i0, n0 = ctx.score_topk(q, prev, k, idx_base=7)
        run = (v0.clone(), i0.clone(), n0)
        val, idx, n = ctx.score_topk(q, c, k, idx_base=idx_base + 10_000, run=(v0, i0, n0))
        wv, wi, wn = _sliced_classic(ctx, q, c, k, chunk, idx_base=idx_base + 10_000, run=run)
    else:
        val, idx, n = ctx.score_topk(q, c, k, idx_base=idx_base)
        wv, wi, wn = _sliced_classic(ctx, q, c, k, chunk, idx_base=idx_base)
    assert n == wn == k, (nq, N, k, d, kind)
    assert torch.equal(val, wv) and torch.equal(idx, wi), (nq, N, k, d, kind, str(dt))


# ---------------------------------------------------------------- refined scorer (round 5) -----
@pytest.mark.parametrize("case", ["random", "anisotropic", "duplicates", "small", "deep", "running", "running_duplicates"])
def test_refined_scorer_returns_the_exact_fp32_topk(ctx, case):
    """sgpt_score_topk_refined: fp32 scores of the fp32 top-k (what the reference's torch.mm + torch.topk compute, util.py:41-43,
    exact_search.py:96-108) through f16 candidate proposals + exact re-scoring.  Against an fp64 product of the same fp32 rows: the
    returned ids are the fp64 top-k except where the fp64 scores of two documents differ by less than fp32 resolution, every
    returned score is the pair's fp32 dot product (2e-6), sorted descending.  'duplicates': blocks of 200 identical documents put
    more equal scores around the k-th best than the head-room holds -- the device-side check must send the chunk to the exact
    pass (report == 1) and the answer stays exact; 'small' (N below k + head-room) is the exact pass outright; 'deep': k = 1001;
    'running': two chunks through the running list equal one pass."""
    g = torch.Generator(device="cpu").manual_seed(11)
    nq, N, d, k = 257, 60000, 768, 11
    base = torch.randn(1, d, generator=g) * (3.0 if case == "anisotropic" else 0.0)
    c = base + torch.randn(N, d, generator=g)
    q = base + torch.randn(nq, d, generator=g)
    if case in ("duplicates", "running_duplicates"):
        c = c[: N // 200].repeat_interleave(200, dim=0)
    if case == "small":
        N = 40
        c = c[:N]
    if case == "deep":
        k, nq = 1001, 64
        q = q[:nq]
    q = torch.nn.functional.normalize(q, dim=1).cuda()
    c = torch.nn.functional.normalize(c, dim=1).cuda()
    kk = min(k, N)
    if case in ("running", "running_duplicates"):      # (with duplicates: the exact pass must start from the list as it was before the chunk)
        h = N // 2 + 17
        r1 = ctx.score_topk_refined(q, c[:h].contiguous(), None, k, idx_base=0)
        val, idx, n, fb = ctx.score_topk_refined(q, c[h:].contiguous(), None, k, idx_base=h, run=(r1[0], r1[1], r1[2]), report=True)
    else:
        val, idx, n, fb = ctx.score_topk_refined(q, c, None, k, idx_base=7 if case == "random" else 0, report=True)
    base_i = 7 if case == "random" else 0
    assert n == kk
    # did the predicated exact pass run?  never for well-spread scores; always when blocks of identical documents sit on the k-th best;
    # concentrated score distributions ('anisotropic': all cosines within +-0.015) and deep lists (k = 1001: 23 slots of head-room)
    # MAY need it -- the worst-case bound on |s16 - s32| (1.1e-3) is then as wide as the gaps between ranks -- and stay exact
    assert fb in (0, 1) and (case != "random" or fb == 0) and ("duplicates" not in case or fb == 1), (case, fb)
    print(f"refined scorer [{case}]: exact pass ran = {fb}")
    val, idx = val.cpu().numpy()[:, :kk], idx.cpu().numpy()[:, :kk] - base_i
    full = (q.double() @ c.double().T).cpu().numpy()
    got = np.take_along_axis(full, idx, 1)
    assert np.abs(val - got).max() < 2e-6                                # the scores ARE the fp32 dot products of their pairs
    assert (np.diff(val, axis=1) <= 0).all()
    kth = -np.sort(-full, axis=1)[:, kk - 1]
    assert (got >= kth[:, None] - 3e-7).all()                            # nothing below the true k-th best (up to fp32 resolution)
    for r in range(0, nq, 37):
        assert len(set(idx[r].tolist())) == kk
    # and the brute-force exact-fp32 scorer agrees (same ids where scores are distinct; values to fp32 rounding)
    bv, bi, _ = ctx.score_topk(q, c, k, idx_base=base_i, dtype=torch.float32)
    bv, bi = bv.cpu().numpy()[:, :kk], bi.cpu().numpy()[:, :kk] - base_i
    assert np.abs(bv - val).max() < 2e-6
    if "duplicates" not in case:
        same = (bi == idx).mean()
        assert same > 0.999, same
