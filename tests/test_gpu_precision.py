"""-m gpu: split-precision ("f16x3") operands -- VERDICT r03 next-1.

The reference runs fp32 everywhere (beir_dense_retriever.py:123,204-205; Transformer.py:38,72): it has no checkpoint on which
11-bit MFMA operands are not enough.  This path does, so every operand class can enter its MFMAs as a hi + lo pair of 16-bit
values (include/sgpt_hip.h::sgpt_model_set_precision), and the default mode picks the classes itself.  Tested here:
  * the split store epilogues of both GEMM kernels (hi == the plain kernel's bits, hi + lo == the fp32 value to ~2^-21,
    the [hi | lo | hi] row layout, the transposed pair) against a PyTorch fp32 product of the same operands;
  * the split-precision scorer layouts (sgpt_split16) against an fp64 product;
  * precision='x3' on the reference's golden fixtures of all three families: embeddings at fp32-mode accuracy;
  * every class on its own (one plan entry at a time) stays at least as close to the reference as the plain path's budget;
  * the probe: clean synthetic checkpoints stay 'plain' (identical bits to precision='plain'), the engineered-outlier
    checkpoint is flagged.  (The full-size outlier / SGPT-1.3B / 2.7B cases: tests/test_gpu_parity_large.py.)"""
import numpy as np
import pytest
import torch

from oracle import sgpt_oracle as O
from helpers import build_model, load_case, maxabs

pytestmark = pytest.mark.gpu


def _ctx():
    from sgpt_amd import get_context
    return get_context("cuda:0")


def gelu_new(u):
    return 0.5 * u * (1.0 + torch.tanh(0.7978845608028654 * (u + 0.044715 * u ** 3)))


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("shape", [(512, 512, 256, True), (768, 256, 192, True), (96, 160, 64, False), (256, 384, 768, False)],
                         ids=["256tile", "256tile-k192", "small-ragged", "small"])
def test_split_store_epilogues(dt, shape):
    M, N, K, force256 = shape
    ctx = _ctx()
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    a = (torch.randn(M, K, generator=g) * 1.5).to(dt).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).cuda()
    bias = (torch.randn(N, generator=g) * 0.1).cuda()
    ref = a.float() @ w.float().T                    # fp32 product of the same 16-bit operands
    old = ctx.set_tile_policy(force256)
    try:
        eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
        for epi, want in (("store", ref), ("gelu", gelu_new(ref + bias))):
            plain = ctx.linear(a, w, bias if epi == "gelu" else None, epi=epi)
            pair = ctx.linear_split(a, w, bias if epi == "gelu" else None, epi=epi)
            tri = ctx.linear_split(a, w, bias if epi == "gelu" else None, epi=epi, triple=True)
            hi, lo = pair[:, :N], pair[:, N:]
            assert torch.equal(hi, plain), (epi, "hi block differs from the plain epilogue")
            assert torch.equal(tri[:, :N], hi) and torch.equal(tri[:, N:2 * N], lo) and torch.equal(tri[:, 2 * N:], hi), epi
            got = hi.float() + lo.float()
            # hi + lo carries the fp32 value to ~eps^2 (relative; f16 lo halves below 2^-24 flush into subnormals: abs floor)
            tol = (want.abs() * (4 * eps * eps) + 2e-7 + (5e-4 if epi == "gelu" else 0) * want.abs())   # gelu: the kernel's fast sigmoid form
            assert bool(((got - want).abs() <= tol + 1e-3 * K / 64 * eps * want.abs().max()).all()), (epi, float((got - want).abs().max()))
            # the pair is far closer to the fp32 value than hi alone, where hi alone is not exact
            e_hi, e_pair = float((hi.float() - want).abs().max()), float((got - want).abs().max())
            assert e_pair <= e_hi
        if M % 128 == 0:
            plain = ctx.linear(a, w, None, epi="vt")
            pair = ctx.linear_split(a, w, None, epi="vt")
            assert torch.equal(pair[0], plain)
            got = pair[0].float() + pair[1].float()
            assert float((got - ref.T).abs().max()) <= float(ref.abs().max()) * 8 * eps * eps + 2e-7 + 1e-3 * K / 64 * eps * float(ref.abs().max())
    finally:
        ctx.set_tile_policy(old)


def test_split16_scorer_layouts():
    ctx = _ctx()
    g = torch.Generator(device="cpu").manual_seed(11)
    nq, nd, d = 37, 1000, 768
    q = torch.nn.functional.normalize(torch.randn(nq, d, generator=g), dim=1)
    c = torch.randn(nd, d, generator=g)
    c[:, [7, 300]] *= 40.0                           # two dominant channels: the case the split scorer exists for
    q[:, [7, 300]] *= 40.0
    q, c = torch.nn.functional.normalize(q, dim=1), torch.nn.functional.normalize(c, dim=1)
    ref = (q.double() @ c.double().T).float().numpy()
    q3, c3 = ctx.split16(q.cuda(), "query"), ctx.split16(c.cuda(), "doc")
    assert q3.shape == (nq, 3 * d) and c3.shape == (nd, 3 * d)
    hi = q.cuda().half()
    assert torch.equal(q3[:, :d], hi) and torch.equal(q3[:, d:2 * d], hi)
    assert torch.equal(c3[:, :d], c.cuda().half()) and torch.equal(c3[:, 2 * d:], c.cuda().half())
    s3 = ctx.scores(q3, c3, dtype=torch.float16).cpu().numpy()
    s1 = ctx.scores(q.cuda().half(), c.cuda().half(), dtype=torch.float16).cpu().numpy()
    e3, e1 = maxabs(s3, ref), maxabs(s1, ref)
    print(f"split scorer: max|cos - ref| {e3:.2e} (plain f16 rows: {e1:.2e})")
    assert e3 < 3e-6 and e1 > 20 * e3
    val, idx, n = ctx.score_topk(q3, c3, 10, dtype=torch.float16)
    want = np.argsort(-ref, axis=1, kind="stable")[:, :10]
    assert n == 10 and np.array_equal(idx.cpu().numpy(), want)


@pytest.mark.parametrize("tag", ["tiny_right", "tiny_left", "tiny_dh128", "cfg1_125m_32x64", "cfg3_125m_specb_s300",
                                 "tiny_gptj_right", "tiny_bloom_left", "tiny_bloom_right"])
def test_x3_mode_reaches_fp32_accuracy_on_the_golden_fixtures(tag):
    fx, cfg_kw, seqs, pad_left, ids, mask = load_case(tag)
    m = build_model(cfg_kw, int(fx["seed"]), float(fx["std"]), "f16", precision="x3")
    plain = build_model(cfg_kw, int(fx["seed"]), float(fx["std"]), "f16")
    plan = m.precision_plan()
    gptj = "n_embd" in cfg_kw
    assert (plan[:, 0] == 2).all() and (plan[:, 2:] == 1).all() and (plan[:, 1] == (0 if gptj else 1)).all()
    ref = fx["emb_weightedmean"]
    got = m.encode_ids(seqs, mode="weightedmean", pad_left=pad_left).cpu().numpy()
    base = plain.encode_ids(seqs, mode="weightedmean", pad_left=pad_left).cpu().numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    e3, e1 = maxabs(got, ref) / scale, maxabs(base, ref) / scale
    dev3 = maxabs(O.cos_sim(got, got), O.cos_sim(ref, ref))
    print(f"{tag} f16x3: max|emb - ref| / max|ref| = {e3:.2e} (plain f16 {e1:.2e}); max|cos - cos_ref| = {dev3:.2e}")
    assert np.isfinite(got).all()
    # GPT-J keeps 16-bit q / k / v / p (rotary in place; head_dim 256 at full size): GEMM operands alone
    assert e3 < (2e-4 if gptj else 2e-5) and dev3 < (1e-4 if gptj else 1e-5)
    assert e3 < e1
    for mode in ("mean", "lasttoken"):
        g2 = m.encode_ids(seqs, mode=mode, pad_left=pad_left).cpu().numpy()
        assert maxabs(g2, fx[f"emb_{mode}"]) / max(1.0, float(np.abs(fx[f"emb_{mode}"]).max())) < (2e-4 if gptj else 2e-5), mode


def test_x3_mode_with_bf16_halves():
    """The same split on bf16 operands (8 + 8 mantissa bits: products to ~2^-16): two orders of magnitude closer to the reference
    than plain bf16 on the 125M-shape fixture -- every LO epilogue and the x3 attention in their bf16 instantiation."""
    fx, cfg_kw, seqs, pad_left, ids, mask = load_case("cfg1_125m_32x64")
    m = build_model(cfg_kw, int(fx["seed"]), float(fx["std"]), "bf16", precision="x3")
    plain = build_model(cfg_kw, int(fx["seed"]), float(fx["std"]), "bf16")
    ref = fx["emb_weightedmean"]
    scale = max(1.0, float(np.abs(ref).max()))
    e3 = maxabs(m.encode_ids(seqs, pad_left=pad_left).cpu().numpy(), ref) / scale
    e1 = maxabs(plain.encode_ids(seqs, pad_left=pad_left).cpu().numpy(), ref) / scale
    print(f"cfg1 bf16x3: max|emb - ref| / max|ref| = {e3:.2e} (plain bf16 {e1:.2e})")
    assert e3 < 3e-4 and e3 < e1 / 20


@pytest.mark.parametrize("tag", ["tiny_right", "cfg1_125m_32x64", "tiny_bloom_left"])
def test_every_class_on_its_own(tag):
    """One plan entry at a time (every block): the launch sequence of each class in isolation -- its producer's split
    epilogue, its consumer's K' = 3 K contraction -- gives a finite result at least as close to the reference as plain f16
    (up to noise), and the bits of a plan are reproducible through get / set."""
    from sgpt_amd import SGPTConfig, SGPTModel
    from helpers import oracle_cfg_weights
    fx, cfg_kw, seqs, pad_left, ids, mask = load_case(tag)
    _, w = oracle_cfg_weights(cfg_kw, int(fx["seed"]), float(fx["std"]))
    scfg = SGPTConfig.from_hf_dict(dict(cfg_kw, model_type="bloom")) if "n_layer" in cfg_kw else SGPTConfig(**cfg_kw)
    m = SGPTModel(scfg, w, device="cuda:0", dtype="f16", precision="auto-class", precise_qk=False)
    try:
        ref = fx["emb_weightedmean"]
        scale = max(1.0, float(np.abs(ref).max()))
        L = scfg.num_layers
        zero = np.zeros((L, 5), dtype=np.int32)
        m.set_precision_plan(zero)
        e_plain = maxabs(m.encode_ids(seqs, pad_left=pad_left).cpu().numpy(), ref) / scale
        for cls, levels in ((0, (1, 2, 3)), (1, (1,)), (2, (1,)), (3, (1,)), (4, (1,))):
            for lv in levels:
                plan = zero.copy()
                plan[:, cls] = lv
                m.set_precision_plan(plan)
                assert np.array_equal(m.precision_plan(), plan)
                got = m.encode_ids(seqs, pad_left=pad_left).cpu().numpy()
                e = maxabs(got, ref) / scale
                print(f"{tag}: class {cls} level {lv}: {e:.2e} (plain {e_plain:.2e})")
                assert np.isfinite(got).all() and e < 1.5 * e_plain + 1e-5, (cls, lv)
                again = m.encode_ids(seqs, pad_left=pad_left).cpu().numpy()
                assert np.array_equal(got, again)
        # a batch of one query-sized call and the same sentences inside a larger call: identical bits under a full plan too
        m.set_precision_plan(m._x3_plan())
        full = m.encode_ids(seqs, pad_left=pad_left).cpu().numpy()
        one = m.encode_ids(seqs[:3], pad_left=None if pad_left is None else pad_left[:3]).cpu().numpy()
        assert np.array_equal(full[:3], one)
    finally:
        m.close()


def test_plan_validation():
    from sgpt_amd import SGPTConfig, SGPTModel
    from helpers import oracle_cfg_weights
    fx, cfg_kw, seqs, pad_left, ids, mask = load_case("tiny_right")
    _, w = oracle_cfg_weights(cfg_kw, int(fx["seed"]), float(fx["std"]))
    scfg = SGPTConfig(**cfg_kw)
    plain = SGPTModel(scfg, w, device="cuda:0", dtype="f16", precision="plain")
    try:
        bad = np.zeros((scfg.num_layers, 5), dtype=np.int32)
        bad[0, 4] = 1
        with pytest.raises(ValueError):          # no split weight copies were kept
            plain.set_precision_plan(bad)
        bad[0, 4] = 0
        bad[0, 0] = 4
        with pytest.raises(ValueError):
            plain.set_precision_plan(bad)
        with pytest.raises(ValueError):
            plain.set_precision_plan(np.zeros(3, dtype=np.int32))
    finally:
        plain.close()
    with pytest.raises(ValueError):
        SGPTModel(scfg, w, device="cuda:0", dtype="fp32", precision="x3")
    fxj, cfgj, *_ = load_case("tiny_gptj_right")
    _, wj = oracle_cfg_weights(cfgj, int(fxj["seed"]), float(fxj["std"]))
    mj = SGPTModel(SGPTConfig.from_hf_dict(dict(cfgj, model_type="gptj")), wj, device="cuda:0", dtype="f16", precision="x3")
    try:
        p = mj.precision_plan()
        p[:, 1] = 1
        with pytest.raises(ValueError):          # rotary embedding: q / k are rotated in place in 16 bits
            mj.set_precision_plan(p)
    finally:
        mj.close()


def test_probe_keeps_clean_checkpoints_plain_and_flags_outliers():
    from sgpt_amd import SGPTConfig, SGPTModel
    cfg = O.NeoConfig(**O.SGPT_125M)
    rng = np.random.default_rng(3)
    seqs = [rng.integers(0, 50256, size=int(n)).tolist() for n in rng.integers(16, 129, size=48)]
    w = O.synth_weights(cfg, seed=1, std=0.02)
    scfg = SGPTConfig(**O.SGPT_125M)
    auto = SGPTModel(scfg, w, device="cuda:0", dtype="f16")            # the default: precision='auto'
    plain = SGPTModel(scfg, w, device="cuda:0", dtype="f16", precision="plain")
    try:
        assert auto.precision == "auto"
        a = auto.encode_ids(seqs).cpu().numpy()
        rep = auto.precision_report
        print("clean 125M crest factors per class (max over blocks):", rep["crest"].max(0).round(1).tolist())
        assert rep["decided"] == "plain" and rep["flagged"] == 0 and not auto.precision_plan().any()
        assert np.array_equal(a, plain.encode_ids(seqs).cpu().numpy())          # same kernels, same bits
        assert (rep["crest"][:, :3] < 9).all() and (rep["crest"][:, 3] < 14).all()
        # a probe that settles on plain gives the [W_hi | W_hi | W_lo] copies back -- 3 x the 16-bit bytes of the
        # four matrices of every block (12 d^2 parameters): 12 blocks x 3 x 2 B x 12 x 768^2
        assert rep["split_weight_bytes_released"] == 12 * 3 * 2 * 12 * 768 * 768
        assert np.array_equal(a, auto.encode_ids(seqs).cpu().numpy())            # nothing the plain plan reads went away
        with pytest.raises(ValueError, match="without split weight copies"):     # ... and a plan that needs them is refused loudly
            auto.set_precision_plan(auto._x3_plan())
        assert auto.release_split_weights() == 0                                 # idempotent
    finally:
        auto.close()
        plain.close()
    wo = O.engineer_outliers({k: v.copy() for k, v in w.items()})
    bad = SGPTModel(scfg, wo, device="cuda:0", dtype="f16")
    cls = SGPTModel(scfg, wo, device="cuda:0", dtype="f16", precision="auto-class")
    try:
        bad.encode_ids(seqs)
        rep = bad.precision_report
        print("outlier 125M crest factors:", rep["crest"].round(0).tolist())
        assert rep["decided"] == "x3" and np.array_equal(bad.precision_plan(), bad._x3_plan())
        assert rep["crest"][3, 3] > 30 and rep["crest"][5, 0] > 15              # block 3's GELU output, the massive channels
        cls.encode_ids(seqs)
        p = cls.precision_plan()
        assert cls.precision_report["decided"] == "classes" and p[3, 4] == 1 and p[2, 4] == 0 and 0 < p.sum() < bad.precision_plan().sum()
    finally:
        bad.close()
        cls.close()
