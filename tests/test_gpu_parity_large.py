"""-m gpu: parity at the FULL shape of BASELINE configs[2], [3], [4] against the REFERENCE stack.

tests/golden/cfg3_neo13b_specb.npz, cfg_neo27b.npz, cfg4_gptj6b.npz, cfg5_bloom7b1.npz hold what the reference computes at
SGPT-1.3B (24 layers, d 2048, specb brackets, documents up to 300 tokens), SGPT-2.7B (32 layers, d 2560, 20 heads of 128),
GPT-J-6B (28 layers, d 4096, head_dim 256, rotary 64) and bloom-7b1 (30 layers, d 4096, 32 heads; left- and right-padded
batches) shape: HF model fp32 eager -> the reference's Pooling.py (weightedmean) -> the reference's util.cos_sim -> the
reference's DenseRetrievalExactSearch top-10 (tests/golden/make_golden_large.py; the weights are regenerated here from the
seed, one numpy stream per tensor).  outlier_125m.npz / outlier_neo13b.npz: the same chain on SGPT-125M / SGPT-1.3B-shape
weights with engineered outliers (a handful of embedding / fc / LayerNorm channels x 100...1000, two GELU outputs beyond the
f16 range) -- the stand-in for real GPT-Neo checkpoints.

Every case goes through the HIP path in ONE sgpt_encode call whose projections all run on the 256x256-tile throughput
kernels (the launch shapes are checked), then through the 16-bit scorer.

north_star bar: embeddings and ranked cosine scores within 1e-3 of the reference CPU path; "embeddings" are compared
NORMALISED (what cosine retrieval consumes; raw pooled rows carry an arbitrary scale -- relative to the row norm they deviate
by the same figure).  Every case's DEFAULT mode (dtype 'f16', precision 'auto', the model's own precise_qk rule) is held to
the bar on both figures; the other rows are reported modes with budgets at ~1.5 x their measured deviation:
  GPT-J-6B / bloom-7b1   plain f16 is a factor of 16 inside the bar (3.8e-5 / 5.8e-5): the probe keeps them plain;
  SGPT-1.3B / 2.7B       random-init GPT-Neo at d >= 2048 has no 1/sqrt(dh) in its attention (HF:gpt_neo:110): logits of std ~9
                         amplify every 16-bit rounding of LayerNorm -> Wq / Wk -> q / k (plain: 8.2e-4 cosine / 1.09e-3
                         embeddings at 1.3B).  The model's structural default (model.default_precise_qk) puts it inside the bar;
                         "f16-qk" = precise_qk=False, the other precise_qk variants are reported beside it;
  outlier_*              the probe sees crest factors of 20-55 (clean: 5-10) and moves the whole model to split-precision
                         operands ("f16x3"); "f16-class" = only the flagged classes; "f16-plain" = what round 3 shipped
                         (1.21e-3 / 7.8e-4 at 125M shape).  Their cosine scores go through the split-precision scorer rows the
                         search host selects for such embeddings (beir.SCORE_SPLIT_CREST).
  bf16 / fp8 storage / fp8 MFMA: reported (SURVEY 7 allows report-only for fp8)."""
import json
import os
import time

import numpy as np
import pytest
import torch

from helpers import GOLDEN, maxabs
from oracle import sgpt_oracle as O

pytestmark = pytest.mark.gpu

BAR = 1e-3
TORCH_DT = {"f16": torch.float16, "bf16": torch.bfloat16, "fp8": torch.bfloat16, "fp8mfma": torch.bfloat16}
# mode suffixes of dtype 'f16': constructor arguments
VARIANTS = {"qk": dict(precise_qk=False, precision="plain"), "full": dict(precise_qk="full", precision="plain"),
            "logits": dict(precise_qk="logits", precision="plain"), "act": dict(precise_qk="act+logits", precision="plain"),
            "fulllogits": dict(precise_qk="full+logits", precision="plain"), "qkvlogits": dict(precise_qk="qkv+logits", precision="plain"),
            "attn": dict(precise_qk="attn", precision="plain"),
            "plain": dict(precision="plain"), "class": dict(precision="auto-class"), "x3": dict(precision="x3")}
# (max |cos - cos_ref|, max |normalised emb - ref|) allowed per case and operand format.  BAR = the north_star bar; every
# other figure is ~1.5 x the deviation measured in round 3 (see the module docstring for the two f16 entries over the bar)
BUDGET = {
    # engineered outliers (oracle.engineer_outliers): "f16" = the default mode (the probe flags the checkpoint -> f16x3)
    "outlier_125m": {"f16": (BAR, BAR), "f16-class": (BAR, BAR), "f16-plain": (1.8e-3, 1.2e-3), "bf16": (1.5e-2, 1.2e-2)},
    "outlier_neo13b": {"f16": (BAR, BAR), "f16-plain": (6e-3, 6e-3)},
    # "f16": the DEFAULT for these models (GPT-Neo, d >= 2048: the structural precise_qk rule) -- embeddings AND cosine scores
    # inside the bar; the other precise_qk variants and the plain projection ("f16-qk") are reported
    "cfg3_neo13b_specb": {"f16": (BAR, BAR), "f16-qk": (BAR, 1.6e-3), "f16-logits": (1.2e-3, 1.3e-3), "f16-act": (BAR, BAR),
                          "f16-full": (BAR, BAR), "f16-fulllogits": (BAR, BAR), "f16-attn": (BAR, BAR), "f16-x3": (BAR, BAR),
                          "bf16": (7.5e-3, 1.0e-2)},
    "cfg_neo27b": {"f16": (BAR, BAR), "f16-qk": (2e-3, 3.5e-3), "f16-logits": (2e-3, 3e-3), "f16-full": (1.5e-3, 2e-3),
                   "f16-fulllogits": (1.5e-3, 1.5e-3), "f16-qkvlogits": (1.5e-3, 1.5e-3), "f16-attn": (1.5e-3, 1.5e-3), "f16-x3": (BAR, BAR)},
    # round 5: SECOND seeds of the two GPT-Neo shapes whose default sits within 20 % of the bar on the first
    # fixture; the 2.7B one with 160-300-token documents (the 256-token local window live at d = 2560 / head_dim 128)
    "cfg3_neo13b_specb_s2": {"f16": (BAR, BAR), "f16-qk": (1.5e-3, 2e-3), "f16-logits": (BAR, 1.3e-3), "f16-act": (BAR, 1.2e-3), "f16-full": (BAR, 1.2e-3),
                             "f16-fulllogits": (BAR, BAR)},
    "cfg_neo27b_s2": {"f16": (BAR, BAR), "f16-qk": (3e-3, 4e-3), "f16-qkvlogits": (BAR, BAR), "f16-attn": (BAR, BAR)},
    "cfg4_gptj6b": {"f16": (BAR, BAR), "bf16": (BAR, BAR), "fp8mfma": (1.0e-2, 1.0e-2)},
    "cfg5_bloom7b1": {"f16": (BAR, BAR), "bf16": (BAR, BAR), "fp8": (1.0e-2, 1.0e-2), "fp8mfma": (1.0e-2, 1.0e-2)},
}
CASES = [(tag, dt) for tag, per in BUDGET.items() for dt in per]

_weights = {}


def case_weights(tag, meta):
    """Seed-regenerated weights of a case, kept for the dtypes of that case only (6 G parameters = 24 GB of host RAM)."""
    if tag not in _weights:
        _weights.clear()
        arch = meta["arch"]
        cfg = {"gpt_neo": O.NeoConfig, "gptj": O.GPTJConfig, "bloom": O.BloomConfig}[arch](**meta["cfg"])
        t = time.time()
        _weights[tag] = O.synth_weights_streams(cfg, seed=meta["seed"], std=meta["std"])
        if meta.get("outliers"):
            O.engineer_outliers(_weights[tag])
        print(f"{tag}: {sum(v.size for v in _weights[tag].values()) / 1e9:.2f} G parameters regenerated in {time.time() - t:.0f} s")
    return _weights[tag]


@pytest.mark.parametrize("tag,dtype", CASES, ids=[f"{t}-{d}" for t, d in CASES])
def test_full_shape_cosine_and_ranked_top10_vs_reference(tag, dtype):
    from sgpt_amd import SGPTConfig, SGPTModel, get_context
    fx = np.load(os.path.join(GOLDEN, f"{tag}.npz"))
    meta = json.loads(str(fx["meta"]))
    w = case_weights(tag, meta)
    arch = meta["arch"]
    scfg = SGPTConfig(**meta["cfg"]) if arch == "gpt_neo" else SGPTConfig.from_hf_dict(dict(meta["cfg"], model_type=arch))
    ctx = get_context("cuda:0")
    lens = fx["lens"].astype(np.int64)
    off = np.concatenate([[0], np.cumsum(lens)])
    seqs = [fx["ids"][off[i]: off[i + 1]].tolist() for i in range(len(lens))]
    pad_left = fx["pad_left"].astype(np.int64).tolist()
    isq = fx["is_query"].astype(bool)
    base, _, variant = dtype.partition("-")
    kw = dict(VARIANTS[variant]) if variant else {}               # no suffix: the model's own defaults
    m = SGPTModel(scfg, w, device="cuda:0", dtype=base, max_tokens_per_call=1 << 17, **kw)
    if not variant and base == "f16":
        assert m.precision == "auto" and (m.precise_qk is not False) == (arch == "gpt_neo" and scfg.hidden_size >= 2048)
    try:
        if dtype == "fp8mfma":
            m.calibrate(seqs[:: max(1, len(seqs) // 16)])                # calibrated on a slice of the case's own inputs
        # every projection of the call must take the 256x256 LDS-DMA kernel: more than half a wave of tiles for the
        # narrowest launch (N = d_model), sgpt_amd/csrc/gemm.hip::launch_gemm16
        from sgpt_amd.model import ALIGN
        alloc = int(((lens + ALIGN - 1) // ALIGN * ALIGN).sum())
        T_pad = (alloc + 255) // 256 * 256
        assert alloc <= m.max_tokens_per_call and (T_pad // 256) * (scfg.hidden_size // 256) * 2 > 256, (alloc, T_pad)
        t = time.time()
        emb = m.encode_ids(seqs, pad_left=pad_left)                      # ONE sgpt_encode call
        torch.cuda.synchronize()
        t_enc = time.time() - t
        shifts = m.range_shifts() if dtype.startswith("f16") else None
        plan = m.precision_plan() if base in ("f16", "bf16") else None
        report = m.precision_report
    finally:
        m.close()
        torch.cuda.empty_cache()
    ref = fx["emb"]
    emb_np = emb.cpu().numpy()
    assert np.isfinite(emb_np).all()
    rel = float((np.abs(emb_np - ref).max(1) / np.linalg.norm(ref, axis=1)).max())
    sdt = TORCH_DT[base]
    en = ctx.l2_normalize(emb)
    e_dev = maxabs(en.cpu().numpy(), O.normalize(ref))
    qi, di = np.nonzero(isq)[0], np.nonzero(~isq)[0]
    qn = en[torch.from_numpy(qi).to(en.device)].contiguous()
    dn = en[torch.from_numpy(di).to(en.device)].contiguous()
    # the search host's rule (beir.DenseRetrievalExactSearch): query rows that a few channels dominate -> split-precision rows
    from sgpt_amd.beir import SCORE_SPLIT_CREST
    q_crest = ctx.row_crest(qn)
    split = q_crest > SCORE_SPLIT_CREST
    q_op = ctx.split16(qn, "query", sdt) if split else ctx._operand(qn, sdt)
    d_op = ctx.split16(dn, "doc", sdt) if split else ctx._operand(dn, sdt)
    cos = ctx.scores(q_op, d_op, dtype=sdt).cpu().numpy()
    c_dev = maxabs(cos, fx["cos"])
    budget, e_budget = BUDGET[tag][dtype]
    k = meta["topk"]
    val, idx, n = ctx.score_topk(q_op, d_op, k, dtype=sdt)
    val, idx = val.cpu().numpy(), idx.cpu().numpy()
    ref_cos, ref_top = fx["cos"], fx["top10"]
    ref_sorted = np.take_along_axis(ref_cos, ref_top, 1)
    overlap = [k - len(set(idx[q].tolist()) - set(ref_top[q].tolist())) for q in range(len(qi))]
    same_rank = int(sum(np.array_equal(idx[q], ref_top[q]) for q in range(len(qi))))
    line = (f"{tag} {dtype}: {len(seqs)} sequences / {alloc} token rows in one call ({t_enc * 1e3:.0f} ms incl. pack); "
            f"max|emb-ref|/||ref|| = {rel:.2e}, normalised max|emb-ref| = {e_dev:.2e}, max|cos-ref| = {c_dev:.2e} over {cos.size} pairs "
            f"(budgets {budget:g} / {e_budget:g}); top-{k} id overlap mean {np.mean(overlap):.2f} min {min(overlap)}, identical ranking "
            f"{same_rank}/{len(qi)}" + (f"; range shifts max {int(shifts.max())}" if shifts is not None else "") +
            (f"; plan entries {int((plan != 0).sum())}/{plan.size}" if plan is not None else "") +
            (f"; probe: {report['decided']}, {report['flagged']} classes flagged, crest max {float(report['crest'].max()):.1f}" if report else "") +
            f"; query-row crest {q_crest:.1f} -> {'split' if split else 'plain'} scorer rows")
    print(line)
    out_dir = os.environ.get("SGPT_PARITY_LOG")
    if out_dir:
        with open(out_dir, "a") as f:
            f.write(json.dumps(dict(case=tag, dtype=dtype, rows=alloc, rel_emb=rel, max_abs_norm_emb=e_dev, max_abs_cos=c_dev,
                                    budget=budget, emb_budget=e_budget, top10_overlap_mean=float(np.mean(overlap)), identical_rank=same_rank,
                                    n_queries=int(len(qi)), n_docs=int(len(di)), encode_ms=t_enc * 1e3,
                                    plan_entries=None if plan is None else int((plan != 0).sum()),
                                    probe=None if not report else report["decided"], split_scorer=bool(split))) + "\n")
    assert np.isfinite(cos).all() and n == k
    assert c_dev < budget and e_dev < e_budget, line
    # ranked top-k through the fused scorer: every returned score within the budget of the reference score of its pair,
    # rank-for-rank scores within the budget of the reference's ranked scores; a document outside the reference top-k may
    # appear only where the reference itself separates it from its k-th hit by less than 2 x budget
    assert maxabs(val, np.take_along_axis(ref_cos, idx, 1)) < budget
    assert maxabs(val, ref_sorted) < budget
    for q in range(len(qi)):
        for doc in set(idx[q].tolist()) - set(ref_top[q].tolist()):
            assert ref_sorted[q, k - 1] - ref_cos[q, doc] < 2 * budget, (q, doc)
    if base == "f16" and meta.get("outliers"):
        # the GELU output of block 3 (two hidden units at ~1e5) was moved under a shift by the guarded re-run; nothing else
        assert shifts[3, 3] >= 2 and int(shifts.sum()) == int(shifts[3, 3]), shifts.tolist()
        if not variant:                                            # the default mode: flagged by the probe, whole model split
            assert report["decided"] == "x3" and (plan[:, 2:] == 1).all() and (plan[:, 0] == 2).all() and split
    elif base == "f16":
        assert shifts is not None and int(shifts.max()) == 0        # std-0.02 random-init weights stay inside the half range
        if not variant:
            assert report["decided"] == "plain"                     # clean checkpoints: only the structural precise_qk rule
