#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: encoded sentences/sec (+ queries/sec @ corpus)
for SGPT-125M, seq_len 128, 16-bit MFMA operands (f16 by default: same MFMA rate as bf16, 3 more mantissa
bits -- the mode that meets the 1e-3 parity bar; `--dtype bf16` times the bf16 variant), cosine top-10,
synthetic corpus (BASELINE configs[1]).

One "step" = one pass of the reference's corpus-chunk loop (custommodels/exact_search.py:80-132)
over a chunk of `--chunk` synthetic documents already resident in HBM as packed token ids:
    encode (GPT-Neo forward + weighted-mean pool + L2 normalise)  ->  bf16 corpus rows
    score nq pre-encoded queries against the chunk, running top-(k+1) merge.
N > 1 (one rank per GPU; `python bench.py --gpus N` re-launches itself under torch.distributed.run when
it was not started by it): every rank owns its own corpus shard (weak scaling: work per GPU fixed);
queries are encoded sharded and all-gathered once over RCCL; at the end the per-rank top-k lists are
all-gathered and merged.  No data-path collective inside a step.  After the timed region the sharded
search is checked against a single-rank search over the same documents.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 MFMA peak
SGPT_125M = dict(vocab_size=50257, max_position_embeddings=2048, hidden_size=768, num_layers=12, num_heads=12,
                 window_size=256)
# other SGPT sizes (not the default metric; `--model`): shapes from SURVEY.md 8
MODELS = {
    "125m": dict(SGPT_125M),
    "1.3b": dict(vocab_size=50257, max_position_embeddings=2048, hidden_size=2048, num_layers=24, num_heads=16, window_size=256),
    "2.7b": dict(vocab_size=50257, max_position_embeddings=2048, hidden_size=2560, num_layers=32, num_heads=20, window_size=256),
    "5.8b": dict(model_type="gptj", vocab_size=50400, n_positions=2048, n_embd=4096, n_layer=28, n_head=16, rotary_dim=64),
    "bloom-7b1": dict(model_type="bloom", vocab_size=250880, hidden_size=4096, n_layer=30, n_head=32),
}


def device_random_weights(cfg, device, seed=1, std=0.02):
    """Random-init weights generated directly in HBM (the 5.8B model is 23 GB in fp32: not worth a host
    round trip for a throughput measurement).  HF state-dict names; same scales as synthetic_weights."""
    g = torch.Generator(device=device).manual_seed(seed)
    d, ffn, gptj = cfg.hidden_size, cfg.intermediate_size, cfg.model_type == "gptj"
    if cfg.model_type == "bloom":
        w = {"word_embeddings.weight": torch.randn((cfg.vocab_size, d), generator=g, device=device) * std,
             "word_embeddings_layernorm.weight": torch.ones(d, device=device), "word_embeddings_layernorm.bias": torch.zeros(d, device=device),
             "ln_f.weight": torch.ones(d, device=device), "ln_f.bias": torch.zeros(d, device=device)}
        for i in range(cfg.num_layers):
            p = f"h.{i}."
            for ln in ("input_layernorm", "post_attention_layernorm"):
                w[p + ln + ".weight"], w[p + ln + ".bias"] = torch.ones(d, device=device), torch.zeros(d, device=device)
            for name, shape in (("self_attention.query_key_value", (3 * d, d)), ("self_attention.dense", (d, d)),
                                ("mlp.dense_h_to_4h", (ffn, d)), ("mlp.dense_4h_to_h", (d, ffn))):
                w[p + name + ".weight"] = torch.randn(shape, generator=g, device=device) * std
                w[p + name + ".bias"] = torch.randn(shape[0], generator=g, device=device) * std
        return w

    def nrm(*shape, s=std, mean=0.0):
        return torch.randn(shape, generator=g, device=device, dtype=torch.float32) * s + mean

    w = {"wte.weight": nrm(cfg.vocab_size, d)}
    if not gptj:
        w["wpe.weight"] = nrm(cfg.max_position_embeddings, d, s=std / 2)
    attn = "attn." if gptj else "attn.attention."
    fc1, fc2 = ("mlp.fc_in", "mlp.fc_out") if gptj else ("mlp.c_fc", "mlp.c_proj")
    for i in range(cfg.num_layers):
        p = f"h.{i}."
        w[p + "ln_1.weight"], w[p + "ln_1.bias"] = nrm(d, s=0.1, mean=1.0), nrm(d, s=0.05)
        if not gptj:
            w[p + "ln_2.weight"], w[p + "ln_2.bias"] = nrm(d, s=0.1, mean=1.0), nrm(d, s=0.05)
            w[p + attn + "out_proj.bias"] = nrm(d)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            w[p + attn + n + ".weight"] = nrm(d, d)
        w[p + fc1 + ".weight"], w[p + fc1 + ".bias"] = nrm(ffn, d), nrm(ffn)
        w[p + fc2 + ".weight"], w[p + fc2 + ".bias"] = nrm(d, ffn), nrm(d)
    w["ln_f.weight"], w["ln_f.bias"] = nrm(d, s=0.1, mean=1.0), nrm(d, s=0.05)
    return w


def flops_per_sentence(S, L=12, d=768):
    """F(S) = L*24*S*d^2 + 2*L*d*S*(S+1) (causal-minimal attention, ffn = 4d)."""
    return L * 24 * S * d * d + 2 * L * d * S * (S + 1)


def cpu_info():
    """CPU model string, physical cores, logical CPUs of the box (lscpu; /proc/cpuinfo as the fallback)."""
    model, phys, logical = None, None, os.cpu_count()
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {ln.split(":", 1)[0].strip(): ln.split(":", 1)[1].strip() for ln in txt.splitlines() if ":" in ln}
        model = kv.get("Model name")
        phys = int(kv["Core(s) per socket"]) * int(kv["Socket(s)"])
    except Exception:  # noqa: BLE001
        try:
            with open("/proc/cpuinfo") as f:
                for ln in f:
                    if ln.startswith("model name"):
                        model = ln.split(":", 1)[1].strip()
                        break
        except OSError:
            pass
    return {"model": model, "physical_cores": phys, "logical_cpus": logical}


def cpu_threads():
    """Threads for the CPU legs: one per physical core (SMT siblings only contend for the FMA units), at most 64 -- a
    128-sentence batch of 768-wide matmuls does not scale past that."""
    info = cpu_info()
    return max(1, min(64, info["physical_cores"] or info["logical_cpus"] or 1))


def hf_cpu_baseline(ocfg, ow, sample, repeats=3, mask=None, hf=None, slice_note=""):
    """The code the reference itself runs on a CPU: HF GPTNeoModel (fp32, eager attention, eval -- the
    un-vendored dependency behind beir_dense_retriever.py:204-205) + the raw weighted-mean pooling of :258-270 +
    F.normalize, on the host cores over a bounded sample; 1 warm-up + `repeats` timed passes, median."""
    from transformers import GPTNeoConfig, GPTNeoModel
    hc = GPTNeoConfig(vocab_size=ocfg.vocab_size, max_position_embeddings=ocfg.max_position_embeddings,
                      hidden_size=ocfg.hidden_size, num_layers=ocfg.num_layers, num_heads=ocfg.num_heads,
                      intermediate_size=ocfg.intermediate_size, window_size=ocfg.window_size,
                      attention_types=[[["global", "local"], ocfg.num_layers // 2]],
                      layer_norm_epsilon=ocfg.layer_norm_epsilon, attention_dropout=0.0, resid_dropout=0.0,
                      embed_dropout=0.0)
    hc._attn_implementation = "eager"
    if hf is None:
        hf = GPTNeoModel(hc).eval()
        hf.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in ow.items()}, strict=False)
    threads = cpu_threads()
    torch.set_num_threads(threads)
    ids = torch.tensor(sample, dtype=torch.long)
    mask = torch.ones_like(ids) if mask is None else torch.tensor(mask, dtype=torch.long)

    def run(i, m):
        with torch.no_grad():
            h = hf(input_ids=i, attention_mask=m).last_hidden_state
            w = torch.arange(1, h.shape[1] + 1, dtype=torch.float32)[None, :, None] * m[:, :, None].float()
            e = (h * w).sum(1) / w.sum(1)
            return torch.nn.functional.normalize(e, dim=1)
    run(ids[:8], mask[:8])                                    # warm-up
    times = []
    for _ in range(repeats):
        t = time.perf_counter()
        emb = run(ids, mask)
        times.append(time.perf_counter() - t)
    dt = float(np.median(times))
    return emb.numpy(), {"value": round(len(sample) / dt, 2), "unit": "sentences/s", "cores": threads, "kind": "reference",
                         "threads_note": "one torch thread per physical core, capped at 64: the 128-sentence sample is [16384, 768] x [768, 768..3072] "
                                         "matmuls + 12-head attention at S = 128 -- on a 128-core host the second 64 threads add synchronisation, not rate "
                                         "(bench.py::cpu_threads)",
                         "sample": f"{len(sample)} sentences x {ids.shape[1]} tokens{slice_note}: HF GPTNeoModel fp32 eager + raw weighted-mean "
                                   f"pooling + normalise (the reference's CPU path), torch CPU, median of {repeats} passes "
                                   f"({', '.join(f'{x:.1f}' for x in times)} s)"}, hf


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_distributed(n):
    """`python bench.py --gpus N` outside torchrun: start N ranks (one per GPU) of this same command line."""
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} HIP device(s) visible")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)          # 25 x 4096 = 102 400 >= configs[1]'s 100k documents
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chunk", type=int, default=4096, help="documents per step (per GPU)")
    ap.add_argument("--call", type=int, default=1024, help="documents per sgpt_encode call")
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--nq", type=int, default=1000)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16", "fp32", "fp8", "fp8mfma"],
                    help="16-bit MFMA operand format (f16: range-guarded, meets the 1e-3 parity bar; bf16); fp32 = exact-fp32 "
                         "MFMA; fp8 = e4m3fn weight storage (SURVEY 8d cfg5), bf16 arithmetic; fp8mfma = fp8 storage + fp8 MFMA "
                         "(v_mfma_f32_16x16x128_f8f6f4) on all four projections of a block")
    ap.add_argument("--model", default="125m", choices=sorted(MODELS), help="SGPT size (default: the BASELINE metric's 125M)")
    ap.add_argument("--precise-qk", choices=["auto", "on", "off", "full", "logits", "act+logits", "full+logits", "qkv+logits", "attn"], default="auto",
                    help="the structural split-precision rule (SGPTModel(precise_qk=...)): auto = the model's default for f16 "
                         "GPT-Neo at d >= 2048 (SGPT-1.3B / 2.7B: the setting that meets the 1e-3 bar there), nothing for the "
                         "125M headline; on / full = Q / K projection over hi + lo pairs; logits = hi + lo pairs inside the "
                         "attention only; act+logits = plus the LayerNorm-1 output split")
    ap.add_argument("--precision", choices=["default", "plain", "auto", "auto-class", "x3"], default="default",
                    help="SGPTModel(precision=...): default = 'auto' for f16 (the first encode probes the checkpoint's operand "
                         "crest factors; a clean one stays on plain 16-bit operands), 'plain' otherwise; x3 = every operand as a hi + lo pair")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-1m", action="store_true", help="skip the queries/sec @ 1M-doc scoring leg")
    ap.add_argument("--cpu-sample", type=int, default=128, help="sentences in the bounded CPU-baseline sample (a slice of the 1024 x seq probe)")
    ap.add_argument("--no-varlen", action="store_true", help="skip the lengths ~U{16..128} leg")
    ap.add_argument("--equal-calls", action="store_true", help="A/B: round 4's equal token budgets per sgpt_encode call instead of the round-aware ones (variable-length leg)")
    ap.add_argument("--no-modes", action="store_true", help="skip the precision-mode leg (f16x3 and exact-fp32 encode rates beside the headline)")
    ap.add_argument("--no-other-models", action="store_true", help="skip the short runs of BASELINE configs[2..4]'s shapes (SGPT-1.3B, 5.8B bf16, bloom-7b1 fp8 MFMA)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        relaunch_distributed(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # SGPT_BENCH_FORCE_DIST=1 (with torchrun --nproc-per-node 1): run the N > 1 code path -- RCCL init, query all-gather,
    # top-k exchange, shard check -- with a world of one, on a single-GPU box
    dist_on = world > 1 or (os.environ.get("SGPT_BENCH_FORCE_DIST") == "1" and "WORLD_SIZE" in os.environ)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)    # RCCL over xGMI

    from sgpt_amd import SGPTConfig, SGPTModel, get_context, synthetic_weights
    ctx = get_context(dev)
    mkw = MODELS[args.model]
    cfg = SGPTConfig.from_hf_dict(mkw) if mkw.get("model_type") in ("gptj", "bloom") else SGPTConfig(**mkw)
    weights = synthetic_weights(cfg, seed=1) if args.model == "125m" else device_random_weights(cfg, dev)
    model = SGPTModel(cfg, weights, device=dev, dtype=args.dtype, max_tokens_per_call=args.call * args.seq,
                      precise_qk={"auto": None, "on": True, "off": False}.get(args.precise_qk, args.precise_qk),
                      **({} if args.precision == "default" else {"precision": args.precision}))
    del weights
    model.round_aware_calls = not args.equal_calls
    torch.cuda.empty_cache()
    d, S, k1 = cfg.hidden_size, args.seq, args.topk + 1      # the reference keeps top_k+1 (exact_search.py:104)
    score_dt = {"fp32": torch.float32, "f16": torch.float16}.get(args.dtype, torch.bfloat16)

    # ---- synthetic inputs, resident in HBM before the clock starts (ids ~ U{0..50255}) ----
    rng = np.random.default_rng(1000 + rank)
    n_steps = args.steps + args.warmup
    calls_per_step = (args.chunk + args.call - 1) // args.call
    packed = []
    for _ in range(n_steps):
        row = []
        left = args.chunk
        for _ in range(calls_per_step):
            nb = min(args.call, left)
            left -= nb
            ids = rng.integers(0, min(50256, cfg.vocab_size), size=(nb, S), dtype=np.int64)
            row.append(model.pack(ids))
        packed.append(row)
    # queries: lengths U{4..32}; encoded sharded by rank, then ONE all-gather
    qrng = np.random.default_rng(7)
    queries = [qrng.integers(0, 50256, size=int(qrng.integers(4, 33))).tolist() for _ in range(args.nq)]
    from sgpt_amd.dist import all_gather_queries, exchange_topk, shard_range, shard_sizes
    qlo, qhi = shard_range(args.nq, rank, world)
    mine = queries[qlo:qhi]

    corpus = torch.empty((n_steps * args.chunk, d), dtype=score_dt, device=dev)   # this rank's shard, stays in HBM
    emb32 = torch.empty((args.chunk, d), dtype=torch.float32, device=dev)

    def encode_queries():
        local_q = model.encode_ids(mine, normalize=True) if mine else torch.empty((0, d), device=dev)
        return all_gather_queries(ctx, local_q, shard_sizes(args.nq, world)) if dist_on else local_q   # ncclAllGather over xGMI (C ABI)

    def step(i, q, run):
        base = i * args.chunk
        o = 0
        for pb in packed[i]:
            model.encode_packed(pb, mode="weightedmean", normalize=True, out=emb32[o: o + pb.B])
            o += pb.B
        rows = corpus[base: base + args.chunk]
        if score_dt != torch.float32:
            ctx.to_16(emb32, score_dt, out=rows)          # corpus rows kept in HBM in the scorer's 16-bit format
        else:
            rows.copy_(emb32)
        return ctx.score_topk(q, rows, k1, idx_base=rank * n_steps * args.chunk + base, run=run, dtype=score_dt)

    def sync():
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(dev)

    q = encode_queries()
    q = ctx._operand(q, score_dt)
    run = None
    for i in range(args.warmup):
        run = step(i, q, run)
    sync()
    ctx.prof_read(reset=True)
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for i in range(args.warmup, n_steps):
        run = step(i, q, run)
    if dist_on:   # exchange step: per-rank top-(k+1) lists -> every rank, merged on the device (sgpt_exchange_topk)
        fv, fi = exchange_topk(ctx, run[0], run[1], k1)
    else:
        fv, fi = run[0], run[1]
    sync()
    dt = time.perf_counter() - t0
    ctx.prof_enable(False)
    n_launch, gemm_ms, gemm_flops = ctx.prof_read(reset=True)
    if dist_on:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(fv[:, : args.topk]).all() and (fi[:, : args.topk] >= 0).all()
    model._check_range()                                    # dtype f16: no activation left the half range (raises otherwise)

    sentences = world * args.steps * args.chunk
    sent_per_s = sentences / dt

    # ---- A/B of the measurement itself: the timed region above carries two hipEventRecord per projection launch
    # (the live GEMM durations of `roofline`); the same steps again with the recording off ----
    event_ab = None
    if world == 1:
        run2 = run
        sync()
        t1 = time.perf_counter()
        for i in range(args.warmup, n_steps):
            run2 = step(i, q, run2)
        sync()
        dt2 = time.perf_counter() - t1
        event_ab = {"sentences_per_s_without_event_records": round(args.steps * args.chunk / dt2, 1), "ms_per_step_without": round(dt2 / args.steps * 1e3, 3),
                    "event_records_in_timed_region": 2 * int(n_launch),
                    "cost_of_recording": f"{(dt / dt2 - 1) * 100:+.2f} % on the step time (same process, second pass over the same steps)"}

    # ---- the same steps with the host side of the boundary inside the clock: every sgpt_encode call first packs its
    # token ids from host memory (numpy [call, S] int64 -> the int32 arena in pinned memory) and copies them over PCIe
    # (one non-blocking H2D per call, two pinned arenas alternating: the copy of call i+1 overlaps the encode of call i).
    # `value` above starts with the packed ids resident in HBM (SURVEY 8d boundary); this is the PCIe-inclusive rate. ----
    host_steps = min(args.steps, 8)
    hrng = np.random.default_rng(3000 + rank)
    host_ids = [[hrng.integers(0, min(50256, cfg.vocab_size), size=(min(args.call, args.chunk - c0), S), dtype=np.int64)
                 for c0 in range(0, args.chunk, args.call)] for _ in range(host_steps)]

    def host_step(i, run):
        o = 0
        for ids in host_ids[i]:
            pb = model.pack(ids)                                    # host pack + pinned async H2D
            model.encode_packed(pb, mode="weightedmean", normalize=True, out=emb32[o: o + pb.B])
            o += pb.B
        rows = corpus[i * args.chunk: (i + 1) * args.chunk]
        if score_dt != torch.float32:
            ctx.to_16(emb32, score_dt, out=rows)
        else:
            rows.copy_(emb32)
        return ctx.score_topk(q, rows, k1, idx_base=rank * n_steps * args.chunk + i * args.chunk, run=run, dtype=score_dt)
    hrun = host_step(0, None)
    sync()
    t = time.perf_counter()
    for i in range(host_steps):
        hrun = host_step(i, hrun)
    sync()
    hdt = time.perf_counter() - t
    if dist_on:
        th = torch.tensor([hdt], dtype=torch.float64, device=dev)
        dist.all_reduce(th, op=dist.ReduceOp.MAX)
        hdt = float(th.item())
    sent_per_s_host = world * host_steps * args.chunk / hdt
    del host_ids

    # ---- N > 1: the sharded search (per-rank score+top-k with a global index base -> exchange -> merge) must equal a
    # single-rank search over the same documents.  Checked on the first `m` rows of every rank's shard. ----
    shard_check = None
    if dist_on:
        m = min(2048, args.chunk)
        mine_rows = corpus[:m].contiguous()
        from sgpt_amd.dist import get_comm
        all_rows = get_comm(ctx).all_gather_rows(mine_rows, [m] * world)
        stride = n_steps * args.chunk                                  # global index base of rank r = r * stride
        sv, si, _ = ctx.score_topk(q, mine_rows, k1, idx_base=rank * stride, dtype=score_dt)
        mv, mi = exchange_topk(ctx, sv, si, k1)
        wv, wi, _ = ctx.score_topk(q, all_rows, k1, idx_base=0, dtype=score_dt)
        wi = (wi // m) * stride + wi % m
        same = bool(torch.equal(mi, wi)) and bool(torch.equal(mv, wv))
        flag = torch.tensor([1 if same else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        assert int(flag.item()) == 1, "sharded search != single-rank search on the same documents"
        shard_check = {"docs": world * m, "nq": args.nq, "identical_to_single_rank": True}

    # ---- queries/sec: scoring + top-k only, against this job's encoded corpus and a 1M-doc synthetic shard (before the
    # variable-length leg below re-uses the corpus rows: a shard that starts with differently distributed documents measures
    # the overflow-recomputation path instead, see scripts/score_bench.py DRIFT=) ----
    def search_once(cmat):
        v, i, _ = ctx.score_topk(q, cmat, k1, idx_base=rank * cmat.shape[0], dtype=score_dt)
        if dist_on:
            v, i = exchange_topk(ctx, v, i, k1)
        return v, i

    def time_search(cmat, reps=5):
        search_once(cmat)
        sync()
        t = time.perf_counter()
        for _ in range(reps):
            search_once(cmat)
        sync()
        return args.nq * reps / (time.perf_counter() - t)

    qps_job = time_search(corpus[args.warmup * args.chunk:])
    qps_1m = qps_1m_enc = projected = k1001 = qps_1m_fp32 = scorer_breakdown = query_latency = None
    qps_enc_by_nq, qps_enc_ll = {}, {}
    if not args.no_1m:
        n1m = 1_000_000 // world                           # 1M-doc corpus sharded over the ranks
        big = torch.empty((n1m, d), dtype=score_dt, device=dev)
        blk = corpus[args.warmup * args.chunk:].float()
        gen = torch.Generator(device=dev).manual_seed(99 + rank)
        for s0 in range(0, n1m, blk.shape[0]):             # synthetic 1M shard: the encoded (anisotropic) embeddings,
            e0 = min(n1m, s0 + blk.shape[0])               # each copy perturbed so that scores are not exact duplicates
            noisy = blk[: e0 - s0] + 0.02 * torch.randn((e0 - s0, d), generator=gen, device=dev)
            big[s0:e0] = torch.nn.functional.normalize(noisy, dim=1).to(score_dt)
        del blk
        qps_1m = time_search(big, reps=3)
        # where a 1 M pass goes (a box-to-box spread of the scorer cannot be read off one number): its GEMM launches
        # (sample + filtered chunks; hipEvent pairs around each) against the whole enqueued pass
        scorer_breakdown = None
        if world == 1:
            ctx.prof_read(reset=True); ctx.prof_enable(True)
            sync()
            t_ = time.perf_counter()
            for _ in range(3):
                search_once(big)
            sync()
            t_pass = (time.perf_counter() - t_) / 3
            ctx.prof_enable(False)
            nl_, gms_, gfl_ = ctx.prof_read(reset=True)
            scorer_breakdown = {"ms_per_pass_with_event_records": round(t_pass * 1e3, 4), "gemm_launches_per_pass": nl_ // 3,
                                "ms_in_gemm_launches": round(gms_ / 3, 4), "ms_select_merge_prologue_and_gaps": round(t_pass * 1e3 - gms_ / 3, 4),
                                "algorithmic_tflop_per_pass": round(2.0 * args.nq * n1m * d / 1e12, 4),
                                "tflops_of_the_gemm_launches": round(2.0 * args.nq * n1m * d / max(gms_ / 3, 1e-9) / 1e9, 1),
                                "note": "launches = prologue-sampled threshold launch + filtered chunks + the predicated (no-op) fallback launches of every chunk"}
        # the same search with the query side included: token ids (host lists) -> pack -> one pinned async H2D ->
        # encode sharded over the ranks + one all-gather -> normalise -> search; nq = 16 / 128 / all
        def qps_incl_encode(nq_sub, reps=3):
            lo_, hi_ = shard_range(nq_sub, rank, world)
            sub = queries[lo_:hi_]

            def once():
                lq = model.encode_ids(sub, normalize=True) if sub else torch.empty((0, d), device=dev)
                qq = all_gather_queries(ctx, lq, shard_sizes(nq_sub, world)) if dist_on else lq
                q2 = ctx._operand(qq, score_dt)
                v2, i2, _ = ctx.score_topk(q2, big, k1, idx_base=rank * big.shape[0], dtype=score_dt)
                if dist_on:
                    v2, i2 = exchange_topk(ctx, v2, i2, k1)
            once()
            sync()
            t = time.perf_counter()
            for _ in range(reps):
                once()
            sync()
            return nq_sub * reps / (time.perf_counter() - t)
        # ---- what one rank of an 8-way sharded 1 M-document search does, measured HERE on one GPU (no 8-GPU node is
        # available to this build): the 125 000-document shard pass at nq = all / 128 with a global index base, the encode of
        # this rank's nq / 8 queries, and the fold of the 8 all-gathered top-(k+1) lists -- next to the single-GPU 1 M pass.
        # Not included: the two RCCL all-gathers over xGMI (3 MB of query rows, 0.1 MB of lists per rank). ----
        if world == 1:
            def t_of(fn, reps=5):
                fn(); sync()
                t_ = time.perf_counter()
                for _ in range(reps):
                    fn()
                sync()
                return (time.perf_counter() - t_) / reps
            shard = big[: n1m // 8]
            q128 = q[:128].contiguous()
            t_1m = args.nq / qps_1m
            # (shard pass and fold are timed as ONE enqueued sequence, as a rank runs them: the fold alone is ~15 us of device
            # time behind ~50 us of Python call overhead, which a stand-alone timing would measure instead)
            gv = torch.randn((8, args.nq, k1), device=dev).sort(dim=2, descending=True).values.contiguous()
            gi = torch.randint(0, n1m, (8, args.nq, k1), device=dev, dtype=torch.int64)

            def rank_search(qq, fold=True):
                ctx.score_topk(qq, shard, k1, idx_base=3 * shard.shape[0], dtype=score_dt)
                if fold:
                    ctx.fold_gathered_topk(gv, gi, k1)
            t_sh = t_of(lambda: rank_search(q, fold=False), reps=10)
            t_sh_fold = t_of(lambda: rank_search(q), reps=10)
            t_fold = max(t_sh_fold - t_sh, 0.0)
            t_1m_128 = t_of(lambda: ctx.score_topk(q128, big, k1, dtype=score_dt), reps=3)
            gv128, gi128 = gv[:, :128].contiguous(), gi[:, :128].contiguous()

            def rank_search_128():
                ctx.score_topk(q128, shard, k1, idx_base=3 * shard.shape[0], dtype=score_dt)
                ctx.fold_gathered_topk(gv128, gi128, k1)
            t_sh_128 = t_of(rank_search_128, reps=10)
            qs8 = queries[: max(1, args.nq // 8)]
            t_qenc_all = t_of(lambda: model.encode_ids(queries, normalize=True), reps=3)
            t_qenc_8th = t_of(lambda: model.encode_ids(qs8, normalize=True), reps=3)
            projected = {"world": 8, "measured_on": "1 GPU (per-rank work of a world of 8; xGMI all-gathers not included)",
                         "nq": args.nq, "docs_total": n1m, "docs_per_rank": int(shard.shape[0]),
                         "ms_1m_pass_single_gpu": round(t_1m * 1e3, 4), "ms_shard_pass": round(t_sh * 1e3, 4),
                         "ms_shard_pass_plus_fold_of_8_lists": round(t_sh_fold * 1e3, 4), "ms_fold_8_lists": round(t_fold * 1e3, 4),
                         "search_speedup": round(t_1m / t_sh_fold, 2),
                         "nq128": {"ms_1m_pass_single_gpu": round(t_1m_128 * 1e3, 4), "ms_shard_pass_plus_fold": round(t_sh_128 * 1e3, 4),
                                   "search_speedup": round(t_1m_128 / t_sh_128, 2)},
                         "ms_query_encode_all": round(t_qenc_all * 1e3, 4), "ms_query_encode_eighth": round(t_qenc_8th * 1e3, 4),
                         "speedup_incl_query_encode": round((t_qenc_all + t_1m) / (t_qenc_8th + t_sh_fold), 2)}
            # ---- the reference driver's real depth: k_values up to 1000 -> top_k + 1 = 1001 kept (beir_dense_retriever.py:440,
            # exact_search.py:104,126), device pass and the nq x 1001 host result-dict assembly behind it ----
            from sgpt_amd.beir import assemble_results
            kk = 1001
            t_k = t_of(lambda: ctx.score_topk(q, big, kk, dtype=score_dt), reps=3)
            v_k, i_k, _ = ctx.score_topk(q, big, kk, dtype=score_dt)
            cids = [f"d{j}" for j in range(n1m)]
            qids = [f"q{j}" for j in range(args.nq)]
            sync()
            from sgpt_amd.beir import _host_ext
            t_asm, t_asm_py = None, None
            for native_ in ((True, False) if _host_ext() is not None else (False,)):
                best = 1e9
                for _ in range(3):
                    t_ = time.perf_counter()
                    res_k = assemble_results(qids, cids, v_k.cpu().numpy(), i_k.cpu().numpy(), native=native_)
                    best = min(best, time.perf_counter() - t_)
                    assert len(res_k) == args.nq and len(res_k[qids[0]]) == kk
                    del res_k
                if native_:
                    t_asm = best
                else:
                    t_asm_py = best
            if t_asm is None:
                t_asm = t_asm_py
            k1001 = {"k": kk, "ms_device_pass": round(t_k * 1e3, 3), "queries_per_sec_device": round(args.nq / t_k, 1),
                     "ms_d2h_and_result_dict": round(t_asm * 1e3, 1),
                     "ms_d2h_and_result_dict_python_form": round(t_asm_py * 1e3, 1),
                     "result_dict": "csrc/host_assemble.c (pre-sized dicts, prefetched id strings)" if _host_ext() is not None
                                    else "Python form (host extension not built)",
                     "queries_per_sec_incl_result_dict": round(args.nq / (t_k + t_asm), 1)}
            del cids, v_k, i_k
        # ---- the scorer a `DenseRetrievalExactSearch()` caller gets BY DEFAULT (score_dtype=torch.float32: fp32 corpus rows,
        # exact-fp32 MFMA 16x16x4 -- 1/16 of the 16-bit rate; beir.py) on the same 1 M documents, next to the 16-bit scorer
        # every other queries/s figure of this line uses ----
        if world == 1 and score_dt != torch.float32:
            big32 = big.float()
            q32 = q.float()[: args.nq].contiguous()
            ctx.score_topk(q32, big32, k1, dtype=torch.float32)
            sync()
            t_ = time.perf_counter()
            for _ in range(2):
                ctx.score_topk(q32, big32, k1, dtype=torch.float32)
            sync()
            qps_1m_fp32 = args.nq * 2 / (time.perf_counter() - t_)
            del big32, q32
        qps_enc_by_nq = {n_: qps_incl_encode(n_, reps=10 if n_ <= 16 else 3) for n_ in sorted({1, 16, 128, args.nq}) if n_ <= args.nq}   # (nq = 1: 1 / value = the latency of one query, ids -> ranked list)
        qps_1m_enc = qps_enc_by_nq[args.nq]
        # the opt-in low-latency mode (k-groups in the small-tile GEMM: not bit-identical across batch sizes), small nq only
        prev_ll = ctx.set_low_latency(True)
        qps_enc_ll = {n_: qps_incl_encode(n_) for n_ in (16, 128) if n_ <= args.nq}
        ctx.set_low_latency(prev_ll)
        # ---- query-sized encodes on their own (round 6: csrc/qgemm.hip), next to the same layouts on the bulk path's small-tile
        # kernels (sgpt_ctx_set_tile_policy(2): what rounds 1-5 ran) -- same box, same process, identical bits ----
        if world == 1 and args.dtype in ("f16", "bf16"):
            def enc_ms(pb_, reps=40):
                for _ in range(5):
                    model.encode_packed(pb_, normalize=True)
                sync()
                t_ = time.perf_counter()
                for _ in range(reps):
                    model.encode_packed(pb_, normalize=True)
                sync()
                return (time.perf_counter() - t_) / reps * 1e3
            query_latency = {"unit": "ms per sgpt_encode call (packed ids resident), weighted-mean pool + normalise", "by_nq": {}}
            for n_ in (1, 16, 125):
                if n_ > args.nq:
                    continue
                pb_ = model.pack(queries[:n_])
                new_ms = enc_ms(pb_)
                prev_pol = ctx.set_tile_policy(2)
                try:
                    old_ms = enc_ms(pb_)
                    same = bool(torch.equal(model.encode_packed(pb_, normalize=True), (ctx.set_tile_policy(0), model.encode_packed(pb_, normalize=True))[1]))
                finally:
                    ctx.set_tile_policy(prev_pol)
                query_latency["by_nq"][str(n_)] = {"token_rows": int(pb_.T_pad), "ms": round(new_ms, 4), "ms_bulk_path_small_tiles": round(old_ms, 4),
                                                   "identical_bits": same}
        if dist_on:
            keys, keys_ll = sorted(qps_enc_by_nq), sorted(qps_enc_ll)
            tq = torch.tensor([qps_1m] + [qps_enc_by_nq[k_] for k_ in keys] + [qps_enc_ll[k_] for k_ in keys_ll],
                              dtype=torch.float64, device=dev)
            dist.all_reduce(tq, op=dist.ReduceOp.MIN)
            qps_1m = float(tq[0].item())
            qps_enc_by_nq = {k_: float(tq[1 + j].item()) for j, k_ in enumerate(keys)}
            qps_enc_ll = {k_: float(tq[1 + len(keys) + j].item()) for j, k_ in enumerate(keys_ll)}
            qps_1m_enc = qps_enc_by_nq[args.nq]
        del big

    # ---- the same step on documents of lengths ~U{16..128} (SURVEY 8d cfg2 asks for both) ----
    varlen = None
    if not args.no_varlen:
        vrng = np.random.default_rng(2000 + rank)
        v_steps = max(1, min(max(4, args.steps // 3), n_steps - 1))
        vpacked, v_tokens, v_flops = [], 0, 0.0
        for _ in range(v_steps + 1):
            lens = vrng.integers(16, S + 1, size=args.chunk)
            docs = [vrng.integers(0, min(50256, cfg.vocab_size), size=int(n)) for n in lens]
            plan = model.plan_batches(lens.astype(np.int64))
            vpacked.append([model.pack([docs[i] for i in sel]) for sel in plan])
            v_tokens += int(lens.sum())
            v_flops += float(sum(flops_per_sentence(int(n), cfg.num_layers, cfg.hidden_size) for n in lens))
        per_step_tokens, per_step_flops = v_tokens / (v_steps + 1), v_flops / (v_steps + 1)

        def vstep(i, run):
            o = 0
            for pb in vpacked[i]:
                model.encode_packed(pb, mode="weightedmean", normalize=True, out=emb32[o: o + pb.B])
                o += pb.B
            rows = corpus[i * args.chunk: (i + 1) * args.chunk]
            if score_dt != torch.float32:
                ctx.to_16(emb32, score_dt, out=rows)
            else:
                rows.copy_(emb32)
            return ctx.score_topk(q, rows, k1, idx_base=i * args.chunk, run=run, dtype=score_dt)
        vrun = vstep(0, None)
        sync()
        t = time.perf_counter()
        for i in range(1, v_steps + 1):
            vrun = vstep(i, vrun)
        sync()
        vdt = time.perf_counter() - t
        if dist_on:
            tv = torch.tensor([vdt], dtype=torch.float64, device=dev)
            dist.all_reduce(tv, op=dist.ReduceOp.MAX)
            vdt = float(tv.item())
        varlen = {"lengths": f"U{{16..{S}}}", "steps": v_steps, "rows_per_call": [int(pb.T_pad) for pb in vpacked[0]], "sentences_per_s": round(world * v_steps * args.chunk / vdt, 1),
                  "tokens_per_s": round(world * v_steps * per_step_tokens / vdt, 1),
                  "mean_len": round(per_step_tokens / args.chunk, 2),
                  "end_to_end_frac_of_mfma_roofline": round(v_steps * per_step_flops / vdt / (PEAK_BF16_TFLOPS * 1e12), 4)}
        del vpacked

    # ---- the precision modes beside the headline: every operand as a hi + lo pair of halves ("f16x3": what the probe selects
    # for an ill-conditioned checkpoint) and the exact-fp32 MFMA mode, the same 1024 x seq encode calls ----
    modes = None
    mode_probe_emb = {}
    if world == 1 and args.model == "125m" and args.dtype == "f16" and not args.no_modes:
        modes = {"f16_sentences_per_s": round(sent_per_s, 1)}
        w2 = synthetic_weights(cfg, seed=1)
        probe_ids = np.random.default_rng(5).integers(0, 50256, size=(1024, S), dtype=np.int64)   # the CPU-baseline leg's probe call
        for tag, kw, calls in (("bf16", dict(dtype="bf16"), 4), ("f16x3", dict(dtype="f16", precision="x3"), 4), ("fp32", dict(dtype="fp32"), 1)):
            m2 = SGPTModel(cfg, w2, device=dev, max_tokens_per_call=args.call * args.seq, **kw)
            if tag == "bf16":       # the dtype BASELINE configs[1] names: its embeddings of the probe call, checked against the CPU reference below
                mode_probe_emb["bf16"] = m2.encode_ids(probe_ids, normalize=True)
            pbs = [m2.pack(np.random.default_rng(77 + j).integers(0, 50256, size=(args.call, S), dtype=np.int64)) for j in range(calls)]
            m2.encode_packed(pbs[0], mode="weightedmean", normalize=True, out=emb32[: args.call])
            sync()
            t = time.perf_counter()
            for pb in pbs:
                m2.encode_packed(pb, mode="weightedmean", normalize=True, out=emb32[: args.call])
            sync()
            modes[f"{tag}_sentences_per_s"] = round(calls * args.call / (time.perf_counter() - t), 1)
            m2.close()
        del w2
        torch.cuda.empty_cache()
        modes["f16x3_over_fp32"] = round(modes["f16x3_sentences_per_s"] / modes["fp32_sentences_per_s"], 2)
        modes["bf16_parity_committed"] = ("BASELINE configs[1] fixture (tests/test_gpu_parity_cfg2.py, profiles/r04_parity_numbers.txt): bf16 max|cos - ref| 2.2e-3, "
                                          "max|normalised emb - ref| 1.3e-3 -- outside the 1e-3 bar, which is why the headline runs IEEE-half operands (3.2e-4 / 1.8e-4)")
        modes["note"] = ("encode + pool only (no scoring), 1024-sentence calls; f16x3 = SGPTModel(precision='x3'): embeddings within ~1e-5 of "
                         "the fp32 reference on the engineered-outlier fixtures (tests/test_gpu_parity_large.py), selected by precision='auto'")

    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (the MFMA GEMM) from live hipEvent timings ----
    gemm_tflops = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    # fp8mfma: all four projections run on the fp8 MFMA (dense peak 5 PFLOP/s, MI355X_MICROARCH.md)
    peak = 157.3 if args.dtype == "fp32" else (5000.0 if args.dtype == "fp8mfma" else PEAK_BF16_TFLOPS)
    # HBM bytes per launch from the committed PMC passes of this same command (profiles/, rocprofv3 --pmc
    # FETCH_SIZE / WRITE_SIZE in their own runs, gfx950 corrections applied there); null if not collected
    traffic, traffic_source = None, None
    mfma_busy = eff_clock = None
    for tname in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", tname)
        if args.dtype in ("bf16", "f16") and args.call * S == 131072 and os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj.get("gemm_avg_hbm_bytes_per_launch")
            mfma_busy, eff_clock = tj.get("gemm_mfma_busy_frac"), tj.get("gemm_effective_clock_ghz")
            traffic_source = f"profiles/{tname} (replayed from the committed rocprofv3 --pmc passes of this command, not measured in this run)"
            break
    # the vendor yardstick of the same five launch shapes (committed measurement, same kernels): what hipBLASLt reaches with NO
    # epilogue, what gemm256d reaches with bias / GELU / residual / transposition fused, and its bare k-loop
    yardstick = None
    ypath = os.path.join(ROOT, "profiles", "r05_hipblaslt_yardstick.txt")
    if os.path.exists(ypath) and args.dtype in ("f16", "bf16"):
        for ln in open(ypath):
            if ln.startswith(args.dtype) and "five launches of a block" in ln:
                cells = [c.split() for c in ln.split("|")[1:]]
                yardstick = {"vendor_gemm_no_epilogue_frac": float(cells[0][2]), "gemm256d_with_fused_epilogues_frac": float(cells[1][2]),
                             "gemm256d_bare_k_loop_frac": float(cells[2][2]),
                             "source": "profiles/r05_hipblaslt_yardstick.txt (scripts/hipblaslt_yardstick.py: torch.matmul -> hipBLASLt vs sgpt_bench_gemm, "
                                       "five projection shapes at 131 072 rows, interleaved rounds, replayed -- not measured in this run)"}
    roofline = {"bound": "mfma", "yardstick": yardstick,
                "kernel": "gemm256 (16-bit operands, 256x256x64 persistent LDS-DMA GEMM, asymmetric 3+2-slot LDS ring; the 4 projection launches per block: QKV in one, out-proj, fc1, fc2)"
                          if args.dtype != "fp32" else "gemm_kernel<float> (exact fp32 MFMA 128x128x32)",
                "achieved": round(gemm_tflops, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(gemm_tflops / peak, 4),
                "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC, avg over the GEMM launch shapes of a block)",
                "traffic_source": traffic_source,
                "mfma_busy": mfma_busy, "effective_clock_ghz": eff_clock,
                "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) and GRBM_GUI_ACTIVE / 8 / kernel duration, time-weighted "
                                  "over the five projection launches, from the same committed PMC passes as `traffic`",
                "algorithmic_flops_per_launch": round(gemm_flops / max(n_launch, 1), 1),
                "launches": n_launch, "avg_launch_ms": round(gemm_ms / max(n_launch, 1), 5),
                "gemm_share_of_step": round(gemm_ms * 1e-3 / dt, 4),
                "end_to_end_frac_of_mfma_roofline": round(
                    sent_per_s / world * flops_per_sentence(S, cfg.num_layers, cfg.hidden_size) / (PEAK_BF16_TFLOPS * 1e12), 4)}

    # ---- CPU baseline + parity probe (rank 0, N = 1 only; bounded: ~10-30 s of CPU work) ----
    # The GPU encodes 1024 sentences in ONE call (131 072 tokens: every projection runs on the 256x256-tile throughput
    # kernels, the ones the timed region uses); every 16th row is checked against the CPU paths on the same ids.
    cpu = None
    if not args.no_cpu_baseline and world == 1 and args.model == "125m":
        from oracle import sgpt_oracle as O      # checker / reported baseline only, never the measured path
        ocfg = O.NeoConfig(**SGPT_125M)
        ow = O.synth_weights(ocfg, seed=1)
        probe = np.random.default_rng(5).integers(0, 50256, size=(1024, S), dtype=np.int64)
        got_all = model.encode_ids(probe, normalize=True)
        stride = max(1, 1024 // args.cpu_sample)
        pick = np.arange(0, 1024, stride)[: args.cpu_sample]
        sample = probe[pick].tolist()
        got = got_all[torch.from_numpy(pick).to(dev)].cpu().numpy()
        O.encode(ow, ocfg, sample[:4], batch_size=4)                    # warm-up
        t = time.perf_counter()
        ce = O.encode(ow, ocfg, sample, batch_size=16, normalize_embeddings=True)
        cdt = time.perf_counter() - t
        # cosine scores as the scorer computes them (rows rounded to the 16-bit corpus format) vs fp32 on the CPU
        g16 = ctx._operand(got_all[torch.from_numpy(pick).to(dev)].contiguous(), score_dt)
        gcos = ctx.scores(g16, g16, dtype=score_dt).cpu().numpy()
        port = {"value": round(len(sample) / cdt, 2), "unit": "sentences/s", "cores": os.cpu_count(), "kind": "port",
                "sample": f"{len(sample)} sentences x {S} tokens, numpy fp32 oracle (oracle/sgpt_oracle.py), "
                          f"encode+pool+normalise, {cdt:.1f}s"}
        parity = {"checked_rows": len(sample), "gpu_call_sentences": 1024, "gpu_dtype": args.dtype,
                  "note": "embeddings = L2-NORMALISED rows (what cosine retrieval consumes; raw pooled rows are O(1-3) and deviate by the same "
                          "figure relative to their norm)",
                  "gpu_vs_oracle_max_abs_emb_diff": float(np.abs(got - ce).max()),
                  "gpu_vs_oracle_max_abs_cos_diff": float(np.abs(gcos - ce @ ce.T).max())}
        try:
            hf_emb, cpu, hf = hf_cpu_baseline(ocfg, ow, sample, slice_note=f" (every {stride}th row of SURVEY 8(d)'s 1024 x {S} slice: "
                                              "three passes over all of it would be minutes of CPU work)")
            parity["gpu_vs_hf_max_abs_emb_diff"] = float(np.abs(got - hf_emb).max())
            parity["gpu_vs_hf_max_abs_cos_diff"] = float(np.abs(gcos - hf_emb @ hf_emb.T).max())
            parity["oracle_vs_hf_max_abs_emb_diff"] = float(np.abs(ce - hf_emb).max())
            if modes is not None and "bf16" in mode_probe_emb:
                gb = mode_probe_emb["bf16"][torch.from_numpy(pick).to(dev)].contiguous()
                gb16 = ctx._operand(gb, torch.bfloat16)
                bcos = ctx.scores(gb16, gb16, dtype=torch.bfloat16).cpu().numpy()
                modes["bf16_live_parity_vs_hf_cpu"] = {"checked_rows": len(sample), "max_abs_normalised_emb_diff": float(np.abs(gb.cpu().numpy() - hf_emb).max()),
                                                       "max_abs_cos_diff": float(np.abs(bcos - hf_emb @ hf_emb.T).max())}
            cpu["checker_port"] = port
            # BASELINE configs[0]: the reference's own CPU-runnable case, 32 sentences of 8..64 tokens (right-padded batch)
            c1rng = np.random.default_rng(0)
            c1 = [c1rng.integers(0, 50256, size=int(c1rng.integers(8, 65))).tolist() for _ in range(32)]
            c1_ids, c1_mask = O.pad_batch(c1, pad_id=O.GPT2_PAD, side="right")
            c1_emb, c1_res, _ = hf_cpu_baseline(ocfg, ow, c1_ids.tolist(), mask=c1_mask.tolist(), hf=hf)
            c1_gpu = model.encode_ids(c1, normalize=True).cpu().numpy()
            c1_res["gpu_vs_hf_max_abs_emb_diff"] = float(np.abs(c1_gpu - c1_emb).max())
            c1_res["sample"] = "BASELINE configs[0]: 32 sentences of 8..64 tokens, one right-padded batch; " + c1_res["sample"]
            cpu["cfg1_32x64"] = c1_res
            del hf
        except Exception as e:  # noqa: BLE001 -- `transformers` missing on the box: the port is the baseline
            cpu = dict(port, hf_error=f"{type(e).__name__}: {e}"[:200])
        # queries/s of the reference's CPU search leg (exact_search.py:96-132 restated: cos_sim + top-k + heap merge per
        # 50k-document chunk) for nq = 128 against a 100k-document fp32 corpus
        crng = np.random.default_rng(11)
        cemb = crng.standard_normal((100_000, d)).astype(np.float32)
        cemb[:, :3] *= 30.0
        qemb = crng.standard_normal((128, d)).astype(np.float32)
        qemb[:, :3] *= 30.0
        qt = []
        for _ in range(3):
            t = time.perf_counter()
            O.exact_search(qemb, [f"q{i}" for i in range(128)], cemb, [f"d{i}" for i in range(100_000)], args.topk, "cos_sim",
                           chunk_size=50_000, backend="torch")
            qt.append(time.perf_counter() - t)
        cpu["search"] = {"value": round(128 / float(np.median(qt)), 1), "unit": "queries/s", "kind": "port",
                         "sample": "nq=128 vs 100k x 768 fp32 documents, cos_sim + top-k + per-query heap merge "
                                   "(oracle.exact_search(backend='torch') = exact_search.py:96-132 restated on the torch CPU "
                                   "primitives the reference calls; the reference's file is Python under /root/reference and cannot "
                                   "travel to the GPU box: scripts/cpu_search_ref_vs_port.py times the reference's own "
                                   "DenseRetrievalExactSearch beside this port on the same inputs in the build container -> "
                                   "profiles/r03_cpu_search_ref_vs_port.txt), median of 3",
                         "cores": torch.get_num_threads()}
        cpu["gpu_vs_cpu_max_abs_emb_diff"] = parity["gpu_vs_oracle_max_abs_emb_diff"]
        cpu["parity"] = parity
        cpu["cpu"] = cpu_info()

    # ---- BASELINE configs[2..4]'s shapes, driver-timed (the sizes where >= 0.50 of the MFMA roofline IS reached
    # had builder-run numbers only): three short runs of this script (3 steps of 1024 documents, encode + pool + score) in
    # processes of their own, after this one has released the GPU memory it no longer needs ----
    other_configs = None
    if world == 1 and args.model == "125m" and not args.no_other_models and not args.no_1m:
        del corpus, emb32
        torch.cuda.empty_cache()
        other_configs = {"note": "python bench.py --model M --dtype D --steps 3 --warmup 1 --chunk 1024 (1024-document steps, seq_len 128, same step "
                                 "definition); frac = end-to-end fraction of the 16-bit dense MFMA peak (fp8mfma: of that SAME 2.5 PFLOP/s peak); "
                                 "parity of each mode at its shape: profiles/r06_models.jsonl, tests/test_gpu_parity_large.py"}
        for tag, mdl, dt_ in (("configs[2] SGPT-1.3B f16 (default precise_qk)", "1.3b", "f16"), ("configs[2] SGPT-1.3B bf16", "1.3b", "bf16"),
                              ("configs[3] SGPT-5.8B bf16", "5.8b", "bf16"), ("configs[4] bloom-7b1 fp8 MFMA", "bloom-7b1", "fp8mfma")):
            try:
                r_ = subprocess.run([sys.executable, os.path.abspath(__file__), "--model", mdl, "--dtype", dt_, "--steps", "3", "--warmup", "1", "--chunk", "1024",
                                     "--no-cpu-baseline", "--no-1m", "--no-varlen", "--no-modes", "--no-other-models"], capture_output=True, text=True, timeout=240)
                ln_ = [x for x in r_.stdout.splitlines() if x.startswith("{")]
                b_ = json.loads(ln_[-1])
                other_configs[tag] = {"sentences_per_s": b_["value"], "ms_per_step": b_["ms_per_step"], "gemm_tflops_live": b_["roofline"]["achieved"],
                                      "end_to_end_frac_of_mfma_roofline": b_["roofline"]["end_to_end_frac_of_mfma_roofline"]}
            except Exception as e:  # noqa: BLE001
                other_configs[tag] = {"error": str(e)[:200]}

    out = {"metric": "encoded sentences/sec (SGPT-125M, seq_len 128, encode + weighted-mean pool + cosine top-10 "
                     "chunk loop)",
           "value": round(sent_per_s, 1), "unit": "sentences/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": ("BASELINE configs[1]: SGPT-125M" if args.model == "125m" else f"SGPT-{args.model.upper()}") +
                                  "-shape random-init weights, " + args.dtype + " MFMA" + (f" + precise_qk={model.precise_qk}" if model.precise_qk else "") +
                                  (" (DEVIATION from configs[1]'s bf16: IEEE-half operands -- same width and MFMA rate class, 3 more "
                                   "mantissa bits, the mode that meets the 1e-3 parity bar; --dtype bf16 times bf16)" if args.dtype == "f16" else "") + ", "
                                  f"{args.steps * args.chunk} docs/GPU x seq_len {S}, nq={args.nq}, cosine top-{args.topk} "
                                  f"(top_k+1 kept), corpus rows {'fp32' if args.dtype == 'fp32' else ('f16' if args.dtype == 'f16' else 'bf16')} in HBM",
                      "docs_per_step": args.chunk, "docs_per_encode_call": args.call, "seq_len": S, "nq": args.nq,
                      "top_k": args.topk, "parallelism": f"corpus-shard x{world}",
                      "scorer": {"this_line": f"{'fp32' if args.dtype == 'fp32' else ('f16' if args.dtype == 'f16' else 'bf16')} corpus rows, 16-bit MFMA scorer "
                                              "(DenseRetrievalExactSearch(score_dtype=torch.float16 / bfloat16))" if args.dtype != "fp32" else "fp32 rows, exact-fp32 MFMA scorer",
                                 "drop_in_default": "DenseRetrievalExactSearch() keeps score_dtype=torch.float32: fp32 corpus rows on the exact-fp32 MFMA "
                                                    "(queries_per_sec_at_1M_corpus_fp32_scorer); the 16-bit scorer is opt-in"},
                      "precision": getattr(model, "precision", "plain"),
                      "precision_probe": None if not model.precision_report else {
                          "decided": model.precision_report["decided"], "flagged_classes": model.precision_report["flagged"],
                          "crest_max_ln1_ctx_ln2_h": [round(float(v), 1) for v in model.precision_report["crest"].max(0)],
                          "limits": model.precision_report["limits"]},
                      "split_plan_entries": int((model.precision_plan() != 0).sum()) if args.dtype in ("f16", "bf16") else 0},
           "value_incl_host_pack_and_h2d": round(sent_per_s_host, 1),
           "host_leg": f"{host_steps} of the same steps with every call's ids packed from host memory and copied over PCIe inside "
                       "the clock (pinned double arena, async H2D overlapping the previous call's encode)",
           "queries_per_sec_at_job_corpus": round(qps_job, 1),
           "job_corpus_docs_per_gpu": args.steps * args.chunk,
           "queries_per_sec_at_1M_corpus": None if qps_1m is None else round(qps_1m, 1),
           "queries_per_sec_at_1M_corpus_fp32_scorer": None if qps_1m_fp32 is None else round(qps_1m_fp32, 1),
           "queries_per_sec_at_1M_corpus_incl_query_encode": None if qps_1m_enc is None else round(qps_1m_enc, 1),
           "queries_per_sec_at_1M_corpus_incl_query_encode_by_nq": {str(k_): round(v_, 1) for k_, v_ in qps_enc_by_nq.items()},
           "queries_per_sec_at_1M_corpus_incl_query_encode_low_latency_mode": {str(k_): round(v_, 1) for k_, v_ in qps_enc_ll.items()},
           "low_latency_mode_note": "round 6: layouts of <= 4096 token rows run csrc/qgemm.hip, which keeps the k-ascending sum and ignores the opt-in "
                                    "k-group mode (sgpt_ctx_set_low_latency): these figures equal the default mode's to within noise",
           "queries_per_sec_at_1M_corpus_k1001": k1001, "projected_8gpu": projected, "precision_modes": modes,
           "varlen": varlen, "shard_check": shard_check,
           "query_encode_latency": query_latency, "scorer_breakdown_1M_pass": scorer_breakdown, "event_record_ab": event_ab,
           "other_configs": other_configs,
           "roofline": roofline, "cpu_baseline": cpu}
    print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
