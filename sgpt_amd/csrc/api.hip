// C ABI (include/sgpt_hip.h): context, packed weights, forward orchestration, scorer.
// Host-side C++ only -- every device op is one of the hand-written kernels in this directory.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/sgpt_hip.h"
#include "common.h"
#include "ctx.h"


struct LayerW {
    void* w_qkv = nullptr;   // [3d, d]  (q rows, k rows, v rows)
    // split-precision copies, rows [W_hi | W_hi | W_lo] of 3 x the input width: w_qkv3 [2d or 3d, 3d] (qk_split alone: the q and k
    // rows; split_weights: q, k and v rows), w_o3 [d, 3d], w_fc3 [ffn, 3d], w_proj3 [d, 3 ffn]
    void *w_qkv3 = nullptr, *w_o3 = nullptr, *w_fc3 = nullptr, *w_proj3 = nullptr;
    void* w_o = nullptr;     // [d, d]
    void* w_fc = nullptr;    // [ffn, d]
    void* w_proj = nullptr;  // [d, ffn]
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *b_o, *b_fc, *b_proj;
    float* b_qkv = nullptr;  // BLOOM: [3d] de-interleaved (q | k | v) projection bias
    // SGPT_FP8W: w_* hold e4m3fn codes, s_* the per-output-channel power-of-two scales
    float *s_qkv = nullptr, *s_o = nullptr, *s_fc = nullptr, *s_proj = nullptr;
    int is_local = 0;
};

struct sgpt_model {
    sgpt_ctx* ctx;
    sgpt_model_desc d;
    std::vector<LayerW> L;
    float *wte = nullptr, *wpe = nullptr, *lnf_g = nullptr, *lnf_b = nullptr;
    float *rot_sin = nullptr, *rot_cos = nullptr;   // GPT-J rotary tables [max_pos, rotary_dim/2]
    float *emb_ln_g = nullptr, *emb_ln_b = nullptr, *alibi = nullptr;   // BLOOM: embedding LayerNorm, ALiBi slopes [H]
    float* zero_bias = nullptr;                      // [max(d, ffn)] zeros: bias-free projections (GPT-J out_proj)
    float* pool_w = nullptr; int pool_w_n = 0;       // learntmean position weights (sgpt_model_set_pool_weights)
    float *lm_w = nullptr, *lm_b = nullptr;           // LM head [vocab, d] (+bias): tied to the embedding unless "lm_head.*" was given
    void* dq[4] = {nullptr, nullptr, nullptr, nullptr};   // SGPT_FP8W / FP8M: bf16 scratch for the current block's qkv / o / fc / proj
    // SGPT_FP8M (fp8 MFMA on the MLP projections): per-block power-of-two scale of the GELU output's e4m3 codes, set by
    // calibration; h_amax = device float bits [n_layers] collected while `calibrating`
    std::vector<float> act_scale;      // [2 * n_layers]: GELU-output scales, then attention-context scales
    unsigned* h_amax = nullptr;        // device float bits [2 * n_layers], same order
    bool calibrating = false;
    // SGPT_F16 range shifts: operand class c of block l is STORED as value * 2^-shift[l * RS_N + c] (classes: RS_*), the
    // consuming launches multiply their fp32 accumulators back (exact).  All 0 until a load-time bound or a run-time
    // magnitude asks for more (sgpt_model_range_adapt).  range_dev: device words [0] = flag (bit 0: an f16 store reached
    // RANGE_LIMIT, bit 1: an e4m3 code saturated), [1 + l * RS_N + c] = fp32 bits of the largest offending magnitude.
    std::vector<int> shift;
    unsigned* range_dev = nullptr;
    std::vector<int> ln_floor;         // [n_layers]: the LayerNorm shift sgpt_model_load derived from the parameters (set_range_shifts may not go below)
    // Precision plan: operand class c of block l enters its consumer as a split-precision (hi + lo) pair when prec[l * PC_N + c]
    // != 0 (classes: PC_*).  crest_dev: device fp32 bits [n_layers * RS_N] collected while `probing` (sgpt_model_precision_probe_*).
    std::vector<int> prec;
    bool split_all = false;            // the split copies of all four matrices exist (sgpt_model_desc.split_weights)
    int qkv3_rows = 0;                 // row blocks of w_qkv3: 2 (q, k: qk_split alone) | 3 (q, k, v: split_weights) | 0 (none)
    unsigned* crest_dev = nullptr;
    bool probing = false;
    std::vector<void*> allocs;
};

// operand classes of a block: LayerNorm-1 output, q | k | v (and the attention context, a convex combination of v rows),
// LayerNorm-2 output, GELU output
enum { RS_LN1 = 0, RS_QKV = 1, RS_LN2 = 2, RS_H = 3, RS_N = 4 };
// precision classes (include/sgpt_hip.h SGPT_PC_*): LayerNorm-1 output -> Q / K (/ V) projection; q | k | v | p inside the attention;
// attention context -> out-projection; LayerNorm-2 output -> fc1; GELU output -> fc2
enum { PC_LN1 = SGPT_PC_LN1, PC_ATT = SGPT_PC_ATT, PC_CTX = SGPT_PC_CTX, PC_LN2 = SGPT_PC_LN2, PC_H = SGPT_PC_H, PC_N = SGPT_PREC_CLASSES };
constexpr int RS_MAX_SHIFT = 40;
static inline float pow2f(int k) { return std::ldexp(1.0f, k); }

namespace {

#define HIPC(ctx, call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return SGPT_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

sgpt_status fail(sgpt_ctx* c, sgpt_status st, const std::string& m) {
    if (c) c->err = m;
    return st;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// environment switches of the measurement scripts: compiled in only with -DSGPT_EXPERIMENTS (libsgpt_hip_exp.so)
#ifdef SGPT_EXPERIMENTS
const char* exp_env(const char* name) { return getenv(name); }
#else
const char* exp_env(const char*) { return nullptr; }
#endif

sgpt_status ensure(sgpt_ctx* c, void** p, size_t* have, size_t need) {
    if (*have >= need) return SGPT_OK;
    if (*p) { HIPC(c, hipDeviceSynchronize()); HIPC(c, hipFree(*p)); *p = nullptr; *have = 0; }
    need = align_up(need + (need >> 3), 1 << 20);
    if (hipMalloc(p, need) != hipSuccess) { *p = nullptr; return fail(c, SGPT_ERR_OOM, "hipMalloc workspace failed"); }
    HIPC(c, hipMemset(*p, 0, need));
    *have = need;
    c->generation++;
    return SGPT_OK;
}

struct Prof {  // brackets one GEMM launch with events when profiling is on
    sgpt_ctx* c; hipStream_t s; bool on; size_t slot = 0;
    Prof(sgpt_ctx* c_, hipStream_t s_, double flops) : c(c_), s(s_), on(c_->prof) {
        if (!on) return;
        if (c->ev_used == c->ev_pool.size()) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            c->ev_pool.emplace_back(a, b);
            c->ev_flops.push_back(0);
        }
        slot = c->ev_used++;
        c->ev_flops[slot] = flops;
        (void)hipEventRecord(c->ev_pool[slot].first, s);
    }
    ~Prof() { if (on) (void)hipEventRecord(c->ev_pool[slot].second, s); }
};

void gemm(sgpt_ctx* c, int dtype, int epi, int out_dtype, const GemmArgs& a0, hipStream_t s) {
    Prof p(c, s, 2.0 * (double)a0.m_valid * a0.N * (a0.k_algo > 0 ? a0.k_algo : a0.K));   // algorithmic FLOPs (split blocks not counted)
    GemmArgs a = a0;
    a.kgroups = c->kgroups; a.force256 = c->force256; a.cu_cap = c->cu_cap;     // per-ctx policies (no process-global state)
    launch_gemm(dtype, epi, out_dtype, a, s);
}

// query-sized projection (qgemm.hip); false = not served, the caller launches gemm()
bool qgemm(sgpt_ctx* c, int dtype, int epi, int out_dtype, const QGemmArgs& a, hipStream_t s) {
    Prof p(c, s, 2.0 * (double)a.g.m_valid * a.g.N * a.g.K);
    return launch_qgemm(dtype, epi, out_dtype, a, s);
}

}  // namespace

extern "C" {

int sgpt_abi_version(void) { return SGPT_ABI_VERSION; }

sgpt_status sgpt_ctx_create(int hip_device, sgpt_ctx** out) {
    if (!out) return SGPT_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || hip_device < 0 || hip_device >= n) return SGPT_ERR_HIP;
    if (hipSetDevice(hip_device) != hipSuccess) return SGPT_ERR_HIP;
    sgpt_ctx* c = new sgpt_ctx();
    c->device = hip_device;
    if (hipMalloc((void**)&c->range_flag, 256) != hipSuccess || hipMemset(c->range_flag, 0, 256) != hipSuccess) {
        delete c;
        return SGPT_ERR_OOM;
    }
    *out = c;
    return SGPT_OK;
}

void sgpt_ctx_destroy(sgpt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->ws) (void)hipFree(c->ws);
    if (c->ws2) (void)hipFree(c->ws2);
    (void)sgpt_comm_destroy(c);
    if (c->ws3) (void)hipFree(c->ws3);
    if (c->ws4) (void)hipFree(c->ws4);
    if (c->range_flag) (void)hipFree(c->range_flag);
    for (auto& e : c->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    delete c;
}

const char* sgpt_last_error(const sgpt_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

uint64_t sgpt_ctx_generation(const sgpt_ctx* c) { return c ? c->generation : 0; }

sgpt_status sgpt_ctx_reserve(sgpt_ctx* c, size_t encode_bytes, size_t score_bytes) {
    if (!c) return SGPT_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    sgpt_status st = ensure(c, &c->ws, &c->ws_bytes, encode_bytes);
    if (st != SGPT_OK) return st;
    return ensure(c, &c->ws2, &c->ws2_bytes, score_bytes);
}

sgpt_status sgpt_range_check(sgpt_ctx* c, int32_t* flagged, int32_t reset, void* stream) {
    if (!c || !flagged) return SGPT_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    int h = 0;
    HIPC(c, hipMemcpyAsync(&h, c->range_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPC(c, hipStreamSynchronize(s));
    if (reset && h) HIPC(c, hipMemsetAsync(c->range_flag, 0, sizeof(int), s));
    *flagged = h;
    return SGPT_OK;
}

sgpt_status sgpt_prof_enable(sgpt_ctx* c, int32_t on) {
    if (!c) return SGPT_ERR_INVALID;
    c->prof = on != 0;
    return SGPT_OK;
}

sgpt_status sgpt_prof_read(sgpt_ctx* c, int64_t* launches, double* ms, double* flops, int32_t reset) {
    if (!c) return SGPT_ERR_INVALID;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipDeviceSynchronize());
    for (size_t i = 0; i < c->ev_used; ++i) {
        float t = 0;
        HIPC(c, hipEventElapsedTime(&t, c->ev_pool[i].first, c->ev_pool[i].second));
        c->prof_ms += t;
        c->prof_flops += c->ev_flops[i];
        c->prof_launches++;
    }
    c->ev_used = 0;
    if (launches) *launches = c->prof_launches;
    if (ms) *ms = c->prof_ms;
    if (flops) *flops = c->prof_flops;
    if (reset) { c->prof_launches = 0; c->prof_ms = 0; c->prof_flops = 0; }
    return SGPT_OK;
}

// ------------------------------------------------------------------------------------------
sgpt_status sgpt_model_load(sgpt_ctx* c, const sgpt_model_desc* d, const sgpt_tensor_view* tv, size_t nt,
                            sgpt_model** out) {
    if (!c || !d || !tv || !out) return SGPT_ERR_INVALID;
    *out = nullptr;
    HIPC(c, hipSetDevice(c->device));
    if (d->arch != SGPT_ARCH_GPTNEO && d->arch != SGPT_ARCH_GPTJ && d->arch != SGPT_ARCH_BLOOM)
        return fail(c, SGPT_ERR_INVALID, "arch must be SGPT_ARCH_GPTNEO, SGPT_ARCH_GPTJ or SGPT_ARCH_BLOOM");
    const bool gptj = d->arch == SGPT_ARCH_GPTJ, bloom = d->arch == SGPT_ARCH_BLOOM;
    const int dm = d->d_model, ffn = d->d_ffn, H = d->n_heads;
    if (dm % 128 || ffn % 128 || H <= 0 || dm % H) return fail(c, SGPT_ERR_INVALID, "d_model and d_ffn must be multiples of 128");
    const int dh = dm / H;
    if (d->compute_dtype != SGPT_F32 && dh != 64 && dh != 128 && dh != 256)
        return fail(c, SGPT_ERR_INVALID, "16-bit attention supports head_dim 64, 128 or 256");
    if (dh > 256 || dh % 4) return fail(c, SGPT_ERR_INVALID, "head_dim must be <= 256 and a multiple of 4");
    if (dm > 4096) return fail(c, SGPT_ERR_INVALID, "d_model > 4096 not supported");
    if (d->compute_dtype != SGPT_BF16 && d->compute_dtype != SGPT_F32 && d->compute_dtype != SGPT_FP8W &&
        d->compute_dtype != SGPT_F16 && d->compute_dtype != SGPT_FP8M)
        return fail(c, SGPT_ERR_INVALID, "bad compute_dtype");
    if (gptj && (d->rotary_dim <= 0 || d->rotary_dim > dh || d->rotary_dim % 2))
        return fail(c, SGPT_ERR_INVALID, "GPT-J needs an even rotary_dim in (0, head_dim]");
    if ((d->qk_split != 0 || d->split_weights != 0) && d->compute_dtype != SGPT_F16 && d->compute_dtype != SGPT_BF16)
        return fail(c, SGPT_ERR_INVALID, "qk_split / split_weights apply to SGPT_F16 / SGPT_BF16 models");

    std::unordered_map<std::string, const sgpt_tensor_view*> byname;
    for (size_t i = 0; i < nt; ++i) byname[tv[i].name] = &tv[i];
    sgpt_model* m = new sgpt_model();
    m->ctx = c;
    m->d = *d;
    m->d.layer_is_local = nullptr;
    const bool fp8 = d->compute_dtype == SGPT_FP8W || d->compute_dtype == SGPT_FP8M;
    const bool f16 = d->compute_dtype == SGPT_F16;
    const bool bf = d->compute_dtype == SGPT_BF16 || f16;          // 16-bit packed weights
    const size_t esz = fp8 ? 1 : (bf ? 2 : 4);
    sgpt_status st = SGPT_OK;
    // SGPT_F16 range audit: stats[0] = max|matmul weight|, [1] = max|LayerNorm gamma|, [2] = max|LayerNorm beta|
    unsigned* stats = nullptr;
    if (f16) {
        if (hipMalloc((void**)&stats, 16) != hipSuccess) { delete m; return fail(c, SGPT_ERR_OOM, "hipMalloc failed"); }
        (void)hipMemsetAsync(stats, 0, 16, 0);
    }

    auto find = [&](const std::string& name, int64_t numel) -> const float* {
        auto it = byname.find(name);
        if (it == byname.end()) { st = fail(c, SGPT_ERR_MISSING, "missing weight tensor: " + name); return nullptr; }
        if (it->second->numel != numel) { st = fail(c, SGPT_ERR_INVALID, "wrong numel for " + name); return nullptr; }
        return it->second->ptr;
    };
    auto dalloc = [&](size_t bytes) -> void* {
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { st = fail(c, SGPT_ERR_OOM, "hipMalloc weights failed"); return nullptr; }
        m->allocs.push_back(p);
        return p;
    };
    auto copy_f32 = [&](const std::string& name, int64_t numel) -> float* {
        const float* src = find(name, numel);
        if (!src) return nullptr;
        float* dst = (float*)dalloc(numel * 4);
        if (!dst) return nullptr;
        if (hipMemcpyAsync(dst, src, numel * 4, hipMemcpyDeviceToDevice, 0) != hipSuccess) st = fail(c, SGPT_ERR_HIP, "memcpy " + name);
        return dst;
    };
    // LayerNorm parameters whose output is rounded to the 16-bit GEMM operand format
    auto copy_ln = [&](const std::string& base, float** g, float** b) {
        *g = copy_f32(base + ".weight", dm);
        *b = copy_f32(base + ".bias", dm);
        if (f16 && *g && *b) { launch_absmax(*g, dm, stats + 1, 0); launch_absmax(*b, dm, stats + 2, 0); }
    };
    // matmul weight [rows, cols] -> packed dtype at row `row_off` of dst (fp8: codes + one scale per row)
    auto pack_rows = [&](const float* src, int64_t rows, int64_t cols, void* dst, int64_t row_off, float* scale) {
        const int64_t off = row_off * cols, numel = rows * cols;
        if (fp8) launch_fp8_quant_rows(src, rows, cols, (uint8_t*)dst + off, scale + row_off, 0);
        else if (bf) {
            launch_f32_to_16(src, numel, (bf16_t*)dst + off, f16 ? DT_F16 : DT_BF16, 0);
            if (f16) launch_absmax(src, numel, stats + 0, 0);
        }
        else if (hipMemcpyAsync((float*)dst + off, src, numel * 4, hipMemcpyDeviceToDevice, 0) != hipSuccess)
            st = fail(c, SGPT_ERR_HIP, "memcpy weight");
    };
    auto pack_w = [&](const std::string& name, int64_t rows, int64_t cols, void* dst, int64_t row_off, float* scale) {
        const float* src = find(name, rows * cols);
        if (src) pack_rows(src, rows, cols, dst, row_off, scale);
    };

    m->wte = copy_f32(bloom ? "word_embeddings.weight" : "wte.weight", (int64_t)d->vocab * dm);
    if (gptj) {
        m->rot_sin = copy_f32("rotary.sin", (int64_t)d->max_pos * (d->rotary_dim / 2));
        m->rot_cos = copy_f32("rotary.cos", (int64_t)d->max_pos * (d->rotary_dim / 2));
    } else if (bloom) {
        m->emb_ln_g = copy_f32("word_embeddings_layernorm.weight", dm);
        m->emb_ln_b = copy_f32("word_embeddings_layernorm.bias", dm);
        m->alibi = copy_f32("alibi.slopes", H);
    } else {
        m->wpe = copy_f32("wpe.weight", (int64_t)d->max_pos * dm);
    }
    // LM head of the cross-encoder scorer (crossencoder/beir/sgptce.py): GPT-Neo / BLOOM tie it to the embedding
    // (HF tie_word_embeddings), GPT-J carries lm_head.weight / lm_head.bias
    m->lm_w = m->wte;
    if (byname.count("lm_head.weight")) {
        m->lm_w = copy_f32("lm_head.weight", (int64_t)d->vocab * dm);
        if (byname.count("lm_head.bias")) m->lm_b = copy_f32("lm_head.bias", d->vocab);
    }
    m->lnf_g = copy_f32("ln_f.weight", dm);
    m->lnf_b = copy_f32("ln_f.bias", dm);
    m->zero_bias = (float*)dalloc((size_t)(ffn > dm ? ffn : dm) * 4);
    if (m->zero_bias && hipMemsetAsync(m->zero_bias, 0, (size_t)(ffn > dm ? ffn : dm) * 4, 0) != hipSuccess)
        st = fail(c, SGPT_ERR_HIP, "memset zero_bias");
    m->L.resize(d->n_layers);
    // HF state-dict names
    //   GPT-Neo h.N.attn.attention.{q,k,v,out}_proj / mlp.c_fc / mlp.c_proj          (HF:gpt_neo:84-87,302-303)
    //   GPT-J   h.N.attn.{q,k,v,out}_proj (no biases) / mlp.fc_in / mlp.fc_out          (HF:gptj:98-101,368-369)
    //   BLOOM   h.N.self_attention.{query_key_value,dense} / mlp.dense_h_to_4h / mlp.dense_4h_to_h,
    //           input_layernorm / post_attention_layernorm                               (HF:bloom:197-198,320-322,351-354)
    const std::string attn = gptj ? "attn." : "attn.attention.";
    const std::string fc1 = bloom ? "mlp.dense_h_to_4h" : (gptj ? "mlp.fc_in" : "mlp.c_fc");
    const std::string fc2 = bloom ? "mlp.dense_4h_to_h" : (gptj ? "mlp.fc_out" : "mlp.c_proj");
    const std::string ln1 = bloom ? "input_layernorm" : "ln_1", ln2 = bloom ? "post_attention_layernorm" : "ln_2";
    float* stage = nullptr;   // BLOOM: fp32 staging for the de-interleaved fused QKV weight
    if (bloom) stage = (float*)dalloc((size_t)3 * dm * dm * 4);
    for (int i = 0; i < d->n_layers && st == SGPT_OK; ++i) {
        const std::string p = "h." + std::to_string(i) + ".";
        LayerW& l = m->L[i];
        l.is_local = (gptj || bloom) ? 0 : (d->layer_is_local ? d->layer_is_local[i] : (i & 1));
        copy_ln(p + ln1, &l.ln1_g, &l.ln1_b);
        if (!gptj) copy_ln(p + ln2, &l.ln2_g, &l.ln2_b);
        else l.ln2_g = l.ln2_b = nullptr;
        if (gptj) l.b_o = m->zero_bias;
        else l.b_o = copy_f32(p + (bloom ? std::string("self_attention.dense.bias") : attn + "out_proj.bias"), dm);
        l.b_fc = copy_f32(p + fc1 + ".bias", ffn);
        l.b_proj = copy_f32(p + fc2 + ".bias", dm);
        l.w_qkv = dalloc((size_t)3 * dm * dm * esz);
        l.w_o = dalloc((size_t)dm * dm * esz);
        l.w_fc = dalloc((size_t)ffn * dm * esz);
        l.w_proj = dalloc((size_t)dm * ffn * esz);
        if (fp8) {
            l.s_qkv = (float*)dalloc((size_t)3 * dm * 4); l.s_o = (float*)dalloc((size_t)dm * 4);
            l.s_fc = (float*)dalloc((size_t)ffn * 4); l.s_proj = (float*)dalloc((size_t)dm * 4);
        }
        if (st != SGPT_OK) break;
        const bool split_all = m->d.split_weights != 0;
        const bool split = m->d.qk_split != 0 || split_all;
        if (split) { l.w_qkv3 = dalloc((size_t)(split_all ? 3 : 2) * dm * 3 * dm * 2); if (!l.w_qkv3) break; }
        if (split_all) {
            l.w_o3 = dalloc((size_t)dm * 3 * dm * 2); l.w_fc3 = dalloc((size_t)ffn * 3 * dm * 2); l.w_proj3 = dalloc((size_t)dm * 3 * ffn * 2);
            if (!l.w_o3 || !l.w_fc3 || !l.w_proj3) break;
        }
        const int dt16 = f16 ? DT_F16 : DT_BF16;
        auto pack3 = [&](const std::string& name, int64_t rows, int64_t cols, void* dst3) {
            const float* src = find(name, rows * cols);
            if (src) launch_pack_split_rows(src, rows, cols, dst3, dt16, 0);
        };
        if (bloom) {
            const float* wq = find(p + "self_attention.query_key_value.weight", (int64_t)3 * dm * dm);
            const float* bq = find(p + "self_attention.query_key_value.bias", (int64_t)3 * dm);
            l.b_qkv = (float*)dalloc((size_t)3 * dm * 4);
            if (!wq || !bq || !l.b_qkv || !stage) break;
            launch_qkv_deinterleave(wq, stage, H, dh, dm, 0);          // rows [h,3,dh] -> [q | k | v]
            launch_qkv_deinterleave(bq, l.b_qkv, H, dh, 1, 0);
            pack_rows(stage, (int64_t)3 * dm, dm, l.w_qkv, 0, l.s_qkv);
            if (split) launch_pack_split_rows(stage, (long)(split_all ? 3 : 2) * dm, dm, l.w_qkv3, dt16, 0);      // q and k (and v) rows
            pack_w(p + "self_attention.dense.weight", dm, dm, l.w_o, 0, l.s_o);
            if (split_all) pack3(p + "self_attention.dense.weight", dm, dm, l.w_o3);
        } else {
            if (split) {
                const float* wq = find(p + attn + "q_proj.weight", (int64_t)dm * dm);
                const float* wk = find(p + attn + "k_proj.weight", (int64_t)dm * dm);
                if (!wq || !wk) break;
                launch_pack_split_rows(wq, dm, dm, l.w_qkv3, dt16, 0);
                launch_pack_split_rows(wk, dm, dm, (bf16_t*)l.w_qkv3 + (size_t)dm * 3 * dm, dt16, 0);
                if (split_all) pack3(p + attn + "v_proj.weight", dm, dm, (bf16_t*)l.w_qkv3 + (size_t)2 * dm * 3 * dm);
            }
            pack_w(p + attn + "q_proj.weight", dm, dm, l.w_qkv, 0, l.s_qkv);
            pack_w(p + attn + "k_proj.weight", dm, dm, l.w_qkv, dm, l.s_qkv);
            pack_w(p + attn + "v_proj.weight", dm, dm, l.w_qkv, (int64_t)2 * dm, l.s_qkv);
            pack_w(p + attn + "out_proj.weight", dm, dm, l.w_o, 0, l.s_o);
            if (split_all) pack3(p + attn + "out_proj.weight", dm, dm, l.w_o3);
        }
        pack_w(p + fc1 + ".weight", ffn, dm, l.w_fc, 0, l.s_fc);
        pack_w(p + fc2 + ".weight", dm, ffn, l.w_proj, 0, l.s_proj);
        if (split_all) { pack3(p + fc1 + ".weight", ffn, dm, l.w_fc3); pack3(p + fc2 + ".weight", dm, ffn, l.w_proj3); }
    }
    if (fp8 && st == SGPT_OK) {
        m->dq[0] = dalloc((size_t)3 * dm * dm * 2); m->dq[1] = dalloc((size_t)dm * dm * 2);
        m->dq[2] = dalloc((size_t)ffn * dm * 2); m->dq[3] = dalloc((size_t)dm * ffn * 2);
    }
    if (d->compute_dtype == SGPT_FP8M && st == SGPT_OK) {
        m->act_scale.assign(2 * (size_t)d->n_layers, 0.0f);   // 0 = not calibrated
        m->h_amax = (unsigned*)dalloc((size_t)d->n_layers * 8);
        if (m->h_amax && hipMemsetAsync(m->h_amax, 0, (size_t)d->n_layers * 8, 0) != hipSuccess) st = fail(c, SGPT_ERR_HIP, "memset");
    }
    m->shift.assign((size_t)d->n_layers * RS_N, 0);
    m->ln_floor.assign((size_t)d->n_layers, 0);
    m->prec.assign((size_t)d->n_layers * PC_N, 0);
    m->split_all = d->split_weights != 0;
    m->qkv3_rows = d->split_weights != 0 ? 3 : (d->qk_split != 0 ? 2 : 0);
    if (d->qk_split != 0)              // the round-3 switch: the Q / K projection of every block contracts over hi + lo pairs
        for (int i = 0; i < d->n_layers; ++i) m->prec[(size_t)i * PC_N + PC_LN1] = 1;
    if (st == SGPT_OK && bf && !fp8) {
        m->crest_dev = (unsigned*)dalloc((size_t)d->n_layers * RS_N * 4);
        if (m->crest_dev && hipMemsetAsync(m->crest_dev, 0, (size_t)d->n_layers * RS_N * 4, 0) != hipSuccess) st = fail(c, SGPT_ERR_HIP, "memset");
    }
    if (st == SGPT_OK) {
        m->range_dev = (unsigned*)dalloc((size_t)(1 + d->n_layers * RS_N) * 4);
        if (m->range_dev && hipMemsetAsync(m->range_dev, 0, (size_t)(1 + d->n_layers * RS_N) * 4, 0) != hipSuccess) st = fail(c, SGPT_ERR_HIP, "memset");
    }
    if (st == SGPT_OK && hipDeviceSynchronize() != hipSuccess) st = fail(c, SGPT_ERR_HIP, "sync after weight pack");
    if (f16) {
        // f16 has 5 exponent bits.  A LayerNorm output is bounded by max|gamma| * sqrt(d) + max|beta| (|x_hat| <= sqrt(d-1)):
        // when that bound can leave the format the LayerNorm outputs are stored under a power-of-two down-shift that the
        // consuming GEMMs undo on their fp32 accumulators (exact; shift[] above).  Activations behind the GEMMs are
        // range-checked on the device at run time (RangeTrack in the store epilogues -> sgpt_model_range_check /
        // sgpt_model_range_adapt).  Only a weight outside the format is refused.
        unsigned h[4] = {0, 0, 0, 0};
        if (st == SGPT_OK && hipMemcpy(h, stats, 12, hipMemcpyDeviceToHost) != hipSuccess) st = fail(c, SGPT_ERR_HIP, "range audit");
        (void)hipFree(stats);
        float wmax, gmax, bmax;
        memcpy(&wmax, &h[0], 4); memcpy(&gmax, &h[1], 4); memcpy(&bmax, &h[2], 4);
        const float bound = gmax * sqrtf((float)dm) + bmax;
        if (st == SGPT_OK && (!(wmax < 65504.f) || !std::isfinite(bound)))
            st = fail(c, SGPT_ERR_RANGE, "SGPT_F16: a matmul weight exceeds the f16 range (or a LayerNorm parameter is not finite); load with SGPT_BF16");
        if (st == SGPT_OK && !(bound < 32768.f)) {
            int k = (int)std::ceil(std::log2(bound / 16384.f));
            k = k < 1 ? 1 : k;
            if (k > RS_MAX_SHIFT) st = fail(c, SGPT_ERR_RANGE, "SGPT_F16: LayerNorm parameters beyond any usable range shift; load with SGPT_BF16");
            for (int i = 0; i < d->n_layers && st == SGPT_OK; ++i) {
                m->shift[(size_t)i * RS_N + RS_LN1] = m->shift[(size_t)i * RS_N + RS_LN2] = k;
                m->ln_floor[i] = k;
            }
        }
    }
    if (st != SGPT_OK) { sgpt_model_free(m); return st; }
    *out = m;
    return SGPT_OK;
}

void sgpt_model_free(sgpt_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->ctx->device);
    (void)hipDeviceSynchronize();
    for (void* p : m->allocs) (void)hipFree(p);
    delete m;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// One forward over the packed token axis.  `layer_out` (fp32 [n_layers+1, B, d]) additionally receives the pooled
// vector of every hidden state on the way (sgpt_encode_layers).
static sgpt_status encode_impl(sgpt_model* m, const int32_t* ids, const int32_t* pos, const int32_t* seq_off,
                               const int32_t* seq_len, const int32_t* pad_left, int32_t B, int32_t T, int32_t max_alloc,
                               int32_t pool_mode, int32_t n_layers_run, int32_t apply_final_ln, int32_t normalize,
                               float* out, float* hidden_out, float* layer_out, float* layer_mean, void* stream) {
    if (!m) return SGPT_ERR_INVALID;
    sgpt_ctx* c = m->ctx;
    if (!ids || !pos || !seq_off || !seq_len || B <= 0 || T <= 0 || T % 32 || max_alloc <= 0 || max_alloc % 2)
        return fail(c, SGPT_ERR_INVALID, "sgpt_encode: bad token layout (T_pad % 32, max_alloc_len % 2)");
    if (n_layers_run < 0 || n_layers_run > m->d.n_layers) return fail(c, SGPT_ERR_INVALID, "sgpt_encode: n_layers_run out of range");
    if (pool_mode < 0 || pool_mode > 3) return fail(c, SGPT_ERR_INVALID, "sgpt_encode: bad pool_mode");
    if (pool_mode == SGPT_POOL_LEARNTMEAN && (out || layer_out || layer_mean)) {
        if (!m->pool_w) return fail(c, SGPT_ERR_MISSING, "sgpt_encode: learntmean needs sgpt_model_set_pool_weights first");
        // with pad_left on the device the longest padded position is not known here: the kernel clamps the table index,
        // and the Python host checks max(pad_left + len) before the call (model.py::_check_learnt)
        // (allocations are 8-row aligned: the longest sequence has at least max_alloc - 7 tokens)
        if (!pad_left && max_alloc - 7 > m->pool_w_n)
            return fail(c, SGPT_ERR_INVALID, "sgpt_encode: fewer learnt position weights than the longest sequence");
    }
    if (max_alloc > 2048) return fail(c, SGPT_ERR_INVALID, "sgpt_encode: sequence longer than 2048 tokens");
    if (!out && !hidden_out && !layer_out && !layer_mean) return fail(c, SGPT_ERR_INVALID, "sgpt_encode: no output requested");
    HIPC(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const int dm = m->d.d_model, ffn = m->d.d_ffn, H = m->d.n_heads, dh = dm / H;
    const bool fp8m = m->d.compute_dtype == SGPT_FP8M;
    const bool fp8 = m->d.compute_dtype == SGPT_FP8W || fp8m;
    const bool bf = m->d.compute_dtype != SGPT_F32;                 // 16-bit MFMA operands (bf16 or f16)
    const int dt = !bf ? SGPT_F32 : (m->d.compute_dtype == SGPT_F16 ? SGPT_F16 : SGPT_BF16);
    const size_t esz = bf ? 2 : 4;
    const size_t SLACK = 64;  // rows of zeroed slack behind buffers the attention key tiles may over-read

    const bool gptj = m->d.arch == SGPT_ARCH_GPTJ;
    // workspace carve (all offsets 256-B aligned)
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_x = carve((size_t)T * dm * 4);                        // residual stream fp32
    // Precision plan (prec[]): a split class is stored as [hi | lo | hi] rows of 3 x its width (the consuming GEMM contracts over
    // all three blocks against [W_hi | W_hi | W_lo]; a consumer that is not split reads the first block alone).  Any split at
    // all: the attention context gets its own buffer (the LayerNorm buffer has 3 d rows then).
    const bool can_split = bf && m->d.compute_dtype != SGPT_FP8W && m->d.compute_dtype != SGPT_FP8M;
    bool any_ln = false, any_att = false, any_ctx = false, any_h = false;
    if (can_split)
        for (int li = 0; li < n_layers_run; ++li) {
            const int* pc = &m->prec[(size_t)li * PC_N];
            any_ln |= pc[PC_LN1] != 0 || pc[PC_LN2] != 0; any_att |= pc[PC_ATT] != 0; any_ctx |= pc[PC_CTX] != 0; any_h |= pc[PC_H] != 0;
        }
    const bool split = any_ln || any_att || any_ctx || any_h;
    const size_t o_a = carve((size_t)T * dm * esz * (any_ln ? 3 : 1));    // LN output (GPT-Neo: also attention ctx)
    const size_t o_c = (gptj || split) ? carve((size_t)T * dm * esz * (any_ctx ? 3 : 1)) : o_a;   // GPT-J: ctx separate (ln_1 output feeds the MLP too)
    const size_t qkv_bytes = ((size_t)T + SLACK) * 3 * dm * esz;
    const size_t o_qkv = carve(qkv_bytes * (any_att ? 2 : 1));           // bf16: [T][2d] qk + V^T [d][T] (x3 attention: the lo halves behind); fp32: [T][3d]
    const size_t o_h = carve((size_t)T * ffn * esz * (any_h ? 3 : 1));                     // MLP hidden (FP8M: e4m3 codes in the same region)
    // FP8M: fp8 MFMA on all four projections when the shapes fit the 256x256x128 kernel and the activation scales are
    // calibrated; otherwise (and while calibrating) the block runs the SGPT_FP8W arithmetic (weights de-quantised to bf16)
    const bool shapes8 = gemm_fp8_shape_ok(T, ffn, dm) && gemm_fp8_shape_ok(T, dm, ffn) && gemm_fp8_shape_ok(T, 2 * dm, dm);
    bool mlp8 = fp8m && !m->calibrating && shapes8;
    if (mlp8)
        for (int li = 0; li < n_layers_run; ++li)
            if (!(m->act_scale[li] > 0.f) || !(m->act_scale[m->d.n_layers + li] > 0.f))
                return fail(c, SGPT_ERR_INVALID, "SGPT_FP8M: activation scales are not set (sgpt_model_calibrate_begin / _end, or sgpt_model_set_act_scales)");
    // Query-sized batches (round 6; qgemm.hip): at most QGEMM_MAX_ROWS token rows, plain 16-bit operands, no probe / calibration
    // pass riding on the forward.  Every projection takes the register-staged deep-prefetch kernel; at d = 512 / 768 / 1024 the two
    // LayerNorms of a sequential block (GPT-Neo, BLOOM) run inside the prologues of the projections they feed: five launches per block
    // instead of seven.  Same arithmetic per element as the bulk path (identical bits); sgpt_ctx_set_tile_policy(1 | 2) keeps the bulk kernels.
    const bool qpath = bf && !fp8 && !split && !(m->probing && m->crest_dev) && !m->calibrating && !c->force256 && !c->no_qpath && T <= QGEMM_MAX_ROWS &&
                       qgemm_shape_ok(T, 3 * dm, dm, EPI_QKV, 2 * dm) && qgemm_shape_ok(T, dm, dm, EPI_BIAS_RESID, 0) &&
                       qgemm_shape_ok(T, ffn, dm, EPI_BIAS_GELU, 0) && qgemm_shape_ok(T, dm, ffn, EPI_BIAS_RESID, 0);
    const bool qln = qpath && !gptj && qgemm_ln_ok(T, 3 * dm, dm, EPI_QKV, 2 * dm) && qgemm_ln_ok(T, ffn, dm, EPI_BIAS_GELU, 0);
    const size_t o_a8 = mlp8 ? carve((size_t)T * dm) : 0;                // FP8M: LayerNorm output as e4m3 codes
    const size_t o_sa = mlp8 ? carve((size_t)T * 4) : 0;                 //       + one scale per row
    const size_t o_lp = (layer_mean && !layer_out) ? carve((size_t)(m->d.n_layers + 1) * B * dm * 4) : 0;
    sgpt_status st = ensure(c, &c->ws, &c->ws_bytes, off);
    if (st != SGPT_OK) return st;
    char* base = (char*)c->ws;
    if (layer_mean && !layer_out) layer_out = (float*)(base + o_lp);     // per-layer pooled vectors, scratch
    float* x = (float*)(base + o_x);
    void* a = base + o_a;
    void* ctx = base + o_c;
    void* qkv = base + o_qkv;
    void* h = base + o_h;
    void* vt = bf ? (void*)((bf16_t*)qkv + ((size_t)T + SLACK) * 2 * dm) : nullptr;

    // Rows that are over-read by the attention key tiles but never written in this call must be finite
    // (a masked key contributes p = 0, and 0 * NaN = NaN): the slack behind the q/k/v buffers, and -- when
    // the attention context has its own buffer (GPT-J) -- the filler rows past the last sequence.  The
    // workspace is reused across calls / dtypes, so stale bytes there can decode to NaN.
    const long att_lo = (long)(qkv_bytes / 2);      // element distance of the lo halves of q | k and of V^T (x3 attention)
    if (bf) {
        HIPC(c, hipMemsetAsync((bf16_t*)qkv + (size_t)T * 2 * dm, 0, SLACK * 2 * dm * esz, s));
        HIPC(c, hipMemsetAsync((bf16_t*)vt + (size_t)T * dm, 0, SLACK * dm * esz, s));
        if (any_att) {
            HIPC(c, hipMemsetAsync((bf16_t*)qkv + att_lo + (size_t)T * 2 * dm, 0, SLACK * 2 * dm * esz, s));
            HIPC(c, hipMemsetAsync((bf16_t*)vt + att_lo + (size_t)T * dm, 0, SLACK * dm * esz, s));
        }
    } else {
        HIPC(c, hipMemsetAsync((float*)qkv + (size_t)T * 3 * dm, 0, SLACK * 3 * dm * esz, s));
    }
    // (query path with the LayerNorm inside the projections: no LayerNorm launch fills the buffer the context shares with it)
    if (gptj || mlp8 || split || qln) HIPC(c, hipMemsetAsync(ctx, 0, (size_t)T * dm * esz * (any_ctx ? 3 : 1), s));   // (fp8: stale bytes would decode to NaN codes)
    launch_embed(ids, pos, m->wte, m->wpe, x, T, dm, m->d.vocab, m->d.max_pos, s);
    if (m->emb_ln_g) launch_layernorm(x, m->emb_ln_g, m->emb_ln_b, x, SGPT_F32, T, dm, m->d.ln_eps, s);   // BLOOM :499
    for (int li = 0; li < n_layers_run; ++li) {
        LayerW l = m->L[li];
        if (layer_out)   // hidden_states[li] = input of block li (HF:gpt_neo:475-478)
            launch_lnf_pool(x, m->lnf_g, m->lnf_b, seq_off, seq_len, pad_left, B, dm, m->d.ln_eps, 0, pool_mode,
                            normalize, m->pool_w, m->pool_w_n, layer_out + (size_t)li * B * dm, s);
        const void* w_fc8 = l.w_fc; const void* w_proj8 = l.w_proj;       // the e4m3 codes (FP8M feeds them to the MFMA directly)
        const void* w_qkv8 = l.w_qkv; const void* w_o8 = l.w_o;
        if (fp8 && !mlp8) {  // this block's weights: e4m3fn codes * 2^k -> bf16, exact; <1 % of the block's time at T >= 16k
            launch_fp8_dequant_rows(l.w_qkv, l.s_qkv, (long)3 * dm, dm, m->dq[0], SGPT_BF16, s);
            launch_fp8_dequant_rows(l.w_o, l.s_o, dm, dm, m->dq[1], SGPT_BF16, s);
            launch_fp8_dequant_rows(l.w_fc, l.s_fc, ffn, dm, m->dq[2], SGPT_BF16, s);
            launch_fp8_dequant_rows(l.w_proj, l.s_proj, dm, ffn, m->dq[3], SGPT_BF16, s);
            l.w_qkv = m->dq[0]; l.w_o = m->dq[1]; l.w_fc = m->dq[2]; l.w_proj = m->dq[3];
        }
        // SGPT_F16 range shifts of this block (all 0 unless the checkpoint needed them): class stored as value * 2^-k
        const bool f16m = dt == SGPT_F16;
        const int* sh = &m->shift[(size_t)li * RS_N];
        const int k_ln1 = f16m ? sh[RS_LN1] : 0, k_qkv = f16m ? sh[RS_QKV] : 0, k_h = f16m ? sh[RS_H] : 0;
        const int k_ln2 = f16m ? (gptj ? sh[RS_LN1] : sh[RS_LN2]) : 0;     // GPT-J: ln_1's output feeds the MLP too
        unsigned* slots = m->range_dev + 1 + (size_t)li * RS_N;
        // this block's precision plan
        const int* pc = &m->prec[(size_t)li * PC_N];
        const int p_ln1 = can_split ? pc[PC_LN1] : 0;                  // 0 | 1 (q, k split) | 2 (q, k, v split) | 3 (q, k: activation split only)
        const bool p_att = can_split && pc[PC_ATT] != 0, p_ctx = can_split && pc[PC_CTX] != 0, p_h = can_split && pc[PC_H] != 0;
        const bool p_ln2 = can_split && pc[PC_LN2] != 0 && (!gptj || p_ln1 != 0);   // GPT-J: fc1 reads ln_1's output (its [hi | lo | hi] rows)
        unsigned* crest = (m->probing && m->crest_dev) ? m->crest_dev + (size_t)li * RS_N : nullptr;
        GemmArgs g{};
        g.A = a; g.lda = dm; g.M = T; g.m_valid = T; g.K = dm; g.ldw = dm;
        g.range_flag = f16m ? (int*)m->range_dev : nullptr;
        AttnArgs at{};
        at.dtype = dt;
        at.seq_off = seq_off; at.B = B; at.H = H; at.dh = dh; at.window = l.is_local ? m->d.window : 0;
        at.scale = m->d.attn_scale * pow2f(2 * k_qkv);      // q and k are both stored down-shifted
        at.max_alloc_len = max_alloc; at.ctx = ctx; at.ldo = dm; at.alibi = m->alibi;
        GemmArgs q{};      // fp8 projections
        q.M = T; q.m_valid = T; q.range_flag = (int*)m->range_dev;
        if (qpath) {
            // ---- query-sized block: [LN1 +] QKV -> (rope) -> attention -> out-proj + residual -> [LN2 +] fc1 + GELU -> fc2 + residual ----
            QGemmArgs qa{};
            qa.g = g; qa.eps = m->d.ln_eps;
            if (qln) { qa.x = x; qa.ln_g = l.ln1_g; qa.ln_b = l.ln1_b; qa.ln_mul = pow2f(-k_ln1); }
            else launch_layernorm(x, l.ln1_g, l.ln1_b, a, dt, T, dm, m->d.ln_eps, s, pow2f(-k_ln1));
            qa.g.W = l.w_qkv; qa.g.N = 3 * dm; qa.g.n_split = 2 * dm; qa.g.out = qkv; qa.g.ldo = 2 * dm; qa.g.out2 = vt; qa.g.ldo2 = T;
            qa.g.bias = l.b_qkv;
            qa.g.in_mul = pow2f(k_ln1); qa.g.out_mul = qa.g.out_mul2 = pow2f(-k_qkv); qa.g.range_amax = f16m ? slots + RS_QKV : nullptr;
            if (!qgemm(c, dt, EPI_QKV, dt, qa, s)) return fail(c, SGPT_ERR_INVALID, "query path: QKV projection not served");
            if (gptj) launch_rope(qkv, dt, 2 * dm, dm, pos, m->rot_sin, m->rot_cos, T, H, dh, m->d.rotary_dim, s);
            at.q = qkv; at.k = (bf16_t*)qkv + dm; at.v = vt; at.ldq = 2 * dm; at.ldvt = T;
            launch_attn_bf16(at, s);
            QGemmArgs qo{};
            qo.g = g;
            qo.g.A = ctx; qo.g.W = l.w_o; qo.g.N = dm; qo.g.out = x; qo.g.ldo = dm; qo.g.bias = l.b_o; qo.g.resid = x;
            qo.g.in_mul = pow2f(k_qkv); qo.g.out_mul = qo.g.out_mul2 = 1.0f;
            if (!qgemm(c, dt, EPI_BIAS_RESID, SGPT_F32, qo, s)) return fail(c, SGPT_ERR_INVALID, "query path: out-projection not served");
            QGemmArgs qf{};
            qf.g = g; qf.eps = m->d.ln_eps;
            if (qln) { qf.x = x; qf.ln_g = l.ln2_g; qf.ln_b = l.ln2_b; qf.ln_mul = pow2f(-k_ln2); }
            else if (!gptj) launch_layernorm(x, l.ln2_g, l.ln2_b, a, dt, T, dm, m->d.ln_eps, s, pow2f(-k_ln2));
            qf.g.W = l.w_fc; qf.g.N = ffn; qf.g.out = h; qf.g.ldo = ffn; qf.g.bias = l.b_fc;
            qf.g.in_mul = pow2f(k_ln2); qf.g.out_mul = pow2f(-k_h); qf.g.range_amax = f16m ? slots + RS_H : nullptr;
            if (!qgemm(c, dt, EPI_BIAS_GELU, dt, qf, s)) return fail(c, SGPT_ERR_INVALID, "query path: fc1 not served");
            QGemmArgs qp{};
            qp.g = g;
            qp.g.A = h; qp.g.lda = ffn; qp.g.W = l.w_proj; qp.g.N = dm; qp.g.K = ffn; qp.g.ldw = ffn; qp.g.out = x; qp.g.ldo = dm;
            qp.g.bias = l.b_proj; qp.g.resid = x;
            qp.g.in_mul = pow2f(k_h); qp.g.out_mul = qp.g.out_mul2 = 1.0f;
            if (!qgemm(c, dt, EPI_BIAS_RESID, SGPT_F32, qp, s)) return fail(c, SGPT_ERR_INVALID, "query path: fc2 not served");
            continue;
        }
        if (mlp8) {
            // ---- attention projections on the fp8 MFMA: a8 = e4m3(LN1(x) / sa[row]) feeds Q, K (row-major bf16) and V^T ----
            launch_layernorm_q8(x, l.ln1_g, l.ln1_b, base + o_a8, (float*)(base + o_sa), nullptr, dt, T, dm, m->d.ln_eps, s);
            q.A = base + o_a8; q.lda = dm; q.a_scale = (const float*)(base + o_sa); q.a_scalar = 1.0f; q.K = dm; q.ldw = dm;
            q.W = w_qkv8; q.w_scale = l.s_qkv; q.N = 2 * dm; q.bias = l.b_qkv; q.out = qkv; q.ldo = 2 * dm;
            { Prof pr(c, s, 2.0 * T * 2.0 * dm * dm); launch_gemm_fp8(EPI_STORE, dt, q, s); }
            q.W = (const uint8_t*)w_qkv8 + (size_t)2 * dm * dm; q.w_scale = l.s_qkv + 2 * dm; q.N = dm;
            q.bias = l.b_qkv ? l.b_qkv + 2 * dm : nullptr; q.out = vt; q.ldo = T;
            { Prof pr(c, s, 2.0 * T * (double)dm * dm); launch_gemm_fp8(EPI_VT, dt, q, s); }
            if (gptj) launch_rope(qkv, dt, 2 * dm, dm, pos, m->rot_sin, m->rot_cos, T, H, dh, m->d.rotary_dim, s);
            at.q = qkv; at.k = (bf16_t*)qkv + dm; at.v = vt; at.ldq = 2 * dm; at.ldvt = T;
            at.out_fp8 = 1; at.out_scale = m->act_scale[m->d.n_layers + li]; at.range_flag = (int*)m->range_dev;
            launch_attn_bf16(at, s);                         // context as e4m3 codes of ctx / s_c, [T][dm] bytes
            q.A = ctx; q.lda = dm; q.a_scale = nullptr; q.a_scalar = m->act_scale[m->d.n_layers + li];
            q.W = w_o8; q.w_scale = l.s_o; q.N = dm; q.K = dm; q.ldw = dm; q.bias = l.b_o; q.resid = x; q.out = x; q.ldo = dm;
            { Prof pr(c, s, 2.0 * T * (double)dm * dm); launch_gemm_fp8(EPI_BIAS_RESID, 0, q, s); }
            q.resid = nullptr;
        } else {
        const long lda1 = p_ln1 ? 3 * dm : dm;                          // row stride of this block's LayerNorm-1 output
        if (p_ln1) launch_layernorm_split(x, l.ln1_g, l.ln1_b, a, dt, T, dm, m->d.ln_eps, s, pow2f(-k_ln1));
        else launch_layernorm(x, l.ln1_g, l.ln1_b, a, dt, T, dm, m->d.ln_eps, s, pow2f(-k_ln1));
        if (crest && bf) launch_crest16(a, T, dm, lda1, dt, crest + RS_LN1, s);
        if (bf) {
            // Q,K -> qk[T][2d] row-major ; V -> V^T[d][T]   (p_att: each also as a lo half, att_lo elements behind)
            g.W = l.w_qkv; g.out = qkv; g.ldo = 2 * dm; g.bias = l.b_qkv;                           // bias: BLOOM only
            g.lda = lda1;
            g.in_mul = pow2f(k_ln1); g.out_mul = g.out_mul2 = pow2f(-k_qkv); g.range_amax = f16m ? slots + RS_QKV : nullptr;
            g.lo_delta = p_att ? att_lo : 0; g.lo_delta2 = p_att ? att_lo : 0;
            if ((p_ln1 == 0 || p_ln1 == 2) && gemm_qkv_one_launch(T, 2 * dm, c->force256 != 0)) {        // query-sized batch: one launch (a launch costs ~8 us there)
                if (p_ln1 == 2) { g.W = l.w_qkv3; g.K = 3 * dm; g.ldw = 3 * dm; g.k_algo = dm; }
                g.N = 3 * dm; g.n_split = 2 * dm; g.out2 = vt; g.ldo2 = T;
                gemm(c, dt, EPI_QKV, dt, g, s);
                g.out2 = nullptr; g.n_split = 0;
            } else if (p_ln1 == 0 && !p_att && g.bias == nullptr && gemm_qkv_bulk(T, 3 * dm, dm, 2 * dm, c->force256 != 0)) {
                // bulk batch, plain operands, no bias (GPT-Neo / GPT-J): q | k | V^T from ONE launch of the 256x256 kernel -- the V
                // column tiles run with swapped operand roles and leave through the same whole-row store epilogue (gemm.hip);
                // the LayerNorm output panel is read once, one launch boundary less per block.  Same sums: identical bits.
                g.N = 3 * dm; g.n_split = 2 * dm; g.out2 = vt; g.ldo2 = T;
                gemm(c, dt, EPI_QKV, dt, g, s);
                g.out2 = nullptr; g.n_split = 0;
            } else {
                // split Q / K: a_hi.W_hi + a_lo.W_hi + a_hi.W_lo as ONE contraction over K' = 3d; V from the hi block alone
                // unless the plan splits it too (p_ln1 == 2)
                // (p_ln1 == 3: the activation alone is split -- the first TWO blocks of both layouts, [a_hi | a_lo] . [W_hi | W_hi])
                if (p_ln1) { g.K = (p_ln1 == 3 ? 2 : 3) * dm; g.ldw = 3 * dm; g.k_algo = dm; g.W = l.w_qkv3; }
                g.N = 2 * dm;
                gemm(c, dt, EPI_STORE, dt, g, s);
                if (p_ln1 == 2) g.W = (bf16_t*)l.w_qkv3 + (size_t)2 * dm * 3 * dm;
                else { g.K = dm; g.ldw = dm; g.k_algo = 0; g.W = (bf16_t*)l.w_qkv + (size_t)2 * dm * dm; }
                g.N = dm; g.out = vt; g.ldo = T;
                g.bias = l.b_qkv ? l.b_qkv + 2 * dm : nullptr;
                gemm(c, dt, EPI_VT, dt, g, s);
            }
            g.K = dm; g.ldw = dm; g.k_algo = 0; g.lda = dm; g.lo_delta = g.lo_delta2 = 0;
            if (gptj) launch_rope(qkv, dt, 2 * dm, dm, pos, m->rot_sin, m->rot_cos, T, H, dh, m->d.rotary_dim, s);
            at.q = qkv; at.k = (bf16_t*)qkv + dm; at.v = vt; at.ldq = 2 * dm; at.ldvt = T;
            at.x3 = p_att ? 1 : 0; at.qk_lo_delta = att_lo; at.v_lo_delta = att_lo;
            at.ldo = p_ctx ? 3 * dm : dm; at.ctx_lo_delta = p_ctx ? dm : 0; at.ctx_hi2_delta = p_ctx ? 2 * dm : 0;
            launch_attn_bf16(at, s);
            if (crest) launch_crest16(ctx, T, dm, at.ldo, dt, crest + RS_QKV, s);
            if (m->calibrating)   // FP8M calibration: range of this block's attention context
                launch_absmax16(ctx, (long)T * dm, dt, m->h_amax + m->d.n_layers + li, s);
        } else {
            g.W = l.w_qkv; g.N = 3 * dm; g.out = qkv; g.ldo = 3 * dm; g.bias = l.b_qkv;
            gemm(c, dt, EPI_STORE, SGPT_F32, g, s);
            if (gptj) launch_rope(qkv, SGPT_F32, 3 * dm, dm, pos, m->rot_sin, m->rot_cos, T, H, dh, m->d.rotary_dim, s);
            at.q = qkv; at.k = (float*)qkv + dm; at.v = (float*)qkv + 2 * dm; at.ldq = 3 * dm;
            launch_attn_f32(at, s);
        }
        // x += ctx . Wo^T (+ bo)      (the context carries v's shift; split context: K' = 3d against [Wo_hi | Wo_hi | Wo_lo])
        g.A = ctx; g.W = l.w_o; g.N = dm; g.K = dm; g.ldw = dm; g.lda = dm; g.out = x; g.ldo = dm; g.bias = l.b_o; g.resid = x;
        if (p_ctx) { g.W = l.w_o3; g.K = 3 * dm; g.ldw = 3 * dm; g.lda = 3 * dm; g.k_algo = dm; }
        g.in_mul = pow2f(k_qkv); g.out_mul = g.out_mul2 = 1.0f; g.range_amax = nullptr;
        gemm(c, dt, EPI_BIAS_RESID, SGPT_F32, g, s);
        g.k_algo = 0;
        }
        // GPT-Neo: x += MLP(LN2(x));  GPT-J (parallel block, HF:gptj:400-411): x += MLP(LN1(x_old)), `a` still holds it
        if (mlp8) {
            // fp8 MFMA: a8 = e4m3(LN(x) / sa[row]);  h8 = e4m3(gelu(a8 . W1_8^T * sa * s1 + b1) / s_h);  x += h8 . W2_8^T * s_h * s2 + b2
            if (!gptj) launch_layernorm_q8(x, l.ln2_g, l.ln2_b, base + o_a8, (float*)(base + o_sa), nullptr, dt, T, dm, m->d.ln_eps, s);
            q.A = base + o_a8; q.lda = dm; q.a_scale = (const float*)(base + o_sa); q.a_scalar = 1.0f;
            q.W = w_fc8; q.ldw = dm; q.w_scale = l.s_fc; q.N = ffn; q.K = dm; q.bias = l.b_fc;
            q.out = h; q.ldo = ffn; q.out_scale = m->act_scale[li];
            { Prof pr(c, s, 2.0 * T * (double)ffn * dm); launch_gemm_fp8(EPI_BIAS_GELU, 0, q, s); }
            q.A = h; q.lda = ffn; q.a_scale = nullptr; q.a_scalar = m->act_scale[li];
            q.W = w_proj8; q.ldw = ffn; q.w_scale = l.s_proj; q.N = dm; q.K = ffn; q.bias = l.b_proj;
            q.resid = x; q.out = x; q.ldo = dm;
            { Prof pr(c, s, 2.0 * T * (double)ffn * dm); launch_gemm_fp8(EPI_BIAS_RESID, 0, q, s); }
        } else {
            // LayerNorm-2 (GPT-J: ln_1's output again, `a` still holds it with its row stride)
            long lda2 = gptj ? (p_ln1 ? 3 * dm : dm) : (p_ln2 ? 3 * dm : dm);
            if (!gptj) {
                if (p_ln2) launch_layernorm_split(x, l.ln2_g, l.ln2_b, a, dt, T, dm, m->d.ln_eps, s, pow2f(-k_ln2));
                else launch_layernorm(x, l.ln2_g, l.ln2_b, a, dt, T, dm, m->d.ln_eps, s, pow2f(-k_ln2));
                if (crest && bf) launch_crest16(a, T, dm, lda2, dt, crest + RS_LN2, s);
            }
            const long ldh = p_h ? 3 * ffn : ffn;
            g.A = a; g.lda = lda2;
            g.W = l.w_fc; g.N = ffn; g.K = dm; g.ldw = dm; g.out = h; g.ldo = ldh; g.bias = l.b_fc; g.resid = nullptr;
            if (p_ln2) { g.W = l.w_fc3; g.K = 3 * dm; g.ldw = 3 * dm; g.k_algo = dm; }
            if (p_h) { g.lo_delta = ffn; g.hi2_delta = 2 * ffn; }       // the GELU output as a [hi | lo | hi] row for fc2
            g.in_mul = pow2f(k_ln2); g.out_mul = pow2f(-k_h); g.range_amax = f16m ? slots + RS_H : nullptr;
            gemm(c, dt, EPI_BIAS_GELU, dt, g, s);
            g.lo_delta = g.hi2_delta = 0; g.k_algo = 0;
            if (crest && bf) launch_crest16(h, T, ffn, ldh, dt, crest + RS_H, s);
            if (m->calibrating) launch_absmax16(h, (long)T * ffn, dt, m->h_amax + li, s);   // FP8M calibration: range of this block's GELU output
            g.A = h; g.lda = ldh; g.W = l.w_proj; g.N = dm; g.K = ffn; g.ldw = ffn; g.out = x; g.ldo = dm;
            if (p_h) { g.W = l.w_proj3; g.K = 3 * ffn; g.ldw = 3 * ffn; g.k_algo = ffn; }
            g.bias = l.b_proj; g.resid = x;
            g.in_mul = pow2f(k_h); g.out_mul = 1.0f; g.range_amax = nullptr;
            gemm(c, dt, EPI_BIAS_RESID, SGPT_F32, g, s);
            g.k_algo = 0;
        }
    }
    if (hidden_out) {
        if (apply_final_ln) launch_layernorm(x, m->lnf_g, m->lnf_b, hidden_out, SGPT_F32, T, dm, m->d.ln_eps, s);
        else HIPC(c, hipMemcpyAsync(hidden_out, x, (size_t)T * dm * 4, hipMemcpyDeviceToDevice, s));
    }
    if (out)
        launch_lnf_pool(x, m->lnf_g, m->lnf_b, seq_off, seq_len, pad_left, B, dm, m->d.ln_eps, apply_final_ln,
                        pool_mode, normalize, m->pool_w, m->pool_w_n, out, s,
                        m->d.compute_dtype == SGPT_F16 ? (int*)m->range_dev : nullptr);
    if (layer_out)
        launch_lnf_pool(x, m->lnf_g, m->lnf_b, seq_off, seq_len, pad_left, B, dm, m->d.ln_eps, apply_final_ln,
                        pool_mode, normalize, m->pool_w, m->pool_w_n, layer_out + (size_t)n_layers_run * B * dm, s);
    if (layer_mean) launch_mean_over_axis0(layer_out, n_layers_run + 1, (long)B * dm, layer_mean, s);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

extern "C" {

sgpt_status sgpt_encode(sgpt_model* m, const int32_t* ids, const int32_t* pos, const int32_t* seq_off,
                        const int32_t* seq_len, const int32_t* pad_left, int32_t B, int32_t T, int32_t max_alloc,
                        int32_t pool_mode, int32_t n_layers_run, int32_t apply_final_ln, int32_t normalize,
                        float* out, float* hidden_out, void* stream) {
    return encode_impl(m, ids, pos, seq_off, seq_len, pad_left, B, T, max_alloc, pool_mode, n_layers_run,
                       apply_final_ln, normalize, out, hidden_out, nullptr, nullptr, stream);
}

sgpt_status sgpt_encode_layers(sgpt_model* m, const int32_t* ids, const int32_t* pos, const int32_t* seq_off,
                               const int32_t* seq_len, const int32_t* pad_left, int32_t B, int32_t T, int32_t max_alloc,
                               int32_t pool_mode, int32_t normalize, float* out_layers, float* out_mean, void* stream) {
    if (!m) return SGPT_ERR_INVALID;
    if (!out_layers && !out_mean) return fail(m->ctx, SGPT_ERR_INVALID, "sgpt_encode_layers: no output requested");
    return encode_impl(m, ids, pos, seq_off, seq_len, pad_left, B, T, max_alloc, pool_mode, m->d.n_layers, 1, normalize,
                       nullptr, nullptr, out_layers, out_mean, stream);
}

sgpt_status sgpt_model_set_pool_weights(sgpt_model* m, const float* w, int32_t n) {
    if (!m) return SGPT_ERR_INVALID;
    sgpt_ctx* c = m->ctx;
    if (!w || n <= 0) return fail(c, SGPT_ERR_INVALID, "sgpt_model_set_pool_weights: bad arguments");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipDeviceSynchronize());      // an in-flight encode may still read the previous table
    if (n > m->pool_w_n) {
        float* p = nullptr;
        if (hipMalloc((void**)&p, (size_t)n * 4) != hipSuccess) return fail(c, SGPT_ERR_OOM, "hipMalloc pool weights failed");
        m->allocs.push_back(p);           // the old, smaller table is released with the model
        m->pool_w = p;
        c->generation++;                  // captured graphs still point at the old table
    }
    m->pool_w_n = n;
    HIPC(c, hipMemcpy(m->pool_w, w, (size_t)n * 4, hipMemcpyDeviceToDevice));
    return SGPT_OK;
}

sgpt_status sgpt_model_calibrate_begin(sgpt_model* m) {
    if (!m) return SGPT_ERR_INVALID;
    sgpt_ctx* c = m->ctx;
    if (m->d.compute_dtype != SGPT_FP8M) return fail(c, SGPT_ERR_INVALID, "calibration applies to SGPT_FP8M models");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipDeviceSynchronize());
    HIPC(c, hipMemset(m->h_amax, 0, (size_t)m->d.n_layers * 8));
    m->calibrating = true;
    return SGPT_OK;
}

sgpt_status sgpt_model_calibrate_end(sgpt_model* m, float margin, float* scales_out) {
    if (!m) return SGPT_ERR_INVALID;
    sgpt_ctx* c = m->ctx;
    if (m->d.compute_dtype != SGPT_FP8M || !m->calibrating) return fail(c, SGPT_ERR_INVALID, "sgpt_model_calibrate_end without _begin");
    m->calibrating = false;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipDeviceSynchronize());
    const int ns = 2 * m->d.n_layers;
    std::vector<float> amax(ns);
    HIPC(c, hipMemcpy(amax.data(), m->h_amax, (size_t)ns * 4, hipMemcpyDeviceToHost));
    if (!(margin >= 1.0f)) margin = 2.0f;
    for (int i = 0; i < ns; ++i) {
        // smallest power of two with margin * amax / scale <= 448 (head-room: later batches may exceed the sample's range;
        // a saturated code raises bit 1 of the range flag)
        const float need = amax[i] * margin / 448.0f;
        float sc = 1.0f;
        if (need > 0.f && std::isfinite(need)) sc = std::exp2(std::ceil(std::log2(need)));
        m->act_scale[i] = sc;
        if (scales_out) scales_out[i] = sc;
    }
    return SGPT_OK;
}

sgpt_status sgpt_model_set_act_scales(sgpt_model* m, const float* scales, int32_t n) {
    if (!m) return SGPT_ERR_INVALID;
    sgpt_ctx* c = m->ctx;
    if (m->d.compute_dtype != SGPT_FP8M || !scales || n != 2 * m->d.n_layers) return fail(c, SGPT_ERR_INVALID, "sgpt_model_set_act_scales: needs 2 * n_layers scales");
    for (int i = 0; i < n; ++i) {
        int e = 0;
        if (!(scales[i] > 0.f) || std::frexp(scales[i], &e) != 0.5f) return fail(c, SGPT_ERR_INVALID, "activation scales must be powers of two");
        m->act_scale[i] = scales[i];
    }
    return SGPT_OK;
}

sgpt_status sgpt_lm_logprobs(sgpt_model* m, const float* hidden, const int32_t* row_idx, const int32_t* targets,
                             int32_t n, float* out_logprob, int32_t* out_greedy, void* stream) {
    if (!m) return SGPT_ERR_INVALID;
    sgpt_ctx* c = m->ctx;
    if (!hidden || !row_idx || !targets || !out_logprob || n <= 0) return fail(c, SGPT_ERR_INVALID, "sgpt_lm_logprobs: bad arguments");
    HIPC(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const int dm = m->d.d_model, V = m->d.vocab;
    const long ldv = (V + 3) / 4 * 4;
    const int R = n < 1024 ? n : 1024;                       // rows per chunk: logits chunk [R, V] fp32 (206 MB at V = 50 257)
    const size_t rows_bytes = align_up((size_t)R * dm * 4, 256), lg_bytes = align_up((size_t)R * ldv * 4, 256);
    sgpt_status st = ensure(c, &c->ws2, &c->ws2_bytes, rows_bytes + lg_bytes);
    if (st != SGPT_OK) return st;
    float* rows = (float*)c->ws2;
    float* logits = (float*)((char*)c->ws2 + rows_bytes);
    for (int r0 = 0; r0 < n; r0 += R) {
        const int nr = (n - r0) < R ? (n - r0) : R;
        launch_gather_rows(hidden, row_idx + r0, nr, dm, rows, s);
        GemmArgs g{};
        g.A = rows; g.lda = dm; g.W = m->lm_w; g.ldw = dm; g.M = nr; g.m_valid = nr; g.N = V; g.K = dm;
        g.out = logits; g.ldo = ldv; g.bias = m->lm_b;
        gemm(c, SGPT_F32, EPI_STORE, SGPT_F32, g, s);          // exact fp32 MFMA: scores are sums of ~30 log-probabilities
        launch_logprob_rows(logits, ldv, V, targets + r0, nr, out_logprob + r0, out_greedy ? out_greedy + r0 : nullptr, s);
    }
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_pool(sgpt_ctx* c, const void* hidden, int32_t dtype, const int32_t* mask, int32_t B, int32_t S,
                      int32_t d, int32_t mode, float* out, void* stream) {
    if (!c || !hidden || !mask || !out || B <= 0 || S <= 0 || d <= 0 || d % 4 || mode < 0 || mode > 2)
        return fail(c, SGPT_ERR_INVALID, "sgpt_pool: bad arguments (d % 4 == 0 required)");
    HIPC(c, hipSetDevice(c->device));
    launch_pool(hidden, dtype, mask, B, S, d, mode, nullptr, out, (hipStream_t)stream);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_pool_learnt(sgpt_ctx* c, const void* hidden, int32_t dtype, const int32_t* mask, int32_t B, int32_t S,
                             int32_t d, const float* pos_weights, float* out, void* stream) {
    if (!c || !hidden || !mask || !out || !pos_weights || B <= 0 || S <= 0 || d <= 0 || d % 4)
        return fail(c, SGPT_ERR_INVALID, "sgpt_pool_learnt: bad arguments (d % 4 == 0 required)");
    HIPC(c, hipSetDevice(c->device));
    launch_pool(hidden, dtype, mask, B, S, d, SGPT_POOL_LEARNTMEAN, pos_weights, out, (hipStream_t)stream);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_fp8_quantize_rows(sgpt_ctx* c, const float* w, int64_t rows, int64_t cols, uint8_t* codes, float* scale,
                                   void* stream) {
    if (!c || !w || !codes || !scale || rows <= 0 || cols <= 0 || cols % 4)
        return fail(c, SGPT_ERR_INVALID, "sgpt_fp8_quantize_rows: bad arguments (cols % 4 == 0 required)");
    HIPC(c, hipSetDevice(c->device));
    launch_fp8_quant_rows(w, rows, cols, codes, scale, (hipStream_t)stream);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_fp8_dequantize_rows(sgpt_ctx* c, const uint8_t* codes, const float* scale, int64_t rows, int64_t cols,
                                     void* out, int32_t out_dtype, void* stream) {
    if (!c || !codes || !scale || !out || rows <= 0 || cols <= 0 || cols % 4 || (out_dtype != SGPT_F32 && out_dtype != SGPT_BF16))
        return fail(c, SGPT_ERR_INVALID, "sgpt_fp8_dequantize_rows: bad arguments (cols % 4 == 0 required)");
    HIPC(c, hipSetDevice(c->device));
    launch_fp8_dequant_rows(codes, scale, rows, cols, out, out_dtype, (hipStream_t)stream);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_l2_normalize(sgpt_ctx* c, const float* in, int64_t n, int32_t d, void* out, int32_t out_dtype,
                              void* stream) {
    if (!c || !in || !out || n <= 0 || d <= 0) return fail(c, SGPT_ERR_INVALID, "sgpt_l2_normalize: bad arguments");
    HIPC(c, hipSetDevice(c->device));
    launch_l2norm(in, n, d, out, out_dtype, (hipStream_t)stream);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_pairwise_scores(sgpt_ctx* c, const float* a, const float* b, int64_t n, int32_t d, int32_t cosine, float* out,
                                 void* stream) {
    if (!c || !a || !b || !out || n <= 0 || d <= 0) return fail(c, SGPT_ERR_INVALID, "sgpt_pairwise_scores: bad arguments");
    HIPC(c, hipSetDevice(c->device));
    launch_pairwise(a, b, n, d, cosine != 0, out, (hipStream_t)stream);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_f32_to_16(sgpt_ctx* c, const float* in, int64_t numel, void* out, int32_t out_dtype, void* stream) {
    if (!c || !in || !out || numel <= 0 || (out_dtype != SGPT_BF16 && out_dtype != SGPT_F16))
        return fail(c, SGPT_ERR_INVALID, "sgpt_f32_to_16: bad arguments");
    HIPC(c, hipSetDevice(c->device));
    launch_f32_to_16(in, numel, out, out_dtype, (hipStream_t)stream);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_f32_to_bf16(sgpt_ctx* c, const float* in, int64_t numel, void* out, void* stream) {
    return sgpt_f32_to_16(c, in, numel, out, SGPT_BF16, stream);
}

static sgpt_status check_score_dims(sgpt_ctx* c, int dtype, int d) {
    if (dtype != SGPT_F32 && dtype != SGPT_BF16 && dtype != SGPT_F16) return fail(c, SGPT_ERR_INVALID, "score: bad dtype");
    const int mult = dtype == SGPT_F32 ? 4 : 8;
    if (d <= 0 || d % mult) return fail(c, SGPT_ERR_INVALID, "score: d must be a multiple of 4 (fp32) / 8 (bf16, f16)");
    return SGPT_OK;
}

sgpt_status sgpt_scores(sgpt_ctx* c, const void* a, const void* b, int32_t dtype, int64_t na, int64_t nb, int32_t d,
                        float* out, int64_t ldo, void* stream) {
    if (!c || !a || !b || !out || na <= 0 || nb <= 0 || ldo < nb || ldo % 4 || na > INT32_MAX || nb > INT32_MAX)
        return fail(c, SGPT_ERR_INVALID, "sgpt_scores: bad arguments");
    sgpt_status st = check_score_dims(c, dtype, d);
    if (st != SGPT_OK) return st;
    HIPC(c, hipSetDevice(c->device));
    GemmArgs g{};
    g.A = a; g.lda = d; g.W = b; g.ldw = d; g.M = (int)na; g.m_valid = (int)na; g.N = (int)nb; g.K = d;
    g.out = out; g.ldo = ldo;
    gemm(c, dtype, EPI_STORE, SGPT_F32, g, (hipStream_t)stream);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

// in_val / in_idx: the running list going in (n_run entries per row, row stride k); run_val / run_idx: the list coming out (they
// may be the same buffers -- the public entry point).  pred_all (device flag or null): every launch of the call is predicated on
// it and the call takes the materialise-and-select path -- the sync-free fallback of sgpt_score_topk_refined.
static sgpt_status score_topk_impl(sgpt_ctx* c, const void* q, const void* corpus, int32_t dtype, int32_t nq, int64_t N,
                                   int32_t d, int32_t k, int64_t idx_base, const float* in_val, const int64_t* in_idx,
                                   float* run_val, int64_t* run_idx, int32_t n_run, int32_t* n_out, void* stream, const int* pred_all) {
    if (!c || !q || !corpus || !run_val || !run_idx || nq <= 0 || N <= 0 || k <= 0 || n_run < 0 || n_run > k)
        return fail(c, SGPT_ERR_INVALID, "sgpt_score_topk: bad arguments");
    sgpt_status st = check_score_dims(c, dtype, d);
    if (st != SGPT_OK) return st;
    HIPC(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    // chunk the corpus so the fp32 score tile [nq, chunk] stays resident in the 256 MiB Infinity Cache, and so
    // that the score GEMM has a whole number of 256-CU waves of 256x256 tiles (query tiles x document tiles)
    const size_t budget = (size_t)160 << 20;
    const int mtq = (nq + 255) / 256;                 // (short query batches, nq <= 64: one 64-row tile, see below)
    long unit = 256;                                  // documents per chunk granule
    { int g = mtq, h = 256; while (h) { int r = g % h; g = h; h = r; } unit = 256L * (256 / g); }
    long chunk = (long)(budget / ((size_t)nq * 4));
    if (chunk >= unit) chunk = chunk / unit * unit;
    else { chunk = chunk / 256 * 256; if (chunk < 256) chunk = 256; }
    if (chunk > 131072) chunk = 131072;
    if (chunk > N) chunk = (long)align_up((size_t)N, 4);
    // bf16 fast path: the 256x256 LDS-DMA GEMM needs the query rows padded to a multiple of 256 (zero rows,
    // never stored) and d % 64 == 0; chunks that are multiples of 256 documents take it, the ragged tail and
    // fp32 go through the 128^2 kernel.
    const bool fast = dtype != SGPT_F32 && d % 64 == 0 && d >= 128;
    // nq <= 64: the 64-query-row scorer tile (score64_kernel) instead of 256 padded rows -- the pass is HBM-bound there
    static const bool no_small_q = exp_env("SGPT_SCORE_NO64") != nullptr;      // A/B switch (experiment build only)
    const int nq_pad = (fast && nq <= 64 && !no_small_q) ? 64 : (nq + 255) / 256 * 256;
    // Threshold-filtered chunks (after the first): see EPI_SCORE_FILTER.  Candidate capacity per query and chunk;
    // the doubling schedule below keeps the expected count at ~k.
    static const bool classic_only = exp_env("SGPT_SCORE_CLASSIC") != nullptr;
    // Capacity and growth: a filtered chunk of len = growth * seen documents expects ~k * growth survivors per query
    // (the threshold is the k-th best of `seen` documents); the merge sorts the candidates in 2048 LDS slots.
    // k <= 64 ("sampled" schedule, round 4): the first thresholds come from a STRIDED SAMPLE of the whole shard instead of its
    // first documents, so (a) nothing is materialised but the sample's own score tile, (b) the expected number of survivors of
    // a chunk, k * len / S, holds whatever the document order does to the score distribution (the reference sorts the corpus
    // by length: a drift along the index used to overflow the lists and pay the materialised recomputation of the chunk), and
    // (c) chunks can therefore be as long as the lists allow instead of doubling: a 125 k-document shard is ONE filtered
    // launch, 1 M documents three (1 + 6 before).  The rank of a sample's k-th best in the population is Gamma(k)-distributed
    // (relative spread 1 / sqrt(k)): lists of `cap` entries take R = cap / (k (1 + 6 / sqrt(k))) times the documents the
    // thresholds were drawn from with ~6 sigma of head-room.
    const bool sampled_k = k <= 64;
    // list capacity of the sampled schedule: the sample is N / ratio documents and ratio grows with cap, so long shards take
    // long lists (1 M documents, k = 11: 2048 entries -> a 15 k-document sample instead of 60 k); short shards keep short
    // ones, because the expected survivors per row and 256-document tile, cap / (2.8 N / 256), is what the filtered GEMM's
    // append path pays for (0.37 at 125 k documents with 512 entries: 236 us per launch against ~190 without appends)
    // nq <= 64 (the HBM-bound 64-row tile): same-box A/B of 512 / 1024 / 2048 entries at 1 M documents (profiles/r04_cap64_ab.txt):
    // nq = 16 0.39 / 0.36 / 0.365 ms, nq = 64 0.41 / 0.375 / 0.39 ms -- the sample is 6 % / 3 % / 1.5 % of the corpus stream, and
    // past 1024 entries the appends cost the streaming tile more than the smaller sample returns
    // (re-measured with the LDS-staged appends, nq = 1000: 125 k documents 0.28 / 0.36 / 0.40 ms with 512 / 1024 / 2048 entries,
    //  250 k documents 0.51 / 0.50 / 0.58 -- the short lists stay)
#ifndef SGPT_SAMPLE_MULT
#define SGPT_SAMPLE_MULT 1.0    // the sample as a multiple of the size the list capacity asks for (A/B builds: the sample is cheap since
#endif                          // its scores stay in the GEMM -- more sampled documents, tighter thresholds, fewer appended survivors)
#ifndef SGPT_FOLD_TAIL_SCORE
#define SGPT_FOLD_TAIL_SCORE 1  // 0: A/B builds -- the materialise-and-select pieces keep their small-tile tail launch
#endif
#ifndef SGPT_FOLD_TAIL
#define SGPT_FOLD_TAIL 1        // 0: the trailing < 256 documents of a shard in a small-tile launch of their own (A/B builds)
#endif
#ifndef SGPT_SAMPLE_TOP2
#define SGPT_SAMPLE_TOP2 1      // 0: the materialised sample tile + select of round 4 (A/B builds)
#endif
#ifndef SGPT_CAP_SHORT
#define SGPT_CAP_SHORT 512
#endif
    const int cap_n = nq <= 64 ? (N >= 400000 ? 1024 : 512) : (N >= 800000 ? 2048 : (N >= 400000 ? 1024 : SGPT_CAP_SHORT));
    const int cap_s = cap_n * (k <= 32 ? 1 : 2) > 2048 ? 2048 : cap_n * (k <= 32 ? 1 : 2);
    const int cap = sampled_k ? cap_s : ((4 * k < 2048 - k) ? 4 * k : 2048 - k);
    const bool half_growth = 2 * cap < 5 * k;          // k > ~340: cap < 2.5 k -> grow by half, expect ~k/2 per chunk
    // k > 64 keeps the first-chunk + doubling schedule.  Two other schedules were measured against doubling and rejected:
    //   round 2: 4x growth with 256-entry lists -- overflowed on corpora whose score distribution drifts along the index and
    //            paid the materialised recomputation;
    //   round 3: 2048-entry lists with growth cap / 6k (k = 11: first chunk + two filtered chunks per 1 M documents) -- ~330
    //            survivors per query and chunk through the epilogue's append path: the filtered GEMM went from 1.45 to 1.58 ms
    //            per pass (nq = 1000), and a drift at 10 % of the corpus from 2.2 to 3.5 ms (profiles/r03_score_schedule.txt).
    const int growth = 1;
    const bool filt = fast && !classic_only && k <= 1024 && chunk % 256 == 0 && N >= 2 * chunk && pred_all == nullptr;
    const bool sampled = filt && sampled_k;
    const double ratio = (double)cap / ((double)k * (1.0 + 6.0 / std::sqrt((double)k)));   // documents per threshold document
    const long n256_all = N / 256 * 256;
    long S = 0, s_stride = 1;                           // sample size (documents) and row stride
    if (sampled) {
        // S = N / ratio: the survivors of the WHOLE shard fit the lists even if no chunk ever raises the thresholds (a corpus
        // whose best documents all come last).  Bounded by a 512 MiB score tile (131 072 documents at nq = 1000; 1 M documents
        // need 60 k): behind that the schedule relies on the merges raising the thresholds, like the doubling one did.
        S = ((long)((double)N / ratio * SGPT_SAMPLE_MULT) + 255) / 256 * 256;     // rounded UP: ratio * S covers the shard
        if (S < 2048) S = 2048;
        const long s_max = (long)(((size_t)512 << 20) / ((size_t)nq * 4)) / 256 * 256;
        if (S > s_max) S = s_max > 2048 ? s_max : 2048;
        if (S > n256_all / 2) S = n256_all / 2 / 256 * 256;
        s_stride = n256_all / S;
        if (s_stride < 1) s_stride = 1;
        // an ODD stride: a corpus with a power-of-two period (bench.py's 1 M shard is perturbed copies of a 40 960-document
        // block: stride 16 met the SAME source documents in every copy -- a third of the effective sample, and lists that
        // overflowed for a few queries in a thousand) is the structure a sample must not resonate with
        if (s_stride > 1 && s_stride % 2 == 0) s_stride -= 1;
        // equally spaced rows have an integer stride: stretch the sample to the END of the shard at that stride (a tail the
        // sample never visits is invisible to the thresholds -- 33 k of 1 M documents at stride 16: a corpus whose best
        // documents sit at the end overflowed the lists of ~3 queries in 1000)
        S = n256_all / s_stride / 256 * 256;
    }

    // A filtered chunk whose candidate lists overflow is recomputed by materialise + select in pieces whose fp32 score tile
    // stays in the Infinity Cache (measured: 1 GiB tiles made the recomputation 6x slower than the plain materialised
    // loop).  Its launches are predicated and exit at once otherwise -- for short query batches one piece covers a whole
    // chunk (nq = 16: 2.6 M documents per 160 MiB tile), so a pass carries 2 predicated launches per chunk, not 2 per 131 072
    // documents.
    long fchunk = (long)(budget / ((size_t)nq * 4)) / 256 * 256;
    // sampled schedule: an overflow needs an adversarial mass of near-equal scores now, so the fallback is sized for few
    // no-op launches (a predicated 256x256 launch that exits at once still costs ~4.4 us, two per piece: ~60 of them were
    // 0.05-0.08 ms of a 1 M-document pass) rather than for the cache: pieces of up to 131 072 documents
    // -- but never beyond a byte budget: the tile is allocated whether or not a fallback ever runs (nq = 10 000 x 131 072 x 4 B
    // would be 5 GB of workspace for launches that exit at once)
    if (sampled && fchunk < 131072) {
        const long by_bytes = (long)(((size_t)512 << 20) / ((size_t)nq * 4)) / 256 * 256;
        const long want = by_bytes < 131072 ? by_bytes : 131072;
        if (fchunk < want) fchunk = want;
    }
    if (fchunk > (long)align_up((size_t)N, 256)) fchunk = (long)align_up((size_t)N, 256);
    if (fchunk < chunk) fchunk = chunk;
    const long n_flags = 64 + N / (1L << 19) + 1;      // one overflow flag per filtered chunk
    const size_t sc_bytes = align_up((size_t)nq * (fchunk > S ? fchunk : S) * 4, 256);
    const size_t tv_bytes = align_up((size_t)nq * k * 4, 256), ti_bytes = align_up((size_t)nq * k * 8, 256);
    const size_t qp_bytes = fast ? align_up((size_t)nq_pad * d * 2, 256) : 0;
    const size_t cv_bytes = filt ? align_up((size_t)nq * cap * 4, 256) : 0, ci_bytes = filt ? align_up((size_t)nq * cap * 8, 256) : 0;
    const size_t cc_bytes = filt ? align_up((size_t)(nq_pad + n_flags) * 4, 256) : 0;  // counters + per-chunk overflow flags
    const size_t th_bytes = sampled ? tv_bytes + ti_bytes + align_up((size_t)nq * 4, 256) : 0;   // sample's list + dense thresholds
    st = ensure(c, &c->ws2, &c->ws2_bytes,
                sc_bytes + 3 * (tv_bytes + ti_bytes) + qp_bytes + cv_bytes + ci_bytes + cc_bytes + th_bytes);
    if (st != SGPT_OK) return st;
    char* base = (char*)c->ws2;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p_ = base + off; off += bytes; return p_; };
    float* sc = (float*)take(sc_bytes);
    float* tv[2]; int64_t* ti[2];
    tv[0] = (float*)take(tv_bytes); tv[1] = (float*)take(tv_bytes);
    ti[0] = (int64_t*)take(ti_bytes); ti[1] = (int64_t*)take(ti_bytes);
    float* sav_v = (float*)take(tv_bytes); int64_t* sav_i = (int64_t*)take(ti_bytes);   // third list: ping-pong partner of a recomputation
    void* qpad = fast ? take(qp_bytes) : nullptr;
    float* cand_v = filt ? (float*)take(cv_bytes) : nullptr;
    long long* cand_i = filt ? (long long*)take(ci_bytes) : nullptr;
    int* cand_cnt = filt ? (int*)take(cc_bytes) : nullptr;
    int* flag = filt ? cand_cnt + nq_pad : nullptr;
    float* th_v = sampled ? (float*)take(tv_bytes) : nullptr;
    int64_t* th_i = sampled ? (int64_t*)take(ti_bytes) : nullptr;
    float* thr_dense = sampled ? (float*)take(align_up((size_t)nq * 4, 256)) : nullptr;
    // prologue: one launch (query rows into their zero-padded tile, counters and flags cleared, an empty running list marked)
    const bool fresh_list = sampled && n_run == 0;
    if (fast && d % 8 == 0 && ((size_t)q & 15) == 0) {
        launch_score_prep(q, qpad, (long)nq * d * 2, (long)nq_pad * d * 2, filt ? cand_cnt : nullptr, filt ? (long)(nq_pad + n_flags) : 0,
                          fresh_list ? (long long*)ti[0] : nullptr, fresh_list ? (long)nq * k : 0, s);
    } else {
        if (fast) {
            if (nq_pad > nq) HIPC(c, hipMemsetAsync((char*)qpad + (size_t)nq * d * 2, 0, (size_t)(nq_pad - nq) * d * 2, s));   // the pad rows only
            HIPC(c, hipMemcpyAsync(qpad, q, (size_t)nq * d * 2, hipMemcpyDeviceToDevice, s));
        }
        if (filt) HIPC(c, hipMemsetAsync(cand_cnt, 0, (size_t)(nq_pad + n_flags) * 4, s));
        if (fresh_list) HIPC(c, hipMemsetAsync(ti[0], 0xff, (size_t)nq * k * 8, s));
    }
    const size_t esz = dtype == SGPT_F32 ? 4 : 2;

    // fp32 scores of documents [c0, c0 + nc) for every query -> sc[nq][ld]
    auto score_tile = [&](long c0, long nc, long ld, const int* pred, long row_stride = 1) {
        GemmArgs g{};
        g.A = q; g.lda = d; g.M = nq; g.m_valid = nq; g.K = d; g.pred = pred;
        g.W = (const char*)corpus + (size_t)c0 * d * esz; g.ldw = (long)d * row_stride; g.N = (int)nc;   // row_stride > 1: a strided sample
        g.out = sc; g.ldo = ld;
        const long na = fast ? nc / 256 * 256 : 0;      // documents the 256-document tile kernels take
        // (round 5, measured and reverted: handing a predicated fallback piece to the register-staged kernel whole -- one no-op launch
        //  instead of two -- made the shard pass 12 % SLOWER: a no-op launch of the persistent 256x256 kernel is 256 workgroups that
        //  exit, one of the small-tile kernel for 125 k documents x 1000 queries is ~8 000.)
        // the < 256 trailing documents ride in the 256x256 launch (clamped rows, GemmArgs.n_valid; their missing columns land in the
        // padding of the score row, which the select does not read): one launch less per piece -- a no-op one in the predicated
        // fallback of every filtered chunk
        const bool fold = SGPT_FOLD_TAIL && SGPT_FOLD_TAIL_SCORE && na > 0 && na < nc && row_stride == 1 && na + 256 <= ld &&
                          gemm_score_tail_foldable(nq_pad, na + 256, d);
        if (na > 0) {
            GemmArgs h = g;
            h.A = qpad; h.M = nq_pad; h.N = (int)(fold ? na + 256 : na); h.n_valid = fold ? (int)nc : 0;
            gemm(c, dtype, EPI_SCORE, SGPT_F32, h, s);
        }
        if (na < nc && !fold) {                         // fp32, or the ragged tail (< 256 documents): register-staged kernel
            g.W = (const char*)corpus + (size_t)(c0 + na) * d * esz; g.N = (int)(nc - na); g.out = sc + na;
            gemm(c, dtype, EPI_SCORE, SGPT_F32, g, s);
        }
    };
    // Materialise-and-select over documents [lo, hi): the reference's chunk loop (exact_search.py:96-132).
    // (pv, pi, have) = running best going in; the last chunk writes (fin_v, fin_i), earlier ones ping-pong.
    auto classic = [&](long lo, long hi, const float* pv, const int64_t* pi, int have, float* fin_v, int64_t* fin_i,
                       const int* pred, long chunk) -> sgpt_status {
        int cur = (pv == tv[0]) ? 1 : 0;
        for (long c0 = lo; c0 < hi; c0 += chunk) {
            const long nc = (hi - c0) < chunk ? (hi - c0) : chunk;
            score_tile(c0, nc, chunk, pred);
            const bool last = c0 + nc >= hi;
            if (last && pv == fin_v) {  // in/out alias on a single-chunk call: stage through the ping-pong buffer
                launch_topk_select(sc, chunk, nc, idx_base + c0, pv, pi, have, k, nq, k, 0, nullptr, tv[cur], ti[cur], s, pred);
                HIPC(c, hipMemcpyAsync(fin_v, tv[cur], (size_t)nq * k * 4, hipMemcpyDeviceToDevice, s));
                HIPC(c, hipMemcpyAsync(fin_i, ti[cur], (size_t)nq * k * 8, hipMemcpyDeviceToDevice, s));
            } else {
                launch_topk_select(sc, chunk, nc, idx_base + c0, pv, pi, have, k, nq, k, 0, nullptr,
                                   last ? fin_v : tv[cur], last ? fin_i : ti[cur], s, pred);
            }
            pv = tv[cur]; pi = ti[cur];
            have = k;  // unused tail slots carry idx = -1 and are ignored by the next merge
            cur ^= 1;
        }
        return SGPT_OK;
    };

    // The same over [lo, hi) with every launch predicated on *pred (a filtered chunk's overflow flag): (pv, pi) = the
    // k-slot running best before the chunk, result in (fin_v, fin_i); pieces of fchunk documents ping-pong through the
    // third list so that nothing is copied (a copy could not be predicated).
    auto recompute = [&](long lo, long hi, const float* pv, const int64_t* pi, float* fin_v, int64_t* fin_i, const int* pred) {
        const long P = (hi - lo + fchunk - 1) / fchunk;
        for (long pc = 0; pc < P; ++pc) {
            const long c0 = lo + pc * fchunk, nc = (hi - c0) < fchunk ? (hi - c0) : fchunk;
            const bool to_fin = (P - 1 - pc) % 2 == 0;
            float* dv = to_fin ? fin_v : sav_v;
            int64_t* di = to_fin ? fin_i : sav_i;
            score_tile(c0, nc, fchunk, pred);
            launch_topk_select(sc, fchunk, nc, idx_base + c0, pv, pi, k, k, nq, k, 0, nullptr, dv, di, s, pred);
            pv = dv; pi = di;
        }
    };

    const float* pv0 = n_run > 0 ? in_val : nullptr;
    const int64_t* pi0 = n_run > 0 ? in_idx : nullptr;
    if (!filt) {
        st = classic(0, N, pv0, pi0, n_run, run_val, run_idx, pred_all, chunk);
        if (st != SGPT_OK) return st;
    } else {
        static const bool no_fallback = exp_env("SGPT_SCORE_NOFALLBACK") != nullptr;   // timing experiments ONLY: unsafe
        int cur = 0, chunk_i = 0;
        long seen = 0;                 // documents behind which the filtered chunks continue
        long eff = 0;                  // documents the current thresholds are the k-th best of
        const long n256 = n256_all;
        if (sampled) {
            // running best going in: tv[0] / ti[0] = the caller's list padded to k columns, or empty (idx -1 everywhere)
            if (n_run > 0) {
                launch_topk_select(in_val, k, 0, 0, in_val, in_idx, n_run, k, nq, k, 0, nullptr, tv[0], ti[0], s);
            }   // (n_run == 0: the prologue marked tv[0] / ti[0] empty -- idx -1, whatever the values say)
            // thresholds: the k-th best of {running best} U {S documents at stride s_stride across the shard}; only the VALUES
            // of this selection are used (its indices are sample positions) -- the filtered chunks below re-score the sampled
            // documents like any other and find them again
            if (SGPT_SAMPLE_TOP2 && nq_pad >= 256 && S % 256 == 0) {
                // the sample's scores never leave the GEMM: 8 floats per query row and 256-document tile (the two best of each
                // wave's 64 documents, EPI_SCORE_TOP2); the k-th best of those and the running list is the threshold
                const long ld2 = 8 * (S / 256);
                GemmArgs h{};
                h.A = qpad; h.lda = d; h.M = nq_pad; h.m_valid = nq; h.K = d;
                h.W = corpus; h.ldw = (long)d * s_stride; h.N = (int)S; h.out = sc; h.ldo = ld2;
                gemm(c, dtype, EPI_SCORE_TOP2, SGPT_F32, h, s);
                launch_topk_select(sc, ld2, ld2, 0, tv[0], ti[0], k, k, nq, k, 0, nullptr, th_v, th_i, s, nullptr, thr_dense);
            } else {
                score_tile(0, S, S, nullptr, s_stride);
                launch_topk_select(sc, S, S, 0, tv[0], ti[0], k, k, nq, k, 0, nullptr, th_v, th_i, s, nullptr, thr_dense);   // (+ the inclusive threshold)
            }
            eff = S;
        } else {
            // first chunk: materialise + select -> the initial thresholds.  One whole wave of tiles is enough (the
            // doubling schedule takes over from there): 16 384 documents for nq = 1000 instead of 32 768
            const long first = unit < chunk ? unit : chunk;
            st = classic(0, first, pv0, pi0, n_run, tv[0], ti[0], nullptr, chunk);
            if (st != SGPT_OK) return st;
            seen = first; eff = first;
        }
        // k > 64, doubling schedule: a filtered chunk is as long as everything seen before it, so a query expects ~k
        // survivors per chunk (k * len / seen) whatever N is; 1 M documents = 1 + 5 launches instead of 31.
        // k <= 64, sampled schedule: `ratio` times the documents the thresholds stand for (see above).
        while (n256 - seen >= 256) {
            long len;
            if (sampled) len = (long)((double)eff * ratio) / 256 * 256;
            else len = half_growth ? (eff / 2 / 256 * 256 > 256 ? eff / 2 / 256 * 256 : 256) : eff * growth;
            if (len < 256) len = 256;
            // chunks of at most 2^19 documents: a merge half-way through a 1 M-document pass tightens the thresholds of the
            // second half (nq = 1000: 380 + 10 appended candidates per query instead of 730; measured 1.62-1.64 against
            // 1.65-1.69 ms, nq = 128 0.51 against 0.54).  Short query batches (the HBM-bound 64-row tile) take the whole shard
            // in ONE launch instead: a merge, three no-op fallback launches and a pipeline ramp less (nq = 16: 0.36 -> 0.35 ms)
            if (len > (1L << 19) && !(sampled && nq <= 64)) len = 1L << 19;
            if (len >= n256 - seen) len = n256 - seen;          // the last chunk takes what is left (no whole-wave rounding: that
            else if (len >= unit) len = len / unit * unit;      //  left a 17 k-document sixth launch behind 1 M documents at nq = 16)
            GemmArgs g{};
            g.A = qpad; g.lda = d; g.M = nq_pad; g.m_valid = nq; g.K = d;
            g.W = (const char*)corpus + (size_t)seen * d * esz; g.ldw = d; g.N = (int)len;
            // sampled schedule: the dense thresholds -- the sample's k-th best, raised by every merge to the running k-th best
            if (sampled) { g.thr = thr_dense; g.thr_ld = 1; }
            else { g.thr = tv[cur] + (k - 1); g.thr_ld = k; }
            g.cand_val = cand_v; g.cand_idx = cand_i; g.cand_cnt = cand_cnt; g.cand_cap = cap; g.idx_base = idx_base + seen;
            // the < 256 trailing documents ride in the last chunk's launch (the 256x256 kernel clamps their rows and masks their
            // columns, GemmArgs.n_valid) when documents are its streamed operand; the 64-row tile keeps its own tail launch
            const bool fold_tail = SGPT_FOLD_TAIL && seen + len == n256 && N > n256 && gemm_score_tail_foldable(nq_pad, len + 256, d);
            if (fold_tail) { g.N = (int)(len + 256); g.n_valid = (int)(len + (N - n256)); }
            gemm(c, dtype, EPI_SCORE_FILTER, SGPT_F32, g, s);
            g.n_valid = 0;
            const long c_lo = seen;
            seen += len;
            if (fold_tail) seen = N;
            eff = seen;                                         // the thresholds now stand for every document of [0, seen)
            if (seen == n256 && seen < N) {
                // ragged tail (< 256 documents): filtered against the same thresholds by the small-tile kernel, its
                // survivors join this chunk's candidate lists -- one merge, no materialise + select round for 72 documents
                g.W = (const char*)corpus + (size_t)seen * d * esz; g.N = (int)(N - seen); g.idx_base = idx_base + seen;
                gemm(c, dtype, EPI_SCORE_FILTER, SGPT_F32, g, s);
                seen = N;
            }
            const bool fin = seen >= N;
            int* cflag = flag + (chunk_i < n_flags ? chunk_i : n_flags - 1);
            float* ov = fin ? run_val : tv[cur ^ 1];
            int64_t* oi = fin ? run_idx : ti[cur ^ 1];
            launch_cand_merge(tv[cur], ti[cur], cand_v, (const int64_t*)cand_i, cand_cnt, cap, nq, k, ov, oi, cflag, s,
                              sampled ? thr_dense : nullptr);
            // A candidate list of this chunk overflowed (document order with a drifting score distribution, a mass of
            // equal scores): the merge result is incomplete -- redo THIS chunk from the pre-chunk best by materialise +
            // select.  Sync-free: predicated on the chunk's device flag.  The thresholds of the next chunk come from the
            // corrected list, so a drift costs one or two recomputed chunks, not the whole call.
            // (Deferring the fallback to ONE predicated whole-shard recomputation per pass for short query batches saves 8 no-op
            // launches = 2 % at nq = 16, and costs 12.9 ms instead of 0.55 when a drift does overflow: a select over 1 M columns
            // has 16 rows of parallelism.  Measured and dropped, profiles/r03_score_defer_ab.txt.)
            if (!no_fallback) recompute(c_lo, seen, tv[cur], ti[cur], ov, oi, cflag);
            cur ^= 1;
            ++chunk_i;
        }
        if (seen < N) {   // (only when no filtered chunk ran: N - first < 256) ragged tail: materialise + select
            st = classic(seen, N, tv[cur], ti[cur], k, run_val, run_idx, nullptr, chunk);
            if (st != SGPT_OK) return st;
        }
    }
    if (n_out) { const int64_t tot = (int64_t)n_run + N; *n_out = (int32_t)(tot < k ? tot : k); }
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_score_topk(sgpt_ctx* c, const void* q, const void* corpus, int32_t dtype, int32_t nq, int64_t N,
                            int32_t d, int32_t k, int64_t idx_base, float* run_val, int64_t* run_idx, int32_t n_run,
                            int32_t* n_out, void* stream) {
    return score_topk_impl(c, q, corpus, dtype, nq, N, d, k, idx_base, run_val, run_idx, run_val, run_idx, n_run, n_out, stream, nullptr);
}

// The exact-fp32 top-k at the 16-bit scorer's speed (round 5): what `torch.mm(q, c.T)` + `torch.topk` of the reference compute in
// fp32 (util.py:41-43, exact_search.py:96-108), found in two stages.
//   1. the 16-bit filtered scorer over f16 copies of the rows takes the k' = k + M best per query (M: head-room, k' <= 64 keeps
//      the sampled schedule);
//   2. every candidate is re-scored in exact fp32 from the fp32 rows (rescore_kernel), merged with the running list, top-k.
// Guarantee: rows L2-normalised (or any rows with |q| |d| <= 1): |s16 - s32| <= eps = 2^-10 (two roundings of relative 2^-11,
// Cauchy-Schwarz) + 2^-25 (|q|_1 + |d|_1) (subnormal flushes) + the two fp32 accumulations ~ 1.1e-3.  A document outside the k'
// candidates has s16 <= u = the k'-th best; if u < t - 2 eps (t = the k-th best s16) its fp32 score is below t - eps, which k
// candidates reach or exceed: it cannot be in the fp32 top-k.  `margin` >= 2 eps is the caller's statement of that bound for its
// rows (2.5e-3 for unit rows; scale by max |q| max |d| for raw dot products).  Queries for which the check fails (masses of
// near-equal scores: duplicated documents) raise a device flag, and the chunk is redone by the exact-fp32 materialise-and-select
// pass, every launch predicated on that flag -- sync-free, exact either way; `refined_fallbacks` counts nothing on the host.
sgpt_status sgpt_score_topk_refined(sgpt_ctx* c, const float* q, const float* corpus32, const void* corpus16, int32_t dtype16,
                                    int32_t nq, int64_t N, int32_t d, int32_t k, int64_t idx_base, float margin,
                                    float* run_val, int64_t* run_idx, int32_t n_run, int32_t* n_out, int32_t* fallback_flag_out,
                                    void* stream) {
    if (!c || !q || !corpus32 || !corpus16 || !run_val || !run_idx || nq <= 0 || N <= 0 || k <= 0 || n_run < 0 || n_run > k ||
        !(margin > 0.f) || (dtype16 != SGPT_F16 && dtype16 != SGPT_BF16) || d % 8)
        return fail(c, SGPT_ERR_INVALID, "sgpt_score_topk_refined: bad arguments (16-bit stage-1 rows, d % 8 == 0, margin > 0)");
    HIPC(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    // head-room: as many extra candidates as the sampled schedule allows (k' <= 64), at least 8; deep lists (k > 56) take k / 8
    int kp = k <= 56 ? 64 : k + (k / 8 > 8 ? k / 8 : 8);
    if (kp > 1024 && k <= 1024) kp = 1024;
    if ((int64_t)kp > N) kp = (int)N;
    if (kp <= k || kp > 1024) {       // nothing to filter with (k >= N) or lists beyond the filtered scorer: the exact pass itself
        if (fallback_flag_out) *fallback_flag_out = -1;
        return score_topk_impl(c, q, corpus32, SGPT_F32, nq, N, d, k, idx_base, run_val, run_idx, run_val, run_idx, n_run, n_out, stream, nullptr);
    }
    const int ldc = kp + k;
    const size_t q16_b = align_up((size_t)nq * d * 2, 256), lv_b = align_up((size_t)nq * kp * 4, 256), li_b = align_up((size_t)nq * kp * 8, 256);
    const size_t cv_b = align_up((size_t)nq * ldc * 4, 256), ci_b = align_up((size_t)nq * ldc * 8, 256);
    const size_t sv_b = align_up((size_t)nq * k * 4, 256), si_b = align_up((size_t)nq * k * 8, 256);
    sgpt_status st = ensure(c, &c->ws4, &c->ws4_bytes, q16_b + lv_b + li_b + cv_b + ci_b + sv_b + si_b + 256);
    if (st != SGPT_OK) return st;
    char* b = (char*)c->ws4;
    void* q16 = b; b += q16_b;
    float* lv = (float*)b; b += lv_b;
    int64_t* li = (int64_t*)b; b += li_b;
    float* cv = (float*)b; b += cv_b;
    int64_t* ci = (int64_t*)b; b += ci_b;
    float* sv = (float*)b; b += sv_b;
    int64_t* si = (int64_t*)b; b += si_b;
    int* flag = (int*)b;
    HIPC(c, hipMemsetAsync(flag, 0, 4, s));
    launch_f32_to_16(q, (int64_t)nq * d, q16, dtype16, s);
    // stage 1: the k' best by 16-bit score (fresh list)
    int32_t n1 = 0;
    st = score_topk_impl(c, q16, corpus16, dtype16, nq, N, d, kp, idx_base, lv, li, lv, li, 0, &n1, stream, nullptr);
    if (st != SGPT_OK) return st;
    launch_refine_check(lv, k, kp, margin, nq, flag, s);
    // stage 2: exact fp32 scores of the candidates | the running list -> top-k
    if (n_run > 0) {      // (kept for the fallback: it starts from the list as it was before this chunk)
        HIPC(c, hipMemcpy2DAsync(sv, (size_t)k * 4, run_val, (size_t)k * 4, (size_t)n_run * 4, nq, hipMemcpyDeviceToDevice, s));
        HIPC(c, hipMemcpy2DAsync(si, (size_t)k * 8, run_idx, (size_t)k * 8, (size_t)n_run * 8, nq, hipMemcpyDeviceToDevice, s));
    }
    launch_rescore(q, corpus32, li, kp, nq, kp, d, idx_base, N, cv, ci, ldc, s);
    launch_list_append(run_val, run_idx, k, n_run, nq, cv, ci, ldc, kp, s);
    launch_topk_select(cv, ldc, 0, 0, cv, ci, kp + n_run, ldc, nq, k, 0, nullptr, run_val, run_idx, s);
    // the predicated exact pass (no-op launches unless a query failed the check)
    st = score_topk_impl(c, q, corpus32, SGPT_F32, nq, N, d, k, idx_base, sv, si, run_val, run_idx, n_run, nullptr, stream, flag);
    if (st != SGPT_OK) return st;
    if (fallback_flag_out) {     // optional, host-synchronising: did the exact pass run? (tests / diagnostics; pass NULL to stay asynchronous)
        HIPC(c, hipMemcpyAsync(fallback_flag_out, flag, 4, hipMemcpyDeviceToHost, s));
        HIPC(c, hipStreamSynchronize(s));
    }
    if (n_out) { const int64_t tot = (int64_t)n_run + N; *n_out = (int32_t)(tot < k ? tot : k); }
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_topk_merge(sgpt_ctx* c, const float* val, const int64_t* idx, int32_t nq, int32_t mcand, int32_t k,
                            const int64_t* exclude_idx, float* out_val, int64_t* out_idx, void* stream) {
    if (!c || !val || !idx || !out_val || !out_idx || nq <= 0 || mcand <= 0 || k <= 0)
        return fail(c, SGPT_ERR_INVALID, "sgpt_topk_merge: bad arguments");
    HIPC(c, hipSetDevice(c->device));
    // all candidates ride in the "previous best" leg of the virtual row (n = 0 fresh scores)
    launch_topk_select(val, mcand, 0, 0, val, idx, mcand, mcand, nq, k, 0, exclude_idx, out_val, out_idx,
                       (hipStream_t)stream);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_topk(sgpt_ctx* c, const float* scores, int32_t nq, int64_t n, int64_t ld, int32_t k, int64_t idx_base,
                      float* out_val, int64_t* out_idx, void* stream) {
    if (!c || !scores || !out_val || !out_idx || nq <= 0 || n <= 0 || ld < n || k <= 0)
        return fail(c, SGPT_ERR_INVALID, "sgpt_topk: bad arguments");
    HIPC(c, hipSetDevice(c->device));
    launch_topk_select(scores, ld, n, idx_base, nullptr, nullptr, 0, 0, nq, k, 1, nullptr, out_val, out_idx,
                       (hipStream_t)stream);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}


sgpt_status sgpt_layernorm_fp8(sgpt_ctx* c, const float* x, const float* gamma, const float* beta, int32_t T, int32_t d, float eps,
                               uint8_t* codes, float* row_scale, void* stream) {
    if (!c || !x || !gamma || !beta || !codes || !row_scale || T <= 0 || d <= 0 || d % 4 || d > 4096)
        return fail(c, SGPT_ERR_INVALID, "sgpt_layernorm_fp8: bad arguments (d % 4 == 0, d <= 4096)");
    HIPC(c, hipSetDevice(c->device));
    launch_layernorm_q8(x, gamma, beta, codes, row_scale, nullptr, SGPT_BF16, T, d, eps, (hipStream_t)stream);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_linear_fp8(sgpt_ctx* c, int32_t epi, int32_t out_dtype, const uint8_t* A, const float* a_scale, float a_scalar,
                            const uint8_t* W, const float* w_scale, const float* bias, const float* resid, void* out,
                            float out_scale, int32_t M, int32_t N, int32_t K, void* stream) {
    if (!c || !A || !W || !w_scale || !out || !(a_scalar > 0.f)) return fail(c, SGPT_ERR_INVALID, "sgpt_linear_fp8: bad arguments");
    if (!gemm_fp8_shape_ok(M, N, K)) return fail(c, SGPT_ERR_INVALID, "sgpt_linear_fp8: M, N, K must be multiples of 256");
    if (epi != EPI_BIAS_GELU && epi != EPI_BIAS_RESID && epi != EPI_STORE && epi != EPI_VT)
        return fail(c, SGPT_ERR_INVALID, "sgpt_linear_fp8: epi 0 (store 16-bit), 1 (bias+gelu -> fp8), 2 (bias+residual -> fp32) or 4 (transposed 16-bit)");
    if ((epi == EPI_STORE || epi == EPI_VT) && out_dtype != SGPT_BF16 && out_dtype != SGPT_F16)
        return fail(c, SGPT_ERR_INVALID, "sgpt_linear_fp8: store epilogues write bf16 or f16");
    if ((epi == EPI_BIAS_GELU || epi == EPI_BIAS_RESID) && !bias) return fail(c, SGPT_ERR_INVALID, "sgpt_linear_fp8: bias required");
    if (epi == EPI_BIAS_RESID && !resid) return fail(c, SGPT_ERR_INVALID, "sgpt_linear_fp8: residual required");
    if (epi == EPI_BIAS_GELU && !(out_scale > 0.f)) return fail(c, SGPT_ERR_INVALID, "sgpt_linear_fp8: out_scale required");
    HIPC(c, hipSetDevice(c->device));
    GemmArgs q{};
    q.A = A; q.lda = K; q.W = W; q.ldw = K; q.M = M; q.m_valid = M; q.N = N; q.K = K; q.out = out; q.ldo = epi == EPI_VT ? M : N;
    q.bias = bias; q.resid = resid; q.a_scale = a_scale; q.a_scalar = a_scalar; q.w_scale = w_scale; q.out_scale = out_scale;
    q.range_flag = c->range_flag;
    { Prof pr(c, (hipStream_t)stream, 2.0 * M * (double)N * K); launch_gemm_fp8(epi, out_dtype, q, (hipStream_t)stream); }
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

int32_t sgpt_ctx_set_low_latency(sgpt_ctx* c, int32_t on) {
    if (!c) return 0;
    const int old = c->kgroups > 1 ? 1 : 0;
    c->kgroups = on ? 2 : 1;
    return old;
}
int32_t sgpt_ctx_set_gemm_cu_cap(sgpt_ctx* c, int32_t n) {
    if (!c) return 0;
    const int old = c->cu_cap;
    c->cu_cap = n > 0 ? n : 0;
    return old;
}
int32_t sgpt_ctx_set_tile_policy(sgpt_ctx* c, int32_t policy) {
    if (!c) return 0;
    const int old = c->force256 ? 1 : c->no_qpath ? 2 : 0;
    c->force256 = policy == 1 ? 1 : 0;
    c->no_qpath = policy == 2 ? 1 : 0;
    return old;
}
#ifdef SGPT_EXPERIMENTS
int32_t sgpt_exp_set_gemm_skew(int32_t cycles) { return set_gemm_skew(cycles); }
int32_t sgpt_exp_set_gemm_w(int32_t on) { return set_gemm_use_w(on); }
#endif

// ---- SGPT_F16 range guard, per model --------------------------------------------------------------------------------
sgpt_status sgpt_model_range_check(sgpt_model* m, int32_t* flagged, int32_t reset, void* stream) {
    if (!m || !flagged) return SGPT_ERR_INVALID;
    sgpt_ctx* c = m->ctx;
    HIPC(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    unsigned h = 0;
    HIPC(c, hipMemcpyAsync(&h, m->range_dev, 4, hipMemcpyDeviceToHost, s));
    HIPC(c, hipStreamSynchronize(s));
    if (reset && h) HIPC(c, hipMemsetAsync(m->range_dev, 0, (size_t)(1 + m->d.n_layers * RS_N) * 4, s));
    *flagged = (int32_t)h;
    return SGPT_OK;
}

sgpt_status sgpt_model_range_adapt(sgpt_model* m, int32_t* n_raised, void* stream) {
    if (!m || !n_raised) return SGPT_ERR_INVALID;
    sgpt_ctx* c = m->ctx;
    *n_raised = 0;
    if (m->d.compute_dtype != SGPT_F16) return fail(c, SGPT_ERR_INVALID, "sgpt_model_range_adapt applies to SGPT_F16 models");
    HIPC(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const int n = m->d.n_layers * RS_N;
    std::vector<unsigned> h((size_t)n + 1);
    HIPC(c, hipMemcpyAsync(h.data(), m->range_dev, (size_t)(n + 1) * 4, hipMemcpyDeviceToHost, s));
    HIPC(c, hipStreamSynchronize(s));
    int raised = 0;
    for (int i = 0; i < n; ++i) {
        if (!h[i + 1]) continue;
        float amax;
        memcpy(&amax, &h[i + 1], 4);
        // the stored magnitude was amax >= RANGE_LIMIT: bring it to <= 16384 (a factor of two of head-room for later
        // batches); infinite (the fp32 value itself overflowed f16 by an unknown factor): 8 binary orders per attempt
        int need = std::isfinite(amax) ? (int)std::ceil(std::log2(amax / 16384.f)) : 8;
        need = need < 1 ? 1 : need;
        if (m->shift[i] + need > RS_MAX_SHIFT)
            return fail(c, SGPT_ERR_RANGE, "SGPT_F16: an activation class needs a range shift beyond 2^40; load with SGPT_BF16");
        m->shift[i] += need;
        ++raised;
    }
    if (raised) {
        HIPC(c, hipMemsetAsync(m->range_dev + 1, 0, (size_t)n * 4, s));
        const unsigned keep = h[0] & ~1u;                 // bit 0 handled; fp8 saturation bits stay
        HIPC(c, hipMemcpyAsync(m->range_dev, &keep, 4, hipMemcpyHostToDevice, s));
        HIPC(c, hipStreamSynchronize(s));
        c->generation++;                                  // captured graphs carry the old factors as kernel arguments
    }
    *n_raised = raised;
    return SGPT_OK;
}

sgpt_status sgpt_model_get_range_shifts(sgpt_model* m, int32_t* shifts, int32_t n) {
    if (!m || !shifts || n != m->d.n_layers * RS_N) return m ? fail(m->ctx, SGPT_ERR_INVALID, "sgpt_model_get_range_shifts: n must be 4 * n_layers") : SGPT_ERR_INVALID;
    for (int i = 0; i < n; ++i) shifts[i] = m->shift[i];
    return SGPT_OK;
}

sgpt_status sgpt_model_set_range_shifts(sgpt_model* m, const int32_t* shifts, int32_t n) {
    if (!m) return SGPT_ERR_INVALID;
    sgpt_ctx* c = m->ctx;
    if (m->d.compute_dtype != SGPT_F16 || !shifts || n != m->d.n_layers * RS_N)
        return fail(c, SGPT_ERR_INVALID, "sgpt_model_set_range_shifts: SGPT_F16 models, n = 4 * n_layers");
    for (int i = 0; i < n; ++i)
        if (shifts[i] < 0 || shifts[i] > RS_MAX_SHIFT) return fail(c, SGPT_ERR_INVALID, "range shifts must lie in [0, 40]");
    // the LayerNorm kernels have no run-time range tracker: their shifts may not go below the bound sgpt_model_load derived
    // from THIS checkpoint's parameters (shifts pinned from another checkpoint would overflow to inf unnoticed)
    for (int i = 0; i < n; ++i) {
        const int cls = i % RS_N, blk = i / RS_N;
        if ((cls == RS_LN1 || cls == RS_LN2) && shifts[i] < m->ln_floor[blk])
            return fail(c, SGPT_ERR_INVALID, "sgpt_model_set_range_shifts: a LayerNorm shift below the bound of this checkpoint's LayerNorm parameters (block " +
                        std::to_string(blk) + ": at least " + std::to_string(m->ln_floor[blk]) + ")");
    }
    for (int i = 0; i < n; ++i) m->shift[i] = shifts[i];
    c->generation++;
    return SGPT_OK;
}

// ---- precision plan: split-precision (hi + lo) operand classes per block ----------------------------------------------
sgpt_status sgpt_model_set_precision(sgpt_model* m, const int32_t* plan, int32_t n) {
    if (!m) return SGPT_ERR_INVALID;
    sgpt_ctx* c = m->ctx;
    const int cd = m->d.compute_dtype;
    if (!plan || n != m->d.n_layers * PC_N) return fail(c, SGPT_ERR_INVALID, "sgpt_model_set_precision: n must be 5 * n_layers");
    const int dh = m->d.d_model / m->d.n_heads;
    bool any = false;
    for (int i = 0; i < n; ++i) {
        const int cls = i % PC_N, v = plan[i];
        if (v < 0 || v > (cls == PC_LN1 ? 3 : 1)) return fail(c, SGPT_ERR_INVALID, "sgpt_model_set_precision: entries are 0 | 1 (LayerNorm-1 class: 0 ... 3)");
        any |= v != 0;
        const bool legacy_ok = cls == PC_LN1 && (v == 1 || v == 3 || (v == 2 && m->qkv3_rows == 3)) && m->L[i / PC_N].w_qkv3 != nullptr;
        if (v != 0 && cls != PC_ATT && !m->split_all && !legacy_ok)      // (the attention class splits activations only)
            return fail(c, SGPT_ERR_INVALID, "sgpt_model_set_precision: the model was loaded without split weight copies (sgpt_model_desc.split_weights)");
        if (cls == PC_ATT && v != 0 && (m->d.arch == SGPT_ARCH_GPTJ || !attn_x3_supported(dh)))
            return fail(c, SGPT_ERR_INVALID, "sgpt_model_set_precision: split-precision attention needs head_dim 64 or 128 and no rotary embedding (GPT-Neo / BLOOM)");
        if (cls == PC_LN2 && m->d.arch == SGPT_ARCH_GPTJ && v != 0 && plan[i - PC_LN2 + PC_LN1] == 0)
            return fail(c, SGPT_ERR_INVALID, "sgpt_model_set_precision: GPT-J's MLP reads ln_1's output: a split fc1 needs a split LayerNorm-1 entry");
    }
    if (any && cd != SGPT_F16 && cd != SGPT_BF16) return fail(c, SGPT_ERR_INVALID, "sgpt_model_set_precision applies to SGPT_F16 / SGPT_BF16 models");
    for (int i = 0; i < n; ++i) m->prec[i] = plan[i];
    c->generation++;                                      // captured graphs carry the old launch sequence
    return SGPT_OK;
}

sgpt_status sgpt_model_get_precision(sgpt_model* m, int32_t* plan, int32_t n) {
    if (!m || !plan || n != m->d.n_layers * PC_N) return m ? fail(m->ctx, SGPT_ERR_INVALID, "sgpt_model_get_precision: n must be 5 * n_layers") : SGPT_ERR_INVALID;
    for (int i = 0; i < n; ++i) plan[i] = m->prec[i];
    return SGPT_OK;
}

// The [W_hi | W_hi | W_lo] copies of sgpt_model_desc.split_weights are 3 x the 16-bit weight bytes on top of the plain copy
// (+35 GB at GPT-J-6B shape).  A model whose probe settled on plain operands gives them back here; a later plan
// that needs them is refused by sgpt_model_set_precision as for a model loaded without them.  The round-3 qk_split copies
// (q and k rows) stay when a plan entry still reads them.
sgpt_status sgpt_model_release_split_weights(sgpt_model* m, int64_t* bytes_freed) {
    if (!m) return SGPT_ERR_INVALID;
    sgpt_ctx* c = m->ctx;
    if (bytes_freed) *bytes_freed = 0;
    if (!m->split_all) return SGPT_OK;
    bool qk_used = false;
    for (int i = 0; i < m->d.n_layers; ++i)
        for (int cls = 0; cls < PC_N; ++cls) {
            const int v = m->prec[(size_t)i * PC_N + cls];
            if (v == 0 || cls == PC_ATT) continue;
            if (cls == PC_LN1) { qk_used = true; continue; }            // any LayerNorm-1 level reads w_qkv3 (level 2: its V rows too): kept whole
            return fail(c, SGPT_ERR_INVALID, "sgpt_model_release_split_weights: the installed precision plan reads the split copies");
        }
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipDeviceSynchronize());
    const size_t dm = m->d.d_model, ffn = m->d.d_ffn;
    int64_t freed = 0;
    auto drop = [&](void*& p, size_t bytes) {
        if (!p) return;
        for (size_t i = 0; i < m->allocs.size(); ++i)
            if (m->allocs[i] == p) { m->allocs[i] = m->allocs.back(); m->allocs.pop_back(); break; }
        (void)hipFree(p);
        p = nullptr;
        freed += (int64_t)bytes;
    };
    for (int i = 0; i < m->d.n_layers; ++i) {
        LayerW& l = m->L[i];
        drop(l.w_o3, dm * 3 * dm * 2); drop(l.w_fc3, ffn * 3 * dm * 2); drop(l.w_proj3, dm * 3 * ffn * 2);
        if (!qk_used) drop(l.w_qkv3, 3 * dm * 3 * dm * 2);
    }
    if (!qk_used) m->qkv3_rows = 0;
    m->split_all = false;
    c->generation++;
    if (bytes_freed) *bytes_freed = freed;
    return SGPT_OK;
}

sgpt_status sgpt_model_precision_probe_begin(sgpt_model* m) {
    if (!m) return SGPT_ERR_INVALID;
    sgpt_ctx* c = m->ctx;
    if (!m->crest_dev) return fail(c, SGPT_ERR_INVALID, "the precision probe applies to SGPT_F16 / SGPT_BF16 models");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipDeviceSynchronize());
    HIPC(c, hipMemset(m->crest_dev, 0, (size_t)m->d.n_layers * RS_N * 4));
    m->probing = true;
    c->generation++;                                      // a graph captured while probing must not outlive the probe
    return SGPT_OK;
}

sgpt_status sgpt_model_precision_probe_end(sgpt_model* m, float* crest_out) {
    if (!m) return SGPT_ERR_INVALID;
    sgpt_ctx* c = m->ctx;
    if (!m->probing) return fail(c, SGPT_ERR_INVALID, "sgpt_model_precision_probe_end without _begin");
    m->probing = false;
    c->generation++;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipDeviceSynchronize());
    if (crest_out) HIPC(c, hipMemcpy(crest_out, m->crest_dev, (size_t)m->d.n_layers * RS_N * 4, hipMemcpyDeviceToHost));
    return SGPT_OK;
}

sgpt_status sgpt_row_crest(sgpt_ctx* c, const void* x, int32_t dtype, int64_t n, int32_t d, int64_t ld, float* crest_out, void* stream) {
    if (!c || !x || !crest_out || n <= 0 || n > INT32_MAX || d <= 0 || d % 2 || ld < d || (dtype != SGPT_BF16 && dtype != SGPT_F16))
        return fail(c, SGPT_ERR_INVALID, "sgpt_row_crest: bad arguments (16-bit rows, d % 2 == 0)");
    HIPC(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    unsigned* slot = (unsigned*)c->range_flag + 16;          // a scratch word of the ctx's 256-byte flag block
    HIPC(c, hipMemsetAsync(slot, 0, 4, s));
    launch_crest16(x, (int)n, d, ld, dtype, slot, s);
    unsigned h = 0;
    HIPC(c, hipMemcpyAsync(&h, slot, 4, hipMemcpyDeviceToHost, s));
    HIPC(c, hipStreamSynchronize(s));
    memcpy(crest_out, &h, 4);
    return SGPT_OK;
}

sgpt_status sgpt_split16(sgpt_ctx* c, const float* in, int64_t n, int32_t d, int32_t layout, void* out, int32_t out_dtype,
                         void* stream) {
    if (!c || !in || !out || n <= 0 || d <= 0 || d % 4 || (layout != 0 && layout != 1) || (out_dtype != SGPT_BF16 && out_dtype != SGPT_F16))
        return fail(c, SGPT_ERR_INVALID, "sgpt_split16: bad arguments (d % 4 == 0, layout 0 | 1, 16-bit out_dtype)");
    HIPC(c, hipSetDevice(c->device));
    launch_split16_rows(in, n, d, layout, out, out_dtype, (hipStream_t)stream);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_linear(sgpt_ctx* c, int32_t dtype, int32_t epi, int32_t out_dtype, const void* A, const void* W,
                        const float* bias, const float* resid, void* out, int32_t M, int32_t N, int32_t K, void* stream) {
    if (!c || !A || !W || !out || M <= 0 || N <= 0 || K <= 0) return fail(c, SGPT_ERR_INVALID, "sgpt_linear: bad arguments");
    const bool in16 = dtype == SGPT_BF16 || dtype == SGPT_F16, o16 = out_dtype == SGPT_BF16 || out_dtype == SGPT_F16;
    if (!in16 && dtype != SGPT_F32) return fail(c, SGPT_ERR_INVALID, "sgpt_linear: bad dtype");
    if (epi != EPI_STORE && epi != EPI_BIAS_GELU && epi != EPI_BIAS_RESID && epi != EPI_VT)
        return fail(c, SGPT_ERR_INVALID, "sgpt_linear: epi must be 0 (store), 1 (bias+gelu), 2 (bias+residual) or 4 (transposed store)");
    if ((epi == EPI_BIAS_GELU || epi == EPI_BIAS_RESID) && !bias) return fail(c, SGPT_ERR_INVALID, "sgpt_linear: bias required");
    if (epi == EPI_BIAS_RESID && (!resid || out_dtype != SGPT_F32)) return fail(c, SGPT_ERR_INVALID, "sgpt_linear: residual epilogue is fp32");
    if (epi == EPI_VT && (!in16 || out_dtype != dtype || M % 128)) return fail(c, SGPT_ERR_INVALID, "sgpt_linear: transposed store is 16-bit, M % 128 == 0");
    if (epi == EPI_BIAS_GELU && out_dtype != dtype) return fail(c, SGPT_ERR_INVALID, "sgpt_linear: gelu output has the operand dtype");
    if (in16 && (o16 ? out_dtype != dtype : false)) return fail(c, SGPT_ERR_INVALID, "sgpt_linear: 16-bit output must match the operand format");
    if (!in16 && o16) return fail(c, SGPT_ERR_INVALID, "sgpt_linear: fp32 operands give fp32 output");
    if (K % (in16 ? 8 : 4) || (epi != EPI_STORE && N % 4)) return fail(c, SGPT_ERR_INVALID, "sgpt_linear: K % 8 (16-bit) / 4 (fp32), N % 4");
    HIPC(c, hipSetDevice(c->device));
    GemmArgs g{};
    g.A = A; g.lda = K; g.W = W; g.ldw = K; g.M = M; g.m_valid = M; g.N = N; g.K = K; g.out = out;
    g.ldo = epi == EPI_VT ? M : N; g.bias = bias; g.resid = resid;
    g.range_flag = out_dtype == SGPT_F16 ? c->range_flag : nullptr;
    gemm(c, dtype, epi, out_dtype, g, (hipStream_t)stream);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_linear_query(sgpt_ctx* c, int32_t dtype, int32_t epi, const void* A, const float* x, const float* ln_gamma,
                              const float* ln_beta, float ln_eps, const void* W, const float* bias, const float* resid, void* out,
                              void* out_vt, int32_t n_split, int32_t M, int32_t N, int32_t K, void* stream) {
    if (!c || !W || !out || M <= 0 || N <= 0 || K <= 0 || (!A && !x)) return fail(c, SGPT_ERR_INVALID, "sgpt_linear_query: bad arguments");
    if (dtype != SGPT_BF16 && dtype != SGPT_F16) return fail(c, SGPT_ERR_INVALID, "sgpt_linear_query: 16-bit operands (SGPT_BF16 | SGPT_F16)");
    if (epi != EPI_STORE && epi != EPI_BIAS_GELU && epi != EPI_BIAS_RESID && epi != EPI_QKV)
        return fail(c, SGPT_ERR_INVALID, "sgpt_linear_query: epi must be 0 (store), 1 (bias+gelu), 2 (bias+residual) or 7 (q | k | V^T)");
    if (x && (!ln_gamma || !ln_beta || (epi != EPI_QKV && epi != EPI_BIAS_GELU)))
        return fail(c, SGPT_ERR_INVALID, "sgpt_linear_query: the LayerNorm prologue feeds epi 7 (QKV) and 1 (fc1 + GELU) and needs gamma / beta");
    if ((epi == EPI_BIAS_GELU || epi == EPI_BIAS_RESID) && !bias) return fail(c, SGPT_ERR_INVALID, "sgpt_linear_query: bias required");
    if (epi == EPI_BIAS_RESID && !resid) return fail(c, SGPT_ERR_INVALID, "sgpt_linear_query: residual required");
    if (epi == EPI_QKV && (!out_vt || n_split <= 0 || n_split >= N)) return fail(c, SGPT_ERR_INVALID, "sgpt_linear_query: epi 7 needs out_vt and 0 < n_split < N");
    HIPC(c, hipSetDevice(c->device));
    QGemmArgs q{};
    q.g.A = A; q.g.lda = K; q.g.W = W; q.g.ldw = K; q.g.M = M; q.g.m_valid = M; q.g.N = N; q.g.K = K; q.g.out = out;
    q.g.ldo = epi == EPI_QKV ? n_split : N; q.g.out2 = out_vt; q.g.ldo2 = M; q.g.n_split = epi == EPI_QKV ? n_split : 0;
    q.g.bias = bias; q.g.resid = resid;
    const int out_dtype = epi == EPI_BIAS_RESID ? SGPT_F32 : dtype;
    q.g.range_flag = out_dtype == SGPT_F16 ? c->range_flag : nullptr;
    if (x) { q.x = x; q.ln_g = ln_gamma; q.ln_b = ln_beta; q.eps = ln_eps; q.g.A = nullptr; }
    if (!qgemm(c, dtype, epi, out_dtype, q, (hipStream_t)stream))
        return fail(c, SGPT_ERR_INVALID, "sgpt_linear_query: shape not served by the query-sized kernels (M % 32, M <= 4096; K / 128 a multiple of 4 or 6; "
                                         "N % 16; LayerNorm prologue: K = 512 | 768 | 1024, N % 32)");
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_linear_split(sgpt_ctx* c, int32_t dtype, int32_t epi, const void* A, const void* W, const float* bias, void* out,
                              int64_t ldo, int64_t lo_delta, int64_t hi2_delta, int32_t M, int32_t N, int32_t K, void* stream) {
    if (!c || !A || !W || !out || M <= 0 || N <= 0 || K <= 0 || lo_delta == 0) return fail(c, SGPT_ERR_INVALID, "sgpt_linear_split: bad arguments");
    if (dtype != SGPT_BF16 && dtype != SGPT_F16) return fail(c, SGPT_ERR_INVALID, "sgpt_linear_split: 16-bit operands");
    if (epi != EPI_STORE && epi != EPI_BIAS_GELU && epi != EPI_VT) return fail(c, SGPT_ERR_INVALID, "sgpt_linear_split: epi 0 (store), 1 (bias+gelu) or 4 (transposed store)");
    if (epi == EPI_BIAS_GELU && !bias) return fail(c, SGPT_ERR_INVALID, "sgpt_linear_split: bias required");
    if (epi == EPI_VT && (M % 128 || hi2_delta != 0)) return fail(c, SGPT_ERR_INVALID, "sgpt_linear_split: transposed store needs M % 128 == 0 and writes hi + lo only");
    if (K % 8 || N % 4 || ldo < (epi == EPI_VT ? M : N)) return fail(c, SGPT_ERR_INVALID, "sgpt_linear_split: K % 8, N % 4, ldo >= row length");
    HIPC(c, hipSetDevice(c->device));
    GemmArgs g{};
    g.A = A; g.lda = K; g.W = W; g.ldw = K; g.M = M; g.m_valid = M; g.N = N; g.K = K; g.out = out; g.ldo = ldo; g.bias = bias;
    g.lo_delta = lo_delta; g.hi2_delta = hi2_delta;
    g.range_flag = dtype == SGPT_F16 ? c->range_flag : nullptr;
    gemm(c, dtype, epi, dtype, g, (hipStream_t)stream);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

sgpt_status sgpt_bench_gemm(sgpt_ctx* c, int32_t dtype, int32_t epi, int32_t out_dtype, int32_t M, int32_t N, int32_t K,
                            int32_t iters, float* ms_out) {
    if (!c || !ms_out || M <= 0 || N <= 0 || K <= 0 || iters <= 0) return fail(c, SGPT_ERR_INVALID, "sgpt_bench_gemm: bad arguments");
    HIPC(c, hipSetDevice(c->device));
    if (dtype == SGPT_FP8M) {   // fp8 MFMA kernel: random fp32 operands quantised row-wise to e4m3 codes + power-of-two scales
        if (!gemm_fp8_shape_ok(M, N, K) || (epi != EPI_BIAS_GELU && epi != EPI_BIAS_RESID && epi != EPI_NONE))
            return fail(c, SGPT_ERR_INVALID, "sgpt_bench_gemm(fp8): shapes % 256, epi 1 | 2 | 5");
        float *Af = nullptr, *Wf = nullptr, *sa = nullptr, *sw = nullptr, *bias = nullptr; uint8_t *A8 = nullptr, *W8 = nullptr; void* O = nullptr;
        HIPC(c, hipMalloc((void**)&Af, (size_t)M * K * 4)); HIPC(c, hipMalloc((void**)&Wf, (size_t)N * K * 4));
        HIPC(c, hipMalloc((void**)&A8, (size_t)M * K)); HIPC(c, hipMalloc((void**)&W8, (size_t)N * K));
        HIPC(c, hipMalloc((void**)&sa, (size_t)M * 4)); HIPC(c, hipMalloc((void**)&sw, (size_t)N * 4));
        HIPC(c, hipMalloc((void**)&bias, (size_t)N * 4)); HIPC(c, hipMalloc(&O, (size_t)M * N * 4));
        launch_fill_rand(Af, (long)M * K, 0, 1u, 1.0f, 0); launch_fill_rand(Wf, (long)N * K, 0, 2u, 0.05f, 0);
        launch_fill_rand(bias, N, 0, 3u, 0.1f, 0); launch_fill_rand((float*)O, (long)M * N, 0, 4u, 1.0f, 0);
        launch_fp8_quant_rows(Af, M, K, A8, sa, 0); launch_fp8_quant_rows(Wf, N, K, W8, sw, 0);
        GemmArgs q{};
        q.A = A8; q.lda = K; q.W = W8; q.ldw = K; q.M = M; q.m_valid = M; q.N = N; q.K = K; q.out = O; q.ldo = N; q.bias = bias;
        q.resid = (const float*)O; q.a_scale = sa; q.a_scalar = 1.0f; q.w_scale = sw; q.out_scale = 0.0625f;
        hipEvent_t e0, e1;
        HIPC(c, hipEventCreate(&e0)); HIPC(c, hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) launch_gemm_fp8(epi, out_dtype, q, 0);
        HIPC(c, hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) launch_gemm_fp8(epi, out_dtype, q, 0);
        HIPC(c, hipEventRecord(e1, 0));
        HIPC(c, hipEventSynchronize(e1));
        float ms = 0;
        HIPC(c, hipEventElapsedTime(&ms, e0, e1));
        *ms_out = ms / iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        (void)hipFree(Af); (void)hipFree(Wf); (void)hipFree(A8); (void)hipFree(W8); (void)hipFree(sa); (void)hipFree(sw); (void)hipFree(bias); (void)hipFree(O);
        HIPC(c, hipGetLastError());
        return SGPT_OK;
    }
    const size_t esz = dtype == SGPT_F32 ? 4 : 2, osz = out_dtype == SGPT_F32 ? 4 : 2;
    void *A = nullptr, *W = nullptr, *O = nullptr; float* bias = nullptr;
    HIPC(c, hipMalloc(&A, (size_t)M * K * esz));
    HIPC(c, hipMalloc(&W, (size_t)N * K * esz));
    HIPC(c, hipMalloc(&O, (size_t)M * N * (epi == EPI_BIAS_RESID ? 4 : osz) + 4096));
    HIPC(c, hipMalloc((void**)&bias, (size_t)N * 4));
    launch_fill_rand(A, (long)M * K, dtype, 1u, 1.0f, 0);
    launch_fill_rand(W, (long)N * K, dtype, 2u, 0.05f, 0);
    launch_fill_rand(bias, N, 0, 3u, 0.1f, 0);
    launch_fill_rand(O, (long)M * N, epi == EPI_BIAS_RESID ? 0 : out_dtype, 4u, 1.0f, 0);
    GemmArgs g{};
    g.A = A; g.lda = K; g.W = W; g.ldw = K; g.M = M; g.m_valid = M; g.N = N; g.K = K; g.out = O;
    g.ldo = epi == EPI_VT ? M : N; g.bias = bias; g.resid = epi == EPI_BIAS_RESID ? (const float*)O : nullptr;
    hipEvent_t e0, e1;
    HIPC(c, hipEventCreate(&e0)); HIPC(c, hipEventCreate(&e1));
    long long* dbg = nullptr;
    if (exp_env("SGPT_GEMM_DBG")) { HIPC(c, hipMalloc((void**)&dbg, 128 * 8)); HIPC(c, hipMemset(dbg, 0, 128 * 8)); g.dbg = dbg; }
    for (int i = 0; i < 3; ++i) launch_gemm(dtype, epi, out_dtype, g, 0);
    HIPC(c, hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch_gemm(dtype, epi, out_dtype, g, 0);
    HIPC(c, hipEventRecord(e1, 0));
    HIPC(c, hipEventSynchronize(e1));
    float ms = 0;
    HIPC(c, hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / iters;
    if (dbg) {
        long long h[128];
        HIPC(c, hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
        for (int tl = 0; tl < 4; ++tl) {
            const long long* r = h + tl * 8;
            const long long* ks = h + 64 + tl * 16;
            fprintf(stderr, "tile %d: k-step starts (rel. to step 0):", tl);
            for (int q = 1; q < 12; ++q) fprintf(stderr, " %lld", ks[q] - ks[0]);
            fprintf(stderr, " | kloop_end %lld  dma_wait +%lld  barrier +%lld  epilogue +%lld | next tile step0 at %lld\n",
                    r[0] - ks[0], r[1] - r[0], r[2] - r[1], r[3] - r[2], h[64 + (tl + 1) * 16] - ks[0]);
        }
        (void)hipFree(dbg);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(O); (void)hipFree(bias);
    HIPC(c, hipGetLastError());
    return SGPT_OK;
}

}  // extern "C"
