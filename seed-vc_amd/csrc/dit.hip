// DiT / UViT estimator and CFM Euler sampler: host-side engine that packs the reference state_dict and
// enqueues the hand-written kernels (kgemm / attention / elementwise) on the caller's HIP stream.
//
// What is restructured relative to the reference's op order (same math, different rounding only):
//  * cond_x_merge_linear(cat[x, prompt_x, cond_projection(mu), style]) is split by input block: only the
//    x block (K = in_channels) is evaluated per step; the prompt / content / style blocks and all biases
//    form a per-utterance static term computed once per call in fp32 (W_c * W_cp is pre-composed).
//  * the classifier-free-guidance "null" twins are extra streams of the same batch (flow_matching.py:86-98);
//    their static term is a constant row.
//  * every timestep-only quantity (t-embeddings, all AdaLN projections, WaveNet cond_layer, FinalLayer
//    modulation) is tabulated for all steps at the start of a call.
//  * residual stream, ODE state, norms, RoPE, softmax statistics and all accumulators are fp32; GEMM and
//    attention operands are fp16 (the reference autocasts the same ops to fp16: inference.py:499).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "model_util.h"

using namespace svc;

namespace {
constexpr int MAX_STREAMS = 3;
constexpr int ROPE_POS = 8192;

float bf16_round(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000u;
    memcpy(&f, &u, 4);
    return f;
}

// torch.linspace(0, 1, n) in fp32 (ATen: start + step*i below the midpoint, end - step*(n-1-i) above)
std::vector<float> linspace01(int n) {
    std::vector<float> v(n);
    const float step = 1.0f / (float)(n - 1);
    const int half = n / 2;
    for (int i = 0; i < n; ++i) v[i] = i < half ? 0.0f + step * (float)i : 1.0f - step * (float)(n - 1 - i);
    return v;
}
}  // namespace

struct svc_dit {
    svc_dit_config_t cfg;
    int D, H, L, C, Dc, S, I, W, NL, WK, npre, C16, C32;
    bool v2, wavenet, adaptive_blocks;   // adaptive_blocks: per-layer modulation used (v1: !time_as_token, v2: always)
    Arena wts, ws;
    // hipGraphs of the Euler loop (all steps of one micro-batch group: estimator body + state update), keyed by everything
    // that is baked into the captured kernel arguments.  The reference's counterpart is `compile_cfm` (torch.compile of the
    // estimator, modules/v2/vc_wrapper.py:116-123).  Only workspace pointers are captured: the kernels touching caller
    // buffers (noise in, conditioning, mel out) stay outside the graph.
    struct StepGraph {
        std::vector<float> key;
        hipGraphExec_t exec = nullptr;
        int seen = 0;                 // calls with this key so far (the first one runs eagerly: lazy kernel attributes)
    };
    std::vector<StepGraph> graphs;
    hipStream_t cap_stream = nullptr;
    // Off by default: measured on MI355X the sampler is GPU-bound at every batch size (B = 1 small: 65.3 ms eager, 67.7 ms
    // replayed -- ~2 k dependent kernels of ~30 us each, the host runs far ahead either way), so replay buys nothing.
    int use_graphs = 0;               // svc_dit_set_graphs(m, 1) / SVC_DIT_GRAPH=1 turn capture + replay on
    void drop_graphs() {
        for (auto& g : graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
        graphs.clear();
    }
    PinnedRing staging;           // per-call host arrays (lengths, time grid) travel through pinned slots: no stream sync
    hipStream_t create_stream;
    int device = -1;              // the device current at creation: calls with another device current are rejected

    struct Layer {
        half_t *wqkv, *wo, *w13, *w2, *wskip;
        float *bskip, *g_attn, *g_ffn;
        half_t* stream = nullptr;   // fused row-panel kernel: [wo | mlp | skip(i+1)? | qkv(i+1)?] fragment stream (fused.hip)
        int stream_slots = 0;
    };
    half_t* stream_pre = nullptr;   // [qkv(0)]
    int stream_pre_slots = 0;
    bool fused_ok = false;
    long fused_min_rows = 10240;    // measured crossover (tiny and small, B = 4 vs 8 utterances x 2 streams): below ~80 row panels the tap-GEMMs win
    std::vector<Layer> layers;
    float* g_final;
    std::vector<int> emit, recv;
    // conditioning path (fp32)
    float *te_w0, *te_b0, *te_w2, *te_b2, *te_freqs;
    float *te2_w0, *te2_b0, *te2_w2, *te2_b2, *te2_freqs;
    float *mod_w, *mod_b;
    int mod_n;            // rows of the stacked modulation projection
    int mod_layer_n;      // per-layer width (v1: 4D = attn 2D | ffn 2D ; v2: 6D)
    // merge / static
    half_t* w_merge_x;    // [Dpad][C16]
    float* w_stat;        // [D][C32 + Dc]  (prompt block | composed content block)
    float* stat_const;    // [D]
    float* w_style_merge; // [D][S] or null
    float *style_in_w, *style_in_b;
    // heads
    half_t *head_wa, *head_w2;
    float *head_b0, *head_b2;
    half_t *w_longskip, *w_conv1, *w_resproj, *w_fl, *w_conv2;
    float *b_longskip, *b_conv1, *b_resproj, *b_fl, *b_conv2;
    float *wn_cond_w, *wn_cond_b, *fl_mod_w, *fl_mod_b;
    std::vector<half_t*> wn_in, wn_res, wn_skip;
    std::vector<float*> wn_res_b, wn_skip_b;
    half_t* wn_tail = nullptr;        // [W][NL W + D] = [skip_0 | ... | skip_{NL-1} | res_projection]: ONE GEMM ends the WaveNet
    float* b_tail = nullptr;          // sum of the skip biases + the res_projection bias
    long acts_stride = 0;             // elements between the per-layer gated-activation buffers
    float* rope;

    // ---- workspace (sized by reserve)
    int cap_streams = 0, cap_B = 0, cap_T = 0, cap_steps = 0;
    int seq_rows = 0, vt_ld = 0;
    float *x32, *xin, *st_term, *v32, *u_rowvec, *tok_style, *prompt32, *wnx32, *flin32;
    half_t *x16, *n16, *qk16, *vt, *ao16, *ff16, *h16, *hm16, *xr16, *wnx16, *acts16, *fl16, *fo16;
    std::vector<half_t*> skip16;
    int *d_kvlen, *d_convlen, *d_plen;
    int* d_convlen_w = nullptr;   // conv lengths relative to the head window
    int win0 = 0;                 // first sequence row the output head is evaluated on (0 = all rows)
    float *d_tvals, *d_tfeat, *d_th, *d_t1, *d_t1s, *d_t2, *d_mod, *d_gcond, *d_flmod, *d_stylevec;
    int microbatch = 0;

    int pack(const StateDict& sd, hipStream_t st);
    int reserve(int n_streams, int B, int T, int n_steps, hipStream_t st);
    int tables(const std::vector<float>& tvals, hipStream_t st);
    int statics(int n_streams, const int (*flags)[3], int B, int T, const float* mu, const float* style_dev,
                hipStream_t st);
    int run_group(const svc_cfm_args_t* a, int b0, int nb, int n_streams, const int (*flags)[3],
                  const std::vector<float>& tvals, const std::vector<float>& dts, float c0, float ca, float cb,
                  int stream_a, int stream_b, hipStream_t st);
    int body(int n_streams, int B, int T, int step, hipStream_t st);
    int body_fused(int n_streams, int B, int T, int step, hipStream_t st);
    int head(int n_streams, int B, int T, int step, hipStream_t st);
};

// --------------------------------------------------------------------------------------------- packing
namespace {

// dense [N][K] fp32 (row stride src_ld, column offset src_col) -> packed [Npad][ldw] at column dst_col
int pack_block(int dtype, const float* src, long src_ld, long src_col, int N, int K, void* dst, long ldw, long dst_col,
               long dst_row0, long dst_row_step, const float* scale, hipStream_t st) {
    return pack_any(dtype, src + src_col, dst, dst_row0 * ldw + dst_col, N, 1, K, src_ld, 0, 1, dst_row_step * ldw, 0, 1,
                    scale, st);
}

float* copy_vec(Arena& ar, const float* src, long n, hipStream_t st) {
    float* d = ar.alloc_n<float>(n, st);
    if (!d) return nullptr;
    if (hipMemcpyAsync(d, src, n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("hipMemcpyAsync failed");
        return nullptr;
    }
    return d;
}

}  // namespace

#define GETW(var, name, ...)                                              \
    const svc_tensor_desc_t* var = sd.get(name);                          \
    if (require_shape(var, name, {__VA_ARGS__})) return 1;

int svc_dit::pack(const StateDict& sd, hipStream_t st) {
    const long Dp = round_up(D, 128);
    auto new16 = [&](long rows, long ld) { return wts.alloc_n<half_t>(round_up(rows, 128) * ld, st); };
    layers.resize(L);
    for (int i = 0; i < L; ++i) {
        const std::string p = "transformer.layers." + std::to_string(i) + ".";
        Layer& ly = layers[i];
        GETW(wqkv, p + "attention.wqkv.weight", 3 * D, D);
        GETW(wo, p + "attention.wo.weight", D, D);
        GETW(w1, p + "feed_forward.w1.weight", I, D);
        GETW(w3, p + "feed_forward.w3.weight", I, D);
        GETW(w2, p + "feed_forward.w2.weight", D, I);
        ly.wqkv = new16(3 * D, D);
        ly.wo = new16(D, D);
        ly.w13 = new16(2 * I, D);
        ly.w2 = new16(D, I);
        if (!ly.wqkv || !ly.wo || !ly.w13 || !ly.w2) return 1;
        if (pack_block(0, wqkv->data, D, 0, 3 * D, D, ly.wqkv, D, 0, 0, 1, nullptr, st)) return 1;
        if (pack_block(0, wo->data, D, 0, D, D, ly.wo, D, 0, 0, 1, nullptr, st)) return 1;
        if (pack_block(0, w1->data, D, 0, I, D, ly.w13, D, 0, 0, 2, nullptr, st)) return 1;   // rows 2j   = w1_j
        if (pack_block(0, w3->data, D, 0, I, D, ly.w13, D, 0, 1, 2, nullptr, st)) return 1;   // rows 2j+1 = w3_j
        if (pack_block(0, w2->data, I, 0, D, I, ly.w2, I, 0, 0, 1, nullptr, st)) return 1;
        const std::string an = v2 ? p + "attention_norm.norm.weight" : p + "attention_norm.norm.weight";
        const std::string fn = v2 ? p + "ffn_norm.weight" : p + "ffn_norm.norm.weight";
        GETW(ga, an, D);
        GETW(gf, fn, D);
        ly.g_attn = copy_vec(wts, ga->data, D, st);
        ly.g_ffn = copy_vec(wts, gf->data, D, st);
        if (!ly.g_attn || !ly.g_ffn) return 1;
        ly.wskip = nullptr;
        ly.bskip = nullptr;
        if (std::find(recv.begin(), recv.end(), i) != recv.end()) {
            GETW(ws_, p + "skip_in_linear.weight", D, 2 * D);
            GETW(bs_, p + "skip_in_linear.bias", D);
            ly.wskip = new16(D, 2 * D);
            if (!ly.wskip) return 1;
            if (pack_block(0, ws_->data, 2 * D, 0, D, 2 * D, ly.wskip, 2 * D, 0, 0, 1, nullptr, st)) return 1;
            ly.bskip = copy_vec(wts, bs_->data, D, st);
            if (!ly.bskip) return 1;
        }
    }
    {
        GETW(gn, "transformer.norm.norm.weight", D);
        g_final = copy_vec(wts, gn->data, D, st);
        if (!g_final) return 1;
    }
    // ---- fragment streams of the fused row-panel kernel (fused.hip): layer i carries wo_i, mlp_i and the NEXT layer's
    // skip linear / QKV projection; the stream of layer L-1 ends after the MLP
    fused_ok = fused_supported(D, I);
    if (fused_ok) {
        auto W = [&](const std::string& k) { return sd.get(k)->data; };
        const int slot_halfs = (D / 16) * 512;
        for (int i = 0; i < L; ++i) {
            const std::string p = "transformer.layers." + std::to_string(i) + ".";
            const bool last = i == L - 1;
            const bool nskip = !last && layers[i + 1].wskip != nullptr;
            const std::string pn = "transformer.layers." + std::to_string(i + 1) + ".";
            const long halfs = fused_stream_halfs(D, I, true, nskip, !last);
            layers[i].stream = wts.alloc_n<half_t>(halfs, st);
            if (!layers[i].stream) return 1;
            const long wr = fused_pack_stream(layers[i].stream, D, I, W(p + "attention.wo.weight"), W(p + "feed_forward.w1.weight"),
                                              W(p + "feed_forward.w3.weight"), W(p + "feed_forward.w2.weight"),
                                              nskip ? W(pn + "skip_in_linear.weight") : nullptr,
                                              last ? nullptr : W(pn + "attention.wqkv.weight"), st);
            if (wr != halfs) { set_error("fused stream packing failed"); return 1; }
            layers[i].stream_slots = (int)(halfs / slot_halfs);
        }
        const long halfs = fused_stream_halfs(D, I, false, false, true);
        stream_pre = wts.alloc_n<half_t>(halfs, st);
        if (!stream_pre) return 1;
        if (fused_pack_stream(stream_pre, D, I, nullptr, nullptr, nullptr, nullptr, nullptr,
                              W("transformer.layers.0.attention.wqkv.weight"), st) != halfs) { set_error("fused stream packing failed"); return 1; }
        stream_pre_slots = (int)(halfs / slot_halfs);
    }
    // ---- stacked modulation projections: [layer0 | layer1 | ... | final]
    mod_layer_n = adaptive_blocks ? (v2 ? 6 * D : 4 * D) : 0;
    mod_n = L * mod_layer_n + 2 * D;
    mod_w = wts.alloc_n<float>((long)mod_n * D, st);
    mod_b = wts.alloc_n<float>(mod_n, st);
    if (!mod_w || !mod_b) return 1;
    auto put_mod = [&](const std::string& wname, const std::string& bname, int rows, long row0) -> int {
        GETW(w, wname, rows, D);
        GETW(b, bname, rows);
        if (pack_f32_launch(w->data, mod_w + row0 * D, rows, 1, D, D, 0, 1, D, 0, 1, nullptr, st)) return 1;
        SVC_CHECK_HIP(hipMemcpyAsync(mod_b + row0, b->data, rows * sizeof(float), hipMemcpyDeviceToDevice, st));
        return 0;
    };
    for (int i = 0; i < L && adaptive_blocks; ++i) {
        const std::string p = "transformer.layers." + std::to_string(i) + ".";
        if (v2) {
            if (put_mod(p + "attention_norm.linear.weight", p + "attention_norm.linear.bias", 6 * D, (long)i * mod_layer_n)) return 1;
        } else {
            if (put_mod(p + "attention_norm.project_layer.weight", p + "attention_norm.project_layer.bias", 2 * D, (long)i * mod_layer_n)) return 1;
            if (put_mod(p + "ffn_norm.project_layer.weight", p + "ffn_norm.project_layer.bias", 2 * D, (long)i * mod_layer_n + 2 * D)) return 1;
        }
    }
    if (v2) {
        if (put_mod("transformer.norm.linear.weight", "transformer.norm.linear.bias", 2 * D, (long)L * mod_layer_n)) return 1;
    } else {
        if (put_mod("transformer.norm.project_layer.weight", "transformer.norm.project_layer.bias", 2 * D, (long)L * mod_layer_n)) return 1;
    }
    // ---- timestep embedders
    auto put_temb = [&](const std::string& p, int dim, float** w0, float** b0, float** w2, float** b2, float** fr) -> int {
        GETW(a, p + ".mlp.0.weight", dim, 256);
        GETW(b, p + ".mlp.0.bias", dim);
        GETW(c, p + ".mlp.2.weight", dim, dim);
        GETW(d, p + ".mlp.2.bias", dim);
        *w0 = copy_vec(wts, a->data, (long)dim * 256, st);
        *b0 = copy_vec(wts, b->data, dim, st);
        *w2 = copy_vec(wts, c->data, (long)dim * dim, st);
        *b2 = copy_vec(wts, d->data, dim, st);
        if (!*w0 || !*b0 || !*w2 || !*b2) return 1;
        if (const auto* f = sd.get(p + ".freqs")) {
            if (require_shape(f, p + ".freqs", {128})) return 1;
            *fr = copy_vec(wts, f->data, 128, st);
        } else {   // v2 recomputes exp(-ln(1e4) i / 128) per call (v2/dit_wrapper.py:41-47)
            std::vector<float> h(128);
            for (int i = 0; i < 128; ++i) h[i] = expf(-logf(10000.0f) * (float)i / 128.0f);
            *fr = wts.alloc_n<float>(128, st);
            if (*fr) SVC_CHECK_HIP(hipMemcpyAsync(*fr, h.data(), 512, hipMemcpyHostToDevice, st));
            SVC_CHECK_HIP(hipStreamSynchronize(st));
        }
        return *fr ? 0 : 1;
    };
    if (put_temb("t_embedder", D, &te_w0, &te_b0, &te_w2, &te_b2, &te_freqs)) return 1;

    // ---- merge linear: split by input block [x (C) | prompt (C) | content (D) | style (S)?]
    const bool style_in_merge = !v2 && cfg.style_condition && !cfg.style_as_token;
    const int Kin = D + 2 * C + (style_in_merge ? S : 0);
    GETW(wm, "cond_x_merge_linear.weight", D, Kin);
    GETW(bm, "cond_x_merge_linear.bias", D);
    GETW(wcp, "cond_projection.weight", D, Dc);
    GETW(bcp, "cond_projection.bias", D);
    w_merge_x = new16(D, C16);
    if (!w_merge_x) return 1;
    if (pack_block(0, wm->data, Kin, 0, D, C, w_merge_x, C16, 0, 0, 1, nullptr, st)) return 1;
    const long Ks = C32 + Dc;
    w_stat = wts.alloc_n<float>(Dp * Ks, st);
    stat_const = wts.alloc_n<float>(D, st);
    if (!w_stat || !stat_const) return 1;
    if (pack_block(1, wm->data, Kin, C, D, C, w_stat, Ks, 0, 0, 1, nullptr, st)) return 1;       // prompt block
    {
        // composed content block: Wcm[d][j] = sum_k Wc[d][k] * Wcp[k][j]  (Wc = merge columns [2C, 2C+D))
        float* wcpT = ws.alloc_n<float>((long)Dc * D, st);   // Wcp^T [Dc][D]
        if (!wcpT) return 1;
        if (pack_f32_launch(wcp->data, wcpT, D, 1, Dc, Dc, 0, 1, 1, 0, D, nullptr, st)) return 1;
        if (small_linear_launch(wm->data + 2 * C, Kin, wcpT, D, nullptr, w_stat + C32, Ks, D, Dc, D, KG_ACT_NONE, st)) return 1;
        // constant: Wc * b_cp + b_merge
        if (small_linear_launch(bcp->data, D, wm->data + 2 * C, Kin, bm->data, stat_const, D, 1, D, D, KG_ACT_NONE, st)) return 1;
    }
    w_style_merge = nullptr;
    if (style_in_merge) {
        w_style_merge = wts.alloc_n<float>((long)D * S, st);
        if (!w_style_merge) return 1;
        if (pack_f32_launch(wm->data + 2 * C + D, w_style_merge, D, 1, S, Kin, 0, 1, S, 0, 1, nullptr, st)) return 1;
    }
    style_in_w = style_in_b = nullptr;
    if (v2 || cfg.style_as_token) {
        GETW(sw, "style_in.weight", D, S);
        GETW(sb, "style_in.bias", D);
        style_in_w = copy_vec(wts, sw->data, (long)D * S, st);
        style_in_b = copy_vec(wts, sb->data, D, st);
        if (!style_in_w || !style_in_b) return 1;
    }
    // ---- heads
    if (!wavenet) {
        GETW(a, "final_mlp.0.weight", D, D);
        GETW(b, "final_mlp.0.bias", D);
        GETW(c, "final_mlp.2.weight", C, D);
        GETW(d, "final_mlp.2.bias", C);
        head_wa = new16(D, D);
        head_w2 = new16(C, D);
        if (!head_wa || !head_w2) return 1;
        if (pack_block(0, a->data, D, 0, D, D, head_wa, D, 0, 0, 1, nullptr, st)) return 1;
        if (pack_block(0, c->data, D, 0, C, D, head_w2, D, 0, 0, 1, nullptr, st)) return 1;
        head_b0 = copy_vec(wts, b->data, D, st);
        head_b2 = copy_vec(wts, d->data, C, st);
        if (!head_b0 || !head_b2) return 1;
    } else {
        if (put_temb("t_embedder2", W, &te2_w0, &te2_b0, &te2_w2, &te2_b2, &te2_freqs)) return 1;
        if (cfg.long_skip_connection) {
            GETW(a, "skip_linear.weight", D, D + C);
            GETW(b, "skip_linear.bias", D);
            w_longskip = new16(D, D + C16);
            if (!w_longskip) return 1;
            if (pack_block(0, a->data, D + C, 0, D, D + C, w_longskip, D + C16, 0, 0, 1, nullptr, st)) return 1;
            b_longskip = copy_vec(wts, b->data, D, st);
            if (!b_longskip) return 1;
        }
        GETW(c1w, "conv1.weight", W, D);
        GETW(c1b, "conv1.bias", W);
        GETW(rpw, "res_projection.weight", W, D);
        GETW(rpb, "res_projection.bias", W);
        GETW(c2w, "conv2.weight", C, W, 1);
        GETW(c2b, "conv2.bias", C);
        w_conv1 = new16(W, D);
        w_resproj = new16(W, D);
        w_conv2 = new16(C, W);
        if (!w_conv1 || !w_resproj || !w_conv2) return 1;
        if (pack_block(0, c1w->data, D, 0, W, D, w_conv1, D, 0, 0, 1, nullptr, st)) return 1;
        if (pack_block(0, rpw->data, D, 0, W, D, w_resproj, D, 0, 0, 1, nullptr, st)) return 1;
        if (pack_block(0, c2w->data, W, 0, C, W, w_conv2, W, 0, 0, 1, nullptr, st)) return 1;
        b_conv1 = copy_vec(wts, c1b->data, W, st);
        b_resproj = copy_vec(wts, rpb->data, W, st);
        b_conv2 = copy_vec(wts, c2b->data, C, st);
        if (!b_conv1 || !b_resproj || !b_conv2) return 1;
        // FinalLayer
        {
            WeightSrc fl;
            if (resolve_weight(sd, "final_layer.linear", wts, st, &fl)) return 1;
            if (require_shape(fl.desc, "final_layer.linear.weight", {W, W})) return 1;
            GETW(flb, "final_layer.linear.bias", W);
            GETW(mw, "final_layer.adaLN_modulation.1.weight", 2 * W, W);
            GETW(mb, "final_layer.adaLN_modulation.1.bias", 2 * W);
            w_fl = new16(W, W);
            if (!w_fl) return 1;
            if (pack_block(0, fl.v, W, 0, W, W, w_fl, W, 0, 0, 1, fl.scale, st)) return 1;
            b_fl = copy_vec(wts, flb->data, W, st);
            fl_mod_w = copy_vec(wts, mw->data, 2L * W * W, st);
            fl_mod_b = copy_vec(wts, mb->data, 2 * W, st);
            if (!b_fl || !fl_mod_w || !fl_mod_b) return 1;
        }
        // WaveNet: GLU-interleaved packing (row 2j = tanh half j, row 2j+1 = sigmoid half j)
        {
            WeightSrc cl;
            if (resolve_weight(sd, "wavenet.cond_layer.conv.conv", wts, st, &cl)) return 1;
            if (require_shape(cl.desc, "wavenet.cond_layer.conv.conv.weight", {2L * W * NL, W, 1})) return 1;
            GETW(clb, "wavenet.cond_layer.conv.conv.bias", 2L * W * NL);
            wn_cond_w = wts.alloc_n<float>(2L * W * NL * W, st);
            wn_cond_b = wts.alloc_n<float>(2L * W * NL, st);
            float* tmp_b = ws.alloc_n<float>(2L * W * NL, st);
            if (!wn_cond_w || !wn_cond_b || !tmp_b) return 1;
            wn_in.resize(NL);
            wn_res.assign(NL, nullptr);
            wn_skip.resize(NL);
            wn_res_b.assign(NL, nullptr);
            wn_skip_b.resize(NL);
            for (int i = 0; i < NL; ++i) {
                const std::string li = std::to_string(i);
                for (int half = 0; half < 2; ++half) {
                    const long src_row0 = 2L * W * i + (long)half * W;
                    // weight rows (with weight-norm scale of the SOURCE row)
                    if (pack_f32_launch(cl.v + src_row0 * W, wn_cond_w + (2L * W * i + half) * W, W, 1, W, W, 0, 1, 2L * W, 0, 1,
                                        cl.scale ? cl.scale + src_row0 : nullptr, st)) return 1;
                    if (pack_f32_launch(clb->data + src_row0, tmp_b + 2L * W * i + half, W, 1, 1, 1, 0, 0, 2, 0, 0, nullptr, st)) return 1;
                }
                WeightSrc in;
                if (resolve_weight(sd, "wavenet.in_layers." + li + ".conv.conv", wts, st, &in)) return 1;
                if (require_shape(in.desc, "wavenet.in_layers." + li + ".conv.conv.weight", {2L * W, W, WK})) return 1;
                GETW(inb, "wavenet.in_layers." + li + ".conv.conv.bias", 2L * W);
                wn_in[i] = new16(2 * W, (long)WK * W);
                float* inb_perm = ws.alloc_n<float>(2L * W, st);
                if (!wn_in[i] || !inb_perm) return 1;
                for (int half = 0; half < 2; ++half) {
                    // src [2W][W][k] -> dst row (2j+half), column tap*W + ci
                    if (pack_f16_launch(in.v + (long)half * W * W * WK, wn_in[i] + (long)half * WK * W, W, WK, W,
                                        (long)W * WK, 1, WK, 2L * WK * W, W, 1, in.scale ? in.scale + (long)half * W : nullptr, st)) return 1;
                    if (pack_f32_launch(inb->data + (long)half * W, inb_perm + half, W, 1, 1, 1, 0, 0, 2, 0, 0, nullptr, st)) return 1;
                }
                // total per-layer additive vector = permuted(cond bias) + permuted(in bias)
                {
                    float* dst = wn_cond_b + 2L * W * i;
                    if (add_rowvec_launch(dst, tmp_b + 2L * W * i, 0, inb_perm, 1, 2 * W, st)) return 1;
                }
                WeightSrc rs;
                const int rs_out = i < NL - 1 ? 2 * W : W;
                if (resolve_weight(sd, "wavenet.res_skip_layers." + li + ".conv.conv", wts, st, &rs)) return 1;
                if (require_shape(rs.desc, "wavenet.res_skip_layers." + li + ".conv.conv.weight", {rs_out, W, 1})) return 1;
                GETW(rsb, "wavenet.res_skip_layers." + li + ".conv.conv.bias", rs_out);
                const long skip_row0 = i < NL - 1 ? W : 0;
                if (i < NL - 1) {
                    wn_res[i] = new16(W, W);
                    if (!wn_res[i]) return 1;
                    if (pack_block(0, rs.v, W, 0, W, W, wn_res[i], W, 0, 0, 1, rs.scale, st)) return 1;
                    wn_res_b[i] = copy_vec(wts, rsb->data, W, st);
                    if (!wn_res_b[i]) return 1;
                }
                wn_skip[i] = new16(W, W);
                if (!wn_skip[i]) return 1;
                if (pack_block(0, rs.v + skip_row0 * W, W, 0, W, W, wn_skip[i], W, 0, 0, 1,
                               rs.scale ? rs.scale + skip_row0 : nullptr, st)) return 1;
                wn_skip_b[i] = copy_vec(wts, rsb->data + skip_row0, W, st);
                if (!wn_skip_b[i]) return 1;
            }
        }
    }
    if (wavenet) {
        // every layer's skip output is only ever summed (wavenet.py:158-163) and then added to res_projection(x_res)
        // (diffusion_transformer.py:531): out = [acts_0 | ... | acts_{NL-1} | x_res] [skip_0 | ... | skip_{NL-1} | res_proj]^T
        // is ONE tap-GEMM over NL + 1 source buffers instead of NL + 1 launches that each re-read and re-write a fp32 plane.
        const long Kt = (long)NL * W + D;
        wn_tail = new16(W, Kt);
        b_tail = wts.alloc_n<float>(W, st);
        if (!wn_tail || !b_tail) return 1;
        for (int i = 0; i < NL; ++i)
            SVC_CHECK_HIP(hipMemcpy2DAsync(wn_tail + (long)i * W, Kt * 2, wn_skip[i], (size_t)W * 2, (size_t)W * 2, W, hipMemcpyDeviceToDevice, st));
        SVC_CHECK_HIP(hipMemcpy2DAsync(wn_tail + (long)NL * W, Kt * 2, w_resproj, (size_t)D * 2, (size_t)D * 2, W, hipMemcpyDeviceToDevice, st));
        if (add_rowvec_launch(b_tail, b_resproj, 0, wn_skip_b[0], 1, W, st)) return 1;
        for (int i = 1; i < NL; ++i)
            if (add_rowvec_launch(b_tail, b_tail, 0, wn_skip_b[i], 1, W, st)) return 1;
    }
    // ---- RoPE table (fp32; v2 rounds it to bf16: v2/dit_model.py:225-234)
    {
        std::vector<float> tab((size_t)ROPE_POS * 32 * 2);
        for (int i = 0; i < 32; ++i) {
            const float e = (float)(2 * i) / 64.0f;
            const float f = 1.0f / powf(10000.0f, e);
            for (int t = 0; t < ROPE_POS; ++t) {
                const float a = (float)t * f;
                float c = (float)cos((double)a), s = (float)sin((double)a);
                if (v2) { c = bf16_round(c); s = bf16_round(s); }
                tab[((size_t)t * 32 + i) * 2] = c;
                tab[((size_t)t * 32 + i) * 2 + 1] = s;
            }
        }
        rope = wts.alloc_n<float>(tab.size(), st);
        if (!rope) return 1;
        SVC_CHECK_HIP(hipMemcpyAsync(rope, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, st));
        SVC_CHECK_HIP(hipStreamSynchronize(st));
    }
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    ws.release();   // pack-time temporaries
    return 0;
}

// --------------------------------------------------------------------------------------------- workspace
int svc_dit::reserve(int n_streams, int B, int T, int n_steps, hipStream_t st) {
    if (n_streams <= cap_streams && B <= cap_B && T <= cap_T && n_steps <= cap_steps) {
        // geometry depends on T: recompute for the current call
        seq_rows = (int)round_up(T + npre, 8);
        vt_ld = (int)round_up(seq_rows, 64);
        return 0;
    }
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    drop_graphs();                // captured workspace pointers die with the workspace
    ws.release();
    cap_streams = std::max(cap_streams, n_streams);
    cap_B = std::max(cap_B, B);
    cap_T = std::max(cap_T, T);
    cap_steps = std::max(cap_steps, n_steps);
    const long rows_seq = round_up(cap_T + npre, 8);
    const long nseq = (long)cap_streams * cap_B;
    const long M = nseq * rows_seq;
    const long vtl = round_up(rows_seq, 64);
    x32 = ws.alloc_n<float>((long)cap_B * rows_seq * C16, st);
    x16 = ws.alloc_n<half_t>(M * C16, st);
    xin = ws.alloc_n<float>(M * D, st);
    st_term = ws.alloc_n<float>(M * D, st);
    v32 = ws.alloc_n<float>(M * C16, st);
    u_rowvec = ws.alloc_n<float>(nseq * D, st);
    tok_style = ws.alloc_n<float>(nseq * D, st);
    prompt32 = ws.alloc_n<float>((long)cap_B * rows_seq * C32, st);
    n16 = ws.alloc_n<half_t>(M * D, st);
    qk16 = ws.alloc_n<half_t>(M * 2 * D, st);
    vt = ws.alloc_n<half_t>(nseq * D * vtl, st);
    ao16 = ws.alloc_n<half_t>(M * D, st);
    ff16 = ws.alloc_n<half_t>(M * I, st);
    h16 = ws.alloc_n<half_t>(M * D, st);
    skip16.assign(emit.size(), nullptr);
    for (auto& p : skip16) {
        p = ws.alloc_n<half_t>(M * D, st);
        if (!p) return 1;
    }
    if (!x32 || !x16 || !xin || !st_term || !v32 || !u_rowvec || !tok_style || !prompt32 || !n16 || !qk16 || !vt ||
        !ao16 || !ff16 || !h16)
        return 1;
    hm16 = xr16 = wnx16 = acts16 = fl16 = fo16 = nullptr;
    wnx32 = flin32 = nullptr;
    if (!wavenet) {
        hm16 = ws.alloc_n<half_t>(M * D, st);
        if (!hm16) return 1;
    } else {
        xr16 = ws.alloc_n<half_t>(M * D, st);
        wnx32 = ws.alloc_n<float>(M * W, st);
        wnx16 = ws.alloc_n<half_t>(M * W, st);
        acts_stride = M * W;
        acts16 = ws.alloc_n<half_t>((long)NL * acts_stride, st);      // one gated-activation plane per layer (tail GEMM)
        flin32 = ws.alloc_n<float>(M * W, st);
        fl16 = ws.alloc_n<half_t>(M * W, st);
        fo16 = ws.alloc_n<half_t>(M * W, st);
        if (!xr16 || !wnx32 || !wnx16 || !acts16 || !flin32 || !fl16 || !fo16) return 1;
    }
    d_kvlen = ws.alloc_n<int>(nseq, st);
    d_convlen = ws.alloc_n<int>(nseq, st);
    d_convlen_w = ws.alloc_n<int>(nseq, st);
    d_plen = ws.alloc_n<int>(cap_B, st);
    const long Sx = cap_steps;
    d_tvals = ws.alloc_n<float>(Sx, st);
    d_tfeat = ws.alloc_n<float>(Sx * 256, st);
    d_th = ws.alloc_n<float>(Sx * std::max(D, W), st);
    d_t1 = ws.alloc_n<float>(Sx * D, st);
    d_t1s = ws.alloc_n<float>(Sx * D, st);
    d_t2 = ws.alloc_n<float>(Sx * std::max(W, 1), st);
    d_mod = ws.alloc_n<float>(Sx * mod_n, st);
    d_gcond = ws.alloc_n<float>(Sx * std::max(2L * W * NL, 1L), st);
    d_flmod = ws.alloc_n<float>(Sx * std::max(2 * W, 1), st);
    d_stylevec = ws.alloc_n<float>((long)cap_B * D, st);
    if (!d_kvlen || !d_convlen || !d_convlen_w || !d_plen || !d_tvals || !d_tfeat || !d_th || !d_t1 || !d_t1s || !d_t2 || !d_mod ||
        !d_gcond || !d_flmod || !d_stylevec)
        return 1;
    seq_rows = (int)round_up(T + npre, 8);
    vt_ld = (int)round_up(seq_rows, 64);
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
}

// --------------------------------------------------------------------------------------------- per-call tables
// Everything that depends on the timestep only (reference: TimestepEmbedder.forward :361-364,
// AdaptiveLayerNorm.project_layer :43-47, WN.cond_layer wavenet.py:142-143, FinalLayer.adaLN_modulation :402).
int svc_dit::tables(const std::vector<float>& tvals, hipStream_t st) {
    const int n = (int)tvals.size();
    {
        float* h = reinterpret_cast<float*>(staging.acquire(n * sizeof(float)));
        if (!h) return 1;
        memcpy(h, tvals.data(), n * sizeof(float));
        SVC_CHECK_HIP(hipMemcpyAsync(d_tvals, h, n * sizeof(float), hipMemcpyHostToDevice, st));
        if (staging.commit(st)) return 1;
    }
    if (timestep_feat_launch(d_tvals, te_freqs, d_tfeat, n, st)) return 1;
    if (small_linear_launch(d_tfeat, 256, te_w0, 256, te_b0, d_th, D, n, D, 256, KG_ACT_SILU, st)) return 1;
    if (small_linear_launch(d_th, D, te_w2, D, te_b2, d_t1, D, n, D, D, KG_ACT_NONE, st)) return 1;
    if (silu_launch(d_t1, d_t1s, (long)n * D, st)) return 1;
    // modulation stack: v1 projects t1 directly, v2 projects SiLU(t1)
    if (small_linear_launch(v2 ? d_t1s : d_t1, D, mod_w, D, mod_b, d_mod, mod_n, n, mod_n, D, KG_ACT_NONE, st)) return 1;
    if (wavenet) {
        if (small_linear_launch(d_tfeat, 256, te2_w0, 256, te2_b0, d_th, W, n, W, 256, KG_ACT_SILU, st)) return 1;
        if (small_linear_launch(d_th, W, te2_w2, W, te2_b2, d_t2, W, n, W, W, KG_ACT_NONE, st)) return 1;
        if (small_linear_launch(d_t2, W, wn_cond_w, W, wn_cond_b, d_gcond, 2L * W * NL, n, 2 * W * NL, W, KG_ACT_NONE, st)) return 1;
        if (small_linear_launch(d_t1s, D, fl_mod_w, W, fl_mod_b, d_flmod, 2 * W, n, 2 * W, W, KG_ACT_NONE, st)) return 1;
    }
    return 0;
}


// --------------------------------------------------------------------------------------------- estimator body
namespace {
KGemmParams gemm_base(int M, int N, int Lout) {
    KGemmParams p;
    memset(&p, 0, sizeof(p));
    p.M = M;
    p.N = N;
    p.Lout = Lout;
    p.a_seq_rows = Lout;
    p.c_seq_rows = Lout;
    p.a_stride = 1;
    p.a_len = Lout;
    p.n_taps = 1;
    p.vec_ok = 1;
    return p;
}
}  // namespace

int svc_dit::body(int n_streams, int B, int T, int step, hipStream_t st) {
    const int nseq = n_streams * B;
    const int M = nseq * seq_rows;
    const float* mod = d_mod + (long)step * mod_n;
    // 1. prefix tokens + zeroed pad rows
    if (prefix_rows_launch(xin, nseq, seq_rows, D, npre, T + npre, d_t1 + (long)step * D, tok_style,
                           cfg.time_as_token ? 1 : 0, st)) return 1;
    // 2. x block of the merge linear + static term
    {
        KGemmParams p = gemm_base(nseq * T, D, T);
        p.a_ptr[0] = x16; p.a_ld[0] = C16; p.a_ktiles[0] = C16 / 64;
        p.a_seq_rows = seq_rows; p.a_len = T;
        p.w = w_merge_x; p.ldw = C16;
        p.c_seq_rows = seq_rows; p.c_off = npre;
        p.c32 = xin; p.ldc32 = D;
        p.res = st_term; p.ldres = D;
        if (kgemm_launch(p, 0, KG_EPI_STORE, st)) return 1;
    }
    if (fused_ok && (long)M >= fused_min_rows) {
        if (body_fused(n_streams, B, T, step, st)) return 1;
        return head(n_streams, B, T, step, st);
    }
    // The head and the LAST transformer layer's query-side work are evaluated on rows >= win0 only: the sampler discards the velocity on prompt frames
    // (flow_matching.py:105-110 zeroes x[..., :prompt_len] after every step), and every head op is row-local except the
    // WaveNet convs, whose receptive field is covered by the halo run_group leaves in front of the shortest prompt.
    const int Lw = seq_rows - win0;
    auto gemm_win = [&](int N) {
        KGemmParams p = gemm_base(nseq * Lw, N, Lw);
        p.a_seq_rows = seq_rows; p.c_seq_rows = seq_rows;
        p.a_off = win0; p.c_off = win0;
        return p;
    };
    size_t emit_i = 0;
    std::vector<int> skip_stack;
    for (int i = 0; i < L; ++i) {
        const bool tail_only = i == L - 1 && win0 > 0;      // K / V still cover every row; queries, wo and the FFN do not

        const Layer& ly = layers[i];
        const float* lmod = adaptive_blocks ? mod + (long)i * mod_layer_n : nullptr;
        const bool is_recv = ly.wskip != nullptr;
        if (is_recv) {
            const int src = skip_stack.back();
            skip_stack.pop_back();
            KGemmParams p = gemm_base(M, D, seq_rows);
            p.n_taps = 2;
            p.a_ptr[0] = h16; p.a_ld[0] = D; p.a_ktiles[0] = D / 64;
            p.a_ptr[1] = skip16[src]; p.a_ld[1] = D; p.a_ktiles[1] = D / 64;
            p.w = ly.wskip; p.ldw = 2 * D;
            p.bias = ly.bskip;
            p.c32 = xin; p.ldc32 = D;
            if (kgemm_launch(p, 0, KG_EPI_STORE, st)) return 1;
        }
        // attention norm
        const float *w_a = nullptr, *b_a = nullptr, *gate_a = nullptr, *w_f = nullptr, *b_f = nullptr, *gate_f = nullptr;
        if (lmod) {
            if (v2) {   // (shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp): v2/dit_model.py:33
                b_a = lmod; w_a = lmod + D; gate_a = lmod + 2 * D;
                b_f = lmod + 3 * D; w_f = lmod + 4 * D; gate_f = lmod + 5 * D;
            } else {    // (weight, bias) split: diffusion_transformer.py:43-48
                w_a = lmod; b_a = lmod + D; w_f = lmod + 2 * D; b_f = lmod + 3 * D;
            }
        }
        if (rmsnorm_mod_launch(xin, D, n16, D, ly.g_attn, w_a, b_a, 0, v2 ? 1 : 0, M, D, seq_rows, 1e-5f, st)) return 1;
        {
            KGemmParams p = gemm_base(M, 3 * D, seq_rows);
            p.a_ptr[0] = n16; p.a_ld[0] = D; p.a_ktiles[0] = D / 64;
            p.w = ly.wqkv; p.ldw = D;
            p.c16 = qk16; p.ldc16 = 2 * D;
            p.rope = rope; p.rope_D = D; p.q_scale = 0.125f * 1.4426950408889634f;
            p.vt = vt; p.vt_seq_stride = (long)D * vt_ld; p.vt_ld = vt_ld; p.vt_mode = attention_vt_mode(nseq, H, seq_rows);
            if (kgemm_launch(p, 0, KG_EPI_QKV_ROPE, st)) return 1;
        }
        {
            AttnParams a;
            memset(&a, 0, sizeof(a));
            a.q = qk16; a.k = qk16 + D; a.ld_qk = 2 * D;
            a.vt = vt; a.vt_seq_stride = (long)D * vt_ld; a.vt_ld = vt_ld; a.vt_perm = attention_vt_mode(nseq, H, seq_rows);
            a.out = ao16; a.ld_out = D;
            a.n_seq = nseq; a.H = H; a.seq_rows = seq_rows; a.Tq = seq_rows;
            a.q_start = tail_only ? win0 : 0;
            a.kv_len = d_kvlen;
            if (attention_launch(a, st)) return 1;
        }
        {
            KGemmParams p = tail_only ? gemm_win(D) : gemm_base(M, D, seq_rows);
            p.a_ptr[0] = ao16; p.a_ld[0] = D; p.a_ktiles[0] = D / 64;
            p.w = ly.wo; p.ldw = D;
            p.gate = gate_a; p.ld_gate = 0;
            p.res = xin; p.ldres = D;
            p.c32 = xin; p.ldc32 = D;
            if (kgemm_launch(p, 0, KG_EPI_STORE, st)) return 1;
        }
        if (rmsnorm_mod_launch(xin, D, n16, D, ly.g_ffn, w_f, b_f, 0, v2 ? 1 : 0, M, D, seq_rows, 1e-5f, st)) return 1;
        {
            KGemmParams p = tail_only ? gemm_win(2 * I) : gemm_base(M, 2 * I, seq_rows);
            p.a_ptr[0] = n16; p.a_ld[0] = D; p.a_ktiles[0] = D / 64;
            p.w = ly.w13; p.ldw = D;
            p.c16 = ff16; p.ldc16 = I;
            if (kgemm_launch(p, 0, KG_EPI_SWIGLU, st)) return 1;
        }
        {
            KGemmParams p = tail_only ? gemm_win(D) : gemm_base(M, D, seq_rows);
            p.a_ptr[0] = ff16; p.a_ld[0] = I; p.a_ktiles[0] = I / 64;
            p.w = ly.w2; p.ldw = I;
            p.gate = gate_f; p.ld_gate = 0;
            p.res = xin; p.ldres = D;
            p.c32 = xin; p.ldc32 = D;
            const bool is_emit = std::find(emit.begin(), emit.end(), i) != emit.end();
            const bool next_recv = i + 1 < L && layers[i + 1].wskip != nullptr;
            if (is_emit) {
                p.c16 = skip16[emit_i]; p.ldc16 = D;
                skip_stack.push_back((int)emit_i);
                ++emit_i;
            } else if (next_recv) {
                p.c16 = h16; p.ldc16 = D;
            }
            if (kgemm_launch(p, 0, KG_EPI_STORE, st)) return 1;
        }
    }
    // final adaptive norm (always modulated: diffusion_transformer.py:142 ; v2 chunk order (scale, shift))
    {
        const float* fm = mod + (long)L * mod_layer_n;
        const float* w = fm;
        const float* b = fm + D;
        if (rmsnorm_mod_launch(xin, D, n16, D, g_final, w, b, 0, v2 ? 1 : 0, M, D, seq_rows, 1e-5f, st)) return 1;
    }
    return head(n_streams, B, T, step, st);
}

// Output head on the rows >= win0 of n16 (the final-norm output): mlp head (tiny / base / v2) or the WaveNet head.
int svc_dit::head(int n_streams, int B, int T, int step, hipStream_t st) {
    const int nseq = n_streams * B;
    const int M = nseq * seq_rows;
    const int Lw = seq_rows - win0;
    auto gemm_win = [&](int N) {
        KGemmParams p = gemm_base(nseq * Lw, N, Lw);
        p.a_seq_rows = seq_rows; p.c_seq_rows = seq_rows;
        p.a_off = win0; p.c_off = win0;
        return p;
    };
    if (!wavenet) {
        {
            KGemmParams p = gemm_win(D);
            p.a_ptr[0] = n16; p.a_ld[0] = D; p.a_ktiles[0] = D / 64;
            p.w = head_wa; p.ldw = D; p.bias = head_b0; p.act = KG_ACT_SILU;
            p.c16 = hm16; p.ldc16 = D;
            if (kgemm_launch(p, 0, KG_EPI_STORE, st)) return 1;
        }
        {
            KGemmParams p = gemm_win(C);
            p.a_ptr[0] = hm16; p.a_ld[0] = D; p.a_ktiles[0] = D / 64;
            p.w = head_w2; p.ldw = D; p.bias = head_b2;
            p.c32 = v32; p.ldc32 = C16;
            if (kgemm_launch(p, 0, KG_EPI_STORE, st)) return 1;
        }
        return 0;
    }
    // ---- WaveNet head (reference: diffusion_transformer.py:524-533, wavenet.py:138-166)
    const half_t* xres16 = n16;
    if (cfg.long_skip_connection) {
        KGemmParams p = gemm_win(D);
        p.n_taps = 2;
        p.a_ptr[0] = n16; p.a_ld[0] = D; p.a_ktiles[0] = D / 64;
        p.a_ptr[1] = x16; p.a_ld[1] = C16; p.a_ktiles[1] = C16 / 64;
        p.w = w_longskip; p.ldw = D + C16; p.bias = b_longskip;
        p.c16 = xr16; p.ldc16 = D;
        if (kgemm_launch(p, 0, KG_EPI_STORE, st)) return 1;
        xres16 = xr16;
    }
    {
        KGemmParams p = gemm_win(W);
        p.a_ptr[0] = xres16; p.a_ld[0] = D; p.a_ktiles[0] = D / 64;
        p.w = w_conv1; p.ldw = D; p.bias = b_conv1;
        p.c32 = wnx32; p.ldc32 = W; p.c16 = wnx16; p.ldc16 = W;
        if (kgemm_launch(p, 0, KG_EPI_STORE, st)) return 1;
    }
    for (int i = 0; i < NL; ++i) {
        int dil = 1;
        for (int j = 0; j < i; ++j) dil *= cfg.wn_dilation_rate;
        {
            // in_layer: k-tap conv with reflect padding (encodec.py:217-227), gated tanh*sigmoid epilogue
            KGemmParams p = gemm_win(2 * W);
            p.n_taps = WK;
            const int total = (WK - 1) * dil;
            const int left = total - total / 2;
            for (int t = 0; t < WK; ++t) {
                p.a_ptr[t] = wnx16; p.a_ld[t] = W; p.a_ktiles[t] = W / 64; p.a_shift[t] = t * dil - left;
            }
            p.pad_mode = KG_PAD_REFLECT;
            p.reflect_min = std::max(left, total - left) + 1;      // pad1d's small-input guard (encodec.py:103-111)
            p.seq_len = d_convlen_w;
            p.w = wn_in[i]; p.ldw = (long)WK * W;
            p.rowvec = d_gcond + (long)step * 2 * W * NL + 2L * W * i; p.ld_rowvec = 0;
            p.c16 = acts16 + (long)i * acts_stride; p.ldc16 = W;
            if (kgemm_launch(p, 0, KG_EPI_TANHSIG, st)) return 1;
        }
        if (i < NL - 1) {
            KGemmParams p = gemm_win(W);
            p.a_ptr[0] = acts16 + (long)i * acts_stride; p.a_ld[0] = W; p.a_ktiles[0] = W / 64;
            p.w = wn_res[i]; p.ldw = W; p.bias = wn_res_b[i];
            p.res = wnx32; p.ldres = W;
            p.c32 = wnx32; p.ldc32 = W; p.c16 = wnx16; p.ldc16 = W;
            if (kgemm_launch(p, 0, KG_EPI_STORE, st)) return 1;
        }
    }
    {   // sum of the NL skip outputs + res_projection(x_res) in one launch (see wn_tail)
        KGemmParams p = gemm_win(W);
        p.n_taps = NL + 1;
        for (int i = 0; i < NL; ++i) {
            p.a_ptr[i] = acts16 + (long)i * acts_stride; p.a_ld[i] = W; p.a_ktiles[i] = W / 64;
        }
        p.a_ptr[NL] = xres16; p.a_ld[NL] = D; p.a_ktiles[NL] = D / 64;
        p.w = wn_tail; p.ldw = (long)NL * W + D; p.bias = b_tail;
        p.c32 = flin32; p.ldc32 = W;
        if (kgemm_launch(p, 0, KG_EPI_STORE, st)) return 1;
    }
    {
        const float* fm = d_flmod + (long)step * 2 * W;      // (shift, scale): diffusion_transformer.py:402
        if (layernorm_mod_launch(flin32, W, fl16, W, fm + W, fm, M, W, 1e-6f, st)) return 1;
    }
    {
        KGemmParams p = gemm_win(W);
        p.a_ptr[0] = fl16; p.a_ld[0] = W; p.a_ktiles[0] = W / 64;
        p.w = w_fl; p.ldw = W; p.bias = b_fl;
        p.c16 = fo16; p.ldc16 = W;
        if (kgemm_launch(p, 0, KG_EPI_STORE, st)) return 1;
    }
    {
        KGemmParams p = gemm_win(C);
        p.a_ptr[0] = fo16; p.a_ld[0] = W; p.a_ktiles[0] = W / 64;
        p.w = w_conv2; p.ldw = W; p.bias = b_conv2;
        p.c32 = v32; p.ldc32 = C16;
        if (kgemm_launch(p, 0, KG_EPI_STORE, st)) return 1;
    }
    return 0;
}

// Transformer stack on the fused row-panel kernel (fused.hip): one launch per layer besides attention.  Layer i's kernel
// also runs the next layer's skip linear, attention norm, QKV projection and RoPE; the last one ends with the final norm.
int svc_dit::body_fused(int n_streams, int B, int T, int step, hipStream_t st) {
    const int nseq = n_streams * B;
    const int M = nseq * seq_rows;
    const float* mod = d_mod + (long)step * mod_n;
    auto mods = [&](int i, const float** w_a, const float** b_a, const float** gate_a, const float** w_f, const float** b_f,
                    const float** gate_f) {
        *w_a = *b_a = *gate_a = *w_f = *b_f = *gate_f = nullptr;
        if (!adaptive_blocks) return;
        const float* lmod = mod + (long)i * mod_layer_n;
        if (v2) {
            *b_a = lmod; *w_a = lmod + D; *gate_a = lmod + 2 * D;
            *b_f = lmod + 3 * D; *w_f = lmod + 4 * D; *gate_f = lmod + 5 * D;
        } else {
            *w_a = lmod; *b_a = lmod + D; *w_f = lmod + 2 * D; *b_f = lmod + 3 * D;
        }
    };
    auto base = [&]() {
        PanelParams p;
        memset(&p, 0, sizeof(p));
        p.M = M; p.Lout = seq_rows; p.seq_rows = seq_rows; p.row_off = 0;
        p.eps = 1e-5f; p.add_one = v2 ? 1 : 0; p.I = I; p.x = xin;
        return p;
    };
    auto set_qkv = [&](PanelParams& p, int layer) {
        const float *w_a, *b_a, *ga, *w_f, *b_f, *gf;
        mods(layer, &w_a, &b_a, &ga, &w_f, &b_f, &gf);
        p.do_qkv = 1;
        p.g_attn = layers[layer].g_attn; p.w_a = w_a; p.b_a = b_a;
        p.rope = rope; p.q_scale = 0.125f * 1.4426950408889634f;
        p.qk = qk16; p.vt = vt; p.vt_seq_stride = (long)D * vt_ld; p.vt_ld = vt_ld; p.vt_mode = attention_vt_mode(nseq, H, seq_rows);
    };
    {
        PanelParams p = base();
        p.wstream = stream_pre; p.n_slots = stream_pre_slots;
        set_qkv(p, 0);
        if (fused_panel_launch(p, D, v2, st)) return 1;
    }
    size_t emit_i = 0;
    std::vector<int> skip_stack;
    for (int i = 0; i < L; ++i) {
        const bool last = i == L - 1;
        const bool tail_only = last && win0 > 0;
        {
            AttnParams a;
            memset(&a, 0, sizeof(a));
            a.q = qk16; a.k = qk16 + D; a.ld_qk = 2 * D;
            a.vt = vt; a.vt_seq_stride = (long)D * vt_ld; a.vt_ld = vt_ld; a.vt_perm = attention_vt_mode(nseq, H, seq_rows);
            a.out = ao16; a.ld_out = D;
            a.n_seq = nseq; a.H = H; a.seq_rows = seq_rows; a.Tq = seq_rows;
            a.q_start = tail_only ? win0 : 0;
            a.kv_len = d_kvlen;
            if (attention_launch(a, st)) return 1;
        }
        PanelParams p = base();
        const float *w_a, *b_a, *ga, *w_f, *b_f, *gf;
        mods(i, &w_a, &b_a, &ga, &w_f, &b_f, &gf);
        p.wstream = layers[i].stream; p.n_slots = layers[i].stream_slots;
        p.do_post = 1; p.ao = ao16;
        p.gate_a = ga; p.gate_f = gf; p.g_ffn = layers[i].g_ffn; p.w_f = w_f; p.b_f = b_f;
        if (std::find(emit.begin(), emit.end(), i) != emit.end()) {
            p.c16 = skip16[emit_i];
            skip_stack.push_back((int)emit_i);
            ++emit_i;
        }
        if (!last) {
            if (layers[i + 1].wskip) {
                p.do_skip = 1;
                p.skip_in = skip16[skip_stack.back()];
                skip_stack.pop_back();
                p.bskip = layers[i + 1].bskip;
            }
            set_qkv(p, i + 1);
        } else {
            const float* fm = mod + (long)L * mod_layer_n;
            p.do_final = 1; p.g_fin = g_final; p.w_fin = fm; p.b_fin = fm + D; p.n16 = n16;
            if (tail_only) { p.Lout = seq_rows - win0; p.row_off = win0; p.M = nseq * p.Lout; }
        }
        if (fused_panel_launch(p, D, v2, st)) return 1;
    }
    return 0;
}

// --------------------------------------------------------------------------------------------- statics
// Per-utterance static part of the merge linear for every stream; flags = {use prompt, use style, use content}.
// Needs prompt32 (token-major prompt, zero beyond each prompt) to be filled already.
int svc_dit::statics(int n_streams, const int (*flags)[3], int B, int T, const float* mu, const float* style_dev,
                     hipStream_t st) {
    const long Ks = C32 + Dc;
    if (w_style_merge)
        if (small_linear_launch(style_dev, S, w_style_merge, S, nullptr, d_stylevec, D, B, D, S, KG_ACT_NONE, st)) return 1;
    for (int j = 0; j < n_streams; ++j) {
        const bool use_p = flags[j][0], use_s = flags[j][1], use_m = flags[j][2];
        float* u = u_rowvec + (long)j * B * D;
        if (add_rowvec_launch(u, (use_s && w_style_merge) ? d_stylevec : nullptr, D, stat_const, B, D, st)) return 1;
        float* dst = st_term + (long)j * B * seq_rows * D;
        {
            KGemmParams p = gemm_base(B * T, D, T);
            p.a_ptr[0] = mu; p.a_ld[0] = Dc; p.a_ktiles[0] = use_m ? Dc / 32 : 0;
            p.a_seq_rows = T; p.a_len = T;
            p.w = w_stat + C32; p.ldw = Ks;
            p.c_seq_rows = seq_rows; p.c_off = npre;
            p.c32 = dst; p.ldc32 = D;
            p.rowvec = u; p.ld_rowvec = D;
            if (kgemm_launch(p, 1, KG_EPI_STORE, st)) return 1;
        }
        if (use_p) {
            KGemmParams p = gemm_base(B * T, D, T);
            p.a_ptr[0] = prompt32; p.a_ld[0] = C32; p.a_ktiles[0] = C32 / 32;
            p.a_seq_rows = seq_rows; p.a_len = T;
            p.w = w_stat; p.ldw = Ks;
            p.c_seq_rows = seq_rows; p.c_off = npre;
            p.c32 = dst; p.ldc32 = D;
            p.res = dst; p.ldres = D;
            if (kgemm_launch(p, 1, KG_EPI_STORE, st)) return 1;
        }
        if (style_in_w) {
            float* tk = tok_style + (long)j * B * D;
            if (use_s) {
                if (small_linear_launch(style_dev, S, style_in_w, S, style_in_b, tk, D, B, D, S, KG_ACT_NONE, st)) return 1;
            } else {
                if (add_rowvec_launch(tk, nullptr, 0, style_in_b, B, D, st)) return 1;
            }
        }
    }
    return 0;
}

namespace {
// x[b][t][:] = 0 for t < prompt_len[b]; refresh the fp16 copies of every stream
__global__ void init_state_kernel(float* __restrict__ x, long ldx, half_t* __restrict__ x16, long ldx16, int rows, int B,
                                  int T, int C, const int* __restrict__ prompt_len, int n_copies, long copy_stride) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * T * C) return;
    const int c = (int)(i % C);
    const long bt = i / C;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const long o = ((long)b * rows + t) * ldx + c;
    float v = x[o];
    if (t < prompt_len[b]) v = 0.f;
    x[o] = v;
    for (int k = 0; k < n_copies; ++k) x16[k * copy_stride + ((long)b * rows + t) * ldx16 + c] = (half_t)v;
}

__global__ void euler_kernel(float* __restrict__ x, long ldx, half_t* __restrict__ x16, long ldx16, int rows,
                             const float* __restrict__ v, long ldv, long v_stream_stride, int v_off, int B, int T, int C,
                             const int* __restrict__ prompt_len, float dt, float c0, float ca, float cb, int stream_a,
                             int stream_b, int n_copies, long copy_stride) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * T * C) return;
    const int c = (int)(i % C);
    const long bt = i / C;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const long vo = ((long)b * rows + v_off + t) * ldv + c;
    float d = c0 * v[vo];
    if (stream_a >= 0) d -= ca * v[vo + stream_a * v_stream_stride];
    if (stream_b >= 0) d -= cb * v[vo + stream_b * v_stream_stride];
    const long xo = ((long)b * rows + t) * ldx + c;
    float nx = x[xo] + dt * d;
    if (t < prompt_len[b]) nx = 0.f;
    x[xo] = nx;
    for (int k = 0; k < n_copies; ++k) x16[k * copy_stride + ((long)b * rows + t) * ldx16 + c] = (half_t)nx;
}

// prompt (B, C, P) -> token-major [B][rows][ld], zero beyond each utterance's prompt length
__global__ void prompt_rows_kernel(const float* __restrict__ src, int C, int P, float* __restrict__ dst, long ld, int rows,
                                   int T, const int* __restrict__ prompt_len, int B) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * T * C) return;
    const int t = (int)(i % T);
    const long bc = i / T;
    const int c = (int)(bc % C), b = (int)(bc / C);
    const float v = (t < prompt_len[b] && t < P) ? src[((long)b * C + c) * P + t] : 0.f;
    dst[((long)b * rows + t) * ld + c] = v;
}
}  // namespace

int svc_dit::run_group(const svc_cfm_args_t* a, int b0, int nb, int n_streams, const int (*flags)[3],
                       const std::vector<float>& tvals, const std::vector<float>& dts, float c0, float ca, float cb,
                       int stream_a, int stream_b, hipStream_t st) {
    const int T = a->T, P = a->P;
    const int n_steps = (int)tvals.size();
    if (reserve(n_streams, nb, T, n_steps, st)) return 1;
    // lengths
    std::vector<int> kv(n_streams * nb), cv(n_streams * nb), pl(nb);
    for (int b = 0; b < nb; ++b) {
        const int len = a->x_lens ? (int)a->x_lens[b0 + b] : T;
        const int plen = a->prompt_lens ? (int)a->prompt_lens[b0 + b] : P;
        SVC_REQUIRE(len >= 1 && len <= T, "x_lens out of range");
        SVC_REQUIRE(plen >= 0 && plen <= P && plen <= len, "prompt_lens out of range");
        pl[b] = plen;
        for (int j = 0; j < n_streams; ++j) {
            kv[j * nb + b] = len + npre;
            cv[j * nb + b] = len;
        }
    }
    // head window: shortest prompt minus the receptive-field halo of the WaveNet convs, rounded down to 8 rows
    {
        int halo = 0;
        if (wavenet) {
            int dil = 1;
            for (int i = 0; i < NL; ++i) {
                const int total = (WK - 1) * dil;
                halo += total - total / 2;
                dil *= cfg.wn_dilation_rate;
            }
        }
        const int min_pl = *std::min_element(pl.begin(), pl.end());
        win0 = std::max(0, npre + min_pl - halo) & ~7;
    }
    std::vector<int> cvw(cv.size());
    for (size_t i = 0; i < cv.size(); ++i) cvw[i] = cv[i] - (win0 - npre > 0 ? win0 - npre : 0);
    {
        const size_t n = kv.size();
        int* h = reinterpret_cast<int*>(staging.acquire((3 * n + pl.size()) * sizeof(int)));
        if (!h) return 1;
        memcpy(h, kv.data(), n * 4); memcpy(h + n, cv.data(), n * 4); memcpy(h + 2 * n, cvw.data(), n * 4);
        memcpy(h + 3 * n, pl.data(), pl.size() * 4);
        SVC_CHECK_HIP(hipMemcpyAsync(d_kvlen, h, n * 4, hipMemcpyHostToDevice, st));
        SVC_CHECK_HIP(hipMemcpyAsync(d_convlen, h + n, n * 4, hipMemcpyHostToDevice, st));
        SVC_CHECK_HIP(hipMemcpyAsync(d_convlen_w, h + 2 * n, n * 4, hipMemcpyHostToDevice, st));
        SVC_CHECK_HIP(hipMemcpyAsync(d_plen, h + 3 * n, pl.size() * 4, hipMemcpyHostToDevice, st));
        if (staging.commit(st)) return 1;
    }
    if (tables(tvals, st)) return 1;

    const long copy_stride = (long)nb * seq_rows * C16;
    const long n_el = (long)nb * T * C;
    // ODE state from z (temperature-scaled), prompt frames zeroed (flow_matching.py:50,76-79)
    if (bct_to_btc_launch(a->z + (long)b0 * C * T, nb, C, T, x32, C16, nullptr, 0, seq_rows, T, a->temperature, st)) return 1;
    hipLaunchKernelGGL(init_state_kernel, dim3(cdiv(n_el, 256)), dim3(256), 0, st, x32, (long)C16, x16, (long)C16, seq_rows,
                       nb, T, C, d_plen, n_streams, copy_stride);
    SVC_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(prompt_rows_kernel, dim3(cdiv(n_el, 256)), dim3(256), 0, st, a->prompt + (long)b0 * C * P, C, P,
                       prompt32, (long)C32, seq_rows, T, d_plen, nb);
    SVC_CHECK_HIP(hipGetLastError());
    if (statics(n_streams, flags, nb, T, a->mu + (long)b0 * T * Dc, a->style + (long)b0 * S, st)) return 1;

    auto steps = [&](hipStream_t s_) -> int {
        for (int s = 0; s < n_steps; ++s) {
            if (body(n_streams, nb, T, s, s_)) return 1;
            hipLaunchKernelGGL(euler_kernel, dim3(cdiv(n_el, 256)), dim3(256), 0, s_, x32, (long)C16, x16, (long)C16, seq_rows,
                               v32, (long)C16, (long)nb * seq_rows * C16, npre, nb, T, C, d_plen, dts[s], c0, ca, cb,
                               stream_a, stream_b, n_streams, copy_stride);
            SVC_CHECK_HIP(hipGetLastError());
        }
        return 0;
    };
    // Everything the loop's kernel arguments depend on besides the (fixed) workspace pointers.
    static const int graph_env = [] { const char* e = getenv("SVC_DIT_GRAPH"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    bool done = false;
    if ((graph_env >= 0 ? graph_env : use_graphs) && !prof_enabled()) {
        std::vector<float> key = {(float)n_streams, (float)nb, (float)T, (float)n_steps, (float)win0, (float)seq_rows, c0, ca, cb,
                                  (float)stream_a, (float)stream_b, (float)fused_min_rows};
        key.insert(key.end(), dts.begin(), dts.end());
        StepGraph* g = nullptr;
        for (auto& c : graphs) if (c.key == key) { g = &c; break; }
        if (!g) {
            if (graphs.size() >= 8) { if (graphs.front().exec) (void)hipGraphExecDestroy(graphs.front().exec); graphs.erase(graphs.begin()); }
            graphs.push_back(StepGraph());
            g = &graphs.back();
            g->key = key;
        }
        if (g->seen++ >= 1 && !g->exec) {
            // captured on a private stream (the caller's may be the legacy default stream, which cannot capture); nothing
            // executes during capture, the instantiated graph is then launched on the caller's stream
            if (!cap_stream && hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); cap_stream = nullptr; }
            hipGraph_t gr = nullptr;
            if (cap_stream && hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int rc = steps(cap_stream);
                const hipError_t e = hipStreamEndCapture(cap_stream, &gr);
                if (rc == 0 && e == hipSuccess && gr && hipGraphInstantiate(&g->exec, gr, nullptr, nullptr, 0) != hipSuccess) g->exec = nullptr;
                if (gr) (void)hipGraphDestroy(gr);
                if (rc || e != hipSuccess) { (void)hipGetLastError(); g->exec = nullptr; if (rc) return 1; }
            } else {
                (void)hipGetLastError();
            }
        }
        if (g->exec) {
            SVC_CHECK_HIP(hipGraphLaunch(g->exec, st));
            done = true;
        }
    }
    if (!done && steps(st)) return 1;
    return btc_to_bct_launch(x32, C16, seq_rows, a->out + (long)b0 * C * T, nb, C, T, st);
}

// --------------------------------------------------------------------------------------------- C ABI
extern "C" {

int svc_dit_create(const svc_dit_config_t* cfg, const svc_tensor_desc_t* weights, int n_weights, void* stream,
                   svc_dit_t** out) {
    SVC_REQUIRE(cfg && weights && out, "null argument");
    SVC_REQUIRE(cfg->hidden_dim == cfg->num_heads * 64, "head_dim must be 64 (hidden_dim = 64 * num_heads)");
    SVC_REQUIRE(cfg->content_dim % 32 == 0, "content_dim must be a multiple of 32");
    SVC_REQUIRE(cfg->in_channels % 8 == 0 && cfg->in_channels <= 128, "in_channels must be a multiple of 8, <= 128");
    svc_dit* m = new svc_dit();
    m->cfg = *cfg;
    m->device = current_device();
    m->v2 = cfg->version == 2;
    m->D = cfg->hidden_dim; m->H = cfg->num_heads; m->L = cfg->depth; m->C = cfg->in_channels;
    m->Dc = cfg->content_dim; m->S = cfg->style_dim;
    {
        const int n_hidden = (int)(2 * (4 * m->D) / 3);
        m->I = (int)round_up(n_hidden, 256);
    }
    m->wavenet = cfg->final_layer_type == 1;
    m->W = m->wavenet ? cfg->wn_hidden_dim : 0;
    m->NL = m->wavenet ? cfg->wn_num_layers : 0;
    m->WK = m->wavenet ? cfg->wn_kernel_size : 0;
    m->npre = (cfg->time_as_token ? 1 : 0) + (cfg->style_as_token ? 1 : 0);
    m->C16 = (int)round_up(m->C, 64);
    m->C32 = (int)round_up(m->C, 32);
    m->adaptive_blocks = m->v2 || !cfg->time_as_token;
    if (m->wavenet) {
        if (m->W % 64 != 0 || m->W != m->D || m->WK * 1 > KG_MAX_TAPS || m->npre != 0) {
            set_error("wavenet head needs wn_hidden_dim == hidden_dim (FinalLayer modulates with t1), no prefix tokens");
            delete m;
            return 1;
        }
    }
    if (!m->v2 && cfg->uvit_skip_connection) {
        for (int i = 0; i < m->L; ++i) {
            if (i < m->L / 2) m->emit.push_back(i);
            if (i > m->L / 2) m->recv.push_back(i);
        }
    }
    hipStream_t st = (hipStream_t)stream;
    StateDict sd(weights, n_weights);
    if (m->pack(sd, st)) {
        delete m;
        return 1;
    }
    *out = m;
    return 0;
}

void svc_dit_destroy(svc_dit_t* m) {
    if (m) {
        m->drop_graphs();
        if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
    }
    delete m;
}

int svc_dit_set_microbatch(svc_dit_t* m, int utterances) {
    SVC_REQUIRE(m && utterances >= 0, "bad argument");
    m->microbatch = utterances;
    return 0;
}

int svc_dit_set_fused_min_rows(svc_dit_t* m, long rows) {
    SVC_REQUIRE(m, "null argument");
    m->fused_min_rows = rows < 0 ? 10240 : rows;
    return 0;
}

int svc_dit_fused_available(svc_dit_t* m) { return m && m->fused_ok ? 1 : 0; }

int svc_dit_set_graphs(svc_dit_t* m, int on) {
    SVC_REQUIRE(m, "null argument");
    m->use_graphs = on ? 1 : 0;
    if (!on) m->drop_graphs();
    return 0;
}

int svc_cfm_sample(svc_dit_t* m, const svc_cfm_args_t* a, void* stream) {
    SVC_REQUIRE(m && a, "null argument");
    SVC_REQUIRE(a->B >= 1 && a->T >= 1 && a->P >= 0 && a->P <= a->T && a->n_timesteps >= 1, "bad sampler shape");
    SVC_REQUIRE(a->T + m->npre <= ROPE_POS, "sequence longer than the RoPE table");
    SVC_REQUIRE(a->mu && (a->prompt || a->P == 0) && a->style && a->z && a->out, "null tensor");
    SVC_REQUIRE(current_device() == m->device, "this handle was created on another device (make it current before the call)");
    hipStream_t st = (hipStream_t)stream;
    const int N = a->n_timesteps;
    // ---- time grid (fp32, like the reference)
    std::vector<float> ts = linspace01(N + 1);
    std::vector<float> tvals(N), dts(N);
    if (!m->v2) {
        // flow_matching.py:70,83,105-106 : dt recomputed from t_span every step, t accumulated
        float t = ts[0];
        for (int s = 1; s <= N; ++s) {
            const float dt = ts[s] - ts[s - 1];
            tvals[s - 1] = t;
            dts[s - 1] = dt;
            t = t + dt;
        }
    } else {
        // v2/cfm.py:47-48,70,126-129 : cosine warp; dt follows the accumulated t
        for (auto& v : ts) v = v + (-1.0f) * (cosf((float)M_PI / 2.0f * v) - 1.0f + v);
        float t = ts[0], dt = ts[1] - ts[0];
        for (int s = 1; s <= N; ++s) {
            tvals[s - 1] = t;
            dts[s - 1] = dt;
            t = t + dt;
            if (s < N) dt = ts[s + 1] - t;
        }
    }
    // ---- guidance streams
    int flags[MAX_STREAMS][3] = {{1, 1, 1}, {0, 0, 0}, {0, 0, 0}};
    int n_streams = 1, sa = -1, sb = -1;
    float c0 = 1.f, ca = 0.f, cb = 0.f;
    const float ra = a->cfg_rate[0], rb = a->cfg_rate[1];
    if (!m->v2) {
        if (ra > 0.f) { n_streams = 2; sa = 1; c0 = 1.f + ra; ca = ra; }          // flow_matching.py:84-101
    } else if (a->random_voice) {                                                 // v2/cfm.py:77-87
        n_streams = 2; flags[0][0] = 0; flags[0][1] = 0; sa = 1; c0 = 1.f + ra; ca = ra;
    } else if (ra == 0.f && rb == 0.f) {
        n_streams = 1;
    } else if (ra == 0.f) {                                                       // similarity only: v2/cfm.py:90-101
        n_streams = 2; flags[1][2] = 1; sa = 1; c0 = 1.f + rb; ca = rb;
    } else if (rb == 0.f) {                                                       // v2/cfm.py:102-112
        n_streams = 2; sa = 1; c0 = 1.f + ra; ca = ra;
    } else {                                                                      // 3-way: v2/cfm.py:113-125
        n_streams = 3; flags[1][2] = 1; sa = 2; sb = 1; c0 = 1.f + ra + rb; ca = ra; cb = rb;
    }
    int mb = m->microbatch > 0 ? m->microbatch : 32;   // measured best on MI355X (tiny, B = 64): 8: 49.8k, 16: 51.7k, 32: 57.2k, 64: 55.7k frames/s
    for (int b0 = 0; b0 < a->B; b0 += mb) {
        const int nb = std::min(mb, a->B - b0);
        if (m->run_group(a, b0, nb, n_streams, flags, tvals, dts, c0, ca, cb, sa, sb, st)) return 1;
    }
    return 0;
}

int svc_dit_forward(svc_dit_t* m, int N, int T, const float* x, const float* prompt_x, const int64_t* x_lens, float t,
                    const float* style, const float* mu, float* out, void* stream) {
    SVC_REQUIRE(m && x && prompt_x && style && mu && out, "null argument");
    SVC_REQUIRE(N >= 1 && T >= 1 && T + m->npre <= ROPE_POS, "bad estimator shape");
    SVC_REQUIRE(current_device() == m->device, "this handle was created on another device (make it current before the call)");
    hipStream_t st = (hipStream_t)stream;
    const int flags[1][3] = {{1, 1, 1}};
    if (m->reserve(1, N, T, 1, st)) return 1;
    std::vector<int> kv(N), cv(N), pl(N);
    for (int b = 0; b < N; ++b) {
        const int len = x_lens ? (int)x_lens[b] : T;
        SVC_REQUIRE(len >= 1 && len <= T, "x_lens out of range");
        kv[b] = len + m->npre; cv[b] = len; pl[b] = T;
    }
    m->win0 = 0;                                    // the estimator seam returns every row
    {
        int* h = reinterpret_cast<int*>(m->staging.acquire(3 * (size_t)N * sizeof(int)));
        if (!h) return 1;
        memcpy(h, kv.data(), N * 4); memcpy(h + N, cv.data(), N * 4); memcpy(h + 2 * N, pl.data(), N * 4);
        SVC_CHECK_HIP(hipMemcpyAsync(m->d_kvlen, h, N * 4, hipMemcpyHostToDevice, st));
        SVC_CHECK_HIP(hipMemcpyAsync(m->d_convlen, h + N, N * 4, hipMemcpyHostToDevice, st));
        SVC_CHECK_HIP(hipMemcpyAsync(m->d_convlen_w, h + N, N * 4, hipMemcpyHostToDevice, st));
        SVC_CHECK_HIP(hipMemcpyAsync(m->d_plen, h + 2 * N, N * 4, hipMemcpyHostToDevice, st));
        if (m->staging.commit(st)) return 1;
    }
    std::vector<float> tv(1, t);
    if (m->tables(tv, st)) return 1;
    const int C = m->C, C16 = m->C16, C32 = m->C32, rows = m->seq_rows;
    if (bct_to_btc_launch(x, N, C, T, m->x32, C16, m->x16, C16, rows, T, 1.0f, st)) return 1;
    const long n_el = (long)N * T * C;
    hipLaunchKernelGGL(prompt_rows_kernel, dim3(cdiv(n_el, 256)), dim3(256), 0, st, prompt_x, C, T, m->prompt32, (long)C32,
                       rows, T, m->d_plen, N);
    SVC_CHECK_HIP(hipGetLastError());
    if (m->statics(1, flags, N, T, mu, style, st)) return 1;
    if (m->body(1, N, T, 0, st)) return 1;
    return btc_to_bct_launch(m->v32 + (long)m->npre * C16, C16, rows, out, N, C, T, st);
}

}  // extern "C"
