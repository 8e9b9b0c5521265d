// InterpolateRegulator (content features / tokens -> `mu`, the sampler's condition) -- SURVEY.md 8(f) row 1.
// Reference: modules/length_regulator.py:90-141 (v1), modules/v2/length_regulator.py:74-105 (v2).
//
//   x = content_in_proj(x) | embedding(tokens)            Linear on the fp32 MFMA | row gather
//   x = nearest-interpolate(x, size = ylens)              row gather, src = min(floor(t * Tin / Tout), Tin - 1)
//   x += f0_embedding(f0_to_coarse(f0)) interpolated | f0_mask          (f0_condition only)
//   n_convs x { Conv1d(C, C, 3, pad 1) -> GroupNorm(1, C) -> Mish }     tap-GEMM (fp32 MFMA) + 2 elementwise kernels
//   Conv1d(C, out, 1) | Identity ; rows >= ylens[b] zeroed
//
// Runs once per utterance, fp32 throughout (it feeds every sampler step).  Batch semantics as everywhere in this
// library: B independent B = 1 runs, each with its own input length and output length (the GroupNorm statistics of
// an utterance cover its own ylens[b] frames only).  Activations are channels-last [B][Tmax][Cpad].
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "conv_util.h"

using namespace svc;

namespace {

// f0 (Hz) -> coarse bin, reproducing f0_to_coarse + clamp (length_regulator.py:15-26,129-130) in fp32.
__device__ __forceinline__ int f0_bin_of(float f0, int n_bins, float a, float b) {
    float mel = 1127.0f * logf(1.0f + f0 / 700.0f);
    if (mel > 0.f) mel = mel * a - b;
    long c = (long)rintf(mel);                 // torch.round: half to even
    c = c > 0 ? c : 0;
    c = c < 1 ? c + 1 : c;
    c = c < n_bins ? c : 0;                    // the reference zeroes overflowing bins (its "+ (>= f0_bin)" term never fires)
    c = c < 0 ? 0 : (c > n_bins - 1 ? n_bins - 1 : c);
    return (int)c;
}

// out[b][t][c] = src row (projected features | embedding row) at the nearest-interpolated index (+ f0 term)
__global__ void lr_gather_kernel(const float* __restrict__ feat, long ld_feat, int tin_max, const long* __restrict__ tokens,
                                 const float* __restrict__ emb, const int* __restrict__ in_lens, const int* __restrict__ ylens,
                                 int interpolate, const float* __restrict__ f0, int tf0_max, const int* __restrict__ f0_lens,
                                 const float* __restrict__ f0_emb, const float* __restrict__ f0_mask, int n_bins, float fa,
                                 float fb, float* __restrict__ out, int tout_max, int C, int ld) {
    const int t = blockIdx.x, b = blockIdx.y;
    float* dst = out + ((long)b * tout_max + t) * ld;
    const int ylen = ylens[b];
    if (t >= ylen) {
        for (int c = threadIdx.x; c < ld; c += blockDim.x) dst[c] = 0.f;
        return;
    }
    const int tin = in_lens[b];
    int src = t;
    if (interpolate) {
        const float scale = (float)tin / (float)ylen;
        src = (int)floorf((float)t * scale);
        src = src < tin - 1 ? src : tin - 1;
    }
    const float* row = tokens ? emb + tokens[(long)b * tin_max + src] * (long)C : feat + ((long)b * tin_max + src) * ld_feat;
    const float* frow = nullptr;
    if (f0_emb) {
        if (f0) {
            const int tf = f0_lens[b];
            const float scale = (float)tf / (float)ylen;
            int sf = (int)floorf((float)t * scale);
            sf = sf < tf - 1 ? sf : tf - 1;
            frow = f0_emb + (long)f0_bin_of(f0[(long)b * tf0_max + sf], n_bins, fa, fb) * C;
        } else {
            frow = f0_mask;
        }
    }
    for (int c = threadIdx.x; c < ld; c += blockDim.x) {
        float v = 0.f;
        if (c < C) {
            v = row[c];
            if (frow) v += frow[c];
        }
        dst[c] = v;
    }
}

// GroupNorm(1, C) statistics of one utterance: sum and sum of squares over its ylens[b] x C values (double)
__global__ __launch_bounds__(256) void lr_gn_stats_kernel(const float* __restrict__ x, int tmax, int ld, int C,
                                                          const int* __restrict__ ylens, double* __restrict__ stats) {
    const int b = blockIdx.y;
    const int len = ylens[b];
    const int rows_per_block = (len + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(r0 + rows_per_block, len);
    double s = 0.0, q = 0.0;
    for (int r = r0 + (threadIdx.x >> 6); r < r1; r += 4) {
        const float* row = x + ((long)b * tmax + r) * ld;
        for (int c = threadIdx.x & 63; c < C; c += 64) {
            const double v = row[c];
            s += v;
            q += v * v;
        }
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && r1 > r0) {
        atomicAdd(&stats[2 * b], sh[0][0]);
        atomicAdd(&stats[2 * b + 1], sh[1][0]);
    }
}

// y = Mish(GroupNorm(x)) in place; rows >= ylens[b] and pad channels are zeroed (they are the next conv's zero padding)
__global__ void lr_gn_mish_kernel(float* __restrict__ x, int tmax, int ld, int C, const int* __restrict__ ylens,
                                  const double* __restrict__ stats, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, float eps) {
    const int t = blockIdx.x, b = blockIdx.y;
    float* row = x + ((long)b * tmax + t) * ld;
    const int len = ylens[b];
    if (t >= len) {
        for (int c = threadIdx.x; c < ld; c += blockDim.x) row[c] = 0.f;
        return;
    }
    const double n = (double)len * C;
    const double mean = stats[2 * b] / n;
    double var = stats[2 * b + 1] / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float mu = (float)mean;
    for (int c = threadIdx.x; c < ld; c += blockDim.x) {
        float o = 0.f;
        if (c < C) {
            const float v = (row[c] - mu) * rstd * gamma[c] + beta[c];
            const float sp = v > 20.0f ? v : log1pf(expf(v));       // F.softplus (threshold 20)
            o = v * tanhf(sp);
        }
        row[c] = o;
    }
}

// out[b][t][0..N) = src[b][t][0..N) for t < ylens[b], else 0 (Identity tail / final mask)
__global__ void lr_mask_copy_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ out, int ld_out, int N,
                                    int tmax, const int* __restrict__ ylens) {
    const int t = blockIdx.x, b = blockIdx.y;
    const bool ok = t < ylens[b];
    const float* s = src + ((long)b * tmax + t) * ld_src;
    float* d = out + ((long)b * tmax + t) * ld_out;
    for (int c = threadIdx.x; c < N; c += blockDim.x) d[c] = ok ? s[c] : 0.f;
}

// nn.Linear weight [N][K] -> ConvW with one tap (fp32 operands)
int pack_linear_f32(const StateDict& sd, const std::string& prefix, int N, int K, Arena& ar, hipStream_t st, ConvW* out) {
    const auto* w = sd.get(prefix + ".weight");
    const auto* b = sd.get(prefix + ".bias");
    if (require_shape(w, prefix + ".weight", {N, K})) return 1;
    if (require_shape(b, prefix + ".bias", {N})) return 1;
    out->vd = 1;
    out->dtype = 1;
    out->k = 1;
    out->cin_pad = cpad(K, 1);
    out->cout = N;
    out->cout_pad = cpad(N, 1);
    out->ldw = out->cin_pad;
    out->w = ar.alloc((size_t)round_up(out->cout_pad, 128) * out->ldw * 4, st);
    out->bias = ar.alloc_n<float>(round_up(out->cout_pad, 8), st);
    if (!out->w || !out->bias) return 1;
    if (pack_f32_launch(w->data, reinterpret_cast<float*>(out->w), N, 1, K, K, 0, 1, out->ldw, 0, 1, nullptr, st)) return 1;
    SVC_CHECK_HIP(hipMemcpyAsync(out->bias, b->data, N * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
}

}  // namespace

struct svc_lr {
    svc_lr_config_t cfg;
    Arena weights, work;
    PinnedRing staging;
    ConvW proj, tail;
    std::vector<ConvW> convs;
    std::vector<const float*> gn_w, gn_b;
    const float* emb = nullptr;
    const float* f0_emb = nullptr;
    const float* f0_mask = nullptr;
    int Cp = 0;
    // workspace
    int cap_B = 0, cap_tin = 0, cap_tout = 0;
    float *xin = nullptr, *h0 = nullptr, *a = nullptr, *y = nullptr;
    int *d_in_lens = nullptr, *d_ylens = nullptr, *d_f0_lens = nullptr;
    double* stats = nullptr;

    int reserve(int B, int tin, int tout, hipStream_t st) {
        if (B <= cap_B && tin <= cap_tin && tout <= cap_tout) return 0;
        SVC_CHECK_HIP(hipStreamSynchronize(st));
        work.release();
        cap_B = std::max(B, cap_B); cap_tin = std::max(tin, cap_tin); cap_tout = std::max(tout, cap_tout);
        const int inp = cfg.is_discrete ? 0 : cpad(cfg.in_channels, 1);
        xin = inp && inp != cfg.in_channels ? work.alloc_n<float>((size_t)cap_B * cap_tin * inp, st) : nullptr;
        h0 = inp ? work.alloc_n<float>((size_t)cap_B * cap_tin * Cp, st) : nullptr;
        const int ldmax = std::max(Cp, cpad(cfg.out_channels, 1));
        a = work.alloc_n<float>((size_t)cap_B * cap_tout * ldmax, st);
        y = work.alloc_n<float>((size_t)cap_B * cap_tout * ldmax, st);
        d_in_lens = work.alloc_n<int>(cap_B, st);
        d_ylens = work.alloc_n<int>(cap_B, st);
        d_f0_lens = work.alloc_n<int>(cap_B, st);
        stats = work.alloc_n<double>((size_t)2 * cap_B * std::max(1, cfg.n_convs), st);
        if ((inp && !h0) || !a || !y || !d_in_lens || !d_ylens || !d_f0_lens || !stats) return 1;
        return 0;
    }
};

extern "C" {

int svc_lr_create(const svc_lr_config_t* cfg, const svc_tensor_desc_t* weights, int n_weights, void* stream, svc_lr_t** out) {
    SVC_REQUIRE(cfg && weights && out, "null argument");
    SVC_REQUIRE(cfg->channels > 0 && cfg->out_channels > 0 && cfg->n_convs >= 0 && cfg->n_convs <= 16, "length regulator config");
    SVC_REQUIRE(cfg->is_discrete ? cfg->codebook_size > 0 : cfg->in_channels > 0, "length regulator input config");
    SVC_REQUIRE(cfg->has_final_conv || cfg->out_channels == cfg->channels, "Identity tail needs out_channels == channels");
    hipStream_t st = (hipStream_t)stream;
    StateDict sd(weights, n_weights);
    auto* m = new svc_lr();
    m->cfg = *cfg;
    const int C = cfg->channels;
    m->Cp = cpad(C, 1);
    auto fail = [&]() { delete m; return 1; };
    // fp32 tables are used in place from copies owned by the handle
    auto own = [&](const std::string& key, std::initializer_list<long> shp, const float** dst) -> int {
        const auto* d = sd.get(key);
        if (require_shape(d, key, shp)) return 1;
        float* p = m->weights.alloc_n<float>(StateDict::numel(d), st);
        if (!p) return 1;
        SVC_CHECK_HIP(hipMemcpyAsync(p, d->data, StateDict::numel(d) * sizeof(float), hipMemcpyDeviceToDevice, st));
        *dst = p;
        return 0;
    };
    if (cfg->is_discrete) {
        if (own("embedding.weight", {cfg->codebook_size, C}, &m->emb)) return fail();
    } else {
        if (pack_linear_f32(sd, "content_in_proj", C, cfg->in_channels, m->weights, st, &m->proj)) return fail();
    }
    if (cfg->f0_condition) {
        if (own("f0_embedding.weight", {cfg->n_f0_bins, C}, &m->f0_emb)) return fail();
        if (own("f0_mask", {1, C}, &m->f0_mask)) return fail();
    }
    m->convs.resize(cfg->n_convs);
    m->gn_w.resize(cfg->n_convs);
    m->gn_b.resize(cfg->n_convs);
    for (int i = 0; i < cfg->n_convs; ++i) {
        const std::string pc = "model." + std::to_string(3 * i), pn = "model." + std::to_string(3 * i + 1);
        if (pack_conv1d(sd, pc, C, C, 3, true, 1, m->weights, st, &m->convs[i])) return fail();
        if (own(pn + ".weight", {C}, &m->gn_w[i])) return fail();
        if (own(pn + ".bias", {C}, &m->gn_b[i])) return fail();
    }
    if (cfg->has_final_conv) {
        const std::string pt = "model." + std::to_string(3 * cfg->n_convs);
        if (pack_conv1d(sd, pt, cfg->out_channels, C, 1, true, 1, m->weights, st, &m->tail)) return fail();
    }
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    *out = m;
    return 0;
}

void svc_lr_destroy(svc_lr_t* m) { delete m; }

int svc_lr_forward(svc_lr_t* m, const float* x, const int64_t* tokens, const int32_t* in_lens, int B, int tin_max,
                   const int32_t* ylens, int tout_max, const float* f0, const int32_t* f0_lens, int tf0_max, float* out,
                   void* stream) {
    SVC_REQUIRE(m && in_lens && ylens && out, "null argument");
    hipStream_t st = (hipStream_t)stream;
    const svc_lr_config_t& c = m->cfg;
    SVC_REQUIRE(c.is_discrete ? tokens != nullptr : x != nullptr, "length regulator: wrong input kind for this model");
    SVC_REQUIRE(!f0 || (c.f0_condition && f0_lens && tf0_max > 0), "f0 given without f0_condition / lengths");
    if (B <= 0 || tout_max <= 0) return 0;
    SVC_REQUIRE(tin_max > 0, "empty content input");
    for (int b = 0; b < B; ++b) {
        SVC_REQUIRE(in_lens[b] > 0 && in_lens[b] <= tin_max, "in_lens out of range");
        SVC_REQUIRE(ylens[b] >= 0 && ylens[b] <= tout_max, "ylens out of range");
        SVC_REQUIRE(c.interpolate || ylens[b] <= in_lens[b], "without interpolation ylens must not exceed the input length");
        SVC_REQUIRE(!f0 || (f0_lens[b] > 0 && f0_lens[b] <= tf0_max), "f0_lens out of range");
    }
    if (m->reserve(B, tin_max, tout_max, st)) return 1;
    {   // the caller's host arrays are copied into a pinned slot: no stream synchronisation here
        int* h = reinterpret_cast<int*>(m->staging.acquire(3 * (size_t)B * sizeof(int)));
        if (!h) return 1;
        memcpy(h, in_lens, B * sizeof(int)); memcpy(h + B, ylens, B * sizeof(int));
        if (f0) memcpy(h + 2 * B, f0_lens, B * sizeof(int));
        SVC_CHECK_HIP(hipMemcpyAsync(m->d_in_lens, h, B * sizeof(int), hipMemcpyHostToDevice, st));
        SVC_CHECK_HIP(hipMemcpyAsync(m->d_ylens, h + B, B * sizeof(int), hipMemcpyHostToDevice, st));
        if (f0) SVC_CHECK_HIP(hipMemcpyAsync(m->d_f0_lens, h + 2 * B, B * sizeof(int), hipMemcpyHostToDevice, st));
        if (m->staging.commit(st)) return 1;
    }
    const int C = c.channels, Cp = m->Cp;

    // 1. content_in_proj on the fp32 MFMA (rows = all input frames)
    if (!c.is_discrete) {
        const float* src = x;
        if (m->xin) {     // in_channels not a multiple of the k-tile: zero-padded copy
            if (pack_f32_launch(x, m->xin, B * tin_max, 1, c.in_channels, c.in_channels, 0, 1, m->proj.cin_pad, 0, 1, nullptr, st)) return 1;
            src = m->xin;
        }
        ConvRun r;
        r.a.hi = src; r.B = 1; r.Lin = B * tin_max; r.Lout = B * tin_max;
        r.c32 = m->h0; r.ldc32 = Cp;
        if (conv1d_run(m->proj, r, st)) return 1;
    }
    // 2. nearest interpolation (+ f0 embedding) -> a
    {
        // a, b of f0_to_coarse are Python doubles in the reference; the tensor arithmetic runs in fp32
        const double mel_min = 1127.0 * log(1.0 + 50.0 / 700.0), mel_max = 1127.0 * log(1.0 + 1100.0 / 700.0);
        const double fa = c.f0_condition ? (c.n_f0_bins - 2) / (mel_max - mel_min) : 0.0;
        const double fb = mel_min * fa - 1.0;
        hipLaunchKernelGGL(lr_gather_kernel, dim3(tout_max, B), dim3(128), 0, st, m->h0, (long)Cp, tin_max,
                           reinterpret_cast<const long*>(tokens), m->emb, m->d_in_lens, m->d_ylens, c.interpolate, f0, tf0_max,
                           m->d_f0_lens, c.f0_condition ? m->f0_emb : nullptr, m->f0_mask, c.n_f0_bins, (float)fa, (float)fb,
                           m->a, tout_max, C, Cp);
        SVC_CHECK_HIP(hipGetLastError());
    }
    // 3. conv stack
    float* cur = m->a;
    float* nxt = m->y;
    if (c.n_convs) SVC_CHECK_HIP(hipMemsetAsync(m->stats, 0, sizeof(double) * 2 * B * c.n_convs, st));
    for (int i = 0; i < c.n_convs; ++i) {
        ConvRun r;
        r.a.hi = cur; r.B = B; r.Lin = tout_max; r.Lout = tout_max; r.pad_left = 1;
        r.seq_len = m->d_ylens;
        r.c32 = nxt; r.ldc32 = Cp;
        if (conv1d_run(m->convs[i], r, st)) return 1;
        double* stt = m->stats + (size_t)2 * B * i;
        hipLaunchKernelGGL(lr_gn_stats_kernel, dim3(32, B), dim3(256), 0, st, nxt, tout_max, Cp, C, m->d_ylens, stt);
        hipLaunchKernelGGL(lr_gn_mish_kernel, dim3(tout_max, B), dim3(128), 0, st, nxt, tout_max, Cp, C, m->d_ylens, stt,
                           m->gn_w[i], m->gn_b[i], 1e-5f);
        SVC_CHECK_HIP(hipGetLastError());
        std::swap(cur, nxt);
    }
    // 4. tail + mask
    const float* fin = cur;
    int ld_fin = Cp;
    if (c.has_final_conv) {
        ConvRun r;
        r.a.hi = cur; r.B = B; r.Lin = tout_max; r.Lout = tout_max;
        r.seq_len = m->d_ylens;
        r.c32 = nxt; r.ldc32 = m->tail.cout_pad;
        if (conv1d_run(m->tail, r, st)) return 1;
        fin = nxt;
        ld_fin = m->tail.cout_pad;
    }
    hipLaunchKernelGGL(lr_mask_copy_kernel, dim3(tout_max, B), dim3(128), 0, st, fin, ld_fin, out, c.out_channels,
                       c.out_channels, tout_max, m->d_ylens);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
