// CAMPPlus style encoder (second half) for gfx950: `campplus_model(feat)` of the reference drivers
// (inference.py:98-101,430) = modules/campplus/DTDNN.py:132-137 -- FCM head (2-D residual convs over the (feature, time)
// plane), TDNN, three CAM dense-TDNN blocks with transit layers, statistics pooling, dense embedding layer -- plus the Kaldi
// fbank front-end the drivers feed it with (torchaudio.compliance.kaldi.fbank, inference.py:418-428).
//
// Everything is fp32 on the MFMA (v_mfma_f32_16x16x4_f32 through the tap-GEMM).  Eval-mode BatchNorms are folded at
// pack time: BN after a conv into its weights + bias, BN before a conv (the pre-activation of the dense layers) into a
// scale/shift + ReLU kernel.  Layouts: the FCM plane is channels-last [B][T + 2][F][32] with F as the tap-GEMM's
// "position" axis (so the stride-2 feature down-sampling is the GEMM's row stride) and whole time steps as its
// "sequences": the 3 x 3 kernel is 9 taps = 3 position shifts x 3 sequence offsets (the t - 1 / t + 1 neighbours are
// pointer offsets of one sequence; the two border sequences of every clip are kept zero).  The TDNN part is channels-last
// [B][T/2][C] with the dense blocks' concatenation as a growing column range of one buffer.  Runs once per reference clip.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "model_util.h"

using namespace svc;

namespace {

constexpr float BN_EPS = 1e-5f;

__global__ void cp_bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                  const float* __restrict__ var, float* __restrict__ scale, float* __restrict__ shift, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = (gamma ? gamma[i] : 1.f) / sqrtf(var[i] + BN_EPS);
    scale[i] = s;
    shift[i] = (beta ? beta[i] : 0.f) - mean[i] * s;
}

// feat [B][T][F] -> plane [B][T + 2][F][32], channel 0 (the other 31 input channels of the 1-channel conv stay zero)
__global__ void cp_feat_plane_kernel(const float* __restrict__ feat, float* __restrict__ plane, int B, int T, int F) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * T * F) return;
    const int f = (int)(i % F);
    const long bt = i / F;
    const int t = (int)(bt % T), b = (int)(bt / T);
    plane[(((long)b * (T + 2) + t + 1) * F + f) * 32] = feat[i];
}

// zero the two border sequences (t = -1 and t = T) of every clip: buf [B][T + 2][row]
__global__ void cp_zero_border_kernel(float* __restrict__ buf, int B, int Tp, long row) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2L * B * row) return;
    const long r = i % row;
    const int which = (int)((i / row) & 1), b = (int)(i / (2 * row));
    buf[((long)b * Tp + (which ? Tp - 1 : 0)) * row + r] = 0.f;
}

// y[r][c] = relu(x[r][c] * s[c] + h[c]) for c < C, 0 for C <= c < Cp     (pre-activation BatchNorm + ReLU)
__global__ void cp_bnrelu_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy, const float* __restrict__ s,
                                 const float* __restrict__ h, long rows, int C, int Cp) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * Cp) return;
    const int c = (int)(i % Cp);
    const long r = i / Cp;
    y[r * ldy + c] = c < C ? fmaxf(x[r * ldx + c] * s[c] + h[c], 0.f) : 0.f;
}

// CAM context: ctx[b][seg][c] = mean_t x[b][t][c] + mean_{t in segment} x[b][t][c]   (layers.py:116-131; the last segment
// averages over the frames it really has: avg_pool1d(ceil_mode=True) does not count the padding)
__global__ void cp_ctx_kernel(const float* __restrict__ x, long ldx, float* __restrict__ ctx, int T2, int C, int seg_len, int n_seg) {
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* xb = x + (long)b * T2 * ldx + c;
    float tot = 0.f;
    for (int s = 0; s < n_seg; ++s) {
        const int t0 = s * seg_len, t1 = min(T2, t0 + seg_len);
        float a = 0.f;
        for (int t = t0; t < t1; ++t) a += xb[(long)t * ldx];
        tot += a;
        ctx[((long)b * n_seg + s) * C + c] = a / (float)(t1 - t0);
    }
    const float mean = tot / (float)T2;
    for (int s = 0; s < n_seg; ++s) ctx[((long)b * n_seg + s) * C + c] += mean;
}

// out[b][t][col0 + c] = y[b][t][c] * m[b][t / seg_len][c]      (m already passed through the sigmoid)
__global__ void cp_gate_kernel(const float* __restrict__ y, long ldy, const float* __restrict__ m, float* __restrict__ out, long ldo,
                               int col0, int B, int T2, int G, int seg_len, int n_seg) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * T2 * G) return;
    const int c = (int)(i % G);
    const long bt = i / G;
    const int t = (int)(bt % T2), b = (int)(bt / T2);
    out[bt * ldo + col0 + c] = y[bt * ldy + c] * m[((long)b * n_seg + t / seg_len) * G + c];
}

// statistics pooling: out[b] = [mean_t x | unbiased std_t x]       (layers.py:26-31)
__global__ void cp_stats_kernel(const float* __restrict__ x, long ldx, float* __restrict__ out, int T2, int C) {
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* xb = x + (long)b * T2 * ldx + c;
    double s = 0.0;
    for (int t = 0; t < T2; ++t) s += (double)xb[(long)t * ldx];
    const double mean = s / T2;
    double v = 0.0;
    for (int t = 0; t < T2; ++t) { const double d = (double)xb[(long)t * ldx] - mean; v += d * d; }
    out[(long)b * 2 * C + c] = (float)mean;
    out[(long)b * 2 * C + C + c] = (float)sqrt(v / (double)(T2 - 1));
}

// ---- Kaldi fbank front-end
// frames [n][nfft]: frame i = wave[i * shift .. + win), DC removed, pre-emphasised, Povey-windowed, zero-padded
__global__ void cp_fbank_frames_kernel(const float* __restrict__ wave, float* __restrict__ frames, int n_frames, int win, int shift,
                                       int nfft, float preemph, const float* __restrict__ window) {
    const int f = blockIdx.x;
    const float* w = wave + (long)f * shift;
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < win; i += 256) s += w[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const float mean = red[0] / (float)win;
    for (int i = threadIdx.x; i < nfft; i += 256) {
        float v = 0.f;
        if (i < win) {
            const float cur = w[i] - mean, prev = w[i > 0 ? i - 1 : 0] - mean;
            v = (cur - preemph * prev) * window[i];
        }
        frames[(long)f * nfft + i] = v;
    }
}

// spec [n][ld_s] = (re | im) -> power [n][ld_p], pad columns zero
__global__ void cp_power_kernel(const float* __restrict__ spec, long ld_s, float* __restrict__ pw, long ld_p, int nb, long n) {
    const long m = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ld_p) return;
    float v = 0.f;
    if (k < nb) {
        const float re = spec[m * ld_s + k], im = spec[m * ld_s + nb + k];
        v = re * re + im * im;
    }
    pw[m * ld_p + k] = v;
}

__global__ void cp_log_kernel(const float* __restrict__ e, long lde, float* __restrict__ out, int n_frames, int n_bins) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n_frames * n_bins) return;
    const int j = (int)(i % n_bins);
    const long f = i / n_bins;
    out[i] = logf(fmaxf(e[f * lde + j], 1.1920928955078125e-07f));
}

struct Gemm {          // one fp32 tap-GEMM launch on channels-last rows
    KGemmParams p;
    Gemm(int M, int N, int Lout) {
        memset(&p, 0, sizeof(p));
        p.M = M; p.N = N; p.Lout = Lout; p.a_seq_rows = Lout; p.c_seq_rows = Lout; p.a_stride = 1; p.a_len = Lout; p.n_taps = 1;
    }
    int run(hipStream_t st) {
        p.vec_ok = (p.N % 8 == 0) && (p.ldc32 % 8 == 0) && (p.ldres % 8 == 0);
        return kgemm_launch(p, 1, KG_EPI_STORE, st);
    }
};

}  // namespace

struct svc_campplus {
    svc_campplus_config_t cfg;
    Arena wts, ws, fb;          // fb: Kaldi-fbank scratch, released before it is regrown
    struct Conv { float* w = nullptr; float* b = nullptr; long ldw = 0; };
    Conv head_conv1, head_conv2, tdnn, dense;
    struct Res { Conv c1, c2, sc; bool has_sc = false; };
    Res res[4];
    struct Layer {
        float *s1, *h1;          // nonlinear1 BatchNorm scale / shift [cin]
        Conv lin1;               // [128][cin_pad] with nonlinear2's BatchNorm folded, bias = its shift
        Conv local;              // [g][k * 128]
        float *l1w, *l1b, *l2w, *l2b;
        int cin;
    };
    std::vector<Layer> layers[4];
    struct Transit { float *s, *h; Conv lin; int cin; } transit[4];
    float *s_out, *h_out;
    int out_ch = 0;
    // fbank
    float *fb_dft = nullptr, *fb_mel = nullptr, *fb_window = nullptr;
    // workspace
    int cap_B = 0, cap_T = 0;
    float *planeA, *planeB, *planeC, *blkA, *blkB, *tmp, *bnout, *ylocal, *ctx, *m1, *m2, *stats;
    long plane_guard = 0;
    int cap_frames = 0;
    float *fr_frames = nullptr, *fr_spec = nullptr, *fr_pow = nullptr, *fr_e = nullptr;

    int pack(const StateDict& sd, hipStream_t st);
    int reserve(int B, int T, hipStream_t st);
    int conv2d(const Conv& c, const float* x, int F, int stride, float* y, const float* resid, int act, int post_act, int B, int T,
               int taps, hipStream_t st);
};

namespace {
int bn_fold(const StateDict& sd, const std::string& p, int n, bool affine, Arena& ar, hipStream_t st, float** scale, float** shift) {
    const svc_tensor_desc_t *g = nullptr, *b = nullptr;
    if (affine) {
        g = sd.get(p + ".weight"); b = sd.get(p + ".bias");
        if (require_shape(g, p + ".weight", {n}) || require_shape(b, p + ".bias", {n})) return 1;
    }
    const auto* m = sd.get(p + ".running_mean");
    const auto* v = sd.get(p + ".running_var");
    if (require_shape(m, p + ".running_mean", {n}) || require_shape(v, p + ".running_var", {n})) return 1;
    *scale = ar.alloc_n<float>(round_up(n, 8), st);
    *shift = ar.alloc_n<float>(round_up(n, 8), st);
    if (!*scale || !*shift) return 1;
    hipLaunchKernelGGL(cp_bn_fold_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, g ? g->data : nullptr, b ? b->data : nullptr, m->data,
                       v->data, *scale, *shift, n);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

// Conv weight [N][Cin][k] (or [N][Cin][kh][kw] with explicit tap -> source-offset list) -> [Npad128][taps * cin_pad], row scale folded
int pack_taps(const float* src, int N, int Cin, int src_taps, const std::vector<int>& tap_src, int cin_pad, const float* scale,
              Arena& ar, hipStream_t st, svc_campplus::Conv* out) {
    const int taps = (int)tap_src.size();
    out->ldw = (long)taps * cin_pad;
    out->w = ar.alloc_n<float>((size_t)round_up(N, 128) * out->ldw, st);
    if (!out->w) return 1;
    for (int t = 0; t < taps; ++t)
        if (pack_f32_launch(src + tap_src[t], out->w + (long)t * cin_pad, N, Cin, 1, (long)Cin * src_taps, src_taps, 0, out->ldw, 1, 0,
                            scale, st)) return 1;
    return 0;
}
}  // namespace

#define GETW(var, name, ...)                                              \
    const svc_tensor_desc_t* var = sd.get(name);                          \
    if (require_shape(var, name, {__VA_ARGS__})) return 1;

int svc_campplus::pack(const StateDict& sd, hipStream_t st) {
    const int mC = cfg.m_channels, F = cfg.feat_dim, init = cfg.init_channels, g = cfg.growth_rate, bn = cfg.bn_size * cfg.growth_rate;
    // 3 x 3 taps in the order tap = dt * 3 + df (dt: time = kernel width, df: feature = kernel height); source offset df * 3 + dt
    std::vector<int> taps9;
    for (int dt = 0; dt < 3; ++dt)
        for (int df = 0; df < 3; ++df) taps9.push_back(df * 3 + dt);
    auto conv3x3 = [&](const std::string& wname, const std::string& bnname, int cin, Conv* c) -> int {
        GETW(w, wname, mC, cin, 3, 3);
        float *s, *h;
        if (bn_fold(sd, bnname, mC, true, wts, st, &s, &h)) return 1;
        c->b = h;
        return pack_taps(w->data, mC, cin, 9, taps9, 32, s, wts, st, c);
    };
    if (conv3x3("head.conv1.weight", "head.bn1", 1, &head_conv1)) return 1;
    int ri = 0;
    for (const char* layer : {"layer1", "layer2"})
        for (int b = 0; b < 2; ++b, ++ri) {
            const std::string p = std::string("head.") + layer + "." + std::to_string(b);
            if (conv3x3(p + ".conv1.weight", p + ".bn1", mC, &res[ri].c1)) return 1;
            if (conv3x3(p + ".conv2.weight", p + ".bn2", mC, &res[ri].c2)) return 1;
            res[ri].has_sc = b == 0;
            if (b == 0) {
                GETW(w, p + ".shortcut.0.weight", mC, mC, 1, 1);
                float *s, *h;
                if (bn_fold(sd, p + ".shortcut.1", mC, true, wts, st, &s, &h)) return 1;
                res[ri].sc.b = h;
                if (pack_taps(w->data, mC, mC, 1, {0}, 32, s, wts, st, &res[ri].sc)) return 1;
            }
        }
    if (conv3x3("head.conv2.weight", "head.bn2", mC, &head_conv2)) return 1;
    // TDNN: input channel of the reference = c * F8 + f (reshape of (B, 32, F/8, T)); ours = f * 32 + c
    const int F8 = F / 8, ch0 = mC * F8;
    {
        GETW(w, "xvector.tdnn.linear.weight", init, ch0, 5);
        float *s, *h;
        if (bn_fold(sd, "xvector.tdnn.nonlinear.batchnorm", init, true, wts, st, &s, &h)) return 1;
        tdnn.b = h;
        tdnn.ldw = 5L * ch0;
        tdnn.w = wts.alloc_n<float>((size_t)round_up(init, 128) * tdnn.ldw, st);
        if (!tdnn.w) return 1;
        for (int t = 0; t < 5; ++t)
            for (int f = 0; f < F8; ++f)
                if (pack_f32_launch(w->data + (long)f * 5 + t, tdnn.w + (long)t * ch0 + f * 32, init, mC, 1, (long)ch0 * 5, (long)F8 * 5, 0,
                                    tdnn.ldw, 1, 0, s, st)) return 1;
    }
    int ch = init;
    for (int bi = 0; bi < cfg.n_blocks; ++bi) {
        const int nl = cfg.block_layers[bi], k = cfg.block_kernel[bi];
        layers[bi].resize(nl);
        for (int i = 0; i < nl; ++i) {
            Layer& ly = layers[bi][i];
            const std::string p = "xvector.block" + std::to_string(bi + 1) + ".tdnnd" + std::to_string(i + 1);
            ly.cin = ch + i * g;
            if (bn_fold(sd, p + ".nonlinear1.batchnorm", ly.cin, true, wts, st, &ly.s1, &ly.h1)) return 1;
            GETW(w1, p + ".linear1.weight", bn, ly.cin, 1);
            float *s2, *h2;
            if (bn_fold(sd, p + ".nonlinear2.batchnorm", bn, true, wts, st, &s2, &h2)) return 1;
            ly.lin1.b = h2;
            if (pack_taps(w1->data, bn, ly.cin, 1, {0}, (int)round_up(ly.cin, 32), s2, wts, st, &ly.lin1)) return 1;
            GETW(wl, p + ".cam_layer.linear_local.weight", g, bn, k);
            std::vector<int> tk;
            for (int t = 0; t < k; ++t) tk.push_back(t);
            if (pack_taps(wl->data, g, bn, k, tk, bn, nullptr, wts, st, &ly.local)) return 1;
            GETW(a, p + ".cam_layer.linear1.weight", bn / 2, bn, 1);
            GETW(ab, p + ".cam_layer.linear1.bias", bn / 2);
            GETW(c2, p + ".cam_layer.linear2.weight", g, bn / 2, 1);
            GETW(cb, p + ".cam_layer.linear2.bias", g);
            ly.l1w = wts.alloc_n<float>((size_t)bn / 2 * bn, st); ly.l1b = wts.alloc_n<float>(bn / 2, st);
            ly.l2w = wts.alloc_n<float>((size_t)g * bn / 2, st); ly.l2b = wts.alloc_n<float>(g, st);
            if (!ly.l1w || !ly.l1b || !ly.l2w || !ly.l2b) return 1;
            SVC_CHECK_HIP(hipMemcpyAsync(ly.l1w, a->data, (size_t)bn / 2 * bn * 4, hipMemcpyDeviceToDevice, st));
            SVC_CHECK_HIP(hipMemcpyAsync(ly.l1b, ab->data, (size_t)bn / 2 * 4, hipMemcpyDeviceToDevice, st));
            SVC_CHECK_HIP(hipMemcpyAsync(ly.l2w, c2->data, (size_t)g * bn / 2 * 4, hipMemcpyDeviceToDevice, st));
            SVC_CHECK_HIP(hipMemcpyAsync(ly.l2b, cb->data, (size_t)g * 4, hipMemcpyDeviceToDevice, st));
        }
        ch += nl * g;
        const std::string p = "xvector.transit" + std::to_string(bi + 1);
        transit[bi].cin = ch;
        if (bn_fold(sd, p + ".nonlinear.batchnorm", ch, true, wts, st, &transit[bi].s, &transit[bi].h)) return 1;
        GETW(w, p + ".linear.weight", ch / 2, ch, 1);
        if (pack_taps(w->data, ch / 2, ch, 1, {0}, ch, nullptr, wts, st, &transit[bi].lin)) return 1;
        ch /= 2;
    }
    out_ch = ch;
    if (bn_fold(sd, "xvector.out_nonlinear.batchnorm", ch, true, wts, st, &s_out, &h_out)) return 1;
    {
        GETW(w, "dense.linear.weight", cfg.embedding_size, 2 * ch, 1);
        float *s, *h;
        if (bn_fold(sd, "dense.nonlinear.batchnorm", cfg.embedding_size, false, wts, st, &s, &h)) return 1;
        dense.b = h;
        dense.ldw = 2L * ch;
        dense.w = wts.alloc_n<float>((size_t)cfg.embedding_size * dense.ldw, st);
        if (!dense.w) return 1;
        if (pack_f32_launch(w->data, dense.w, cfg.embedding_size, 2 * ch, 1, 2L * ch, 1, 0, dense.ldw, 1, 0, s, st)) return 1;
    }
    // ---- Kaldi fbank constants: 25 ms / 10 ms frames at 16 kHz, 512-point DFT, 80 Kaldi-mel triangles from 20 Hz to Nyquist
    {
        const int win = 400, nfft = 512, nb = nfft / 2 + 1, bins = cfg.feat_dim;
        std::vector<float> wv(win), basis((size_t)round_up(2 * nb, 128) * nfft, 0.f), mel((size_t)round_up(bins, 128) * round_up(nb, 32), 0.f);
        for (int i = 0; i < win; ++i) wv[i] = powf(0.5f - 0.5f * cosf(2.0f * (float)M_PI * (float)i / (float)(win - 1)), 0.85f);
        for (int k = 0; k < nb; ++k)
            for (int n = 0; n < nfft; ++n) {
                const double ang = 2.0 * M_PI * (double)(((long)k * n) % nfft) / (double)nfft;
                basis[(size_t)k * nfft + n] = (float)cos(ang);
                basis[(size_t)(nb + k) * nfft + n] = (float)(-sin(ang));
            }
        auto melf = [](float f) { return 1127.0f * logf(1.0f + f / 700.0f); };
        const float mlo = melf(20.0f), mhi = melf(8000.0f), delta = (mhi - mlo) / (float)(bins + 1);
        const long ldm = round_up(nb, 32);
        for (int b = 0; b < bins; ++b) {
            const float left = mlo + b * delta, center = mlo + (b + 1.0f) * delta, right = mlo + (b + 2.0f) * delta;
            for (int k = 0; k < nfft / 2; ++k) {
                const float m = melf(16000.0f / nfft * (float)k);
                const float up = (m - left) / (center - left), down = (right - m) / (right - center);
                mel[(size_t)b * ldm + k] = std::max(0.0f, std::min(up, down));
            }
        }
        fb_window = wts.alloc_n<float>(win, st);
        fb_dft = wts.alloc_n<float>(basis.size(), st);
        fb_mel = wts.alloc_n<float>(mel.size(), st);
        if (!fb_window || !fb_dft || !fb_mel) return 1;
        SVC_CHECK_HIP(hipMemcpyAsync(fb_window, wv.data(), wv.size() * 4, hipMemcpyHostToDevice, st));
        SVC_CHECK_HIP(hipMemcpyAsync(fb_dft, basis.data(), basis.size() * 4, hipMemcpyHostToDevice, st));
        SVC_CHECK_HIP(hipMemcpyAsync(fb_mel, mel.data(), mel.size() * 4, hipMemcpyHostToDevice, st));
        SVC_CHECK_HIP(hipStreamSynchronize(st));
    }
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
}

int svc_campplus::reserve(int B, int T, hipStream_t st) {
    if (B <= cap_B && T <= cap_T) return 0;
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    ws.release();
    cap_B = std::max(B, cap_B); cap_T = std::max(T, cap_T);
    const long Tp = cap_T + 2, F = cfg.feat_dim;
    plane_guard = F * 32;                                   // one zero sequence in front of / behind every plane buffer
    const long plane = (long)cap_B * Tp * F * 32 + 2 * plane_guard;
    planeA = ws.alloc_n<float>(plane, st); planeB = ws.alloc_n<float>(plane, st); planeC = ws.alloc_n<float>(plane, st);
    const long T2 = (cap_T - 1) / 2 + 1, rows = (long)cap_B * T2;
    int cmax = cfg.init_channels, ch = cfg.init_channels;
    for (int bi = 0; bi < cfg.n_blocks; ++bi) { ch += cfg.block_layers[bi] * cfg.growth_rate; cmax = std::max(cmax, ch); ch /= 2; }
    const long nseg = cdiv(T2, cfg.seg_len), bn = cfg.bn_size * cfg.growth_rate;
    blkA = ws.alloc_n<float>(rows * cmax, st); blkB = ws.alloc_n<float>(rows * cmax, st); tmp = ws.alloc_n<float>(rows * cmax, st);
    bnout = ws.alloc_n<float>(rows * bn, st); ylocal = ws.alloc_n<float>(rows * cfg.growth_rate, st);
    ctx = ws.alloc_n<float>((long)cap_B * nseg * bn, st); m1 = ws.alloc_n<float>((long)cap_B * nseg * bn, st);
    m2 = ws.alloc_n<float>((long)cap_B * nseg * bn, st); stats = ws.alloc_n<float>((long)cap_B * 2 * cmax, st);
    if (!planeA || !planeB || !planeC || !blkA || !blkB || !tmp || !bnout || !ylocal || !ctx || !m1 || !m2 || !stats) return 1;
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
}

// x, y: plane buffers (pointing past the guard) [B][T + 2][F][32] / [B][T + 2][Fo][32]; taps = 9 (3 x 3, pad 1) or 1 (1 x 1)
int svc_campplus::conv2d(const Conv& c, const float* x, int F, int stride, float* y, const float* resid, int act, int post_act, int B,
                         int T, int taps, hipStream_t st) {
    const int Tp = T + 2, Fo = taps == 9 ? (F + 2 - 3) / stride + 1 : (F - 1) / stride + 1;
    Gemm g(B * Tp * Fo, cfg.m_channels, Fo);
    g.p.a_seq_rows = F; g.p.a_len = F; g.p.a_stride = stride; g.p.pad_mode = KG_PAD_ZERO;
    g.p.n_taps = taps;
    for (int t = 0; t < taps; ++t) {
        const int dt = taps == 9 ? t / 3 : 1, df = taps == 9 ? t % 3 : 1;
        g.p.a_ptr[t] = x + (long)(dt - 1) * F * 32;          // the t - 1 / t + 1 neighbour = one whole sequence away
        g.p.a_ld[t] = 32; g.p.a_ktiles[t] = 1; g.p.a_shift[t] = df - 1;
    }
    g.p.w = c.w; g.p.ldw = c.ldw; g.p.bias = c.b;
    g.p.c32 = y; g.p.ldc32 = 32;
    g.p.res = resid; g.p.ldres = 32;
    g.p.act = act; g.p.post_relu = post_act != 0; g.p.act_slope = 0.f;
    if (g.run(st)) return 1;
    hipLaunchKernelGGL(cp_zero_border_kernel, dim3(cdiv(2L * B * Fo * 32, 256)), dim3(256), 0, st, y, B, Tp, (long)Fo * 32);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" {

int svc_campplus_create(const svc_campplus_config_t* cfg, const svc_tensor_desc_t* weights, int n_weights, void* stream,
                        svc_campplus_t** out) {
    SVC_REQUIRE(cfg && weights && out, "null argument");
    SVC_REQUIRE(cfg->m_channels == 32 && cfg->feat_dim % 8 == 0 && cfg->n_blocks >= 1 && cfg->n_blocks <= 4, "CAMPPlus: FCM with 32 channels");
    SVC_REQUIRE(cfg->bn_size * cfg->growth_rate % 32 == 0 && cfg->growth_rate % 8 == 0 && cfg->init_channels % 32 == 0, "CAMPPlus: channel multiples");
    auto* m = new svc_campplus();
    m->cfg = *cfg;
    StateDict sd(weights, n_weights);
    if (m->pack(sd, (hipStream_t)stream)) { delete m; return 1; }
    *out = m;
    return 0;
}

void svc_campplus_destroy(svc_campplus_t* m) { delete m; }

int svc_campplus_forward(svc_campplus_t* m, const float* feat, int B, int T, float* out, void* stream) {
    SVC_REQUIRE(m && feat && out && B >= 1 && T >= 8, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (m->reserve(B, T, st)) return 1;
    const auto& c = m->cfg;
    const int F = c.feat_dim, Tp = T + 2, LR = KG_ACT_LRELU;       // leaky ReLU with slope 0 = ReLU
    float* A = m->planeA + m->plane_guard;
    float* Bp = m->planeB + m->plane_guard;
    float* Cp = m->planeC + m->plane_guard;
    const size_t plane_bytes = ((size_t)B * Tp * F * 32 + 2 * m->plane_guard) * 4;
    SVC_CHECK_HIP(hipMemsetAsync(m->planeA, 0, plane_bytes, st));
    SVC_CHECK_HIP(hipMemsetAsync(m->planeB, 0, plane_bytes, st));
    SVC_CHECK_HIP(hipMemsetAsync(m->planeC, 0, plane_bytes, st));
    hipLaunchKernelGGL(cp_feat_plane_kernel, dim3(cdiv((long)B * T * F, 256)), dim3(256), 0, st, feat, A, B, T, F);
    SVC_CHECK_HIP(hipGetLastError());
    // FCM head (DTDNN.py:39-52)
    if (m->conv2d(m->head_conv1, A, F, 1, Bp, nullptr, LR, 0, B, T, 9, st)) return 1;       // x = Bp, F
    float *x = Bp, *t1 = A, *t2 = Cp;
    int Fc = F;
    for (int ri = 0; ri < 4; ++ri) {
        const auto& r = m->res[ri];
        const int stride = r.has_sc ? 2 : 1, Fo = (Fc + 2 - 3) / stride + 1;
        if (m->conv2d(r.c1, x, Fc, stride, t1, nullptr, LR, 0, B, T, 9, st)) return 1;
        const float* shortcut = x;
        if (r.has_sc) {
            if (m->conv2d(r.sc, x, Fc, stride, t2, nullptr, 0, 0, B, T, 1, st)) return 1;
            shortcut = t2;
        }
        // out = relu(bn2(conv2(.)) + shortcut): residual added in the epilogue, ReLU after it.  In-place when the shortcut
        // is x itself would alias reads of neighbours with writes, so the result goes to a third buffer.
        float* dst = r.has_sc ? x : t2;
        if (m->conv2d(r.c2, t1, Fo, 1, dst, shortcut, 0, LR, B, T, 9, st)) return 1;
        if (!r.has_sc) std::swap(x, t2);
        Fc = Fo;
    }
    if (m->conv2d(m->head_conv2, x, Fc, 2, t1, nullptr, LR, 0, B, T, 9, st)) return 1;
    const int F8 = (Fc + 2 - 3) / 2 + 1, ch0 = 32 * F8;
    SVC_REQUIRE(F8 == F / 8, "CAMPPlus: feature axis bookkeeping");
    // TDNN (k 5, stride 2, pad 2) on [B][T + 2][ch0], rows 1 .. T of every clip
    const int T2 = (T + 4 - 5) / 2 + 1, rows = B * T2, g = c.growth_rate, bn = c.bn_size * c.growth_rate;
    int cmax = c.init_channels, chs = c.init_channels;
    for (int bi = 0; bi < c.n_blocks; ++bi) { chs += c.block_layers[bi] * g; cmax = std::max(cmax, chs); chs /= 2; }
    float *blk = m->blkA, *blk_next = m->blkB;
    {
        Gemm gm(rows, c.init_channels, T2);
        gm.p.a_seq_rows = Tp; gm.p.a_off = 1; gm.p.a_len = T; gm.p.a_stride = 2; gm.p.pad_mode = KG_PAD_ZERO;
        gm.p.n_taps = 5;
        for (int t = 0; t < 5; ++t) { gm.p.a_ptr[t] = t1; gm.p.a_ld[t] = ch0; gm.p.a_ktiles[t] = ch0 / 32; gm.p.a_shift[t] = t - 2; }
        gm.p.w = m->tdnn.w; gm.p.ldw = m->tdnn.ldw; gm.p.bias = m->tdnn.b; gm.p.act = LR;
        gm.p.c32 = blk; gm.p.ldc32 = cmax;
        if (gm.run(st)) return 1;
    }
    const int n_seg = cdiv(T2, c.seg_len);
    int ch = c.init_channels;
    for (int bi = 0; bi < c.n_blocks; ++bi) {
        const int k = c.block_kernel[bi], dil = c.block_dilation[bi];
        for (auto& ly : m->layers[bi]) {
            const int cinp = (int)round_up(ly.cin, 32);
            hipLaunchKernelGGL(cp_bnrelu_kernel, dim3(cdiv((long)rows * cinp, 256)), dim3(256), 0, st, blk, (long)cmax, m->tmp, (long)cinp,
                               ly.s1, ly.h1, (long)rows, ly.cin, cinp);
            {   // linear1 (1 x 1) + nonlinear2 (BatchNorm folded) + ReLU
                Gemm gm(rows, bn, T2);
                gm.p.a_ptr[0] = m->tmp; gm.p.a_ld[0] = cinp; gm.p.a_ktiles[0] = cinp / 32;
                gm.p.w = ly.lin1.w; gm.p.ldw = ly.lin1.ldw; gm.p.bias = ly.lin1.b; gm.p.act = LR;
                gm.p.c32 = m->bnout; gm.p.ldc32 = bn;
                if (gm.run(st)) return 1;
            }
            {   // CAM: local k-tap conv
                Gemm gm(rows, g, T2);
                gm.p.n_taps = k; gm.p.pad_mode = KG_PAD_ZERO;
                for (int t = 0; t < k; ++t) { gm.p.a_ptr[t] = m->bnout; gm.p.a_ld[t] = bn; gm.p.a_ktiles[t] = bn / 32; gm.p.a_shift[t] = (t - (k - 1) / 2) * dil; }
                gm.p.w = ly.local.w; gm.p.ldw = ly.local.ldw;
                gm.p.c32 = m->ylocal; gm.p.ldc32 = g;
                if (gm.run(st)) return 1;
            }
            hipLaunchKernelGGL(cp_ctx_kernel, dim3(cdiv(bn, 128), B), dim3(128), 0, st, m->bnout, (long)bn, m->ctx, T2, bn, c.seg_len, n_seg);
            SVC_CHECK_HIP(hipGetLastError());
            if (small_linear_launch(m->ctx, bn, ly.l1w, bn, ly.l1b, m->m1, bn / 2, B * n_seg, bn / 2, bn, LR, st)) return 1;
            if (small_linear_launch(m->m1, bn / 2, ly.l2w, bn / 2, ly.l2b, m->m2, g, B * n_seg, g, bn / 2, KG_ACT_SIGMOID, st)) return 1;
            hipLaunchKernelGGL(cp_gate_kernel, dim3(cdiv((long)rows * g, 256)), dim3(256), 0, st, m->ylocal, (long)g, m->m2, blk, (long)cmax,
                               ly.cin, B, T2, g, c.seg_len, n_seg);
            SVC_CHECK_HIP(hipGetLastError());
        }
        ch += (int)m->layers[bi].size() * g;
        hipLaunchKernelGGL(cp_bnrelu_kernel, dim3(cdiv((long)rows * ch, 256)), dim3(256), 0, st, blk, (long)cmax, m->tmp, (long)ch,
                           m->transit[bi].s, m->transit[bi].h, (long)rows, ch, ch);
        Gemm gm(rows, ch / 2, T2);
        gm.p.a_ptr[0] = m->tmp; gm.p.a_ld[0] = ch; gm.p.a_ktiles[0] = ch / 32;
        gm.p.w = m->transit[bi].lin.w; gm.p.ldw = m->transit[bi].lin.ldw;
        gm.p.c32 = blk_next; gm.p.ldc32 = cmax;
        if (gm.run(st)) return 1;
        std::swap(blk, blk_next);
        ch /= 2;
    }
    hipLaunchKernelGGL(cp_bnrelu_kernel, dim3(cdiv((long)rows * ch, 256)), dim3(256), 0, st, blk, (long)cmax, m->tmp, (long)ch, m->s_out,
                       m->h_out, (long)rows, ch, ch);
    hipLaunchKernelGGL(cp_stats_kernel, dim3(cdiv(ch, 128), B), dim3(128), 0, st, m->tmp, (long)ch, m->stats, T2, ch);
    SVC_CHECK_HIP(hipGetLastError());
    return small_linear_launch(m->stats, 2 * ch, m->dense.w, m->dense.ldw, m->dense.b, out, c.embedding_size, B, c.embedding_size, 2 * ch,
                               KG_ACT_NONE, st);
}

int svc_kaldi_fbank_frames(int n_samples) { return n_samples >= 400 ? 1 + (n_samples - 400) / 160 : 0; }

int svc_kaldi_fbank(svc_campplus_t* m, const float* wave, int n_samples, float* out, void* stream) {
    SVC_REQUIRE(m && wave && out, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int win = 400, shift = 160, nfft = 512, nb = nfft / 2 + 1, bins = m->cfg.feat_dim;
    const int n = svc_kaldi_fbank_frames(n_samples);
    SVC_REQUIRE(n >= 1, "waveform shorter than one 25 ms frame");
    const long ld_s = round_up(2 * nb, 8), ld_p = round_up(nb, 32), ld_e = round_up(bins, 32);
    if (n > m->cap_frames) {
        SVC_CHECK_HIP(hipStreamSynchronize(st));
        m->fb.release();                                    // the stream is idle: nothing reads the old buffers any more
        m->cap_frames = 0;
        m->fr_frames = m->fb.alloc_n<float>((size_t)n * nfft, st);
        m->fr_spec = m->fb.alloc_n<float>((size_t)n * ld_s, st);
        m->fr_pow = m->fb.alloc_n<float>((size_t)n * ld_p, st);
        m->fr_e = m->fb.alloc_n<float>((size_t)n * ld_e, st);
        if (!m->fr_frames || !m->fr_spec || !m->fr_pow || !m->fr_e) return 1;
        m->cap_frames = n;
    }
    hipLaunchKernelGGL(cp_fbank_frames_kernel, dim3(n), dim3(256), 0, st, wave, m->fr_frames, n, win, shift, nfft, 0.97f, m->fb_window);
    SVC_CHECK_HIP(hipGetLastError());
    {
        Gemm g(n, 2 * nb, n);
        g.p.a_ptr[0] = m->fr_frames; g.p.a_ld[0] = nfft; g.p.a_ktiles[0] = nfft / 32;
        g.p.w = m->fb_dft; g.p.ldw = nfft;
        g.p.c32 = m->fr_spec; g.p.ldc32 = ld_s;
        g.p.vec_ok = 1;
        if (kgemm_launch(g.p, 1, KG_EPI_STORE, st)) return 1;
    }
    hipLaunchKernelGGL(cp_power_kernel, dim3(cdiv(ld_p, 128), n), dim3(128), 0, st, m->fr_spec, ld_s, m->fr_pow, ld_p, nb, (long)n);
    SVC_CHECK_HIP(hipGetLastError());
    {
        Gemm g(n, bins, n);
        g.p.a_ptr[0] = m->fr_pow; g.p.a_ld[0] = ld_p; g.p.a_ktiles[0] = (int)(ld_p / 32);
        g.p.w = m->fb_mel; g.p.ldw = ld_p;
        g.p.c32 = m->fr_e; g.p.ldc32 = ld_e;
        g.p.vec_ok = 1;
        if (kgemm_launch(g.p, 1, KG_EPI_STORE, st)) return 1;
    }
    hipLaunchKernelGGL(cp_log_kernel, dim3(cdiv((long)n * bins, 256)), dim3(256), 0, st, m->fr_e, ld_e, out, n, bins);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
