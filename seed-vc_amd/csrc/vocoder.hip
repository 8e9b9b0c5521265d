// BigVGAN-v2 and HiFT mel -> waveform vocoders on the tap-GEMM (channels-last activations).
//
// Layout: every activation is [B][L][Cpad] (channels contiguous, Cpad = C rounded up to the k-tile: 32
// fp32 / 64 fp16, pad channels are kept at zero).  Conv1d = k-tap GEMM over shifted rows (zero pad),
// ConvTranspose1d(k = 2s, p = s/2) = 3-tap GEMM with N = s * Cpad_out whose output row q holds output
// samples s*q .. s*q+s-1, i.e. it IS the channels-last up-sampled tensor.  precision 0 runs the
// contractions on the fp32 MFMA (exact fp32 products, needed for the 1e-4 waveform RMS parity bar);
// precision 1 feeds fp16 operands (fp32 accumulate); precision 2 ("fp16x3") splits every operand into fp16
// hi + lo planes and issues the three significant products hi*hi + hi*lo + lo*hi on the fp16 MFMA (fp32
// accumulate): ~2^-22 relative product error (fp32-class results) at 3/16 of the fp32-MFMA cycles.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "conv_util.h"

using namespace svc;

namespace {


// ---- small kernels -------------------------------------------------------------------------------
// snake parameter preparation: mode 0 BigVGAN log-scale snakebeta (alpha, beta), 1 log-scale snake (alpha only),
// 2 HiFT linear snake.  out a[c], inv_b[c]
__global__ void snake_params_kernel(const float* alpha, const float* beta, float* a, float* inv_b, int C, int mode) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float al = alpha[c], be = beta ? beta[c] : alpha[c];
    if (mode != 2) { al = expf(al); be = expf(be); }
    a[c] = al;
    inv_b[c] = 1.0f / (be + 1e-9f);
}

// (B, C, S) fp32 -> channels-last [B][S][ld] in fp32 or fp16, pad channels zero
template <typename OutT>
__global__ void mel_to_cl_kernel(const float* __restrict__ mel, OutT* __restrict__ dst, OutT* __restrict__ dst_lo, int B, int C,
                                 int S, int ld) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * S * ld) return;
    const int c = (int)(i % ld);
    const long bs = i / ld;
    const int s = (int)(bs % S), b = (int)(bs / S);
    const float v = c < C ? mel[((long)b * C + c) * S + s] : 0.f;
    const OutT h = (OutT)v;
    dst[i] = h;
    if (dst_lo) dst_lo[i] = (OutT)(v - (float)h);
}

// elementwise channels-last: y = f(x) with pad channels zeroed. mode 2 leaky relu, 3 identity (cast)
template <typename OutT>
__global__ void ew_cl_kernel(const float* __restrict__ x, OutT* __restrict__ y, OutT* __restrict__ y_lo, long rows, int C, int ld,
                             int mode, float slope) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ld) return;
    const int c = (int)(i % ld);
    float v = c < C ? x[i] : 0.f;
    if (mode == 2) v = v > 0.f ? v : v * slope;
    const OutT h = (OutT)v;
    y[i] = h;
    if (y_lo) y_lo[i] = (OutT)(v - (float)h);
}

__global__ void copy_row_kernel(float* x, int B, int rows, int ld, int src_row, int dst_row) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * ld) return;
    const int b = i / ld, c = i - b * ld;
    x[((long)b * rows + dst_row) * ld + c] = x[((long)b * rows + src_row) * ld + c];
}

// ---- HiFT source / STFT / iSTFT ---------------------------------------------------------------------
// per (b, harmonic): exclusive prefix over frames of up * (double) v_f, v_f = fp32((f0 * (h+1)) / sr)
__global__ void hift_phase_prefix_kernel(const float* __restrict__ f0, double* __restrict__ prefix, int B, int S, int NH, int up,
                                         float sr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * NH) return;
    const int b = i / NH, h = i - b * NH;
    double acc = 0.0;
    for (int f = 0; f < S; ++f) {
        prefix[((long)b * NH + h) * S + f] = acc;
        const float v = (f0[(long)b * S + f] * (float)(h + 1)) / sr;
        // the reference adds the value `up` times in a double accumulator (torch.cumsum on CPU floats);
        // up * v is exact in double, so this differs from the sequential sum by < 1e-13 relative
        acc += (double)up * (double)v;
    }
}

// merged source s[b][n] = tanh(sum_h lin_w[h] * (sine_h * uv + noise_amp * noise_h) + lin_b)
__global__ void hift_source_kernel(const float* __restrict__ f0, const double* __restrict__ prefix, const float* __restrict__ phase0,
                                   const float* __restrict__ noise, const float* __restrict__ lin_w, const float* __restrict__ lin_b,
                                   float* __restrict__ s_out, int B, int S, int NH, int up, float sr, float sine_amp, float noise_std,
                                   float voiced_thr) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long Lw = (long)S * up;
    if (i >= B * Lw) return;
    const int b = (int)(i / Lw);
    const long n = i - (long)b * Lw;
    const int f = (int)(n / up), r = (int)(n - (long)f * up);
    const float f0v = f0[(long)b * S + f];
    const float uv = f0v > voiced_thr ? 1.f : 0.f;
    const float namp = uv * noise_std + (1.f - uv) * sine_amp / 3.f;
    const float two_pi = 6.283185307179586f;
    float acc = lin_b[0];
    for (int h = 0; h < NH; ++h) {
        const float v = (f0v * (float)(h + 1)) / sr;
        const double cum = prefix[((long)b * NH + h) * S + f] + (double)(r + 1) * (double)v;
        const float cf = (float)cum;
        const float frac = fmodf(cf, 1.0f);
        const float theta = two_pi * frac;
        const float ph = h == 0 ? 0.f : phase0[(long)b * NH + h];
        const float sine = sine_amp * sinf(theta + ph);
        const float val = sine * uv + namp * noise[((long)b * NH + h) * Lw + n];
        acc += lin_w[h] * val;
    }
    s_out[i] = tanhf(acc);
}

// centred, reflect-padded STFT (n_fft 16, hop 4, periodic Hann): out [B][F][ld]: channels 0..8 real, 9..17 imag
__global__ void hift_stft_kernel(const float* __restrict__ s, float* __restrict__ out, int B, long Lw, int F, int ld) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * F) return;
    const int b = (int)(i / F), fr = (int)(i - (long)b * F);
    float x[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        long q = 4L * fr - 8 + n;
        if (q < 0) q = -q;
        if (q >= Lw) q = 2 * (Lw - 1) - q;
        const float w = 0.5f - 0.5f * cosf(6.283185307179586f * (float)n / 16.0f);
        x[n] = s[(long)b * Lw + q] * w;
    }
    float* o = out + i * ld;
    for (int k = 0; k < 9; ++k) {
        float re = 0.f, im = 0.f;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int kn = (k * n) & 15;
            const float ang = 6.283185307179586f * (float)kn / 16.0f;
            re += x[n] * cosf(ang);
            im -= x[n] * sinf(ang);
        }
        o[k] = re;
        o[9 + k] = im;
    }
    for (int c = 18; c < ld; ++c) o[c] = 0.f;
}

// conv_post output [B][F][ld] (9 log-magnitudes | 9 phase pre-activations) -> windowed time frames [B][F][16]
__global__ void hift_frames_kernel(const float* __restrict__ x, float* __restrict__ frames, long BF, int ld) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BF) return;
    const float* v = x + i * ld;
    float re[9], im[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float mag = fminf(expf(v[k]), 1e2f);
        const float ph = sinf(v[9 + k]);
        re[k] = mag * cosf(ph);
        im[k] = mag * sinf(ph);
    }
    for (int n = 0; n < 16; ++n) {
        float acc = re[0] + ((n & 1) ? -re[8] : re[8]);
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            const int kn = (k * n) & 15;
            const float ang = 6.283185307179586f * (float)kn / 16.0f;
            acc += 2.f * (re[k] * cosf(ang) - im[k] * sinf(ang));
        }
        const float w = 0.5f - 0.5f * cosf(6.283185307179586f * (float)n / 16.0f);
        frames[i * 16 + n] = acc * (1.0f / 16.0f) * w;
    }
}

// overlap-add + window-envelope normalisation + trim + clamp: out [B][Lw]
__global__ void hift_ola_kernel(const float* __restrict__ frames, float* __restrict__ out, int B, int F, long Lw, float limit) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * Lw) return;
    const int b = (int)(i / Lw);
    const long n = i - (long)b * Lw;
    const long p = n + 8;                         // index in the untrimmed signal
    float acc = 0.f, env = 0.f;
    const long f_hi = p / 4;
    for (int j = 0; j < 4; ++j) {
        const long f = f_hi - j;
        if (f < 0 || f >= F) continue;
        const int o = (int)(p - 4 * f);
        if (o < 0 || o >= 16) continue;
        const float w = 0.5f - 0.5f * cosf(6.283185307179586f * (float)o / 16.0f);
        acc += frames[((long)b * F + f) * 16 + o];
        env += w * w;
    }
    float v = acc / env;
    v = fminf(fmaxf(v, -limit), limit);
    out[i] = v;
}

__global__ void abs_copy_kernel(const float* __restrict__ src, long ld, float* __restrict__ dst, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = fabsf(src[i * ld]);
}

// writable operand planes of a model (hi, and lo in split mode)
struct ActOut {
    void* hi = nullptr;
    void* lo = nullptr;
    ActBuf in() const { ActBuf a; a.hi = hi; a.lo = lo; return a; }
};

int ew_cl(const float* x, ActOut y, int vd, long rows, int C, int ld, int mode, float slope, hipStream_t st) {
    const long n = rows * ld;
    if (vd != 1)
        hipLaunchKernelGGL(ew_cl_kernel<half_t>, dim3(cdiv(n, 256)), dim3(256), 0, st, x, (half_t*)y.hi,
                           is_split(vd) ? (half_t*)y.lo : nullptr, rows, C, ld, mode, slope);
    else
        hipLaunchKernelGGL(ew_cl_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, st, x, (float*)y.hi, (float*)nullptr, rows, C, ld,
                           mode, slope);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

int mel_to_cl(const float* mel, ActOut dst, int vd, int B, int C, int S, int ld, hipStream_t st) {
    const long n = (long)B * S * ld;
    if (vd != 1)
        hipLaunchKernelGGL(mel_to_cl_kernel<half_t>, dim3(cdiv(n, 256)), dim3(256), 0, st, mel, (half_t*)dst.hi,
                           is_split(vd) ? (half_t*)dst.lo : nullptr, B, C, S, ld);
    else
        hipLaunchKernelGGL(mel_to_cl_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, st, mel, (float*)dst.hi, (float*)nullptr, B, C, S, ld);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

struct SnakeP {
    float* a = nullptr;
    float* inv_b = nullptr;
};

int make_snake(const StateDict& sd, const std::string& alpha_key, const std::string& beta_key, int C, int mode, Arena& ar,
               hipStream_t st, SnakeP* out) {
    const auto* al = sd.get(alpha_key);
    if (require_shape(al, alpha_key, {C})) return 1;
    const float* be = nullptr;
    if (!beta_key.empty()) {
        const auto* b = sd.get(beta_key);
        if (require_shape(b, beta_key, {C})) return 1;
        be = b->data;
    }
    out->a = ar.alloc_n<float>(C, st);
    out->inv_b = ar.alloc_n<float>(C, st);
    if (!out->a || !out->inv_b) return 1;
    hipLaunchKernelGGL(snake_params_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, al->data, be, out->a, out->inv_b, C, mode);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

// One residual pair-stack "AMPBlock1 / ResBlock": for d in dils: xt = act(y); xt = c1(xt); xt = act(xt); xt = c2(xt); y = xt + y
struct ResBlockW {
    int k = 0, ch = 0, ndil = 0;
    int dil[3] = {1, 1, 1};
    ConvW c1[3], c2[3];
    SnakeP a1[3], a2[3];
};

}  // namespace

// ================================================================================================ BigVGAN
struct svc_bigvgan {
    svc_bigvgan_config_t cfg;
    int dtype;          // kgemm dtype: 1 = f32 (precision 0), 0 = f16 (precision 1)
    Arena wts, ws;
    ConvW conv_pre, conv_post;
    std::vector<ConvW> ups;
    std::vector<ResBlockW> blocks;
    SnakeP act_post;
    float taps[12];
    int cap_B = 0, cap_S = 0;
    ActOut mel_a, act_a;
    float *x, *y, *t, *xsum;
    int microbatch = 0;

    int reserve(int B, int S, hipStream_t st);
    int run(const float* mel, int B, int S, float* out, hipStream_t st);
};

namespace {
int act_cl_any(const float* x, int ld, ActOut y, int vd, const float* taps, const SnakeP& sp, int B, int C, int L, int mode,
               float slope, hipStream_t st, int lo_fmt = 0) {
    return act_cl_launch(x, ld, y.hi, is_split(vd) ? y.lo : nullptr, ld, vd != 1, taps, sp.a, sp.inv_b, B, C, L, mode, slope, st,
                         is_split(vd) ? lo_fmt : 0);
}

// runs one residual stack on stream buffer `x_in` (read only) producing the block output either into `y`
// (intermediate pairs) and, for the last pair, (y_last) * out_scale + res2 -> final_dst
int resblock_run(const ResBlockW& rb, int dtype, const float* taps, int act_mode, const float* x_in, float* y, float* t, ActOut act_a,
                 int B, int L, float out_scale, const float* res2, float* final_dst, hipStream_t st, ActOut act_b = ActOut()) {
    const int f16 = dtype;      // operand mode handed to the activation writers
    const int ld = cpad(rb.ch, dtype);
    const float* cur = x_in;
    // Pointwise Snake (HiFT) with fp16 operand planes: only the first activation is a kernel of its own; every later
    // one is the epilogue of the conv that produces its input (conv1 -> a2, conv2 -> next pair's a1), writing the hi / lo
    // planes straight into the other operand buffer.  The fp32 intermediate `t` disappears.
    const bool fuse = act_mode == 1 && dtype != 1 && act_b.hi != nullptr;
    // vd == 3: whoever writes a conv's operand planes (an activation kernel or the previous conv's epilogue) writes the lo
    // plane in the format that conv will read: fp8 byte pairs when it runs as fp16 + fp8 corrections (conv_p8_ok)
    auto fuse_into = [&](ConvRun& r, const SnakeP& sp, const ActOut& dst, bool next_p8) {
        r.post_a = sp.a; r.post_ib = sp.inv_b; r.post_n = rb.ch;
        r.c16 = reinterpret_cast<half_t*>(dst.hi); r.ldc16 = ld;
        r.c16_lo = is_split(dtype) ? reinterpret_cast<half_t*>(dst.lo) : nullptr;
        r.c16_lo_fmt = next_p8 ? 1 : 0;
    };
    for (int d = 0; d < rb.ndil; ++d) {
        const bool p8_1 = dtype == 3 && conv_p8_ok(rb.c1[d], L, rb.dil[d]);
        const bool p8_2 = dtype == 3 && conv_p8_ok(rb.c2[d], L, 1);
        if (!fuse || d == 0) {
            if (act_cl_any(cur, ld, act_a, f16, taps, rb.a1[d], B, rb.ch, L, act_mode, 0.f, st, p8_1)) return 1;
        }
        ConvRun r1;
        r1.a = act_a.in(); r1.B = B; r1.Lin = L; r1.Lout = L; r1.dilation = rb.dil[d];
        r1.pad_left = (rb.k * rb.dil[d] - rb.dil[d]) / 2;
        r1.p8 = p8_1;
        if (fuse) fuse_into(r1, rb.a2[d], act_b, p8_2);
        else { r1.c32 = t; r1.ldc32 = ld; }
        if (conv1d_run(rb.c1[d], r1, st)) return 1;
        if (!fuse) {
            if (act_cl_any(t, ld, act_a, f16, taps, rb.a2[d], B, rb.ch, L, act_mode, 0.f, st, p8_2)) return 1;
        }
        ConvRun r2;
        r2.a = fuse ? act_b.in() : act_a.in(); r2.B = B; r2.Lin = L; r2.Lout = L; r2.dilation = 1;
        r2.pad_left = (rb.k - 1) / 2;
        r2.p8 = p8_2;
        r2.res = cur; r2.ldres = ld;
        const bool last = d == rb.ndil - 1;
        if (last) {
            r2.out_scale = out_scale;
            r2.res2 = res2; r2.ldres2 = ld;
            r2.c32 = final_dst; r2.ldc32 = ld;
        } else {
            r2.c32 = y; r2.ldc32 = ld;
            if (fuse) fuse_into(r2, rb.a1[d + 1], act_a, dtype == 3 && conv_p8_ok(rb.c1[d + 1], L, rb.dil[d + 1]));
        }
        if (conv1d_run(rb.c2[d], r2, st)) return 1;
        cur = y;
    }
    return 0;
}
}  // namespace

int svc_bigvgan::reserve(int B, int S, hipStream_t st) {
    if (B <= cap_B && S <= cap_S) return 0;
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    ws.release();
    cap_B = std::max(B, cap_B);
    cap_S = std::max(S, cap_S);
    long L = cap_S;
    long max_el = L * cpad(cfg.upsample_initial_channel, dtype);
    int ch = cfg.upsample_initial_channel;
    for (int i = 0; i < cfg.num_upsamples; ++i) {
        L *= cfg.upsample_rates[i];
        ch /= 2;
        max_el = std::max(max_el, L * cpad(ch, dtype));
    }
    max_el *= cap_B;
    mel_a.hi = ws.alloc((size_t)cap_B * cap_S * cpad(cfg.num_mels, dtype) * vesize(dtype), st);
    mel_a.lo = ws.alloc((size_t)cap_B * cap_S * cpad(cfg.num_mels, dtype) * vesize(dtype), st);
    act_a.hi = ws.alloc((size_t)max_el * vesize(dtype), st);
    act_a.lo = ws.alloc((size_t)max_el * vesize(dtype), st);
    x = ws.alloc_n<float>(max_el, st);
    y = ws.alloc_n<float>(max_el, st);
    t = ws.alloc_n<float>(max_el, st);
    xsum = ws.alloc_n<float>(max_el, st);
    if (!mel_a.hi || !mel_a.lo || !act_a.hi || !act_a.lo || !x || !y || !t || !xsum) return 1;
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
}

int svc_bigvgan::run(const float* mel, int B, int S, float* out, hipStream_t st) {
    const int vd = dtype;
    if (mel_to_cl(mel, mel_a, vd, B, cfg.num_mels, S, cpad(cfg.num_mels, vd), st)) return 1;
    int ch = cfg.upsample_initial_channel;
    long L = S;
    {
        ConvRun r;
        r.a = mel_a.in(); r.B = B; r.Lin = S; r.Lout = S; r.pad_left = 3;
        r.c32 = xsum; r.ldc32 = cpad(ch, vd);
        if (conv1d_run(conv_pre, r, st)) return 1;
    }
    const int nk = cfg.num_kernels;
    for (int i = 0; i < cfg.num_upsamples; ++i) {
        // transposed conv reads the previous stage output (xsum); fp16 / split modes go through a cast copy
        ActBuf a;
        a.hi = xsum;
        if (vd != 1) {
            if (ew_cl(xsum, act_a, vd, (long)B * L, ch, cpad(ch, vd), 3, 0.f, st)) return 1;
            a = act_a.in();
        }
        if (convT_run(ups[i], a, B, (int)L, x, 0, 0, st)) return 1;
        L *= cfg.upsample_rates[i];
        ch /= 2;
        for (int j = 0; j < nk; ++j) {
            // x = mean_j block_j(x): block j's last conv writes (y_j) / nk + (j > 0 ? xsum : 0) into xsum
            if (resblock_run(blocks[i * nk + j], vd, taps, 0, x, y, t, act_a, B, (int)L, 1.0f / (float)nk,
                             j > 0 ? xsum : nullptr, xsum, st)) return 1;
        }
    }
    if (act_cl_any(xsum, cpad(ch, vd), act_a, vd, taps, act_post, B, ch, (int)L, 0, 0.f, st)) return 1;
    {
        ConvRun r;
        r.a = act_a.in(); r.B = B; r.Lin = (int)L; r.Lout = (int)L; r.pad_left = 3;
        r.c32 = out; r.ldc32 = 1; r.n_override = 1;
        r.act = cfg.use_tanh_at_final ? KG_ACT_TANH : KG_ACT_CLAMP;
        r.act_slope = 1.0f;
        if (conv1d_run(conv_post, r, st)) return 1;
    }
    return 0;
}

namespace {
int pack_resblock(const StateDict& sd, const std::string& p, int ch, int k, const int* dils, int ndil, int dtype, int snake_mode,
                  bool bigvgan_names, bool has_beta, Arena& ar, hipStream_t st, ResBlockW* rb) {
    rb->k = k;
    rb->ch = ch;
    rb->ndil = ndil;
    for (int d = 0; d < ndil; ++d) {
        rb->dil[d] = dils[d];
        const std::string ds = std::to_string(d);
        if (pack_conv1d(sd, p + ".convs1." + ds, ch, ch, k, true, dtype, ar, st, &rb->c1[d])) return 1;
        if (pack_conv1d(sd, p + ".convs2." + ds, ch, ch, k, true, dtype, ar, st, &rb->c2[d])) return 1;
        if (bigvgan_names) {
            const std::string a1 = p + ".activations." + std::to_string(2 * d) + ".act";
            const std::string a2 = p + ".activations." + std::to_string(2 * d + 1) + ".act";
            if (make_snake(sd, a1 + ".alpha", has_beta ? a1 + ".beta" : "", ch, snake_mode, ar, st, &rb->a1[d])) return 1;
            if (make_snake(sd, a2 + ".alpha", has_beta ? a2 + ".beta" : "", ch, snake_mode, ar, st, &rb->a2[d])) return 1;
        } else {
            if (make_snake(sd, p + ".activations1." + ds + ".alpha", "", ch, 2, ar, st, &rb->a1[d])) return 1;
            if (make_snake(sd, p + ".activations2." + ds + ".alpha", "", ch, 2, ar, st, &rb->a2[d])) return 1;
        }
    }
    return 0;
}
}  // namespace

// ================================================================================================ HiFT
struct svc_hift {
    svc_hift_config_t cfg;
    int dtype;
    int microbatch = 0;
    Arena wts, ws;
    ConvW f0_convs[5];
    float *f0_lin_w, *f0_lin_b, *src_lin_w, *src_lin_b;
    ConvW conv_pre, conv_post, ups[2], src_down[2];
    ResBlockW src_rb[2];
    std::vector<ResBlockW> blocks;
    int up_total;
    int cap_B = 0, cap_S = 0;
    ActOut mel_a, act_a, act_b, stft_a;
    void *f0_a, *f0_b;
    float *f0_buf, *s_buf, *stft32, *x, *y, *t, *xsum, *si, *post, *frames;
    double* prefix;

    int reserve(int B, int S, hipStream_t st);
    int run(const float* mel, const float* f0, const float* phase0, const float* noise, int B, int S, float* out, float* f0_out,
            hipStream_t st);
};

int svc_hift::reserve(int B, int S, hipStream_t st) {
    if (B <= cap_B && S <= cap_S) return 0;
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    ws.release();
    cap_B = std::max(B, cap_B);
    cap_S = std::max(S, cap_S);
    const long Bc = cap_B, Sc = cap_S;
    const int bc = cfg.base_channels;
    const long F = Sc * (up_total / cfg.istft_hop) + 1;       // frames of the source STFT / output iSTFT
    long max_el = Sc * cpad(bc, dtype);
    {
        long L = Sc;
        int ch = bc;
        for (int i = 0; i < cfg.num_upsamples; ++i) {
            L = L * cfg.upsample_rates[i] + (i == cfg.num_upsamples - 1 ? 1 : 0);
            ch /= 2;
            max_el = std::max(max_el, L * cpad(ch, dtype));
        }
    }
    max_el *= Bc;
    const int fc = cfg.f0_cond_channels;
    mel_a.hi = ws.alloc((size_t)Bc * Sc * cpad(cfg.in_channels, dtype) * vesize(dtype), st);
    mel_a.lo = ws.alloc((size_t)Bc * Sc * cpad(cfg.in_channels, dtype) * vesize(dtype), st);
    f0_a = ws.alloc((size_t)Bc * Sc * cpad(fc, 1) * 4, st);
    f0_b = ws.alloc((size_t)Bc * Sc * cpad(fc, 1) * 4, st);
    act_a.hi = ws.alloc((size_t)max_el * vesize(dtype), st);
    act_a.lo = ws.alloc((size_t)max_el * vesize(dtype), st);
    act_b.hi = ws.alloc((size_t)max_el * vesize(dtype), st);
    act_b.lo = ws.alloc((size_t)max_el * vesize(dtype), st);
    stft_a.hi = ws.alloc((size_t)Bc * F * 64 * vesize(dtype), st);
    stft_a.lo = ws.alloc((size_t)Bc * F * 64 * vesize(dtype), st);
    f0_buf = ws.alloc_n<float>(Bc * Sc, st);
    s_buf = ws.alloc_n<float>(Bc * Sc * up_total, st);
    stft32 = ws.alloc_n<float>(Bc * F * 64, st);
    x = ws.alloc_n<float>(max_el, st);
    y = ws.alloc_n<float>(max_el, st);
    t = ws.alloc_n<float>(max_el, st);
    xsum = ws.alloc_n<float>(max_el, st);
    si = ws.alloc_n<float>(max_el, st);
    post = ws.alloc_n<float>(Bc * F * 64, st);
    frames = ws.alloc_n<float>(Bc * F * 16, st);
    prefix = ws.alloc_n<double>(Bc * (cfg.nb_harmonics + 1) * Sc, st);
    if (!mel_a.hi || !mel_a.lo || !f0_a || !f0_b || !act_a.hi || !act_a.lo || !act_b.hi || !act_b.lo || !stft_a.hi || !stft_a.lo || !f0_buf || !s_buf || !stft32 || !x || !y || !t || !xsum || !si ||
        !post || !frames || !prefix)
        return 1;
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
}

int svc_hift::run(const float* mel, const float* f0_in, const float* phase0, const float* noise, int B, int S, float* out,
                  float* f0_out, hipStream_t st) {
    const int vd = dtype;
    const int NH = cfg.nb_harmonics + 1;
    const long Lw = (long)S * up_total;
    const int F = (int)(Lw / cfg.istft_hop) + 1;
    const int bc = cfg.base_channels;
    // ---- f0 (predictor runs in fp32 MFMA regardless of `precision`: the phase integrates f0 over seconds)
    const float* f0 = f0_in;
    if (!f0) {
        const int fc = cfg.f0_cond_channels, fld = cpad(fc, 1);
        ActOut f0in;
        f0in.hi = f0_a;
        if (mel_to_cl(mel, f0in, 1, B, cfg.in_channels, S, cpad(cfg.in_channels, 1), st)) return 1;
        void* src = f0_a;
        void* dst = f0_b;
        for (int i = 0; i < 5; ++i) {
            ConvRun r;
            r.a.hi = src; r.B = B; r.Lin = S; r.Lout = S; r.pad_left = 1;
            r.c32 = (float*)dst; r.ldc32 = fld; r.act = KG_ACT_ELU;
            if (conv1d_run(f0_convs[i], r, st)) return 1;
            std::swap(src, dst);
        }
        // classifier: Linear(fc -> 1) then abs
        if (small_linear_launch((const float*)src, fld, f0_lin_w, fc, f0_lin_b, f0_buf, 1, B * S, 1, fc, KG_ACT_NONE, st)) return 1;
        hipLaunchKernelGGL(abs_copy_kernel, dim3(cdiv((long)B * S, 256)), dim3(256), 0, st, f0_buf, 1L, f0_buf, (long)B * S);
        SVC_CHECK_HIP(hipGetLastError());
        f0 = f0_buf;
    }
    if (f0_out) SVC_CHECK_HIP(hipMemcpyAsync(f0_out, f0, (size_t)B * S * 4, hipMemcpyDeviceToDevice, st));
    // ---- harmonic source + its STFT
    hipLaunchKernelGGL(hift_phase_prefix_kernel, dim3(cdiv(B * NH, 64)), dim3(64), 0, st, f0, prefix, B, S, NH, up_total,
                       (float)cfg.sampling_rate);
    SVC_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(hift_source_kernel, dim3(cdiv((long)B * Lw, 256)), dim3(256), 0, st, f0, prefix, phase0, noise, src_lin_w,
                       src_lin_b, s_buf, B, S, NH, up_total, (float)cfg.sampling_rate, cfg.nsf_alpha, cfg.nsf_sigma,
                       cfg.nsf_voiced_threshold);
    SVC_CHECK_HIP(hipGetLastError());
    const int ld18 = cpad(cfg.istft_n_fft + 2, dtype);
    hipLaunchKernelGGL(hift_stft_kernel, dim3(cdiv((long)B * F, 256)), dim3(256), 0, st, s_buf, stft32, B, Lw, F, ld18);
    SVC_CHECK_HIP(hipGetLastError());
    ActBuf stft_in;
    stft_in.hi = stft32;
    if (vd != 1) {
        if (ew_cl(stft32, stft_a, vd, (long)B * F, ld18, ld18, 3, 0.f, st)) return 1;
        stft_in = stft_a.in();
    }
    // ---- main path
    if (mel_to_cl(mel, mel_a, vd, B, cfg.in_channels, S, cpad(cfg.in_channels, dtype), st)) return 1;
    {
        ConvRun r;
        r.a = mel_a.in(); r.B = B; r.Lin = S; r.Lout = S; r.pad_left = 3;
        r.c32 = xsum; r.ldc32 = cpad(bc, dtype);
        if (conv1d_run(conv_pre, r, st)) return 1;
    }
    int ch = bc;
    long L = S;
    const int nk = cfg.num_kernels;
    for (int i = 0; i < cfg.num_upsamples; ++i) {
        const bool last_up = i == cfg.num_upsamples - 1;
        if (ew_cl(xsum, act_a, vd, (long)B * L, ch, cpad(ch, dtype), 2, cfg.lrelu_slope, st)) return 1;
        const int u = cfg.upsample_rates[i];
        const long Lnew = L * u + (last_up ? 1 : 0);
        ch /= 2;
        const int ld = cpad(ch, dtype);
        if (last_up) {
            // ReflectionPad1d((1, 0)) (generator.py:413-414): each utterance's up-sampled rows land at row offset 1
            // of its (L*u + 1)-row sequence, then row 0 = row 2 (reflect).  One launch per utterance because the
            // one-row shift is not a whole number of GEMM output rows (each holds u samples).
            const size_t a_step = (size_t)L * ups[i].cin_pad * vesize(dtype);
            for (int b = 0; b < B; ++b) {
                ActBuf ab;
                ab.hi = (const char*)act_a.hi + b * a_step;
                ab.lo = (const char*)act_a.lo + b * a_step;
                if (convT_run(ups[i], ab, 1, (int)L, x + ((long)b * Lnew + 1) * ld, 0, 0, st)) return 1;
            }
            hipLaunchKernelGGL(copy_row_kernel, dim3(cdiv((long)B * ld, 256)), dim3(256), 0, st, x, B, (int)Lnew, ld, 2, 0);
            SVC_CHECK_HIP(hipGetLastError());
        } else {
            if (convT_run(ups[i], act_a.in(), B, (int)L, x, 0, 0, st)) return 1;
        }
        L = Lnew;
        // source fusion: si = source_resblock(source_down(s_stft)); x = x + si  (generator.py:416-419)
        {
            ConvRun r;
            r.a = stft_in; r.B = B; r.Lin = F; r.Lout = (int)L;
            const int dk = src_down[i].k;
            if (dk == 1) { r.stride = 1; r.pad_left = 0; }
            else { r.stride = dk / 2; r.pad_left = dk / 4; }
            r.c32 = si; r.ldc32 = ld;
            if (conv1d_run(src_down[i], r, st)) return 1;
        }
        // x <- x + source_resblock(si): last conv of the stack adds res2 = x and writes x
        if (resblock_run(src_rb[i], dtype, nullptr, 1, si, y, t, act_a, B, (int)L, 1.0f, x, x, st, act_b)) return 1;
        for (int j = 0; j < nk; ++j) {
            if (resblock_run(blocks[i * nk + j], dtype, nullptr, 1, x, y, t, act_a, B, (int)L, 1.0f / (float)nk,
                             j > 0 ? xsum : nullptr, xsum, st, act_b)) return 1;
        }
    }
    if (ew_cl(xsum, act_a, vd, (long)B * L, ch, cpad(ch, dtype), 2, 0.01f, st)) return 1;     // F.leaky_relu default slope
    {
        ConvRun r;
        r.a = act_a.in(); r.B = B; r.Lin = (int)L; r.Lout = (int)L; r.pad_left = 3;
        r.c32 = post; r.ldc32 = ld18;
        if (conv1d_run(conv_post, r, st)) return 1;
    }
    hipLaunchKernelGGL(hift_frames_kernel, dim3(cdiv((long)B * F, 256)), dim3(256), 0, st, post, frames, (long)B * F, ld18);
    SVC_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(hift_ola_kernel, dim3(cdiv((long)B * Lw, 256)), dim3(256), 0, st, frames, out, B, F, Lw, cfg.audio_limit);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

// ================================================================================================ C ABI
extern "C" {

int svc_bigvgan_create(const svc_bigvgan_config_t* cfg, const svc_tensor_desc_t* weights, int n_weights, void* stream,
                       svc_bigvgan_t** out) {
    SVC_REQUIRE(cfg && weights && out, "null argument");
    SVC_REQUIRE(cfg->num_upsamples >= 1 && cfg->num_upsamples <= 8 && cfg->num_kernels >= 1 && cfg->num_kernels <= 4, "config");
    hipStream_t st = (hipStream_t)stream;
    svc_bigvgan* m = new svc_bigvgan();
    m->cfg = *cfg;
    m->dtype = cfg->precision == 1 ? 0 : (cfg->precision == 2 ? 2 : (cfg->precision == 3 ? 3 : 1));   // operand mode: fp32 | fp16 | fp16x3 | fp16 + fp8 corrections
    StateDict sd(weights, n_weights);
    auto fail = [&]() { delete m; return 1; };
    const int c0 = cfg->upsample_initial_channel;
    if (pack_conv1d(sd, "conv_pre", c0, cfg->num_mels, 7, true, m->dtype, m->wts, st, &m->conv_pre)) return fail();
    m->ups.resize(cfg->num_upsamples);
    int ch = c0;
    const int snake_mode = cfg->snake_logscale ? (cfg->snakebeta ? 0 : 1) : 2;
    for (int i = 0; i < cfg->num_upsamples; ++i) {
        if (pack_convT(sd, "ups." + std::to_string(i) + ".0", ch, ch / 2, cfg->upsample_kernel_sizes[i], cfg->upsample_rates[i],
                       m->dtype, m->wts, st, &m->ups[i])) return fail();
        ch /= 2;
        for (int j = 0; j < cfg->num_kernels; ++j) {
            ResBlockW rb;
            if (pack_resblock(sd, "resblocks." + std::to_string(i * cfg->num_kernels + j), ch, cfg->resblock_kernel_sizes[j],
                              cfg->resblock_dilation_sizes[j], 3, m->dtype, snake_mode, true, cfg->snakebeta != 0, m->wts, st, &rb))
                return fail();
            m->blocks.push_back(rb);
        }
    }
    if (make_snake(sd, "activation_post.act.alpha", cfg->snakebeta ? "activation_post.act.beta" : "", ch, snake_mode, m->wts, st,
                   &m->act_post)) return fail();
    if (pack_conv1d(sd, "conv_post", 1, ch, 7, cfg->use_bias_at_final != 0, m->dtype, m->wts, st, &m->conv_post)) return fail();
    {
        const auto* f = sd.get("activation_post.upsample.filter");
        if (!f || StateDict::numel(f) != 12) { set_error("missing activation_post.upsample.filter (12 taps)"); return fail(); }
        if (hipMemcpy(m->taps, f->data, 48, hipMemcpyDeviceToHost) != hipSuccess) { set_error("filter copy failed"); return fail(); }
    }
    if (hipStreamSynchronize(st) != hipSuccess) { set_error("sync failed"); return fail(); }
    *out = m;
    return 0;
}

void svc_bigvgan_destroy(svc_bigvgan_t* m) { delete m; }

int svc_bigvgan_set_microbatch(svc_bigvgan_t* m, int utterances) {
    SVC_REQUIRE(m && utterances >= 0, "bad argument");
    m->microbatch = utterances;
    return 0;
}

int svc_bigvgan_forward(svc_bigvgan_t* m, const float* mel, int B, int S, float* out, void* stream) {
    SVC_REQUIRE(m && mel && out && B >= 1 && S >= 1, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    long total = 1;
    for (int i = 0; i < m->cfg.num_upsamples; ++i) total *= m->cfg.upsample_rates[i];
    // measured, vocoder alone, B = 32 x S = 430 (round 3, fp16p8): 4: 132.6, 8: 119.1, 16: 112.9, 32: 108.8 ms (round 1, small B = 64
    // end to end: 4: 27.3k, 8: 28.2k, 16: 28.5k frames/s)
    const int mb = m->microbatch > 0 ? m->microbatch : 32;
    for (int b0 = 0; b0 < B; b0 += mb) {
        const int nb = std::min(mb, B - b0);
        if (m->reserve(nb, S, st)) return 1;
        if (m->run(mel + (long)b0 * m->cfg.num_mels * S, nb, S, out + (long)b0 * S * total, st)) return 1;
    }
    return 0;
}

int svc_hift_create(const svc_hift_config_t* cfg, const svc_tensor_desc_t* weights, int n_weights, void* stream,
                    svc_hift_t** out) {
    SVC_REQUIRE(cfg && weights && out, "null argument");
    SVC_REQUIRE(cfg->num_upsamples == 2 && cfg->istft_n_fft == 16 && cfg->istft_hop == 4 && cfg->nb_harmonics + 1 <= 16,
                "HiFT: only the 2-stage, n_fft 16 / hop 4 generator of configs/hifigan.yml is supported");
    hipStream_t st = (hipStream_t)stream;
    svc_hift* m = new svc_hift();
    m->cfg = *cfg;
    m->dtype = cfg->precision == 1 ? 0 : (cfg->precision == 2 ? 2 : (cfg->precision == 3 ? 3 : 1));   // operand mode: fp32 | fp16 | fp16x3 | fp16 + fp8 corrections
    m->up_total = cfg->istft_hop;
    for (int i = 0; i < cfg->num_upsamples; ++i) m->up_total *= cfg->upsample_rates[i];
    StateDict sd(weights, n_weights);
    auto fail = [&]() { delete m; return 1; };
    const int bc = cfg->base_channels, fc = cfg->f0_cond_channels;
    for (int i = 0; i < 5; ++i) {
        if (pack_conv1d(sd, "f0_predictor.condnet." + std::to_string(2 * i), fc, i == 0 ? cfg->in_channels : fc, 3, true, 1,
                        m->wts, st, &m->f0_convs[i])) return fail();
    }
    {
        const auto* w = sd.get("f0_predictor.classifier.weight");
        const auto* b = sd.get("f0_predictor.classifier.bias");
        const auto* lw = sd.get("m_source.l_linear.weight");
        const auto* lb = sd.get("m_source.l_linear.bias");
        if (require_shape(w, "f0_predictor.classifier.weight", {1, fc}) || require_shape(b, "f0_predictor.classifier.bias", {1}) ||
            require_shape(lw, "m_source.l_linear.weight", {1, cfg->nb_harmonics + 1}) || require_shape(lb, "m_source.l_linear.bias", {1}))
            return fail();
        m->f0_lin_w = m->wts.alloc_n<float>(fc, st);
        m->f0_lin_b = m->wts.alloc_n<float>(1, st);
        m->src_lin_w = m->wts.alloc_n<float>(16, st);
        m->src_lin_b = m->wts.alloc_n<float>(1, st);
        if (!m->f0_lin_w || !m->f0_lin_b || !m->src_lin_w || !m->src_lin_b) return fail();
        (void)hipMemcpyAsync(m->f0_lin_w, w->data, fc * 4, hipMemcpyDeviceToDevice, st);
        (void)hipMemcpyAsync(m->f0_lin_b, b->data, 4, hipMemcpyDeviceToDevice, st);
        (void)hipMemcpyAsync(m->src_lin_w, lw->data, (cfg->nb_harmonics + 1) * 4, hipMemcpyDeviceToDevice, st);
        (void)hipMemcpyAsync(m->src_lin_b, lb->data, 4, hipMemcpyDeviceToDevice, st);
    }
    if (pack_conv1d(sd, "conv_pre", bc, cfg->in_channels, 7, true, m->dtype, m->wts, st, &m->conv_pre)) return fail();
    int ch = bc;
    // downsample rates of the source branch: cumulative products of [1] + reversed(ups)[:-1], reversed (generator.py:349-351)
    int down_rate[2] = {cfg->upsample_rates[1], 1};
    for (int i = 0; i < 2; ++i) {
        if (pack_convT(sd, "ups." + std::to_string(i), ch, ch / 2, cfg->upsample_kernel_sizes[i], cfg->upsample_rates[i], m->dtype,
                       m->wts, st, &m->ups[i])) return fail();
        ch /= 2;
        const int u = down_rate[i];
        const int dk = u == 1 ? 1 : 2 * u;
        if (pack_conv1d(sd, "source_downs." + std::to_string(i), ch, cfg->istft_n_fft + 2, dk, true, m->dtype, m->wts, st,
                        &m->src_down[i])) return fail();
        if (pack_resblock(sd, "source_resblocks." + std::to_string(i), ch, cfg->source_resblock_kernel_sizes[i],
                          cfg->source_resblock_dilation_sizes[i], 3, m->dtype, 2, false, false, m->wts, st, &m->src_rb[i]))
            return fail();
        for (int j = 0; j < cfg->num_kernels; ++j) {
            ResBlockW rb;
            if (pack_resblock(sd, "resblocks." + std::to_string(i * cfg->num_kernels + j), ch, cfg->resblock_kernel_sizes[j],
                              cfg->resblock_dilation_sizes[j], 3, m->dtype, 2, false, false, m->wts, st, &rb)) return fail();
            m->blocks.push_back(rb);
        }
    }
    if (pack_conv1d(sd, "conv_post", cfg->istft_n_fft + 2, ch, 7, true, m->dtype, m->wts, st, &m->conv_post)) return fail();
    if (hipStreamSynchronize(st) != hipSuccess) { set_error("sync failed"); return fail(); }
    *out = m;
    return 0;
}

void svc_hift_destroy(svc_hift_t* m) { delete m; }

int svc_hift_set_microbatch(svc_hift_t* m, int utterances) {
    SVC_REQUIRE(m && utterances >= 0, "bad argument");
    m->microbatch = utterances;
    return 0;
}

int svc_hift_forward(svc_hift_t* m, const float* mel, const float* f0, const float* phase0, const float* noise, int B, int S,
                     float* out, float* f0_out, void* stream) {
    SVC_REQUIRE(m && mel && phase0 && noise && out && B >= 1 && S >= 1, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int NH = m->cfg.nb_harmonics + 1;
    const long Lw = (long)S * m->up_total;
    // measured, vocoder alone, B = 32 x S = 430 (round 3, fp16p8): 4: 37.4, 8: 29.5, 16: 28.2, 32: 26.9 ms
    const int mb = m->microbatch > 0 ? m->microbatch : 32;
    for (int b0 = 0; b0 < B; b0 += mb) {
        const int nb = std::min(mb, B - b0);
        if (m->reserve(nb, S, st)) return 1;
        if (m->run(mel + (long)b0 * m->cfg.in_channels * S, f0 ? f0 + (long)b0 * S : nullptr, phase0 + (long)b0 * NH,
                   noise + (long)b0 * NH * Lw, nb, S, out + (long)b0 * Lw, f0_out ? f0_out + (long)b0 * S : nullptr, st))
            return 1;
    }
    return 0;
}

// ---- op-level entry points for the parity tests -------------------------------------------------------
int svc_op_conv1d(const float* x, const float* w, const float* bias, float* y, int B, int L, int Cin, int Cout, int k,
                  int dilation, int stride, int pad_left, int Lout, int pad_mode, int dtype_in, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int dtype = dtype_in;    // 0 f16, 1 f32, 2 f16x3
    Arena ar;
    svc_tensor_desc_t d[2];
    d[0].name = "c.weight"; d[0].data = w; d[0].ndim = 3; d[0].shape[0] = Cout; d[0].shape[1] = Cin; d[0].shape[2] = k; d[0].shape[3] = 1;
    d[1].name = "c.bias"; d[1].data = bias; d[1].ndim = 1; d[1].shape[0] = Cout; d[1].shape[1] = d[1].shape[2] = d[1].shape[3] = 1;
    StateDict sd(d, bias ? 2 : 1);
    ConvW cw;
    if (pack_conv1d(sd, "c", Cout, Cin, k, bias != nullptr, dtype, ar, st, &cw)) return 1;
    void* a = ar.alloc((size_t)B * L * cw.cin_pad * vesize(dtype), st);
    void* a_lo = ar.alloc((size_t)B * L * cw.cin_pad * vesize(dtype), st);
    float* c = ar.alloc_n<float>((size_t)B * Lout * cw.cout_pad, st);
    if (!a || !a_lo || !c) return 1;
    if (pack_any(gdt(dtype), x, a, 0, B * L, 1, Cin, Cin, 0, 1, cw.cin_pad, 0, 1, nullptr, st)) return 1;
    if (is_split(dtype))
        if (pack_f16_lo_launch(x, (half_t*)a_lo, B * L, 1, Cin, Cin, 0, 1, cw.cin_pad, 0, 1, nullptr, st)) return 1;
    ConvRun r;
    r.a.hi = a; r.a.lo = a_lo; r.B = B; r.Lin = L; r.Lout = Lout; r.dilation = dilation; r.stride = stride; r.pad_left = pad_left; r.pad_mode = pad_mode;
    r.c32 = c; r.ldc32 = cw.cout_pad;
    if (conv1d_run(cw, r, st)) return 1;
    if (pack_f32_launch(c, y, B * Lout, 1, Cout, cw.cout_pad, 0, 1, Cout, 0, 1, nullptr, st)) return 1;
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
}

int svc_op_conv_transpose1d(const float* x, const float* w, const float* bias, float* y, int B, int L, int Cin, int Cout, int k,
                            int stride, int dtype_in, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int dtype = dtype_in;
    SVC_REQUIRE(bias != nullptr, "bias required");
    Arena ar;
    svc_tensor_desc_t d[2];
    d[0].name = "c.weight"; d[0].data = w; d[0].ndim = 3; d[0].shape[0] = Cin; d[0].shape[1] = Cout; d[0].shape[2] = k; d[0].shape[3] = 1;
    d[1].name = "c.bias"; d[1].data = bias; d[1].ndim = 1; d[1].shape[0] = Cout; d[1].shape[1] = d[1].shape[2] = d[1].shape[3] = 1;
    StateDict sd(d, 2);
    ConvW cw;
    if (pack_convT(sd, "c", Cin, Cout, k, stride, dtype, ar, st, &cw)) return 1;
    void* a = ar.alloc((size_t)B * L * cw.cin_pad * vesize(dtype), st);
    void* a_lo = ar.alloc((size_t)B * L * cw.cin_pad * vesize(dtype), st);
    float* c = ar.alloc_n<float>((size_t)B * L * stride * cw.cout_pad, st);
    if (!a || !a_lo || !c) return 1;
    if (pack_any(gdt(dtype), x, a, 0, B * L, 1, Cin, Cin, 0, 1, cw.cin_pad, 0, 1, nullptr, st)) return 1;
    if (is_split(dtype))
        if (pack_f16_lo_launch(x, (half_t*)a_lo, B * L, 1, Cin, Cin, 0, 1, cw.cin_pad, 0, 1, nullptr, st)) return 1;
    ActBuf ab;
    ab.hi = a; ab.lo = a_lo;
    if (convT_run(cw, ab, B, L, c, 0, 0, st)) return 1;
    if (pack_f32_launch(c, y, B * L * stride, 1, Cout, cw.cout_pad, 0, 1, Cout, 0, 1, nullptr, st)) return 1;
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
}

}  // extern "C"
