// Fused anti-aliased SnakeBeta activation: 2x Kaiser-sinc up-sampling -> x + sin^2(a x)/b -> 2x down-sampling,
// replicate padding, one HBM read and one HBM write per element (algorithmic bytes 2*B*C*L*sizeof(T)).
//
// Closed form (verified against Activation1d in tests/golden/act.npz):
//   xp(j) = x[clamp(j-5)], u[m] = 2 sum_j xp(j) f[m+15-2j], s = u + sin^2(a u) / (b + 1e-9),
//   y[i]  = sum_t f[t] s[clamp(2i + t - 5)].
// Two layouts share the math:
//   * channels-first rows [B][C][L]   -- drop-in for the reference's CUDA extension seam
//     (anti_alias_activation_cuda.cu:43-246); one block per 1024-sample row segment staged in LDS.
//   * channels-last [B][L][ld]        -- the vocoder's internal layout (lanes = channels, coalesced);
//     each thread slides a register window along time.
#include <hip/hip_bf16.h>

#include "common.h"
#include "kernels.h"

namespace svc {
namespace {

constexpr int TI = 1024;   // outputs per block (channels-first kernel)

template <typename T> __device__ __forceinline__ float ld_f(const T* p) { return (float)*p; }
template <> __device__ __forceinline__ float ld_f<__hip_bfloat16>(const __hip_bfloat16* p) { return __bfloat162float(*p); }
template <typename T> __device__ __forceinline__ void st_f(T* p, float v) { *p = (T)v; }
template <> __device__ __forceinline__ void st_f<__hip_bfloat16>(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }

struct Taps { float f[12]; };

// sin^2 for the public seam: arbitrary caller data, so arguments beyond the polynomial's verified range take libm's path
__device__ __forceinline__ float sin_sq_any(float z) {
    if (__builtin_expect(fabsf(z) > 48.f, 0)) {
        const float sn = sinf(z);
        return sn * sn;
    }
    return sin_sq(z);
}

// Scalar form: any L / alignment (rows whose length is not a multiple of 4 cannot take aligned 16-byte accesses).
template <typename T>
__global__ __launch_bounds__(256) void aa_act_rows_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          const float* __restrict__ up12, const float* __restrict__ dn12,
                                                          const float* __restrict__ log_alpha,
                                                          const float* __restrict__ log_beta, int C, int L) {
    Taps up, dn;
#pragma unroll
    for (int i = 0; i < 12; ++i) { up.f[i] = up12[i]; dn.f[i] = dn12[i]; }
    __shared__ float sx[TI + 10];
    __shared__ float ss[2 * TI + 12];
    const int c = blockIdx.y, b = blockIdx.z;
    const int i0 = blockIdx.x * TI;
    const long row = ((long)b * C + c) * L;
    const float a = expf(log_alpha[c]);
    const float inv_b = 1.0f / (expf(log_beta[c]) + 1e-9f);
    const int n_out = min(TI, L - i0);

    for (int k = threadIdx.x; k < n_out + 10; k += 256) {
        int q = i0 - 5 + k;
        q = q < 0 ? 0 : (q > L - 1 ? L - 1 : q);
        sx[k] = ld_f(x + row + q);
    }
    __syncthreads();
    // s[m] for m in [2 i0 - 5, 2 (i0 + n_out) + 4]
    const int m_base = 2 * i0 - 5;
    for (int mm = threadIdx.x; mm < 2 * n_out + 10; mm += 256) {
        const int m = m_base + mm;
        float s = 0.f;
        if (m >= 0 && m < 2 * L) {
            // taps f[m + 15 - 2 j] in [0, 11]; largest j = (m + 15) >> 1
            const int jhi = (m + 15) >> 1;
            const int t0 = (m + 15) & 1;           // first tap index (0 for odd m, 1 for even m)
            float u = 0.f;
#pragma unroll
            for (int e = 0; e < 6; ++e) u += sx[jhi - e - i0] * up.f[t0 + 2 * e];
            u *= 2.0f;
            s = u + inv_b * sin_sq_any(a * u);
        }
        ss[mm] = s;
    }
    __syncthreads();
    for (int ii = threadIdx.x; ii < n_out; ii += 256) {
        const int i = i0 + ii;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            int k = 2 * i + t - 5;
            k = k < 0 ? 0 : (k > 2 * L - 1 ? 2 * L - 1 : k);
            acc += dn.f[t] * ss[k - m_base];
        }
        st_f(y + row + i, acc);
    }
}

// Vector form (L % 4 == 0: every row and every 4-sample group starts on a 16-byte (fp32) / 8-byte (16-bit) boundary).
// A block owns TI = 1024 outputs of one row, a thread 4 consecutive outputs Q .. Q + 3 (Q = i0 + 4 t):
//   stage 1  x[Q .. Q+3] -> LDS with ONE vector load per lane (16 B fp32 / 8 B fp16, bf16), 8-sample clamped halo each side;
//   stage 2  the 8 up-sampled activations s[2Q .. 2Q+7] from x[Q-3 .. Q+6] (three ds_read_b128) -> LDS (two ds_write_b128);
//            s halo (5 + 5 values) and out-of-row groups by the generic clamped form;
//   stage 3  y[Q .. Q+3] from s[2Q-5 .. 2Q+12] (six ds_read_b128), one vector store per lane.
// LDS instructions per output 3.25 (was 26), VALU ~55 (libm sinf was ~45 alone); lane-consecutive 16-byte LDS accesses are
// conflict-free.  Same closed form, fp32 accumulation; the sums run in another order than the scalar form (not bit-equal).
template <typename T> struct Vec4IO;
template <> struct Vec4IO<float> {
    static __device__ __forceinline__ float4v ld(const float* p) { return *reinterpret_cast<const float4v*>(p); }
    static __device__ __forceinline__ void st(float* p, float4v v) { *reinterpret_cast<float4v*>(p) = v; }
};
template <> struct Vec4IO<half_t> {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ float4v ld(const half_t* p) {
        const h4 h = *reinterpret_cast<const h4*>(p);
        return (float4v){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
    static __device__ __forceinline__ void st(half_t* p, float4v v) {
        *reinterpret_cast<h4*>(p) = (h4){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    }
};
template <> struct Vec4IO<__hip_bfloat16> {
    static __device__ __forceinline__ float4v ld(const __hip_bfloat16* p) {
        const uint2 r = *reinterpret_cast<const uint2*>(p);
        return (float4v){__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                         __uint_as_float(r.y & 0xffff0000u)};
    }
    static __device__ __forceinline__ void st(__hip_bfloat16* p, float4v v) {
        __hip_bfloat16 o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __float2bfloat16(v[i]);
        *reinterpret_cast<uint2*>(p) = *reinterpret_cast<const uint2*>(o);
    }
};

template <typename T>
__global__ __launch_bounds__(256) void aa_act_rows4_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                           const float* __restrict__ up12, const float* __restrict__ dn12,
                                                           const float* __restrict__ log_alpha,
                                                           const float* __restrict__ log_beta, int C, int L) {
    Taps up, dn;
#pragma unroll
    for (int i = 0; i < 12; ++i) { up.f[i] = up12[i]; dn.f[i] = dn12[i]; }
    // sx[k] = x[clamp(i0 - 8 + k)], k in [0, TI + 16);   ss[k] = s[clamp(2 i0 - 8 + k)], k in [0, 2 TI + 24) (edges unused)
    __shared__ __attribute__((aligned(16))) float sx[TI + 16];
    __shared__ __attribute__((aligned(16))) float ss[2 * TI + 24];
    const int c = blockIdx.y, b = blockIdx.z;
    const int i0 = blockIdx.x * TI;
    const long row = ((long)b * C + c) * L;
    const float a = expf(log_alpha[c]);
    const float inv_b = 1.0f / (expf(log_beta[c]) + 1e-9f);
    const int t = threadIdx.x;
    const int Q = i0 + 4 * t;
    const T* xr = x + row;
    if (Q < L) *reinterpret_cast<float4v*>(sx + 8 + 4 * t) = Vec4IO<T>::ld(xr + Q);
    else if (Q < L + 8) *reinterpret_cast<float4v*>(sx + 8 + 4 * t) = (float4v){1.f, 1.f, 1.f, 1.f} * ld_f(xr + L - 1);
    if (t < 16) {                                                   // halo: 8 below, 8 above the block's 1024 samples
        const int k = t < 8 ? t : TI + t;                           // sx index
        int q = i0 - 8 + k;
        q = q < 0 ? 0 : (q > L - 1 ? L - 1 : q);
        sx[k] = ld_f(xr + q);
    }
    __syncthreads();
    const int Lm = 2 * L - 1;
    auto s_at = [&](int m) -> float {                               // generic: any m, clamped to the row
        m = m < 0 ? 0 : (m > Lm ? Lm : m);
        const int q = m >> 1, odd = m & 1;
        const float* xs = sx + (q + 2 + odd - (i0 - 8));            // x[q + 2 + odd - e], taps f[(1 - odd) + 2 e]
        float u = 0.f;
#pragma unroll
        for (int e = 0; e < 6; ++e) u += xs[-e] * (odd ? up.f[2 * e] : up.f[2 * e + 1]);
        u *= 2.0f;
        return u + inv_b * sin_sq_any(a * u);
    };
    if (Q + 3 < L) {
        // x[Q-4 .. Q+7]; s[2q] = 2 sum_e x[q+2-e] f[1+2e], s[2q+1] = 2 sum_e x[q+3-e] f[2e] for q = Q .. Q+3
        float xv[12];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float4v v = *reinterpret_cast<const float4v*>(sx + 4 * t + 4 + 4 * j);
            xv[4 * j] = v[0]; xv[4 * j + 1] = v[1]; xv[4 * j + 2] = v[2]; xv[4 * j + 3] = v[3];
        }
        float sv[8];
#pragma unroll
        for (int dq = 0; dq < 4; ++dq) {
            float ue = 0.f, uo = 0.f;
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                ue += xv[4 + dq + 2 - e] * up.f[2 * e + 1];
                uo += xv[4 + dq + 3 - e] * up.f[2 * e];
            }
            ue *= 2.0f; uo *= 2.0f;
            sv[2 * dq] = ue + inv_b * sin_sq_any(a * ue);
            sv[2 * dq + 1] = uo + inv_b * sin_sq_any(a * uo);
        }
        *reinterpret_cast<float4v*>(ss + 8 + 8 * t) = (float4v){sv[0], sv[1], sv[2], sv[3]};
        *reinterpret_cast<float4v*>(ss + 12 + 8 * t) = (float4v){sv[4], sv[5], sv[6], sv[7]};
    } else if (Q < L + 4) {                                         // first group past the row end: clamped values
#pragma unroll
        for (int k = 0; k < 8; ++k) ss[8 + 8 * t + k] = s_at(2 * Q + k);
    }
    if (t >= 64 && t < 74) {                                        // s halo: m = 2 i0 - 5 .. 2 i0 - 1 and 2 (i0 + TI) .. + 4
        const int k = t - 64;
        const int mm = k < 5 ? 3 + k : 2 * TI + 8 + (k - 5);        // ss index
        ss[mm] = s_at(2 * i0 - 8 + mm);
    }
    __syncthreads();
    if (Q < L) {
        // s[2Q-8 .. 2Q+15] = ss[8 t .. 8 t + 23]; y[Q + d] = sum_t f[t] s[2 (Q + d) + t - 5] = sw[3 + 2 d + t]
        float sw[24];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float4v v = *reinterpret_cast<const float4v*>(ss + 8 * t + 4 * j);
            sw[4 * j] = v[0]; sw[4 * j + 1] = v[1]; sw[4 * j + 2] = v[2]; sw[4 * j + 3] = v[3];
        }
        float4v o;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 12; ++k) acc += dn.f[k] * sw[3 + 2 * d + k];
            o[d] = acc;
        }
        Vec4IO<T>::st(y + row + Q, o);
    }
}


// ---- channels-last: x [B][L][ld] fp32 -> y fp32 and/or fp16 (the next conv's A operand).
// Thread = (channel, time segment of SEG outputs); a rolling window of s values lives in registers.
// mode 0: anti-aliased snake (a, inv_b per channel); mode 1: plain snake x + inv_b sin^2(a x) (HiFT);
// mode 2: leaky relu with `slope`.
constexpr int SEG = 64;

template <typename OutT>
__global__ __launch_bounds__(256) void act_cl_kernel(const float* __restrict__ x, long ldx, OutT* __restrict__ y,
                                                     OutT* __restrict__ ylo, long ldy, Taps ft, const float* __restrict__ pa, const float* __restrict__ pinvb,
                                                     int C, int L, int mode, float slope, int lo_fmt) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int seg = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    const int i0 = seg * SEG;
    if (c >= ldy || i0 >= L) return;
    const float* xr = x + (long)b * L * ldx + c;
    OutT* yr = y + (long)b * L * ldy + c;
    OutT* yl = ylo ? ylo + (long)b * L * ldy + c : nullptr;
    const int i1 = min(i0 + SEG, L);
    // lo plane: the fp16 residual, or (lo_fmt 1, fp16 planes only) the fp8 byte pair of the fp16 + fp8-corrections convs
    auto put_lo = [&](OutT* dst, float v, OutT h) {
        if constexpr (sizeof(OutT) == 2) {
            if (lo_fmt) { *reinterpret_cast<unsigned short*>(dst) = (unsigned short)lo_pair_p8((float)h, v - (float)h); return; }
        }
        *dst = (OutT)(v - (float)h);
    };
    auto put = [&](int i, float v) {
        const OutT h = (OutT)v;
        yr[(long)i * ldy] = h;
        if (yl) put_lo(yl + (long)i * ldy, v, h);
    };
    if (c >= C) {   // pad channels of the channels-last layout stay zero
        for (int i = i0; i < i1; ++i) put(i, 0.f);
        return;
    }
    if (mode != 0) {
        const float a = mode == 1 ? pa[c] : 0.f, ib = mode == 1 ? pinvb[c] : 0.f;
        // 8 independent row loads in flight per thread (the loop is latency-, not bandwidth-bound otherwise)
        for (int ib8 = i0; ib8 < i1; ib8 += 8) {
            float v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v8[u] = ib8 + u < i1 ? xr[(long)(ib8 + u) * ldx] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (ib8 + u >= i1) break;
                const float v = v8[u];
                float o;
                if (mode == 1) o = v + ib * sin_sq(a * v);
                else o = v > 0.f ? v : v * slope;
                put(ib8 + u, o);
            }
        }
        return;
    }
    // sin on the transcendental unit (v_sin_f32 takes revolutions): one 8-cycle instruction instead of the 12-instruction
    // polynomial -- PMC showed this kernel at 83 % VALU issue (89 instructions per output), so instruction count is its
    // time.  BigVGAN S = 430 against the reference: 1.726e-6 with it, 1.728e-6 with the polynomial (fp16x3).
    const float a = pa[c] * 0.15915494309189535f, ib = pinvb[c];
    const int Lm = 2 * L - 1;
    auto snake = [&](float u) -> float {
        const float sn = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(a * u));     // fract: v_sin_f32 is defined on +-256 revolutions only
        return u + ib * sn * sn;
    };
    float f2[12];     // up-sampling taps with the x2 (ratio) folded in: exact
#pragma unroll
    for (int t = 0; t < 12; ++t) f2[t] = 2.0f * ft.f[t];
    float sw[12];     // s window of output i: sw[k] = s[clamp(2 i - 5 + k)], k = 0..11
    float xw[6];      // x window feeding the next two s values: xw[k] = x[clamp(i + 1 + k)], k = 0..5
    // Six outputs per trip: the s window advances by 2 and the x window by 1 per output, so after 6 outputs both
    // circular buffers are back at their base and every index below is a compile-time constant (no register shuffling).
    // logical sw[t] of output u = sw[(2u + t) % 12], logical xw[k] = xw[(u + k) % 6].
    auto trip = [&](int ib0, const float (&xn)[6], auto&& put_at) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int i = ib0 + u;
            if (i >= i1) break;
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < 12; ++t) acc += ft.f[t] * sw[(2 * u + t) % 12];
            put_at(i, acc);
            if (i + 1 < i1) {
                // new s[2i+7] (odd m: taps 0,2,..,10) and s[2i+8] (even m: taps 1,3,..,11), both from x[i+1 .. i+6]
                float u1 = 0.f, u2 = 0.f;
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    u1 += xw[(u + 5 - e) % 6] * f2[2 * e];
                    u2 += xw[(u + 5 - e) % 6] * f2[2 * e + 1];
                }
                const float s1 = (2 * i + 7 <= Lm) ? snake(u1) : sw[(2 * u + 11) % 12];
                const float s2 = (2 * i + 8 <= Lm) ? snake(u2) : s1;
                sw[(2 * u) % 12] = s1;          // the two oldest entries become logical sw[10], sw[11] of the next output
                sw[(2 * u + 1) % 12] = s2;
                xw[u % 6] = xn[u];
            }
        }
    };
    if (i0 >= 8 && i0 + SEG + 16 <= L) {
        // Interior segment (all but the first and last of a sequence): no index is ever clamped, so the rows are walked
        // with two running pointers -- the generic path below pays a clamp and a 64-bit multiply per load and per store,
        // which was ~45 % of this kernel's instructions.  Same arithmetic in the same order: bit-identical.
        float xrow[12];                                  // x[i0 - 5 .. i0 + 6]
        const float* xp = xr + (long)(i0 - 5) * ldx;
#pragma unroll
        for (int k = 0; k < 12; ++k) { xrow[k] = *xp; xp += ldx; }      // leaves xp at row i0 + 7
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            // m = 2 i0 - 5 + k: jhi = i0 + 5 + (k >> 1), t0 = k & 1; x[jhi - 5 - e] = xrow[(k >> 1) + 5 - e]
            float u = 0.f;
#pragma unroll
            for (int e = 0; e < 6; ++e) u += xrow[(k >> 1) + 5 - e] * f2[(k & 1) + 2 * e];
            sw[k] = snake(u);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) xw[k] = xrow[6 + k];
        OutT* yp = yr + (long)i0 * ldy;
        OutT* ylp = yl ? yl + (long)i0 * ldy : nullptr;
        auto put_walk = [&](int, float v) {
            const OutT h = (OutT)v;
            *yp = h;
            yp += ldy;
            if (ylp) { put_lo(ylp, v, h); ylp += ldy; }
        };
        for (int ib0 = i0; ib0 < i1; ib0 += 6) {
            float xn[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) { xn[q] = *xp; xp += ldx; }     // rows ib0 + 7 .. ib0 + 12 <= i0 + 72 < L
            trip(ib0, xn, put_walk);
        }
        return;
    }
    auto xat = [&](int q) -> float {
        q = q < 0 ? 0 : (q > L - 1 ? L - 1 : q);
        return xr[(long)q * ldx];
    };
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        int m = 2 * i0 - 5 + k;
        m = m < 0 ? 0 : (m > Lm ? Lm : m);
        const int jhi = (m + 15) >> 1, t0 = (m + 15) & 1;
        float u = 0.f;
#pragma unroll
        for (int e = 0; e < 6; ++e) u += xat(jhi - 5 - e) * f2[t0 + 2 * e];
        sw[k] = snake(u);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) xw[k] = xat(i0 + 1 + k);
    for (int ib0 = i0; ib0 < i1; ib0 += 6) {
        float xn[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) xn[q] = xat(ib0 + 7 + q);
        trip(ib0, xn, put);
    }
}


// fp16-output variant of the POINTWISE modes (1, 2) with lanes = channel PAIRS: 8-byte loads and 4-byte (half2) stores
// of the hi and lo planes instead of 2-byte stores.
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void act_cl2_kernel(const float* __restrict__ x, long ldx, half_t* __restrict__ y,
                                                      half_t* __restrict__ ylo, long ldy, Taps ft, const float* __restrict__ pa,
                                                      const float* __restrict__ pinvb, int C, int L, int mode, float slope, int lo_fmt) {
    const int c = (blockIdx.x * 64 + (threadIdx.x & 63)) * 2;
    const int seg = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    const int i0 = seg * SEG;
    if (c >= ldy || i0 >= L) return;
    const float* xr = x + (long)b * L * ldx + c;
    half_t* yr = y + (long)b * L * ldy + c;
    half_t* yl = ylo ? ylo + (long)b * L * ldy + c : nullptr;
    const int i1 = min(i0 + SEG, L);
    const bool ok0 = c < C, ok1 = c + 1 < C;              // pad channels of the channels-last layout stay zero
    auto put = [&](int i, float2v v) {
        if (!ok0) v[0] = 0.f;
        if (!ok1) v[1] = 0.f;
        const half2v h = {(half_t)v[0], (half_t)v[1]};
        *reinterpret_cast<half2v*>(yr + (long)i * ldy) = h;
        if (yl) {
            if (lo_fmt) {
                *reinterpret_cast<unsigned*>(yl + (long)i * ldy) =
                    lo_pair_p8((float)h[0], v[0] - (float)h[0]) | (lo_pair_p8((float)h[1], v[1] - (float)h[1]) << 16);
            } else {
                const half2v l = {(half_t)(v[0] - (float)h[0]), (half_t)(v[1] - (float)h[1])};
                *reinterpret_cast<half2v*>(yl + (long)i * ldy) = l;
            }
        }
    };
    if (!ok0) {
        for (int i = i0; i < i1; ++i) put(i, (float2v){0.f, 0.f});
        return;
    }
    const float a0 = mode != 2 ? pa[c] : 0.f, a1 = (mode != 2 && ok1) ? pa[c + 1] : 0.f;
    const float b0 = mode != 2 ? pinvb[c] : 0.f, b1 = (mode != 2 && ok1) ? pinvb[c + 1] : 0.f;
    auto snake = [&](float2v u) -> float2v {
        return (float2v){u[0] + b0 * sin_sq(a0 * u[0]), u[1] + b1 * sin_sq(a1 * u[1])};
    };
    auto ld2 = [&](int q) -> float2v { return *reinterpret_cast<const float2v*>(xr + (long)q * ldx); };
    if (mode != 0) {
        for (int ib8 = i0; ib8 < i1; ib8 += 8) {
            float2v v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v8[u] = ib8 + u < i1 ? ld2(ib8 + u) : (float2v){0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (ib8 + u >= i1) break;
                const float2v v = v8[u];
                float2v o;
                if (mode == 1) o = snake(v);
                else o = (float2v){v[0] > 0.f ? v[0] : v[0] * slope, v[1] > 0.f ? v[1] : v[1] * slope};
                put(ib8 + u, o);
            }
        }
        return;
    }
}

}  // namespace

int aa_act_rows_launch(const void* x, void* y, const float* up, const float* dn, const float* log_alpha,
                       const float* log_beta, int B, int C, int L, int dtype, hipStream_t st) {
    dim3 grid(cdiv(L, TI), C, B);
    if (B * C == 0 || L == 0) return 0;
    const bool al16 = ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
    if (L % 4 == 0 && al16 && dtype >= 0 && dtype <= 2) {
        if (dtype == 0)
            hipLaunchKernelGGL(aa_act_rows4_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (float*)y, up, dn, log_alpha, log_beta, C, L);
        else if (dtype == 1)
            hipLaunchKernelGGL(aa_act_rows4_kernel<half_t>, grid, dim3(256), 0, st, (const half_t*)x, (half_t*)y, up, dn, log_alpha, log_beta, C, L);
        else
            hipLaunchKernelGGL(aa_act_rows4_kernel<__hip_bfloat16>, grid, dim3(256), 0, st, (const __hip_bfloat16*)x, (__hip_bfloat16*)y, up, dn,
                               log_alpha, log_beta, C, L);
        SVC_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (dtype == 0)
        hipLaunchKernelGGL(aa_act_rows_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (float*)y, up, dn, log_alpha, log_beta, C, L);
    else if (dtype == 1)
        hipLaunchKernelGGL(aa_act_rows_kernel<half_t>, grid, dim3(256), 0, st, (const half_t*)x, (half_t*)y, up, dn, log_alpha, log_beta, C, L);
    else if (dtype == 2)
        hipLaunchKernelGGL(aa_act_rows_kernel<__hip_bfloat16>, grid, dim3(256), 0, st, (const __hip_bfloat16*)x, (__hip_bfloat16*)y, up, dn, log_alpha, log_beta, C, L);
    else {
        set_error("anti_alias_act: dtype must be 0 (fp32), 1 (fp16) or 2 (bf16)");
        return 1;
    }
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

int act_cl_launch(const float* x, long ldx, void* y, void* y_lo, long ldy, int out_f16, const float* taps12_host, const float* a,
                  const float* inv_b, int B, int C, int L, int mode, float slope, hipStream_t st, int lo_fmt) {
    Taps ft;
    for (int i = 0; i < 12; ++i) ft.f[i] = taps12_host ? taps12_host[i] : 0.f;
    dim3 grid(cdiv(ldy, 64), cdiv(cdiv(L, SEG), 4), B);
    // pointwise modes only: the anti-aliased window doubles its registers per thread in the pair form and measured slower
    if (out_f16 && mode != 0 && (ldx % 2) == 0 && (ldy % 2) == 0) {
        dim3 grid2(cdiv(ldy, 128), cdiv(cdiv(L, SEG), 4), B);
        hipLaunchKernelGGL(act_cl2_kernel, grid2, dim3(256), 0, st, x, ldx, (half_t*)y, (half_t*)y_lo, ldy, ft, a, inv_b, C, L, mode, slope, lo_fmt);
    } else if (out_f16)
        hipLaunchKernelGGL(act_cl_kernel<half_t>, grid, dim3(256), 0, st, x, ldx, (half_t*)y, (half_t*)y_lo, ldy, ft, a, inv_b, C, L, mode, slope, lo_fmt);
    else
        hipLaunchKernelGGL(act_cl_kernel<float>, grid, dim3(256), 0, st, x, ldx, (float*)y, (float*)nullptr, ldy, ft, a, inv_b, C, L, mode, slope, 0);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace svc
