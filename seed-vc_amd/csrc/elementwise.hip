// HBM-bound kernels of the DiT / CFM path: adaptive RMSNorm / LayerNorm (wave-shuffle reductions),
// ODE state update with classifier-free guidance, layout changes, timestep features.
#include "common.h"
#include "kernels.h"

namespace svc {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One wave per row.  y = rmsnorm(x) * gamma * (add_one + w) + b   -> fp16
__global__ __launch_bounds__(256) void rmsnorm_mod_kernel(const float* __restrict__ x, long ldx, half_t* __restrict__ y,
                                                          long ldy, const float* __restrict__ gamma,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          long ld_wb, int add_one, int rows, int D, int seq_rows,
                                                          float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float2* xr = reinterpret_cast<const float2*>(x + (long)row * ldx);
    const int n2 = D >> 1;
    // every operand is requested before the reduction: one memory round trip per row instead of two dependent ones
    // (the kernel only runs on small launches, where that latency is what it costs)
    const int seq = row / seq_rows;
    const float2* g2 = reinterpret_cast<const float2*>(gamma);
    const float2* w2 = w ? reinterpret_cast<const float2*>(w + (long)seq * ld_wb) : nullptr;
    const float2* b2 = b ? reinterpret_cast<const float2*>(b + (long)seq * ld_wb) : nullptr;
    const float one = add_one ? 1.f : 0.f;
    float2 v[8], g[8], ww[8], bb[8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = lane + 64 * i;
        if (e < n2) {
            v[i] = xr[e];
            g[i] = g2[e];
            ww[i] = w2 ? w2[e] : float2{0.f, 0.f};
            bb[i] = b2 ? b2[e] : float2{0.f, 0.f};
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (lane + 64 * i < n2) ss += v[i].x * v[i].x + v[i].y * v[i].y;
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)D + eps);
    half2v* yr = reinterpret_cast<half2v*>(y + (long)row * ldy);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = lane + 64 * i;
        if (e < n2) {
            float o0 = v[i].x * rs * g[i].x, o1 = v[i].y * rs * g[i].y;
            if (w2) { o0 *= (one + ww[i].x); o1 *= (one + ww[i].y); }
            if (b2) { o0 += bb[i].x; o1 += bb[i].y; }
            yr[e] = (half2v){(half_t)o0, (half_t)o1};
        }
    }
}

// LayerNorm without affine, then * (1 + scale) + shift  -> fp16   (DiT FinalLayer)
__global__ __launch_bounds__(256) void layernorm_mod_kernel(const float* __restrict__ x, long ldx, half_t* __restrict__ y,
                                                            long ldy, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int rows, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float2* xr = reinterpret_cast<const float2*>(x + (long)row * ldx);
    const int n2 = D >> 1;
    float2 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = lane + 64 * i;
        if (e < n2) { v[i] = xr[e]; s += v[i].x + v[i].y; }
    }
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = lane + 64 * i;
        if (e < n2) { const float a = v[i].x - mean, c = v[i].y - mean; ss += a * a + c * c; }
    }
    const float rs = rsqrtf(wave_sum(ss) / (float)D + eps);
    const float2* sc2 = reinterpret_cast<const float2*>(scale);
    const float2* sh2 = reinterpret_cast<const float2*>(shift);
    half2v* yr = reinterpret_cast<half2v*>(y + (long)row * ldy);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = lane + 64 * i;
        if (e < n2) {
            const float2 sc = sc2[e], sh = sh2[e];
            yr[e] = (half2v){(half_t)((v[i].x - mean) * rs * (1.f + sc.x) + sh.x),
                             (half_t)((v[i].y - mean) * rs * (1.f + sc.y) + sh.y)};
        }
    }
}

// (B, C, T) fp32 channel-major -> token-major [B][T_rows][ld] fp32 and/or fp16 (zero where t >= t_valid)
__global__ void bct_to_btc_kernel(const float* __restrict__ src, int B, int C, int T_src, float* dst32, long ld32,
                                  half_t* dst16, long ld16, int seq_rows, int t_valid, float scale) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        tile[i][tx] = (c < C && t < T_src && t < t_valid) ? src[((long)b * C + c) * T_src + t] * scale : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        if (t < seq_rows && c < C) {
            const float v = tile[tx][i];
            if (dst32) dst32[((long)b * seq_rows + t) * ld32 + c] = v;
            if (dst16) dst16[((long)b * seq_rows + t) * ld16 + c] = (half_t)v;
        }
    }
}

// token-major [B][seq_rows][ld] fp32 -> (B, C, T) fp32
__global__ void btc_to_bct_kernel(const float* __restrict__ src, long ld, int seq_rows, float* __restrict__ dst, int B,
                                  int C, int T) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        tile[i][tx] = (t < T && c < C) ? src[((long)b * seq_rows + t) * ld + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        if (c < C && t < T) dst[((long)b * C + c) * T + t] = tile[tx][i];
    }
}

__global__ void cast_rows_kernel(const float* __restrict__ src, long lds_, half_t* __restrict__ dst, long ldd, int rows,
                                 int cols) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    dst[(long)r * ldd + c] = (half_t)src[(long)r * lds_ + c];
}

// sinusoidal timestep features: out[s][0:128] = cos(1000 t_s f), out[s][128:256] = sin(...)
__global__ void timestep_feat_kernel(const float* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ out,
                                     int n) {
    const int s = blockIdx.x, i = threadIdx.x;   // 128 threads
    if (s >= n) return;
    const float a = 1000.0f * t[s] * freqs[i];
    out[s * 256 + i] = cosf(a);
    out[s * 256 + 128 + i] = sinf(a);
}

__global__ void silu_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = x[i]; y[i] = v / (1.f + expf(-v)); }
}

// per step: write prefix-token rows (time / style) of every stream's sequences and zero the pad rows.
// xin [n_stream*B][seq_rows][D]; tok_time [D] (this step); tok_style [n_stream][B][D]
__global__ void prefix_rows_kernel(float* __restrict__ xin, int n_seq, int seq_rows, int D, int n_prefix, int t_rows,
                                   const float* __restrict__ tok_time, const float* __restrict__ tok_style,
                                   int time_first) {
    const int seq = blockIdx.x;
    const int n_special = n_prefix + (seq_rows - t_rows);
    for (int idx = threadIdx.x; idx < n_special * D; idx += blockDim.x) {
        const int r = idx / D, d = idx - r * D;
        int row;
        float v = 0.f;
        if (r < n_prefix) {
            row = r;
            const bool is_time = time_first && r == 0;
            v = is_time ? tok_time[d] : tok_style[(long)seq * D + d];
        } else {
            row = t_rows + (r - n_prefix);
        }
        xin[((long)seq * seq_rows + row) * D + d] = v;
    }
}

// x += dt * ((1 + w_a + w_b) * v0 - w_a * v_a - w_b * v_b); zero the prompt region; refresh the fp16 copy.
// v: [n_stream][B][seq_rows_v][ldv] (stream 0 = fully conditional); x: [B][T_rows][ldx]
__global__ void euler_cfg_kernel(float* __restrict__ x, long ldx, half_t* __restrict__ x16, long ldx16, int x_rows,
                                 const float* __restrict__ v, long ldv, long v_stream_stride, int v_rows, int B, int T,
                                 int C, const int* __restrict__ prompt_len, float dt, float c0, float ca, float cb,
                                 int stream_a, int stream_b) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * T * C;
    if (i >= total) return;
    const int c = (int)(i % C);
    const long bt = i / C;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const long vrow = ((long)b * v_rows + t) * ldv + c;
    float d = c0 * v[vrow];
    if (stream_a >= 0) d -= ca * v[vrow + stream_a * v_stream_stride];
    if (stream_b >= 0) d -= cb * v[vrow + stream_b * v_stream_stride];
    const long xo = ((long)b * x_rows + t) * ldx + c;
    float nx = x[xo] + dt * d;
    if (t < prompt_len[b]) nx = 0.f;
    x[xo] = nx;
    x16[((long)b * x_rows + t) * ldx16 + c] = (half_t)nx;
}

// generic strided pack copy with optional per-dim0 scale: dst[i0*d0 + i1*d1 + i2*d2] = src[i0*s0+i1*s1+i2*s2]*scale[i0]
// lo_part (fp16 only): store the residual v - float(half(v)) instead of half(v) (split-precision operands)
template <typename OutT>
__global__ void pack_kernel(const float* __restrict__ src, OutT* __restrict__ dst, int n0, int n1, int n2, long s0,
                            long s1, long s2, long d0, long d1, long d2, const float* __restrict__ scale, int lo_part) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n0 * n1 * n2;
    if (i >= total) return;
    const int i2 = (int)(i % n2);
    const long r = i / n2;
    const int i1 = (int)(r % n1), i0 = (int)(r / n1);
    float v = src[i0 * s0 + i1 * s1 + i2 * s2];
    if (scale) v *= scale[i0];
    if (lo_part) v = v - (float)(half_t)v;
    dst[i0 * d0 + i1 * d1 + i2 * d2] = (OutT)v;
}

// weight-norm scale: out[i] = g[i] / ||v[i, :]||   (one block per row)
__global__ void wn_scale_kernel(const float* __restrict__ g, const float* __restrict__ v, long row_elems, float* __restrict__ out) {
    __shared__ float red[256];
    const int i = blockIdx.x;
    float ss = 0.f;
    for (long e = threadIdx.x; e < row_elems; e += blockDim.x) { const float a = v[i * row_elems + e]; ss += a * a; }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[i] = g[i] / sqrtf(red[0]);
}

// small dense fp32 mat-vec batch used at pack time / per call for conditioning vectors:
// out[r][n] = act(bias[n] + sum_k in[r][k] * W[n][k])      (one wave per output element group)
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ in, long ld_in, const float* __restrict__ W,
                                                           long ldw, const float* __restrict__ bias, float* __restrict__ out,
                                                           long ld_out, int R, int N, int K, int act) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= R * N) return;
    const int r = wave / N, n = wave - r * N;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += in[(long)r * ld_in + k] * W[(long)n * ldw + k];
    s = wave_sum(s);
    if (lane == 0) {
        if (bias) s += bias[n];
        if (act == KG_ACT_SILU) s = s / (1.f + expf(-s));
        else if (act != KG_ACT_NONE) s = act_apply(s, act, 0.f);
        out[(long)r * ld_out + n] = s;
    }
}

__global__ void add_vec_kernel(float* __restrict__ dst, const float* __restrict__ a, const float* __restrict__ b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = a[i] + (b ? b[i] : 0.f);
}

// dst[seq][n] = a[seq][n] + bvec[n]   (per-sequence additive vectors for the merge GEMM)
__global__ void add_rowvec_kernel(float* __restrict__ dst, const float* __restrict__ a, long lda, const float* __restrict__ bvec,
                                  int n_seq, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seq * n) return;
    const int s = i / n, c = i - s * n;
    dst[(long)s * n + c] = (a ? a[(long)s * lda + c] : 0.f) + bvec[c];
}

}  // namespace

#define LAUNCH_CHECK() SVC_CHECK_HIP(hipGetLastError())

int rmsnorm_mod_launch(const float* x, long ldx, half_t* y, long ldy, const float* gamma, const float* w, const float* b,
                       long ld_wb, int add_one, int rows, int D, int seq_rows, float eps, hipStream_t st) {
    SVC_REQUIRE(D % 2 == 0 && D <= 1024 && ldx % 2 == 0 && ldy % 2 == 0, "rmsnorm shape");
    hipLaunchKernelGGL(rmsnorm_mod_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, ldx, y, ldy, gamma, w, b, ld_wb,
                       add_one, rows, D, seq_rows, eps);
    LAUNCH_CHECK();
    return 0;
}

int layernorm_mod_launch(const float* x, long ldx, half_t* y, long ldy, const float* scale, const float* shift, int rows,
                         int D, float eps, hipStream_t st) {
    SVC_REQUIRE(D % 2 == 0 && D <= 1024, "layernorm shape");
    hipLaunchKernelGGL(layernorm_mod_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, ldx, y, ldy, scale, shift, rows,
                       D, eps);
    LAUNCH_CHECK();
    return 0;
}

int bct_to_btc_launch(const float* src, int B, int C, int T_src, float* dst32, long ld32, half_t* dst16, long ld16,
                      int seq_rows, int t_valid, float scale, hipStream_t st) {
    dim3 grid(cdiv(seq_rows, 32), cdiv(C, 32), B);
    hipLaunchKernelGGL(bct_to_btc_kernel, grid, dim3(256), 0, st, src, B, C, T_src, dst32, ld32, dst16, ld16, seq_rows,
                       t_valid, scale);
    LAUNCH_CHECK();
    return 0;
}

int btc_to_bct_launch(const float* src, long ld, int seq_rows, float* dst, int B, int C, int T, hipStream_t st) {
    dim3 grid(cdiv(T, 32), cdiv(C, 32), B);
    hipLaunchKernelGGL(btc_to_bct_kernel, grid, dim3(256), 0, st, src, ld, seq_rows, dst, B, C, T);
    LAUNCH_CHECK();
    return 0;
}

int cast_rows_launch(const float* src, long lds_, half_t* dst, long ldd, int rows, int cols, hipStream_t st) {
    const long n = (long)rows * cols;
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, src, lds_, dst, ldd, rows, cols);
    LAUNCH_CHECK();
    return 0;
}

int timestep_feat_launch(const float* t, const float* freqs, float* out, int n, hipStream_t st) {
    hipLaunchKernelGGL(timestep_feat_kernel, dim3(n), dim3(128), 0, st, t, freqs, out, n);
    LAUNCH_CHECK();
    return 0;
}

int silu_launch(const float* x, float* y, long n, hipStream_t st) {
    hipLaunchKernelGGL(silu_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, x, y, n);
    LAUNCH_CHECK();
    return 0;
}

int prefix_rows_launch(float* xin, int n_seq, int seq_rows, int D, int n_prefix, int t_rows, const float* tok_time,
                       const float* tok_style, int time_first, hipStream_t st) {
    if (n_prefix + (seq_rows - t_rows) == 0) return 0;
    hipLaunchKernelGGL(prefix_rows_kernel, dim3(n_seq), dim3(256), 0, st, xin, n_seq, seq_rows, D, n_prefix, t_rows,
                       tok_time, tok_style, time_first);
    LAUNCH_CHECK();
    return 0;
}

int euler_cfg_launch(float* x, long ldx, half_t* x16, long ldx16, int x_rows, const float* v, long ldv,
                     long v_stream_stride, int v_rows, int B, int T, int C, const int* prompt_len, float dt, float c0,
                     float ca, float cb, int stream_a, int stream_b, hipStream_t st) {
    const long n = (long)B * T * C;
    hipLaunchKernelGGL(euler_cfg_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, x, ldx, x16, ldx16, x_rows, v, ldv,
                       v_stream_stride, v_rows, B, T, C, prompt_len, dt, c0, ca, cb, stream_a, stream_b);
    LAUNCH_CHECK();
    return 0;
}

int pack_f16_launch(const float* src, half_t* dst, int n0, int n1, int n2, long s0, long s1, long s2, long d0, long d1,
                    long d2, const float* scale, hipStream_t st) {
    const long n = (long)n0 * n1 * n2;
    if (n == 0) return 0;
    hipLaunchKernelGGL(pack_kernel<half_t>, dim3(cdiv(n, 256)), dim3(256), 0, st, src, dst, n0, n1, n2, s0, s1, s2, d0,
                       d1, d2, scale, 0);
    LAUNCH_CHECK();
    return 0;
}

int pack_f16_lo_launch(const float* src, half_t* dst, int n0, int n1, int n2, long s0, long s1, long s2, long d0, long d1,
                       long d2, const float* scale, hipStream_t st) {
    const long n = (long)n0 * n1 * n2;
    if (n == 0) return 0;
    hipLaunchKernelGGL(pack_kernel<half_t>, dim3(cdiv(n, 256)), dim3(256), 0, st, src, dst, n0, n1, n2, s0, s1, s2, d0,
                       d1, d2, scale, 1);
    LAUNCH_CHECK();
    return 0;
}

int pack_f32_launch(const float* src, float* dst, int n0, int n1, int n2, long s0, long s1, long s2, long d0, long d1,
                    long d2, const float* scale, hipStream_t st) {
    const long n = (long)n0 * n1 * n2;
    if (n == 0) return 0;
    hipLaunchKernelGGL(pack_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, st, src, dst, n0, n1, n2, s0, s1, s2, d0,
                       d1, d2, scale, 0);
    LAUNCH_CHECK();
    return 0;
}

int wn_scale_launch(const float* g, const float* v, int rows, long row_elems, float* out, hipStream_t st) {
    hipLaunchKernelGGL(wn_scale_kernel, dim3(rows), dim3(256), 0, st, g, v, row_elems, out);
    LAUNCH_CHECK();
    return 0;
}

int small_linear_launch(const float* in, long ld_in, const float* W, long ldw, const float* bias, float* out, long ld_out,
                        int R, int N, int K, int act, hipStream_t st) {
    const long waves = (long)R * N;
    if (waves == 0) return 0;
    hipLaunchKernelGGL(small_linear_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, st, in, ld_in, W, ldw, bias, out, ld_out,
                       R, N, K, act);
    LAUNCH_CHECK();
    return 0;
}

int add_rowvec_launch(float* dst, const float* a, long lda, const float* bvec, int n_seq, int n, hipStream_t st) {
    hipLaunchKernelGGL(add_rowvec_kernel, dim3(cdiv((long)n_seq * n, 256)), dim3(256), 0, st, dst, a, lda, bvec, n_seq, n);
    LAUNCH_CHECK();
    return 0;
}

}  // namespace svc
