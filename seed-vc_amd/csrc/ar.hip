// v2 AR model (NaiveTransformer, GQA 12q/2kv, KV cache): `forward_generate` (prefill and one-token decode step)
// and the top-p / repetition-penalty / exponential-race sampler, as HIP kernels for gfx950.
//
// The one-token decode step is HBM-bound (every weight byte is read once per token), so its linears are
// wave-per-output-row GEMV kernels streaming fp16 weights with 16-byte loads; prefill (S > 8 rows) reuses the
// MFMA tap-GEMM on the same packed weights.  The whole decode step (38 kernels: three per layer, see "three launches
// per layer" below) is captured once into a hipGraph and replayed per token (the reference's answer to launch overhead
// is torch.compile "reduce-overhead", modules/v2/vc_wrapper.py:105-114); positions live in device memory and are
// advanced inside the graph, so a replay needs no host-side argument update.
//
// reference: modules/v2/ar.py:239-267 (forward_generate), :75-93 (KVCache.update), :503-567 (Attention),
//            :600-651 (RMSNorm, bf16 RoPE table), :712-763 (sample / logits_to_probs / exponential race).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "model_util.h"

using namespace svc;

namespace {

// Device-resident state of the generate loop: the captured per-token graph (decode step -> rank -> sample, which also
// embeds the drawn token and advances the positions) reads everything that changes from token to token from here, so one graph replay per token needs no host argument.
struct GenState {
    const float* noise;      // [max_new][V] Exp(1) draws, row t for token t
    int* toks;               // [max_new] generated tokens
    int cnt;                 // index of the token being generated (>= 1 inside the loop)
    int min_before_eos, eos;
    float temperature, top_p, rep_pen;
};

// Cross-lane sums on the DPP path (one VALU instruction per step, no LDS crossbar round trip: `__shfl_xor` compiles to
// ds_bpermute_b32, ~100+ cycles each, and the one-token kernels are chains of such latencies).  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
__device__ __forceinline__ float row8_sum_f(float v) {      // every lane of an aligned 8-lane group gets the group's sum
    v += dpp_f<DPP_XOR1>(v); v += dpp_f<DPP_XOR2>(v); v += dpp_f<DPP_HALF_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float row16_sum_f(float v) {     // ... of an aligned 16-lane group (a DPP row)
    v = row8_sum_f(v); v += dpp_f<DPP_ROW_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {      // every lane gets the wave's sum
    v = row16_sum_f(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
}

// out[s][n] = (res ? res[s][n] : 0) + sum_k x[s][k] * W[n][k]        (one wave per output column, S <= 8 rows)
// GLU: rows (2j, 2j+1) of W are (w1_j, w3_j): out16[s][j] = silu(a) * b
template <bool GLU>
__global__ __launch_bounds__(256) void gemv_kernel(const half_t* __restrict__ x, long ldx, const half_t* __restrict__ W, long ldw,
                                                   const float* res, long ldres, float* out32, half_t* out16, long ldo,
                                                   int S, int N, int K) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int n_out = GLU ? N / 2 : N;
    if (wave >= n_out) return;
    const half_t* w0 = W + (long)(GLU ? 2 * wave : wave) * ldw;
    const half_t* w1 = w0 + ldw;
    float acc0[8], acc1[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) acc0[s] = acc1[s] = 0.f;
    for (int k0 = lane * 8; k0 < K; k0 += 512) {
        const half8 a = *reinterpret_cast<const half8*>(w0 + k0);
        half8 b;
        if (GLU) b = *reinterpret_cast<const half8*>(w1 + k0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s < S) {
                const half8 xv = *reinterpret_cast<const half8*>(x + (long)s * ldx + k0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc0[s] += (float)xv[j] * (float)a[j];
                    if (GLU) acc1[s] += (float)xv[j] * (float)b[j];
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s < S) {
            const float a = wave_sum_f(acc0[s]);
            const float b = GLU ? wave_sum_f(acc1[s]) : 0.f;
            if (lane == 0) {
                if (GLU) {
                    out16[(long)s * ldo + wave] = (half_t)((a / (1.f + __expf(-a))) * b);
                } else {
                    float o = a;
                    if (res) o += res[(long)s * ldres + wave];
                    if (out32) out32[(long)s * ldo + wave] = o;
                    if (out16) out16[(long)s * ldo + wave] = (half_t)o;
                }
            }
        }
    }
}


// ---- decode-step GEMV (S <= 8 rows): one wave per PAIR of output rows (2w, 2w+1), fp16 weights streamed with
// 16-byte loads, fp32 accumulate.  NORM fuses the preceding RMSNorm (x fp32, rstd recomputed per wave: K floats
// from L2 -- cheaper than a launch).  Epilogues: PLAIN (+residual), GLU (rows = (w1_j, w3_j) -> silu(a) * b) and
// QKV (rows = one RoPE pair: rotate with the bf16 table, q -> q_out, k / v -> scattered into the KV cache).
enum { GV_PLAIN = 0, GV_GLU = 1, GV_QKV = 2 };
struct GemvArgs {
    const void* x; long ldx;            // NORM ? fp32 : fp16
    const float* gamma; float eps;
    const half_t* W; long ldw;
    const float* res; long ldres;
    float* out32; half_t* out16; long ldo;
    int S, N, K;
    // QKV
    float *q_out, *kc, *vc;
    const float* rope;
    const int* pos;
    int H, Hkv, Lmax;
};

template <bool NORM, int EPI>
__global__ __launch_bounds__(256) void gemv_pair_kernel(const GemvArgs a) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int r0 = 2 * wave;
    if (r0 >= a.N) return;
    const bool has1 = r0 + 1 < a.N;
    const half_t* w0 = a.W + (long)r0 * a.ldw;
    const half_t* w1 = w0 + (has1 ? a.ldw : 0);
    float rstd[8];
    if constexpr (NORM) {
        const float* xf = reinterpret_cast<const float*>(a.x);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float ss = 0.f;
            if (s < a.S)
                for (int k0 = lane * 4; k0 < a.K; k0 += 256) {
                    const float4v v = *reinterpret_cast<const float4v*>(xf + (long)s * a.ldx + k0);
                    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                }
            rstd[s] = rsqrtf(wave_sum_f(ss) / (float)a.K + a.eps);
        }
    }
    float acc0[8], acc1[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) acc0[s] = acc1[s] = 0.f;
    for (int k0 = lane * 8; k0 < a.K; k0 += 512) {
        const half8 wa = *reinterpret_cast<const half8*>(w0 + k0);
        const half8 wb = *reinterpret_cast<const half8*>(w1 + k0);
        float g[8];
        if constexpr (NORM) {
            const float4v g0 = *reinterpret_cast<const float4v*>(a.gamma + k0);
            const float4v g1 = *reinterpret_cast<const float4v*>(a.gamma + k0 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { g[j] = g0[j]; g[4 + j] = g1[j]; }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s < a.S) {
                float xv[8];
                if constexpr (NORM) {
                    const float* xf = reinterpret_cast<const float*>(a.x) + (long)s * a.ldx + k0;
                    const float4v x0 = *reinterpret_cast<const float4v*>(xf);
                    const float4v x1 = *reinterpret_cast<const float4v*>(xf + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { xv[j] = x0[j] * rstd[s] * g[j]; xv[4 + j] = x1[j] * rstd[s] * g[4 + j]; }
                } else {
                    const half8 xh = *reinterpret_cast<const half8*>(reinterpret_cast<const half_t*>(a.x) + (long)s * a.ldx + k0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[j] = (float)xh[j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc0[s] += xv[j] * (float)wa[j];
                    acc1[s] += xv[j] * (float)wb[j];
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s < a.S) {
            const float v0 = wave_sum_f(acc0[s]);
            const float v1 = wave_sum_f(acc1[s]);
            if (lane == 0) {
                if constexpr (EPI == GV_GLU) {
                    a.out16[(long)s * a.ldo + wave] = (half_t)((v0 / (1.f + __expf(-v0))) * v1);
                } else if constexpr (EPI == GV_PLAIN) {
                    float o0 = v0, o1 = v1;
                    if (a.res) { o0 += a.res[(long)s * a.ldres + r0]; if (has1) o1 += a.res[(long)s * a.ldres + r0 + 1]; }
                    if (a.out32) { a.out32[(long)s * a.ldo + r0] = o0; if (has1) a.out32[(long)s * a.ldo + r0 + 1] = o1; }
                    if (a.out16) { a.out16[(long)s * a.ldo + r0] = (half_t)o0; if (has1) a.out16[(long)s * a.ldo + r0 + 1] = (half_t)o1; }
                } else {
                    const int D = a.H * 64, kvd = a.Hkv * 64;
                    const int ip = a.pos[s], kp = a.pos[a.S + s];
                    if (r0 < D + kvd) {
                        const int pair = (r0 & 63) >> 1;
                        const float cs = a.rope[((long)ip * 32 + pair) * 2], sn = a.rope[((long)ip * 32 + pair) * 2 + 1];
                        const float o0 = v0 * cs - v1 * sn, o1 = v1 * cs + v0 * sn;
                        if (r0 < D) {
                            a.q_out[(long)s * D + r0] = o0;
                            a.q_out[(long)s * D + r0 + 1] = o1;
                        } else {
                            const int ek = r0 - D;
                            float* dst = a.kc + ((long)(ek >> 6) * a.Lmax + kp) * 64 + (ek & 63);
                            dst[0] = o0;
                            dst[1] = o1;
                        }
                    } else {
                        const int ev = r0 - D - kvd;
                        float* dst = a.vc + ((long)(ev >> 6) * a.Lmax + kp) * 64 + (ev & 63);
                        dst[0] = v0;
                        dst[1] = v1;
                    }
                }
            }
        }
    }
}

template <bool NORM, int EPI>
int gemv_pair_launch(const GemvArgs& a, hipStream_t st) {
    const int waves = (a.N + 1) / 2;
    hipLaunchKernelGGL((gemv_pair_kernel<NORM, EPI>), dim3(cdiv(waves, 4)), dim3(256), 0, st, a);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

// RoPE (bf16-rounded table) on q and k, scatter k / v into the cache at kv_pos.  qkv [S][D + 2 kvd] fp32.
__global__ void ar_rope_cache_kernel(const float* __restrict__ qkv, long ldq, float* __restrict__ q_out, float* __restrict__ kc,
                                     float* __restrict__ vc, const float* __restrict__ rope, const int* __restrict__ pos, int S,
                                     int H, int Hkv, int Lmax) {
    // pos[0..S) = input_pos (RoPE), pos[S..2S) = kv_pos (cache slot)
    const int s = blockIdx.x;
    const int D = H * 64, kvd = Hkv * 64;
    const float* row = qkv + (long)s * ldq;
    const int ip = pos[s], kp = pos[S + s];
    for (int i = threadIdx.x; i < (D + 2 * kvd) / 2; i += blockDim.x) {
        const int e = 2 * i;                       // even element index within [q | k | v]
        const float x0 = row[e], x1 = row[e + 1];
        if (e < D + kvd) {
            const int pair = (e & 63) >> 1;
            const float cs = rope[((long)ip * 32 + pair) * 2], sn = rope[((long)ip * 32 + pair) * 2 + 1];
            const float o0 = x0 * cs - x1 * sn, o1 = x1 * cs + x0 * sn;
            if (e < D) {
                q_out[(long)s * D + e] = o0;
                q_out[(long)s * D + e + 1] = o1;
            } else {
                const int ek = e - D, hk = ek >> 6, d = ek & 63;
                float* dst = kc + ((long)hk * Lmax + kp) * 64 + d;
                dst[0] = o0;
                dst[1] = o1;
            }
        } else {
            const int ev = e - D - kvd, hv = ev >> 6, d = ev & 63;
            float* dst = vc + ((long)hv * Lmax + kp) * 64 + d;
            dst[0] = x0;
            dst[1] = x1;
        }
    }
}

// One 1024-thread block per (token s, head h): softmax(q k^T / 8 over cache slots j <= kv_pos[s]) v -> y16 [S][D].
// Scores: one thread per key (16 independent 16-byte loads in flight per thread); PV: lane = d, 16 key slices.
__global__ __launch_bounds__(1024) void ar_attn_kernel(const float* __restrict__ q, const float* __restrict__ kc,
                                                       const float* __restrict__ vc, half_t* __restrict__ y, const int* __restrict__ pos,
                                                       int S, int H, int Hkv, int Lmax) {
    extern __shared__ float sm[];               // scores [Lmax] then 16 x 64 partial outputs
    __shared__ float red[16];
    const int s = blockIdx.x, h = blockIdx.y;
    const int hk = h / (H / Hkv);
    const int n_keys = pos[S + s] + 1;          // causal row of the mask: slots 0 .. kv_pos
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float4v qv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) qv[i] = *reinterpret_cast<const float4v*>(q + ((long)s * H + h) * 64 + i * 4);
    const float* kbase = kc + (long)hk * Lmax * 64;
    float mx = -1e30f;
    for (int j = tid; j < n_keys; j += 1024) {
        const float4v* kr = reinterpret_cast<const float4v*>(kbase + (long)j * 64);
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float4v kv = kr[i]; d += qv[i][0] * kv[0] + qv[i][1] * kv[1] + qv[i][2] * kv[2] + qv[i][3] * kv[3]; }
        d *= 0.125f;
        sm[j] = d;
        mx = fmaxf(mx, d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    __syncthreads();
    float ls = 0.f;
    for (int j = tid; j < n_keys; j += 1024) {
        const float p = expf(sm[j] - m);
        sm[j] = p;
        ls += p;
    }
    ls = wave_sum_f(ls);
    if (lane == 0) red[wave] = ls;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += red[i];
    const float inv = 1.0f / tot;
    // o[d] = sum_j p_j v_j[d]: lane = d, 16 waves split the keys
    const float* vbase = vc + (long)hk * Lmax * 64;
    float acc = 0.f;
#pragma unroll 4
    for (int j = wave; j < n_keys; j += 16) acc += sm[j] * vbase[(long)j * 64 + lane];
    float* part = sm + Lmax;
    part[wave * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0) {
        float o = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) o += part[i * 64 + lane];
        y[((long)s * H + h) * 64 + lane] = (half_t)(o * inv);
    }
}

// ------------------------------------------------------------------------------------------------ S = 1 decode step
// Building blocks of the four-launch form (SVC_AR_DEC=1; the default three-launch form further down reuses dec_ffn13 /
// dec_w2 and replaces dec_qkv / dec_attn by dec_w2qkv / dec_attn2).  The step is bound by launch boundaries and memory
// round trips, not bandwidth (13.4 MB of weights per layer); every kernel issues EVERY load a wave needs before anything
// waits:
//   dec_qkv   : attention_norm + wqkv GEMV + RoPE + KV-cache scatter          (one wave per RoPE row pair)
//   dec_attn  : one workgroup per head: scores, softmax, PV over the valid cache prefix, then that head's slice of wo
//               (768 x 64) -> partial residual vectors part[head][D]          (wo's own launch disappears)
//   dec_ffn13 : h = h_in + sum_heads part[head] (summed once per workgroup in LDS; workgroup 0 stores it to the other
//               residual buffer), ffn_norm, w1/w3 GEMV, SwiGLU -> ff16        (4 rows per wave)
//   dec_w2    : h_out = h + w2 GEMV                                            (2 rows per wave)
// A wave owns whole weight rows; lane l owns the 16-byte chunks l, l + 64, ... of every row (and of the input vector),
// so the input never goes through LDS and one wave reduction per row finishes it.
constexpr int DEC_MAXC = 5;          // chunks of 8 elements per lane: reductions up to 64 * 8 * 5 = 2560 long

template <int NR>
struct DecW { half8 w[NR][DEC_MAXC]; };

// Every weight chunk of the NR rows; rows past the end re-read the last row.  KC = the reduction length when it is known
// at compile time (the ar_base sizes), 0 = runtime K.  No lane is predicated: a chunk index past the row is CLAMPED (the
// lane re-reads the last chunk, an L1 hit) and its input element is zeroed in dec_dot -- a predicated load is a branch
// per request with conservative vmcnt(0) waits at the joins, which serialised these one-round-trip kernels; whole chunks
// past the row are skipped by a wave-uniform test (compile-time with KC).
template <int NR, int KC = 0>
__device__ __forceinline__ void dec_load_w(DecW<NR>& r, const half_t* __restrict__ W, long ldw, int row0, int n_rows, int K, int lane) {
    const int nch = KC ? KC >> 3 : K >> 3;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const half_t* wr = W + (long)(row0 + q < n_rows ? row0 + q : n_rows - 1) * ldw;
#pragma unroll
        for (int i = 0; i < DEC_MAXC; ++i) {
            if (64 * i < nch) {
                const int c = lane + 64 * i;
                r.w[q][i] = *reinterpret_cast<const half8*>(wr + 8 * (c < nch ? c : nch - 1));
            }
        }
    }
}

struct DecG { float4v g0[DEC_MAXC], g1[DEC_MAXC]; };      // a lane's chunks of the RMSNorm weight

template <int KC = 0>
__device__ __forceinline__ void dec_load_g(DecG& g, const float* __restrict__ gamma, int K, int lane) {
    const int nch = KC ? KC >> 3 : K >> 3;
#pragma unroll
    for (int i = 0; i < DEC_MAXC; ++i) {
        if (64 * i < nch) {
            const int c = lane + 64 * i, cc = c < nch ? c : nch - 1;
            g.g0[i] = *reinterpret_cast<const float4v*>(gamma + 8 * cc);
            g.g1[i] = *reinterpret_cast<const float4v*>(gamma + 8 * cc + 4);
        }
    }
}

// input chunks (global or LDS) -> optional RMSNorm (rstd from this wave's own sum of squares) * gamma -> NR dot products.
// `pg`: the norm weight already in registers (requested before a barrier), or null to load it here.
template <int NR, bool XF16, int KC = 0>
__device__ __forceinline__ void dec_dot(const DecW<NR>& r, int K, const void* x, const float* gamma, float eps, bool norm,
                                        float (&out)[NR], int lane, const DecG* pg = nullptr) {
    float xv[DEC_MAXC][8];
    const int nch = KC ? KC >> 3 : K >> 3;
    const int Kk = KC ? KC : K;
    DecG gl;
    if (norm && !pg) dec_load_g<KC>(gl, gamma, K, lane);
    const DecG& g = pg ? *pg : gl;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < DEC_MAXC; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[i][j] = 0.f;
        if (64 * i < nch) {
            const int c = lane + 64 * i, cc = c < nch ? c : nch - 1;
            if constexpr (XF16) {
                const half8 h = *reinterpret_cast<const half8*>(reinterpret_cast<const half_t*>(x) + 8 * cc);
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[i][j] = c < nch ? (float)h[j] : 0.f;
            } else {
                const float* xf = reinterpret_cast<const float*>(x) + 8 * cc;
                const float4v a = *reinterpret_cast<const float4v*>(xf), b = *reinterpret_cast<const float4v*>(xf + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { xv[i][j] = c < nch ? a[j] : 0.f; xv[i][4 + j] = c < nch ? b[j] : 0.f; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += xv[i][j] * xv[i][j];
        }
    }
    if (norm) {
        const float rstd = rsqrtf(wave_sum_f(ss) / (float)Kk + eps);
#pragma unroll
        for (int i = 0; i < DEC_MAXC; ++i) {
            if (64 * i < nch) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { xv[i][j] *= rstd * g.g0[i][j]; xv[i][4 + j] *= rstd * g.g1[i][j]; }
            }
        }
    }
    float acc[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        acc[q] = 0.f;
#pragma unroll
        for (int i = 0; i < DEC_MAXC; ++i) {
            if (64 * i < nch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[q] += xv[i][j] * (float)r.w[q][i][j];
            }
        }
    }
    // the NR reductions are independent DPP chains (the compiler interleaves them)
#pragma unroll
    for (int q = 0; q < NR; ++q) out[q] = wave_sum_f(acc[q]);
}

template <int KD>
__global__ __launch_bounds__(64) void dec_qkv_kernel(const float* __restrict__ h, const float* __restrict__ gamma, float eps,
                                                     const half_t* __restrict__ W, int K, int N, float* __restrict__ q_out,
                                                     float* __restrict__ kc, float* __restrict__ vc, const float* __restrict__ rope,
                                                     const int* __restrict__ pos, int H, int Hkv, int Lmax) {
    const int lane = threadIdx.x, r0 = 2 * blockIdx.x;
    if (r0 >= N) return;
    DecW<2> w;
    dec_load_w<2, KD>(w, W, K, r0, N, K, lane);
    // epilogue operands are fetched now, not after the reduction (each would be one more exposed round trip)
    const int ip = pos[0], kp = pos[1];
    const int pair = (r0 & 63) >> 1;
    const float cs = rope[((long)ip * 32 + pair) * 2], sn = rope[((long)ip * 32 + pair) * 2 + 1];
    float v[2];
    dec_dot<2, false, KD>(w, K, h, gamma, eps, true, v, lane);
    if (lane == 0) {
        const int D = H * 64, kvd = Hkv * 64;
        if (r0 < D + kvd) {
            const float o0 = v[0] * cs - v[1] * sn, o1 = v[1] * cs + v[0] * sn;
            if (r0 < D) {
                q_out[r0] = o0;
                q_out[r0 + 1] = o1;
            } else {
                const int ek = r0 - D;
                float* dst = kc + ((long)(ek >> 6) * Lmax + kp) * 64 + (ek & 63);
                dst[0] = o0;
                dst[1] = o1;
            }
        } else {
            const int ev = r0 - D - kvd;
            float* dst = vc + ((long)(ev >> 6) * Lmax + kp) * 64 + (ev & 63);
            dst[0] = v[0];
            dst[1] = v[1];
        }
    }
}

// One 1024-thread workgroup per head.  part[h][n] = sum_d wo[n][64 h + d] * y_h[d]  (y_h rounded to fp16 like the
// stand-alone path).  Latency-shaped: thread (g = tid / 16, c = tid % 16) owns float4 column c of the key AND value rows
// g, g + 64, ... (8 per batch of 512 keys) and requests all of them -- and its piece of the wo slice -- before anything
// waits, so a whole batch costs one memory round trip; scores are 16-lane shuffle sums, the softmax is the online form
// across batches (one for contexts up to 512 keys), P.V is accumulated from the registers already held, and the 64
// groups are summed through LDS.  Every global access is a coalesced 256-byte row (128-byte row slice for wo).
__global__ __launch_bounds__(1024) void dec_attn_kernel(const float* __restrict__ q, const float* __restrict__ kc,
                                                        const float* __restrict__ vc, const half_t* __restrict__ wo,
                                                        float* __restrict__ part, const int* __restrict__ pos, int H, int Hkv, int Lmax) {
    __shared__ __attribute__((aligned(16))) float pacc[64 * 64];     // per key group: 64 output columns
    __shared__ float red[16], yv[64];
    const int h = blockIdx.x, D = H * 64;
    const int hk = h / (H / Hkv);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = tid >> 4, c = tid & 15;
    // this head's wo column slice goes global -> LDS by LDS-DMA right away (needed last; in registers it cost 32 VGPRs
    // and pushed the kernel into scratch)
    constexpr int WO_TRIPS = 8;                 // wo rows up to 1024: row n = 8 lanes x 16 bytes, 128 rows per trip
    __shared__ __attribute__((aligned(16))) half_t wo_s[1024 * 64];
    {
        typedef __attribute__((address_space(1))) const void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int i = 0; i < WO_TRIPS; ++i) {
            const int n = i * 128 + (tid >> 3);
            const int nn = n < D ? n : D - 1;
            if (i * 128 < D)
                __builtin_amdgcn_global_load_lds((gptr_t)(wo + (long)nn * D + 64 * h + 8 * (tid & 7)),
                                                 (lptr_t)(wo_s + (i * 128 + wave_u * 8) * 64), 16, 0, 0);
        }
    }
    const int n_keys = pos[1] + 1;
    const float4v qv = *reinterpret_cast<const float4v*>(q + (long)h * 64 + 4 * c);
    const float* kbase = kc + (long)hk * Lmax * 64 + 4 * c;
    const float* vbase = vc + (long)hk * Lmax * 64 + 4 * c;
    constexpr int KB = 8;                       // keys per thread per batch
    float m_run = -1e30f, l_run = 0.f;
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < n_keys; j0 += 64 * KB) {
        float4v kv[KB], vv[KB];
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const int j = j0 + g + 64 * i;
            const long o = (long)(j < n_keys ? j : 0) * 64;
            kv[i] = *reinterpret_cast<const float4v*>(kbase + o);
            vv[i] = *reinterpret_cast<const float4v*>(vbase + o);
        }
        float sc[KB];
#pragma unroll
        for (int i = 0; i < KB; ++i) sc[i] = qv[0] * kv[i][0] + qv[1] * kv[i][1] + qv[2] * kv[i][2] + qv[3] * kv[i][3];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
            for (int i = 0; i < KB; ++i) sc[i] += __shfl_xor(sc[i], o);
        float bm = -1e30f;
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            sc[i] = j0 + g + 64 * i < n_keys ? sc[i] * 0.125f : -1e30f;
            bm = fmaxf(bm, sc[i]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bm = fmaxf(bm, __shfl_xor(bm, o));
        __syncthreads();                         // red[] of the previous batch has been read
        if (lane == 0) red[wave] = bm;
        __syncthreads();
        float m_new = m_run;
#pragma unroll
        for (int i = 0; i < 16; ++i) m_new = fmaxf(m_new, red[i]);
        const float scale = expf(m_run - m_new);
        l_run *= scale;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] *= scale;
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const float p = j0 + g + 64 * i < n_keys ? expf(sc[i] - m_new) : 0.f;
            l_run += p;                          // every lane of a group carries the same p: the sum is taken from lane c == 0
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += p * vv[i][r];
        }
        m_run = m_new;
    }
    // sum over the 64 key groups
    *reinterpret_cast<float4v*>(pacc + g * 64 + 4 * c) = acc;
    float ls = c == 0 ? l_run : 0.f;
    ls = wave_sum_f(ls);
    __syncthreads();
    if (lane == 0) red[wave] = ls;
    __syncthreads();
    if (tid < 64) {
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += red[i];
        float o = 0.f;
#pragma unroll 8
        for (int i = 0; i < 64; ++i) o += pacc[i * 64 + tid];
        yv[tid] = (float)(half_t)(o / tot);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's part of the wo slice has landed
    __syncthreads();
    const float4v y0 = *reinterpret_cast<const float4v*>(yv + 8 * (tid & 7)), y1 = *reinterpret_cast<const float4v*>(yv + 8 * (tid & 7) + 4);
#pragma unroll
    for (int i = 0; i < WO_TRIPS; ++i) {
        const int n = i * 128 + (tid >> 3);
        float o = 0.f;
        if (n < D) {
            const half8 wr = *reinterpret_cast<const half8*>(wo_s + n * 64 + 8 * (tid & 7));
#pragma unroll
            for (int j = 0; j < 4; ++j) o += y0[j] * (float)wr[j] + y1[j] * (float)wr[4 + j];
        }
        o += __shfl_xor(o, 1); o += __shfl_xor(o, 2); o += __shfl_xor(o, 4);
        if (n < D && (tid & 7) == 0) part[(long)h * D + n] = o;
    }
}

// 256 threads = 4 waves x 4 rows (2 SwiGLU outputs each).  h = h_in + sum_p part[p] is summed once per workgroup; the weight
// rows are requested before that prologue.
template <int NP, int KD>
__global__ __launch_bounds__(256) void dec_ffn13_kernel(const float* __restrict__ h_in, const float* __restrict__ part, int n_part,
                                                        float* __restrict__ h_out, const float* __restrict__ gamma, float eps,
                                                        const half_t* __restrict__ W, int K, int N, half_t* __restrict__ ff) {
    __shared__ __attribute__((aligned(16))) float hs[64 * 8 * DEC_MAXC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = (blockIdx.x * 4 + wave) * 4;
    DecW<4> w;
    dec_load_w<4, KD>(w, W, K, r0 < N ? r0 : 0, N, K, lane);
    DecG g;                                     // requested before the barrier (after it: one more exposed L2 round trip)
    dec_load_g<KD>(g, gamma, K, lane);
    for (int c = tid; c < (K >> 2); c += 256) {
        float4v a = *reinterpret_cast<const float4v*>(h_in + 4 * c);
        if constexpr (NP > 0) {
            float4v b[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) b[p] = *reinterpret_cast<const float4v*>(part + (long)p * K + 4 * c);
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] += b[p][j];
        } else {
            for (int p = 0; p < n_part; ++p) {
                const float4v b = *reinterpret_cast<const float4v*>(part + (long)p * K + 4 * c);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] += b[j];
            }
        }
        *reinterpret_cast<float4v*>(hs + 4 * c) = a;
        if (blockIdx.x == 0) *reinterpret_cast<float4v*>(h_out + 4 * c) = a;
    }
    __syncthreads();
    if (r0 >= N) return;
    float v[4];
    dec_dot<4, false, KD>(w, K, hs, gamma, eps, true, v, lane, &g);
    if (lane == 0) {
        ff[r0 >> 1] = (half_t)((v[0] / (1.f + __expf(-v[0]))) * v[1]);
        if (r0 + 2 < N) ff[(r0 >> 1) + 1] = (half_t)((v[2] / (1.f + __expf(-v[2]))) * v[3]);
    }
}

// out[n] = res[n] + sum_k W[n][k] x16[k]   (w2 + residual; one wave per 2 rows)
template <int KI>
__global__ __launch_bounds__(64) void dec_w2_kernel(const half_t* __restrict__ x, const half_t* __restrict__ W, int K, int N,
                                                    const float* __restrict__ res, float* __restrict__ out) {
    const int lane = threadIdx.x, r0 = 2 * blockIdx.x;
    if (r0 >= N) return;
    DecW<2> w;
    dec_load_w<2, KI>(w, W, K, r0, N, K, lane);
    const float res0 = res[r0], res1 = r0 + 1 < N ? res[r0 + 1] : 0.f;
    float v[2];
    dec_dot<2, true, KI>(w, K, x, nullptr, 0.f, false, v, lane);
    if (lane == 0) {
        out[r0] = res0 + v[0];
        if (r0 + 1 < N) out[r0 + 1] = res1 + v[1];
    }
}

// ---- three launches per layer -------------------------------------------------------------------------------------
// The attention RMSNorm scale is ONE scalar per token, so it commutes out of the QKV projection:
//     qkv = wqkv (gamma * h / rms(h)) = (Wq' h) / rms(h),   Wq' = wqkv diag(gamma),
// and h = h_mid + w2 ff of the previous layer makes  Wq' h = Wq' h_mid + (Wq' w2) ff  -- with W' = Wq' w2 composed in fp32
// at pack time, the previous layer's w2 GEMV and this layer's QKV GEMV read the same inputs (ff, h_mid) and become ONE
// launch; the 1 / rms(h) factor, RoPE and the KV-cache write move into the attention kernel, which reads h anyway.
// A dependent launch costs ~6 us on this machine whatever it does; the second matrix costs 4.7 MB more weights per layer.

// layer 0: qkv_raw[n] = sum_k Wq'[n][k] h[k]   (no preceding w2)
template <int KD>
__global__ __launch_bounds__(64) void dec_qkvraw_kernel(const float* __restrict__ h, const half_t* __restrict__ Wq, int K, int N,
                                                        float* __restrict__ out) {
    const int lane = threadIdx.x, r0 = 2 * blockIdx.x;
    if (r0 >= N) return;
    DecW<2> w;
    dec_load_w<2, KD>(w, Wq, K, r0, N, K, lane);
    float v[2];
    dec_dot<2, false, KD>(w, K, h, nullptr, 0.f, false, v, lane);
    if (lane == 0) {
        out[r0] = v[0];
        if (r0 + 1 < N) out[r0 + 1] = v[1];
    }
}

// layers >= 1.  Rows [0, D): h_out = h_mid + w2 ff (the previous layer's output = this layer's input);
// rows [D, D + N): qkv_raw = W' ff + Wq' h_mid, Wc = [W' | Wq'] row-wise (ld = I + D).
template <int KD, int KI>
__global__ __launch_bounds__(64) void dec_w2qkv_kernel(const half_t* __restrict__ ff, const float* __restrict__ h_mid,
                                                       const half_t* __restrict__ W2, const half_t* __restrict__ Wc, int I, int D, int N,
                                                       float* __restrict__ h_out, float* __restrict__ qkv_out) {
    const int lane = threadIdx.x, r0 = 2 * blockIdx.x;
    if (r0 < D) {
        DecW<2> w;
        dec_load_w<2, KI>(w, W2, I, r0, D, I, lane);
        const float res0 = h_mid[r0], res1 = r0 + 1 < D ? h_mid[r0 + 1] : 0.f;
        float v[2];
        dec_dot<2, true, KI>(w, I, ff, nullptr, 0.f, false, v, lane);
        if (lane == 0) {
            h_out[r0] = res0 + v[0];
            if (r0 + 1 < D) h_out[r0 + 1] = res1 + v[1];
        }
        return;
    }
    const int rq = r0 - D;
    if (rq >= N) return;
    DecW<2> wa, wb;
    dec_load_w<2, KI>(wa, Wc, (long)I + D, rq, N, I, lane);
    dec_load_w<2, KD>(wb, Wc + I, (long)I + D, rq, N, D, lane);
    float va[2], vb[2];
    dec_dot<2, true, KI>(wa, I, ff, nullptr, 0.f, false, va, lane);
    dec_dot<2, false, KD>(wb, D, h_mid, nullptr, 0.f, false, vb, lane);
    if (lane == 0) {
        qkv_out[rq] = va[0] + vb[0];
        if (rq + 1 < N) qkv_out[rq + 1] = va[1] + vb[1];
    }
}

// Attention of the three-launch form, spread over the chip: grid = (DEC_NS wo-row slices) x (heads), 512 threads.
// q / k / v arrive unnormalised (qkv_raw).  Every workgroup of head h recomputes that head's softmax over the valid cache
// prefix [0, kv_pos] -- <= 4096 x 64 fp32 keys and values, read from L2 / Infinity Cache -- and applies 1 / DEC_NS of the
// head's wo column slice (D / DEC_NS rows x 64 columns, held in registers), so 96 workgroups carry the 12 heads of ar_base
// instead of 12 (one CU per head moved 154 KB of K / V + 98 KB of wo and ran eight 1024-thread barriers).
//   * every request that does not depend on `pos` goes out first: the first two batches of cache rows (by position,
//     clamped to the cache, NOT to the valid prefix), the wo rows, the layer input, q / k / v;
//   * 1 / rms(h): every wave reduces the layer input h on its own (D floats from L2, one DPP tree: no barrier);
//   * q and the new k are rotated (bf16-rounded table, position input_pos); the new key / value are used from registers
//     for position kv_pos and stored into the cache by slice 0 of the first head of each KV group;
//   * EIGHT lanes per key (thread = key slot tid / 8, column octet c = tid % 8: columns 32 r + 4 c .. + 3, r = 0, 1, so the
//     eight lanes of a key read one whole 128-byte line per request): a score is 8 FMAs + three DPP adds and the softmax
//     bookkeeping is replicated 8x, not 16x as with one float4 column per lane (in-kernel timestamps: the key loop took
//     1.5 us of the 6.6 us a workgroup lives; four lanes per key needs 48 more registers for q / k / v and spilled);
//     64 slots x 4 keys = 256 keys per batch, two batches in flight; each slot runs its OWN online softmax (no cross-wave
//     exchange per batch);
//   * merge: the 8 slots of a wave by DPP row rotations (one (max, l, acc) per 16-lane row and wave-uniform max), the 32
//     rows through LDS in two short stages that use every thread (the serial 32-term sum of 64 threads took 2.2 us);
//   * part[h][n] = sum_d wo[n][64 h + d] y[d] for this slice's rows n (y rounded to fp16 like the stand-alone path).
constexpr int DEC_NS = 8;           // wo row slices per head
constexpr int DA_KB = 4;            // keys per thread per batch (64 slots x 4 = 256 keys)
constexpr int DA_WO = 2;            // wo rows per thread: 64 rows per pass, D / DEC_NS <= 128 rows (D <= 1024)
constexpr int DPP_ROR8 = 0x128;
struct DaKey { float4v r[2]; };     // a lane's eighth of a 64-float row: columns 32 r + 4 c .. + 3
__global__ __launch_bounds__(512) void dec_attn2_kernel(const float* __restrict__ hres, const float* __restrict__ qkv_raw, float eps,
                                                        const float* __restrict__ rope, float* __restrict__ kc, float* __restrict__ vc,
                                                        const half_t* __restrict__ wo, float* __restrict__ part,
                                                        const int* __restrict__ pos, int H, int Hkv, int Lmax) {
    __shared__ __attribute__((aligned(16))) float pacc[32 * 64];     // per 16-lane row: 64 output columns
    __shared__ __attribute__((aligned(16))) float red[8 * 64];
    __shared__ float pm[32], pl[32], redl[8], yv[64];
    const int sl = blockIdx.x, h = blockIdx.y, D = H * 64, kvd = Hkv * 64;
    const int hk = h / (H / Hkv);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = tid >> 3, c = tid & 7;
    const int rows_per = (D + DEC_NS - 1) / DEC_NS;
    // this slice's wo rows: row n = 8 lanes x 16 bytes of columns [64 h, 64 h + 64)
    half8 wr[DA_WO];
#pragma unroll
    for (int i = 0; i < DA_WO; ++i) {
        const int n = sl * rows_per + i * 64 + (tid >> 3);
        const int nn = n < D ? n : D - 1;
        wr[i] = *reinterpret_cast<const half8*>(wo + (long)nn * D + 64 * h + 8 * (tid & 7));
    }
    // wave-uniform bases + 32-bit lane offsets: the requests take the (SGPR base, VGPR offset, immediate) form
    const float* kbase = kc + (long)hk * Lmax * 64;
    const float* vbase = vc + (long)hk * Lmax * 64;
    DaKey ka[DA_KB], va[DA_KB], kb[DA_KB], vb[DA_KB];
    // Cache rows are requested by position only: the first two batches do not wait for `pos` to arrive.  Rows past kv_pos
    // hold zeros or stale FINITE values of an earlier run (the cache is zero-initialised and only ever written with
    // computed keys / values); their scores are masked and their p is exactly 0.
    auto load_batch = [&](DaKey (&kx)[DA_KB], DaKey (&vx)[DA_KB], int j0) {    // position kv_pos is patched from registers
#pragma unroll
        for (int i = 0; i < DA_KB; ++i) {
            const int j = j0 + slot + 64 * i;
            const unsigned o = (unsigned)(j < Lmax ? j : Lmax - 1) * 64u + 4u * (unsigned)c;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                kx[i].r[r] = *reinterpret_cast<const float4v*>(kbase + o + 32 * r);
                vx[i].r[r] = *reinterpret_cast<const float4v*>(vbase + o + 32 * r);
            }
        }
    };
    load_batch(ka, va, 0);
    load_batch(kb, vb, 64 * DA_KB);
    __builtin_amdgcn_sched_barrier(0);          // the cache rows go out first: nothing below is hoisted above their requests
    const int ip = pos[0], kp = pos[1];
    const int n_keys = kp + 1;
    // 1 / rms of the layer input, per wave: D <= 1024 floats = up to 4 float4 per lane, all requested at once
    float4v hx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = lane + 64 * i;                // clamped address + weight 0: a predicated load would be a branch with a
        hx[i] = *reinterpret_cast<const float4v*>(hres + 4 * (q < (D >> 2) ? q : 0));      // vmcnt(0) wait behind the cache rows
        if (q >= (D >> 2)) hx[i] = (float4v){0.f, 0.f, 0.f, 0.f};
    }
    DaKey q4, k4, v4;
    float4v cs[2];                              // (cos, sin) of the rotation pairs 16 r + 2 c, 16 r + 2 c + 1
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        q4.r[r] = *reinterpret_cast<const float4v*>(qkv_raw + (long)h * 64 + 32 * r + 4 * c);
        k4.r[r] = *reinterpret_cast<const float4v*>(qkv_raw + D + (long)hk * 64 + 32 * r + 4 * c);
        v4.r[r] = *reinterpret_cast<const float4v*>(qkv_raw + D + kvd + (long)hk * 64 + 32 * r + 4 * c);
        cs[r] = *reinterpret_cast<const float4v*>(rope + ((long)ip * 32 + 16 * r + 2 * c) * 2);
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) ss += hx[i][0] * hx[i][0] + hx[i][1] * hx[i][1] + hx[i][2] * hx[i][2] + hx[i][3] * hx[i][3];
    ss = wave_sum_f(ss);
    const float rstd = rsqrtf(ss / (float)D + eps);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { q4.r[r][j] *= rstd; k4.r[r][j] *= rstd; v4.r[r][j] *= rstd; }
        const float4v q0 = q4.r[r], k0 = k4.r[r], t = cs[r];
        q4.r[r][0] = q0[0] * t[0] - q0[1] * t[1]; q4.r[r][1] = q0[1] * t[0] + q0[0] * t[1];
        q4.r[r][2] = q0[2] * t[2] - q0[3] * t[3]; q4.r[r][3] = q0[3] * t[2] + q0[2] * t[3];
        k4.r[r][0] = k0[0] * t[0] - k0[1] * t[1]; k4.r[r][1] = k0[1] * t[0] + k0[0] * t[1];
        k4.r[r][2] = k0[2] * t[2] - k0[3] * t[3]; k4.r[r][3] = k0[3] * t[2] + k0[2] * t[3];
    }
    if (sl == 0 && h % (H / Hkv) == 0 && slot == 0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            *reinterpret_cast<float4v*>(kc + ((long)hk * Lmax + kp) * 64 + 32 * r + 4 * c) = k4.r[r];
            *reinterpret_cast<float4v*>(vc + ((long)hk * Lmax + kp) * 64 + 32 * r + 4 * c) = v4.r[r];
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) q4.r[r] *= 0.125f;      // 1 / sqrt(64) folded into q (exact: a power of two)
    float m_run = -1e30f, l_run = 0.f;
    DaKey acc;
#pragma unroll
    for (int r = 0; r < 2; ++r) acc.r[r] = (float4v){0.f, 0.f, 0.f, 0.f};
    auto process = [&](DaKey (&kx)[DA_KB], DaKey (&vx)[DA_KB], int j0) {
        float sc[DA_KB];
#pragma unroll
        for (int i = 0; i < DA_KB; ++i) {
            if (j0 + slot + 64 * i == kp) { kx[i] = k4; vx[i] = v4; }
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r)
                a += q4.r[r][0] * kx[i].r[r][0] + q4.r[r][1] * kx[i].r[r][1] + q4.r[r][2] * kx[i].r[r][2] + q4.r[r][3] * kx[i].r[r][3];
            sc[i] = a;
        }
#pragma unroll
        for (int i = 0; i < DA_KB; ++i) sc[i] = row8_sum_f(sc[i]);
        float m_new = m_run;
#pragma unroll
        for (int i = 0; i < DA_KB; ++i) {
            sc[i] = j0 + slot + 64 * i < n_keys ? sc[i] : -1e30f;
            m_new = fmaxf(m_new, sc[i]);
        }
        const float scale = __expf(m_run - m_new);
        l_run *= scale;
#pragma unroll
        for (int r = 0; r < 2; ++r) acc.r[r] *= scale;
#pragma unroll
        for (int i = 0; i < DA_KB; ++i) {
            const float p = j0 + slot + 64 * i < n_keys ? __expf(sc[i] - m_new) : 0.f;
            l_run += p;                          // the eight lanes of a slot carry the same p
#pragma unroll
            for (int r = 0; r < 2; ++r) acc.r[r] += p * vx[i].r[r];
        }
        m_run = m_new;
    };
    for (int j0 = 0; j0 < n_keys; j0 += 2 * 64 * DA_KB) {
        process(ka, va, j0);
        if (j0 + 2 * 64 * DA_KB < n_keys) load_batch(ka, va, j0 + 2 * 64 * DA_KB);
        if (j0 + 64 * DA_KB < n_keys) {
            process(kb, vb, j0 + 64 * DA_KB);
            if (j0 + 3 * 64 * DA_KB < n_keys) load_batch(kb, vb, j0 + 3 * 64 * DA_KB);
        }
    }
    // merge, stage 0: the 8 slots of this wave.  Wave-uniform maximum, then per 16-lane row (2 slots) sums by a rotation:
    // lane c of a row ends up with the row's sum for its column octet.
    float mw = fmaxf(m_run, dpp_f<DPP_ROR8>(m_run));
    mw = fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mw), 0)),
                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mw), 16))),
               fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mw), 32)),
                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mw), 48))));
    {
        const float scale = __expf(m_run - mw);
        l_run *= scale;
        l_run += dpp_f<DPP_ROR8>(l_run);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = acc.r[r][j] * scale;
                a += dpp_f<DPP_ROR8>(a);
                acc.r[r][j] = a;
            }
    }
    const int g = tid >> 4;                     // 16-lane row index: 32 per workgroup
    if ((tid & 15) < 8) {
#pragma unroll
        for (int r = 0; r < 2; ++r) *reinterpret_cast<float4v*>(pacc + g * 64 + 32 * r + 4 * c) = acc.r[r];
        if ((tid & 15) == 0) { pm[g] = mw; pl[g] = l_run; }
    }
    __syncthreads();
    {
        // stage 1: every thread; wave w folds the four rows 4 w .. 4 w + 3 (they share pm) for column d = lane
        float M = pm[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) M = fmaxf(M, pm[4 * i]);
        const float w = __expf(pm[4 * wave] - M);
        const float o = pacc[(4 * wave) * 64 + lane] + pacc[(4 * wave + 1) * 64 + lane] + pacc[(4 * wave + 2) * 64 + lane] +
                        pacc[(4 * wave + 3) * 64 + lane];
        red[wave * 64 + lane] = w * o;
        if (lane == 0) redl[wave] = w * (pl[4 * wave] + pl[4 * wave + 1] + pl[4 * wave + 2] + pl[4 * wave + 3]);
    }
    __syncthreads();
    if (tid < 64) {
        float o = 0.f, tot = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { o += red[i * 64 + tid]; tot += redl[i]; }
        yv[tid] = (float)(half_t)(o / tot);
    }
    __syncthreads();
    const float4v y0 = *reinterpret_cast<const float4v*>(yv + 8 * (tid & 7)), y1 = *reinterpret_cast<const float4v*>(yv + 8 * (tid & 7) + 4);
#pragma unroll
    for (int i = 0; i < DA_WO; ++i) {
        const int nl = i * 64 + (tid >> 3);
        const int n = sl * rows_per + nl;
        float o = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) o += y0[j] * (float)wr[i][j] + y1[j] * (float)wr[i][4 + j];
        o = row8_sum_f(o);
        if (nl < rows_per && n < D && (tid & 7) == 0) part[(long)h * D + n] = o;
    }
}

// logits[n] = sum_k W[n][k] norm(h)[k]   (final norm + output head; 4 rows per wave)
template <int KD>
__global__ __launch_bounds__(256) void dec_head_kernel(const float* __restrict__ h, const float* __restrict__ gamma, float eps,
                                                       const half_t* __restrict__ W, int K, int N, float* __restrict__ logits) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = (blockIdx.x * 4 + wave) * 4;
    if (r0 >= N) return;
    DecW<4> w;
    dec_load_w<4, KD>(w, W, K, r0, N, K, lane);
    float v[4];
    dec_dot<4, false, KD>(w, K, h, gamma, eps, true, v, lane);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (r0 + r < N) logits[r0 + r] = v[r];
    }
}

__global__ void advance_pos_kernel(int* pos, int* cnt) {   // S = 1: {input_pos, kv_pos} += 1 (ar.py:402-403)
    if (threadIdx.x < 2) pos[threadIdx.x] += 1;
    if (cnt && threadIdx.x == 2) cnt[0] += 1;
}

// x = embeddings[previous token]  (embed_base of the token sampled by the previous step, ar.py:188-193,414)
__global__ void ar_embed_kernel(const float* __restrict__ emb, const GenState* __restrict__ gs, float* __restrict__ x, int D) {
    const long t = gs->toks[gs->cnt - 1];
    for (int c = threadIdx.x; c < D; c += blockDim.x) x[c] = emb[t * D + c];
}

// ---- sampler: one block, vocab <= 4096.  reference: ar.py:731-763 + :723-727
constexpr int SORT_N = 4096;
// Sampler stage 1 (many workgroups): repetition penalty + suppression, then the RANK of every logit in the descending
// order torch.sort gives (ties: lower index first) by counting -- workgroup b ranks tokens 16 b .. 16 b + 15, 16 lanes per
// token, each lane counting over a 1/16 stride of the vocabulary held in LDS.  Writes the sorted (value, index) pairs and
// the penalised logits.  V^2 comparisons spread over V / 16 workgroups (one CU alone needs ~100 us for them; the 78-stage
// single-workgroup bitonic network this replaces took 74 us).
__global__ __launch_bounds__(256) void ar_rank_kernel(const float* __restrict__ logits, int V, const int* __restrict__ prev, int n_prev,
                                                      int suppress, float rep_pen, const GenState* __restrict__ gs,
                                                      float* __restrict__ skey, int* __restrict__ sidx, float* __restrict__ lgp) {
    if (gs) {   // generate loop: token gs->cnt; the repetition penalty sees previous_tokens[0] only (ar.py:442-444)
        prev = gs->toks; n_prev = 1;
        suppress = gs->cnt < gs->min_before_eos ? gs->eos : -1;
        rep_pen = gs->rep_pen;
    }
    __shared__ __attribute__((aligned(16))) float lg[SORT_N];
    const int tid = threadIdx.x;
    for (int i = tid; i < SORT_N; i += 256) lg[i] = i < V ? logits[i] : -INFINITY;
    __syncthreads();
    // repetition penalty from the ORIGINAL logits (gather, transform, scatter: duplicates write the same value)
    for (int i = tid; i < n_prev; i += 256) {
        const int t = prev[i];
        const float sc = logits[t];
        lg[t] = sc < 0.f ? sc * rep_pen : sc / rep_pen;
    }
    __syncthreads();
    if (tid == 0 && suppress >= 0) lg[suppress] = -INFINITY;
    __syncthreads();
    const int i = blockIdx.x * 16 + (tid >> 4), l = tid & 15;
    const float mine = i < V ? lg[i] : -INFINITY;
    int rk = 0;
    for (int j = l; j < V; j += 16) {
        const float o = lg[j];
        rk += (o > mine) || (o == mine && j < i);
    }
    rk += __shfl_xor(rk, 1); rk += __shfl_xor(rk, 2); rk += __shfl_xor(rk, 4); rk += __shfl_xor(rk, 8);
    if (l == 0 && i < V) {
        skey[rk] = mine;
        sidx[rk] = i;
        lgp[i] = mine;
    }
}

// Sampler stage 2 (one workgroup): softmax over the sorted logits, top-p cut, temperature softmax, exponential race.
__global__ __launch_bounds__(1024) void ar_sample_kernel(const float* __restrict__ lgp, int V, const float* __restrict__ skey,
                                                         const int* __restrict__ sidx, float temperature, float top_p,
                                                         const float* __restrict__ exp_noise, int* __restrict__ idx_out,
                                                         float* __restrict__ probs_out, GenState* __restrict__ gs,
                                                         const float* __restrict__ emb, float* __restrict__ next_x, int D,
                                                         int* __restrict__ pos) {
    if (gs) {
        const int t = gs->cnt;
        temperature = gs->temperature; top_p = gs->top_p;
        exp_noise = gs->noise + (size_t)t * V;
        idx_out = gs->toks + t;
    }
    __shared__ float key[SORT_N];
    __shared__ int idx[SORT_N];
    __shared__ float lg[SORT_N];       // penalised logits in vocabulary order, later reused
    // block-wide reductions: DPP inside a wave, 16 slots through LDS, ONE barrier each (every reduction has its own slots,
    // so nothing has to wait for the previous one to be read out); the tree reductions this replaces were ~50 barriers
    // of 16 waves per token (~4 us of the 10 us this kernel took)
    __shared__ float r_mx[16], r_sum[16], r_best[16];
    __shared__ int r_idx[16];
    __shared__ double r_scan[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto wave_max_f = [](float v) {
        v = fmaxf(v, dpp_f<DPP_XOR1>(v)); v = fmaxf(v, dpp_f<DPP_XOR2>(v));
        v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v)); v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
        return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)),
                           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16))),
                     fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)),
                           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48))));
    };
    for (int i = tid; i < SORT_N; i += 1024) {
        lg[i] = i < V ? lgp[i] : -INFINITY;
        key[i] = i < V ? skey[i] : -INFINITY;
        idx[i] = i < V ? sidx[i] : i;
    }
    __syncthreads();
    // softmax of the sorted logits, cumulative sum (double, like torch.cumsum on CPU floats), top-p mask
    const float m = key[0];
    // chunked scan: thread t (< 1024) owns sorted elements 4t..4t+3
    float e4[4];
    double local = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) { e4[r] = expf(key[4 * tid + r] - m); local += (double)e4[r]; }
    double incl = local;               // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    if (lane == 63) r_scan[wave] = incl;
    __syncthreads();
    double base = 0.0, total = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const double w = r_scan[i];
        if (i < wave) base += w;
        total += w;
    }
    double run = base + incl - local;  // sum of everything before this thread's first element
    for (int r = 0; r < 4; ++r) {
        const int sidx = 4 * tid + r;
        run += (double)e4[r];
        const float cum = (float)(run / total);
        const bool remove = sidx > 0 && cum > top_p;
        if (idx[sidx] < V) lg[idx[sidx]] = remove ? -INFINITY : lg[idx[sidx]];
    }
    __syncthreads();
    // final softmax over kept logits / temperature
    const float tinv = 1.0f / fmaxf(temperature, 1e-5f);
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 1024) mx = fmaxf(mx, lg[i] * tinv);
    mx = wave_max_f(mx);
    if (lane == 0) r_mx[wave] = mx;
    __syncthreads();
    float m2 = r_mx[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m2 = fmaxf(m2, r_mx[i]);
    float sum = 0.f;
    for (int i = tid; i < V; i += 1024) { const float e = expf(lg[i] * tinv - m2); key[i] = e; sum += e; }
    sum = wave_sum_f(sum);
    if (lane == 0) r_sum[wave] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += r_sum[i];
    const float inv = 1.0f / tot;
    // exponential race: argmax probs / q (ties: the lower index)
    float best = -1.f;
    int besti = 0;
    for (int i = tid; i < V; i += 1024) {
        const float p = key[i] * inv;
        if (probs_out) probs_out[i] = p;
        const float r = p / exp_noise[i];
        if (r > best) { best = r; besti = i; }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(besti, o);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if (lane == 0) { r_best[wave] = best; r_idx[wave] = besti; }
    __syncthreads();
    best = r_best[0]; besti = r_idx[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        const float ob = r_best[i];
        const int oi = r_idx[i];
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    idx[0] = besti;                    // every thread holds the same winner; the tail below reads idx[0]
    __syncthreads();
    if (tid == 0) idx_out[0] = idx[0];
    if (gs && next_x) {
        // generate loop: this workgroup also prepares the next step -- embedding row of the token just drawn into the
        // residual buffer (ar.py:188-193,414), positions and token counter advanced (ar.py:402-403) -- which saves the
        // embed, copy and advance launches of every token (a dependent launch costs ~4.5 us whatever it does)
        const long tk = idx[0];
        for (int c = tid; c < D; c += 1024) next_x[c] = emb[tk * D + c];
        if (tid == 0) { pos[0] += 1; pos[1] += 1; gs->cnt += 1; }
    }
}

}  // namespace

struct svc_ar {
    svc_ar_config_t cfg;
    int D, H, Hkv, L, I, V, Lmax, kvd, Nqkv;
    Arena wts, ws;
    struct Layer {
        half_t *wqkv, *wo, *w13, *w2;
        half_t* wc = nullptr;     // three-launch form: layer 0 [Nqkv][D] = wqkv diag(gamma); layers >= 1 [Nqkv][I + D] = [Wq' w2_prev | Wq']
        float *g_attn, *g_ffn;
        float *kc, *vc;
    };
    std::vector<Layer> layers;
    float* g_final;
    half_t* w_out;
    float* rope;
    int cap_S = 0;
    float *h32, *qkv32, *q32, *logits;
    float *h32b, *part;           // S = 1 step: second residual buffer, per-head wo partials
    half_t *n16, *y16, *ff16, *x16;
    int* d_pos;
    GenState* d_gen = nullptr;    // generate-loop state (device)
    float *d_skey = nullptr, *d_lgp = nullptr;   // sampler stage 1 -> stage 2
    int* d_sidx = nullptr;
    int sample(const float* lg, const int* prev, int n_prev, int suppress, float temperature, float top_p, float rep_pen,
               const float* exp_noise, int* idx_out, float* probs_out, GenState* gs, hipStream_t st, bool prepare_next = false);
    // decode graph
    hipGraphExec_t graph = nullptr;
    hipGraphExec_t gen_graph = nullptr;   // step -> rank -> sample (+ next embedding, advance), driven by d_gen
    float* gx = nullptr;          // staged input of the captured step
    float* emb = nullptr;         // model.embeddings.weight [V][D] fp32 (generate loop only)
    int ensure_graph();
    int ensure_gen_graph();
    int run1(const float* x, const int* d_positions, float* logits_out, hipStream_t st);
    int run1_fused(const float* x, const int* d_positions, float* logits_out, hipStream_t st);
    bool have_wc = false;

    int reserve(int S, hipStream_t st);
    int run(const float* x, int S, const int* d_positions, float* logits_out, hipStream_t st);
};

namespace {
int lin(const half_t* x, const half_t* W, long ldw, const float* res, float* out32, half_t* out16, long ldo, int S, int N, int K,
        bool glu, hipStream_t st) {
    if (S <= 8) {
        const int n_out = glu ? N / 2 : N;
        if (glu)
            hipLaunchKernelGGL(gemv_kernel<true>, dim3(cdiv(n_out, 4)), dim3(256), 0, st, x, (long)K, W, ldw, res, ldo, out32, out16, ldo, S, N, K);
        else
            hipLaunchKernelGGL(gemv_kernel<false>, dim3(cdiv(n_out, 4)), dim3(256), 0, st, x, (long)K, W, ldw, res, ldo, out32, out16, ldo, S, N, K);
        SVC_CHECK_HIP(hipGetLastError());
        return 0;
    }
    KGemmParams p;
    memset(&p, 0, sizeof(p));
    p.M = S; p.N = N; p.Lout = S; p.a_seq_rows = S; p.c_seq_rows = S; p.a_stride = 1; p.a_len = S;
    p.n_taps = 1; p.a_ptr[0] = x; p.a_ld[0] = K; p.a_ktiles[0] = K / 64;
    p.w = W; p.ldw = ldw;
    p.res = res; p.ldres = ldo;
    p.c32 = out32; p.ldc32 = ldo;
    p.c16 = out16; p.ldc16 = ldo;
    p.vec_ok = (N % 8 == 0) && (ldo % 8 == 0);
    return kgemm_launch(p, 0, glu ? KG_EPI_SWIGLU : KG_EPI_STORE, st);
}
}  // namespace

int svc_ar::reserve(int S, hipStream_t st) {
    if (S <= cap_S) return 0;
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    ws.release();
    cap_S = std::max(S, 8);
    const long Sr = cap_S;
    h32 = ws.alloc_n<float>(Sr * D, st);
    qkv32 = ws.alloc_n<float>(Sr * Nqkv, st);
    q32 = ws.alloc_n<float>(Sr * D, st);
    logits = ws.alloc_n<float>(round_up(V, 8), st);
    n16 = ws.alloc_n<half_t>(Sr * D, st);
    y16 = ws.alloc_n<half_t>(Sr * D, st);
    ff16 = ws.alloc_n<half_t>(Sr * I, st);
    x16 = ws.alloc_n<half_t>(Sr * D, st);
    d_pos = ws.alloc_n<int>(2 * Sr, st);
    gx = ws.alloc_n<float>(D, st);
    h32b = ws.alloc_n<float>(D, st);
    part = ws.alloc_n<float>((long)H * D, st);
    d_gen = reinterpret_cast<GenState*>(ws.alloc(sizeof(GenState), st));
    d_skey = ws.alloc_n<float>(SORT_N, st);
    d_sidx = ws.alloc_n<int>(SORT_N, st);
    d_lgp = ws.alloc_n<float>(SORT_N, st);
    if (!d_skey || !d_sidx || !d_lgp) return 1;
    if (!h32 || !qkv32 || !q32 || !logits || !n16 || !y16 || !ff16 || !x16 || !d_pos || !gx || !h32b || !part || !d_gen) return 1;
    if (graph) { (void)hipGraphExecDestroy(graph); graph = nullptr; }
    if (gen_graph) { (void)hipGraphExecDestroy(gen_graph); gen_graph = nullptr; }
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
}

// The one-token GEMV kernels are instantiated for the ar_base sizes (dim 768, intermediate 2304: reduction lengths known at
// compile time, dead chunks pruned) and for runtime sizes (<0, 0>: any dim <= 2560 that is a multiple of 8).
#define SVC_DEC_KD (D == 768 && I == 2304 ? 768 : 0)
#define SVC_DEC_LAUNCH1(kern, WHICH, grid, block, lds, st, ...)                                                  \
    do {                                                                                                          \
        if (SVC_DEC_KD) { constexpr int KD = 768, KI = 2304; (void)KD; (void)KI; hipLaunchKernelGGL((kern<WHICH>), grid, block, lds, st, __VA_ARGS__); } \
        else { constexpr int KD = 0, KI = 0; (void)KD; (void)KI; hipLaunchKernelGGL((kern<WHICH>), grid, block, lds, st, __VA_ARGS__); }                 \
    } while (0)
#define SVC_DEC_LAUNCH2(kern, grid, block, lds, st, ...)                                                         \
    do {                                                                                                          \
        if (SVC_DEC_KD) hipLaunchKernelGGL((kern<768, 2304>), grid, block, lds, st, __VA_ARGS__);                 \
        else hipLaunchKernelGGL((kern<0, 0>), grid, block, lds, st, __VA_ARGS__);                                 \
    } while (0)
#define SVC_DEC_LAUNCH_NP(kern, NP, grid, block, lds, st, ...)                                                   \
    do {                                                                                                          \
        if (SVC_DEC_KD) hipLaunchKernelGGL((kern<NP, 768>), grid, block, lds, st, __VA_ARGS__);                   \
        else hipLaunchKernelGGL((kern<NP, 0>), grid, block, lds, st, __VA_ARGS__);                                \
    } while (0)

// One-token step on the four-launches-per-layer kernels (dec_*).
int svc_ar::run1(const float* x, const int* d_positions, float* logits_out, hipStream_t st) {
    if (x != h32) SVC_CHECK_HIP(hipMemcpyAsync(h32, x, (size_t)D * 4, hipMemcpyDeviceToDevice, st));
    for (int i = 0; i < L; ++i) {
        const Layer& ly = layers[i];
        SVC_DEC_LAUNCH1(dec_qkv_kernel, KD, dim3(Nqkv / 2), dim3(64), 0, st, h32, ly.g_attn, cfg.norm_eps, ly.wqkv, D, Nqkv, q32, ly.kc,
                           ly.vc, rope, d_positions, H, Hkv, Lmax);
        hipLaunchKernelGGL(dec_attn_kernel, dim3(H), dim3(1024), 0, st, q32, ly.kc, ly.vc, ly.wo, part, d_positions, H, Hkv, Lmax);
        if (H == 12)
            SVC_DEC_LAUNCH_NP(dec_ffn13_kernel, 12, dim3(cdiv(2 * I, 16)), dim3(256), 0, st, h32, part, H, h32b, ly.g_ffn, cfg.norm_eps,
                               ly.w13, D, 2 * I, ff16);
        else
            SVC_DEC_LAUNCH_NP(dec_ffn13_kernel, 0, dim3(cdiv(2 * I, 16)), dim3(256), 0, st, h32, part, H, h32b, ly.g_ffn, cfg.norm_eps,
                               ly.w13, D, 2 * I, ff16);
        SVC_DEC_LAUNCH1(dec_w2_kernel, KI, dim3(D / 2), dim3(64), 0, st, ff16, ly.w2, I, D, h32b, h32);
        SVC_CHECK_HIP(hipGetLastError());
    }
    SVC_DEC_LAUNCH1(dec_head_kernel, KD, dim3(cdiv(V, 16)), dim3(256), 0, st, h32, g_final, cfg.norm_eps, w_out, D, V, logits_out);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

// One-token step, three launches per layer (dec_qkvraw | dec_w2qkv, dec_attn2, dec_ffn13) + the last w2 + head.
int svc_ar::run1_fused(const float* x, const int* d_positions, float* logits_out, hipStream_t st) {
    if (x != h32) SVC_CHECK_HIP(hipMemcpyAsync(h32, x, (size_t)D * 4, hipMemcpyDeviceToDevice, st));
    for (int i = 0; i < L; ++i) {
        const Layer& ly = layers[i];
        if (i == 0)
            SVC_DEC_LAUNCH1(dec_qkvraw_kernel, KD, dim3(cdiv(Nqkv, 2)), dim3(64), 0, st, h32, ly.wc, D, Nqkv, qkv32);
        else
            SVC_DEC_LAUNCH2(dec_w2qkv_kernel, dim3(cdiv(D + Nqkv, 2)), dim3(64), 0, st, ff16, h32b, layers[i - 1].w2, ly.wc, I, D, Nqkv, h32,
                               qkv32);
        hipLaunchKernelGGL(dec_attn2_kernel, dim3(DEC_NS, H), dim3(512), 0, st, h32, qkv32, cfg.norm_eps, rope, ly.kc, ly.vc, ly.wo, part,
                           d_positions, H, Hkv, Lmax);
        if (H == 12)
            SVC_DEC_LAUNCH_NP(dec_ffn13_kernel, 12, dim3(cdiv(2 * I, 16)), dim3(256), 0, st, h32, part, H, h32b, ly.g_ffn, cfg.norm_eps,
                               ly.w13, D, 2 * I, ff16);
        else
            SVC_DEC_LAUNCH_NP(dec_ffn13_kernel, 0, dim3(cdiv(2 * I, 16)), dim3(256), 0, st, h32, part, H, h32b, ly.g_ffn, cfg.norm_eps,
                               ly.w13, D, 2 * I, ff16);
        SVC_CHECK_HIP(hipGetLastError());
    }
    SVC_DEC_LAUNCH1(dec_w2_kernel, KI, dim3(D / 2), dim3(64), 0, st, ff16, layers[L - 1].w2, I, D, h32b, h32);
    SVC_DEC_LAUNCH1(dec_head_kernel, KD, dim3(cdiv(V, 16)), dim3(256), 0, st, h32, g_final, cfg.norm_eps, w_out, D, V, logits_out);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

int svc_ar::run(const float* x, int S, const int* d_positions, float* logits_out, hipStream_t st) {
    // SVC_AR_DEC: 0 = generic path, 1 = four launches per layer, 2 (default) = three launches per layer
    static const int dec_mode = [] { const char* e = getenv("SVC_AR_DEC"); return e ? atoi(e) : 2; }();
    if (S == 1 && dec_mode && D <= 1024 && D % 8 == 0 && I % 8 == 0 && D <= 64 * 8 * DEC_MAXC && I <= 64 * 8 * DEC_MAXC && Nqkv % 2 == 0)
        return dec_mode >= 2 && have_wc ? run1_fused(x, d_positions, logits_out, st) : run1(x, d_positions, logits_out, st);
    if (x != h32) SVC_CHECK_HIP(hipMemcpyAsync(h32, x, (size_t)S * D * 4, hipMemcpyDeviceToDevice, st));
    const size_t attn_lds = ((size_t)Lmax + 1024) * sizeof(float);
    const bool fused = S <= 8;      // decode step: 5 launches per layer (norm / RoPE / cache scatter live in the GEMVs)
    for (int i = 0; i < L; ++i) {
        const Layer& ly = layers[i];
        if (fused) {
            GemvArgs a;
            memset(&a, 0, sizeof(a));
            a.x = h32; a.ldx = D; a.gamma = ly.g_attn; a.eps = cfg.norm_eps; a.W = ly.wqkv; a.ldw = D; a.S = S; a.N = Nqkv; a.K = D;
            a.q_out = q32; a.kc = ly.kc; a.vc = ly.vc; a.rope = rope; a.pos = d_positions; a.H = H; a.Hkv = Hkv; a.Lmax = Lmax;
            if (gemv_pair_launch<true, GV_QKV>(a, st)) return 1;
        } else {
            if (rmsnorm_mod_launch(h32, D, n16, D, ly.g_attn, nullptr, nullptr, 0, 0, S, D, S, cfg.norm_eps, st)) return 1;
            if (lin(n16, ly.wqkv, D, nullptr, qkv32, nullptr, Nqkv, S, Nqkv, D, false, st)) return 1;
            hipLaunchKernelGGL(ar_rope_cache_kernel, dim3(S), dim3(256), 0, st, qkv32, (long)Nqkv, q32, ly.kc, ly.vc, rope, d_positions,
                               S, H, Hkv, Lmax);
            SVC_CHECK_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(ar_attn_kernel, dim3(S, H), dim3(1024), attn_lds, st, q32, ly.kc, ly.vc, y16, d_positions, S, H, Hkv, Lmax);
        SVC_CHECK_HIP(hipGetLastError());
        if (fused) {
            GemvArgs a;
            memset(&a, 0, sizeof(a));
            a.x = y16; a.ldx = D; a.W = ly.wo; a.ldw = D; a.res = h32; a.ldres = D; a.out32 = h32; a.ldo = D; a.S = S; a.N = D; a.K = D;
            if (gemv_pair_launch<false, GV_PLAIN>(a, st)) return 1;
            memset(&a, 0, sizeof(a));
            a.x = h32; a.ldx = D; a.gamma = ly.g_ffn; a.eps = cfg.norm_eps; a.W = ly.w13; a.ldw = D; a.out16 = ff16; a.ldo = I;
            a.S = S; a.N = 2 * I; a.K = D;
            if (gemv_pair_launch<true, GV_GLU>(a, st)) return 1;
            memset(&a, 0, sizeof(a));
            a.x = ff16; a.ldx = I; a.W = ly.w2; a.ldw = I; a.res = h32; a.ldres = D; a.out32 = h32; a.ldo = D; a.S = S; a.N = D; a.K = I;
            if (gemv_pair_launch<false, GV_PLAIN>(a, st)) return 1;
        } else {
            if (lin(y16, ly.wo, D, h32, h32, nullptr, D, S, D, D, false, st)) return 1;
            if (rmsnorm_mod_launch(h32, D, n16, D, ly.g_ffn, nullptr, nullptr, 0, 0, S, D, S, cfg.norm_eps, st)) return 1;
            if (lin(n16, ly.w13, D, nullptr, nullptr, ff16, I, S, 2 * I, D, true, st)) return 1;
            if (lin(ff16, ly.w2, I, h32, h32, nullptr, D, S, D, I, false, st)) return 1;
        }
    }
    // last token only (ar.py:255-259): final RMSNorm fused into the output GEMV
    GemvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = h32 + (long)(S - 1) * D; a.ldx = D; a.gamma = g_final; a.eps = cfg.norm_eps; a.W = w_out; a.ldw = D;
    a.out32 = logits_out; a.ldo = V; a.S = 1; a.N = V; a.K = D;
    return gemv_pair_launch<true, GV_PLAIN>(a, st);
}

int svc_ar::ensure_graph() {
    svc_ar* m = this;
    if (!m->graph) {
        hipStream_t cs;
        SVC_CHECK_HIP(hipStreamCreate(&cs));
        hipGraph_t g = nullptr;
        SVC_CHECK_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        int rc = m->run(m->gx, 1, m->d_pos, m->logits, cs);
        if (!rc) {
            hipLaunchKernelGGL(advance_pos_kernel, dim3(1), dim3(64), 0, cs, m->d_pos, (int*)nullptr);
            if (hipGetLastError() != hipSuccess) rc = 1;
        }
        const hipError_t e = hipStreamEndCapture(cs, &g);
        if (rc || e != hipSuccess) {
            if (g) (void)hipGraphDestroy(g);
            (void)hipStreamDestroy(cs);
            if (!rc) set_error(std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
            return 1;
        }
        SVC_CHECK_HIP(hipGraphInstantiate(&m->graph, g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
        (void)hipStreamDestroy(cs);
    }
    return 0;
}

int svc_ar::sample(const float* lg, const int* prev, int n_prev, int suppress, float temperature, float top_p, float rep_pen,
                   const float* exp_noise, int* idx_out, float* probs_out, GenState* gs, hipStream_t st, bool prepare_next) {
    hipLaunchKernelGGL(ar_rank_kernel, dim3(cdiv(V, 16)), dim3(256), 0, st, lg, V, prev, n_prev, suppress, rep_pen, gs, d_skey, d_sidx, d_lgp);
    hipLaunchKernelGGL(ar_sample_kernel, dim3(1), dim3(1024), 0, st, d_lgp, V, d_skey, d_sidx, temperature, top_p, exp_noise, idx_out,
                       probs_out, gs, prepare_next ? emb : nullptr, prepare_next ? h32 : nullptr, D, d_pos);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

int svc_ar::ensure_gen_graph() {
    if (gen_graph) return 0;
    hipStream_t cs;
    SVC_CHECK_HIP(hipStreamCreate(&cs));
    hipGraph_t g = nullptr;
    SVC_CHECK_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    // the step runs in place on h32, which holds the embedding of the previous token (written by svc_ar_generate for the
    // first step and by the sampler of every step for the next one)
    int rc = run(h32, 1, d_pos, logits, cs);
    if (!rc) rc = sample(logits, nullptr, 0, -1, 1.f, 1.f, 1.f, nullptr, nullptr, nullptr, d_gen, cs, true);
    const hipError_t e = hipStreamEndCapture(cs, &g);
    if (rc || e != hipSuccess) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipStreamDestroy(cs);
        if (!rc) set_error(std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        return 1;
    }
    SVC_CHECK_HIP(hipGraphInstantiate(&gen_graph, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    (void)hipStreamDestroy(cs);
    return 0;
}

extern "C" {

int svc_ar_create(const svc_ar_config_t* cfg, const svc_tensor_desc_t* weights, int n_weights, void* stream, svc_ar_t** out) {
    SVC_REQUIRE(cfg && weights && out, "null argument");
    SVC_REQUIRE(cfg->head_dim == 64 && cfg->dim == cfg->n_head * 64 && cfg->n_head % cfg->n_local_heads == 0, "AR: head_dim 64, GQA");
    SVC_REQUIRE(cfg->dim % 64 == 0 && cfg->intermediate_size % 64 == 0 && cfg->vocab_size <= SORT_N, "AR: shape limits");
    SVC_REQUIRE(cfg->max_seq_len >= 8 && cfg->max_seq_len <= 8192, "AR: max_seq_len");
    hipStream_t st = (hipStream_t)stream;
    svc_ar* m = new svc_ar();
    m->cfg = *cfg;
    m->D = cfg->dim; m->H = cfg->n_head; m->Hkv = cfg->n_local_heads; m->L = cfg->n_layer; m->I = cfg->intermediate_size;
    m->V = cfg->vocab_size; m->Lmax = cfg->max_seq_len; m->kvd = m->Hkv * 64; m->Nqkv = m->D + 2 * m->kvd;
    StateDict sd(weights, n_weights);
    auto fail = [&]() { delete m; return 1; };
    const int D = m->D, I = m->I, V = m->V;
    auto pack16 = [&](const std::string& name, int N, int K, long row0, long row_step, half_t* dst, long ld) -> int {
        const auto* w = sd.get(name);
        if (require_shape(w, name, {N, K})) return 1;
        return pack_f16_launch(w->data, dst + row0 * ld, N, 1, K, K, 0, 1, row_step * ld, 0, 1, nullptr, st);
    };
    auto vec = [&](const std::string& name, int n) -> float* {
        const auto* w = sd.get(name);
        if (require_shape(w, name, {n})) return nullptr;
        float* d = m->wts.alloc_n<float>(n, st);
        if (d) (void)hipMemcpyAsync(d, w->data, n * 4, hipMemcpyDeviceToDevice, st);
        return d;
    };
    m->layers.resize(m->L);
    for (int i = 0; i < m->L; ++i) {
        const std::string p = "model.layers." + std::to_string(i) + ".";
        auto& ly = m->layers[i];
        ly.wqkv = m->wts.alloc_n<half_t>(round_up(m->Nqkv, 128) * (long)D, st);
        ly.wo = m->wts.alloc_n<half_t>(round_up(D, 128) * (long)D, st);
        ly.w13 = m->wts.alloc_n<half_t>(round_up(2 * I, 128) * (long)D, st);
        ly.w2 = m->wts.alloc_n<half_t>(round_up(D, 128) * (long)I, st);
        ly.kc = m->wts.alloc_n<float>((long)m->Hkv * m->Lmax * 64, st);
        ly.vc = m->wts.alloc_n<float>((long)m->Hkv * m->Lmax * 64, st);
        if (!ly.wqkv || !ly.wo || !ly.w13 || !ly.w2 || !ly.kc || !ly.vc) return fail();
        if (pack16(p + "attention.wqkv.weight", m->Nqkv, D, 0, 1, ly.wqkv, D)) return fail();
        if (pack16(p + "attention.wo.weight", D, D, 0, 1, ly.wo, D)) return fail();
        if (pack16(p + "feed_forward.w1.weight", I, D, 0, 2, ly.w13, D)) return fail();
        if (pack16(p + "feed_forward.w3.weight", I, D, 1, 2, ly.w13, D)) return fail();
        if (pack16(p + "feed_forward.w2.weight", D, I, 0, 1, ly.w2, I)) return fail();
        ly.g_attn = vec(p + "attention_norm.weight", D);
        ly.g_ffn = vec(p + "ffn_norm.weight", D);
        if (!ly.g_attn || !ly.g_ffn) return fail();
    }
    {   // three-launch decode form: Wq' = wqkv diag(gamma_attn) and, for layers >= 1, W' = Wq' w2_prev composed in fp32
        const int Nq = m->Nqkv;
        Arena tmp;
        float* wq = tmp.alloc_n<float>((size_t)round_up(Nq, 128) * D, st);           // Wq'            [Nq][D]
        float* w2t = tmp.alloc_n<float>((size_t)round_up(I, 128) * D, st);           // w2_prev^T      [I][D]
        float* wp = tmp.alloc_n<float>((size_t)round_up(Nq, 128) * I, st);           // W' = Wq' w2    [Nq][I]
        bool ok = wq && w2t && wp && D % 32 == 0 && I % 8 == 0;
        for (int i = 0; ok && i < m->L; ++i) {
            const std::string p = "model.layers." + std::to_string(i) + ".";
            auto& ly = m->layers[i];
            const auto* wqkv = sd.get(p + "attention.wqkv.weight");
            // columns scaled by gamma: dim0 of the pack = column index
            ok = ok && !pack_f32_launch(wqkv->data, wq, D, 1, Nq, 1, 0, D, 1, 0, D, ly.g_attn, st);
            if (i == 0) {
                ly.wc = m->wts.alloc_n<half_t>((size_t)round_up(Nq, 2) * D, st);
                ok = ok && ly.wc && !pack_f16_launch(wq, ly.wc, Nq, 1, D, D, 0, 1, D, 0, 1, nullptr, st);
                continue;
            }
            const auto* w2 = sd.get("model.layers." + std::to_string(i - 1) + ".feed_forward.w2.weight");        // [D][I]
            ok = ok && !pack_f32_launch(w2->data, w2t, I, 1, D, 1, 0, I, D, 0, 1, nullptr, st);                    // -> [I][D]
            if (!ok) break;
            KGemmParams g;
            memset(&g, 0, sizeof(g));
            g.M = Nq; g.N = I; g.Lout = Nq; g.a_seq_rows = Nq; g.c_seq_rows = Nq; g.a_stride = 1; g.a_len = Nq;
            g.n_taps = 1; g.a_ptr[0] = wq; g.a_ld[0] = D; g.a_ktiles[0] = D / 32;       // fp32 k-tiles of 32 elements (128 bytes)
            g.w = w2t; g.ldw = D; g.c32 = wp; g.ldc32 = I; g.vec_ok = 1;
            ok = ok && !kgemm_launch(g, 1, KG_EPI_STORE, st);                            // exact fp32 fma chain (v_mfma_f32_16x16x4_f32)
            const long ldc = (long)I + D;
            ly.wc = m->wts.alloc_n<half_t>((size_t)round_up(Nq, 2) * ldc, st);
            ok = ok && ly.wc && !pack_f16_launch(wp, ly.wc, Nq, 1, I, I, 0, 1, ldc, 0, 1, nullptr, st) &&
                 !pack_f16_launch(wq, ly.wc + I, Nq, 1, D, D, 0, 1, ldc, 0, 1, nullptr, st);
        }
        if (hipStreamSynchronize(st) != hipSuccess) ok = false;                           // tmp is freed on scope exit
        m->have_wc = ok;
        if (!ok) (void)hipGetLastError();
    }
    m->g_final = vec("model.norm.weight", D);
    m->w_out = m->wts.alloc_n<half_t>(round_up(V, 128) * (long)D, st);
    if (!m->g_final || !m->w_out) return fail();
    if (pack16("model.output.weight", V, D, 0, 1, m->w_out, D)) return fail();
    {   // bf16-rounded RoPE table over max_seq_len positions (ar.py:624-632)
        std::vector<float> tab((size_t)m->Lmax * 64);
        for (int i = 0; i < 32; ++i) {
            const float f = 1.0f / powf(cfg->rope_base, (float)(2 * i) / 64.0f);
            for (int t = 0; t < m->Lmax; ++t) {
                const float a = (float)t * f;
                float c = (float)cos((double)a), s = (float)sin((double)a);
                uint32_t u;
                memcpy(&u, &c, 4); u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000u; memcpy(&c, &u, 4);
                memcpy(&u, &s, 4); u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000u; memcpy(&s, &u, 4);
                tab[((size_t)t * 32 + i) * 2] = c;
                tab[((size_t)t * 32 + i) * 2 + 1] = s;
            }
        }
        m->rope = m->wts.alloc_n<float>(tab.size(), st);
        if (!m->rope) return fail();
        if (hipMemcpyAsync(m->rope, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { set_error("rope upload failed"); return fail(); }
    }
    if (const auto* e = sd.get("model.embeddings.weight")) {       // optional: only svc_ar_generate needs it
        if (require_shape(e, "model.embeddings.weight", {V, D})) return fail();
        m->emb = m->wts.alloc_n<float>((size_t)V * D, st);
        if (!m->emb) return fail();
        if (hipMemcpyAsync(m->emb, e->data, (size_t)V * D * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            set_error("embedding copy failed");
            return fail();
        }
    }
    if (hipStreamSynchronize(st) != hipSuccess) { set_error("sync failed"); return fail(); }
    *out = m;
    return 0;
}

void svc_ar_destroy(svc_ar_t* m) {
    if (m && m->graph) (void)hipGraphExecDestroy(m->graph);
    if (m && m->gen_graph) (void)hipGraphExecDestroy(m->gen_graph);
    delete m;
}

int svc_ar_reset(svc_ar_t* m, void* stream) {
    SVC_REQUIRE(m, "null argument");
    for (auto& ly : m->layers) {
        SVC_CHECK_HIP(hipMemsetAsync(ly.kc, 0, (size_t)m->Hkv * m->Lmax * 64 * 4, (hipStream_t)stream));
        SVC_CHECK_HIP(hipMemsetAsync(ly.vc, 0, (size_t)m->Hkv * m->Lmax * 64 * 4, (hipStream_t)stream));
    }
    return 0;
}

int svc_ar_forward_generate(svc_ar_t* m, const float* x, int S, const int64_t* input_pos, const int64_t* kv_pos, float* logits_out,
                            void* stream) {
    SVC_REQUIRE(m && x && input_pos && kv_pos && logits_out && S >= 1, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (m->reserve(S, st)) return 1;
    std::vector<int> pos(2 * S);
    for (int s = 0; s < S; ++s) {
        SVC_REQUIRE(input_pos[s] >= 0 && input_pos[s] < m->Lmax && kv_pos[s] >= 0 && kv_pos[s] < m->Lmax, "position out of range");
        pos[s] = (int)input_pos[s];
        pos[S + s] = (int)kv_pos[s];
    }
    SVC_CHECK_HIP(hipMemcpyAsync(m->d_pos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice, st));
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    return m->run(x, S, m->d_pos, logits_out, st);
}

// One-token decode step replayed from a hipGraph.  The first call (or a call with set_pos != 0) sets the device
// positions {input_pos, kv_pos}; every replay advances both by one, as NaiveWrapper.generate does (ar.py:402-403).
int svc_ar_decode_step(svc_ar_t* m, const float* x, int set_pos, int64_t input_pos, int64_t kv_pos, float* logits_out, void* stream) {
    SVC_REQUIRE(m && x && logits_out, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (m->reserve(1, st)) return 1;
    if (set_pos) {
        SVC_REQUIRE(input_pos >= 0 && input_pos < m->Lmax && kv_pos >= 0 && kv_pos < m->Lmax, "position out of range");
        const int pos[2] = {(int)input_pos, (int)kv_pos};
        SVC_CHECK_HIP(hipMemcpyAsync(m->d_pos, pos, 8, hipMemcpyHostToDevice, st));
        SVC_CHECK_HIP(hipStreamSynchronize(st));
    }
    if (m->ensure_graph()) return 1;
    SVC_CHECK_HIP(hipMemcpyAsync(m->gx, x, (size_t)m->D * 4, hipMemcpyDeviceToDevice, st));
    SVC_CHECK_HIP(hipGraphLaunch(m->graph, st));
    SVC_CHECK_HIP(hipMemcpyAsync(logits_out, m->logits, (size_t)m->V * 4, hipMemcpyDeviceToDevice, st));
    return 0;
}

// Whole generation loop of NaiveWrapper.generate (modules/v2/ar.py:382-422) for B = 1: prefill, then one captured
// decode step + on-device sampling per token; the host only looks at the tokens every `check_every` steps (EOS check),
// so there is no per-token synchronisation.  Tokens produced speculatively after an EOS are discarded.
int svc_ar_generate(svc_ar_t* m, const float* x_prefill, int S, const int64_t* input_pos, const int64_t* kv_pos,
                    const float* exp_noise, int max_new, int min_tokens_before_eos, float temperature, float top_p,
                    float repetition_penalty, int check_every, int32_t* tokens_out, int32_t* n_tokens, void* stream) {
    SVC_REQUIRE(m && x_prefill && input_pos && kv_pos && exp_noise && tokens_out && n_tokens && S >= 1 && max_new >= 1, "bad argument");
    SVC_REQUIRE(m->emb, "svc_ar_generate needs model.embeddings.weight in the state dict given to svc_ar_create");
    hipStream_t st = (hipStream_t)stream;
    const int V = m->V, eos = V - 1;
    if (check_every < 1) check_every = 16;
    // prefill + first token (EOS suppressed, no previous tokens: ar.py:399-401)
    if (m->reserve(S, st)) return 1;       // m->logits exists from here on
    if (svc_ar_forward_generate(m, x_prefill, S, input_pos, kv_pos, m->logits, stream)) return 1;
    if (m->sample(m->logits, nullptr, 0, eos, temperature, top_p, repetition_penalty, exp_noise, tokens_out, nullptr, nullptr, st)) return 1;
    if (m->reserve(1, st)) return 1;
    const int pos[2] = {(int)input_pos[S - 1] + 1, (int)kv_pos[S - 1] + 1};
    SVC_CHECK_HIP(hipMemcpyAsync(m->d_pos, pos, 8, hipMemcpyHostToDevice, st));
    GenState gs;
    gs.noise = exp_noise; gs.toks = tokens_out; gs.cnt = 1; gs.min_before_eos = min_tokens_before_eos; gs.eos = eos;
    gs.temperature = temperature; gs.top_p = top_p; gs.rep_pen = repetition_penalty;
    SVC_CHECK_HIP(hipMemcpyAsync(m->d_gen, &gs, sizeof(gs), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(ar_embed_kernel, dim3(1), dim3(256), 0, st, m->emb, m->d_gen, m->h32, m->D);   // input of the first step
    SVC_CHECK_HIP(hipGetLastError());
    SVC_CHECK_HIP(hipStreamSynchronize(st));
    if (m->ensure_gen_graph()) return 1;
    std::vector<int32_t> host(max_new);
    int n = 1, checked = 1, t = 1;
    bool done = false;
    while (!done) {
        const int t_end = std::min(max_new, t + check_every);
        for (; t < t_end; ++t) {
            if (pos[0] + (t - 1) >= m->Lmax || pos[1] + (t - 1) >= m->Lmax) { done = true; break; }   // cache / RoPE table exhausted
            SVC_CHECK_HIP(hipGraphLaunch(m->gen_graph, st));      // decode step on embed(token t-1) -> sample token t, embed it, advance
        }
        if (t > checked) {
            SVC_CHECK_HIP(hipMemcpyAsync(host.data() + checked, tokens_out + checked, (size_t)(t - checked) * 4, hipMemcpyDeviceToHost, st));
            SVC_CHECK_HIP(hipStreamSynchronize(st));
            for (int i = checked; i < t; ++i) {
                if (host[i] == eos) { done = true; break; }
                n = i + 1;
            }
            checked = t;
        }
        if (t >= max_new) done = true;
    }
    *n_tokens = n;
    return 0;
}

int svc_ar_sample(svc_ar_t* m, const float* logits, const int32_t* prev_tokens, int n_prev, int suppress_token, float temperature,
                  float top_p, float repetition_penalty, const float* exp_noise, int32_t* idx_out, float* probs_out, void* stream) {
    SVC_REQUIRE(m && logits && exp_noise && idx_out && n_prev >= 0 && (n_prev == 0 || prev_tokens), "bad argument");
    if (m->reserve(1, (hipStream_t)stream)) return 1;
    return m->sample(logits, prev_tokens, n_prev, suppress_token, temperature, top_p, repetition_penalty, exp_noise, idx_out, probs_out,
                     nullptr, (hipStream_t)stream);
}

}  // extern "C"
