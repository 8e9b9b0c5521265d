// Stride-1 Conv1d on channels-last activations with the activation tile RESIDENT in LDS across the taps.
//
// The tap-GEMM (kgemm.hip) streams an [rows][64 B] activation tile from L2 for every (tap, sub-product) k-tile; for a
// k-tap conv that is the same rows, shifted, k (x2 in split precision) times -- and the L2 -> LDS stream is what bounds
// its main loop ("Known headroom").  Here a workgroup owns 256 consecutive output positions of ONE sequence and
// 128 output channels; per 64-channel chunk it loads the 256 + (k-1) dil input rows once (hi and lo planes), then walks
// the taps reading shifted row windows of that tile from LDS, while only the weight tiles (16 KB each) stream through a
// 3-stage LDS-DMA ring.  L2 -> LDS traffic per FLOP drops ~3.7x for an 11-tap split-precision conv.
//
// Products per (chunk, tap): fp16 mode a * w ; split mode a_hi * w_hi + a_hi * w_lo + a_lo * w_hi (weights packed
// [w_hi | w_lo | w_hi] per tap as for the tap-GEMM); NSUB = 2, "fp16 + fp8 corrections": a_hi * w_hi on the fp16 MFMA and
// both correction products in ONE block-scaled fp8 MFMA of K = 128 on the byte-pair planes (lo_pair_p8, common.h) --
// two units of matrix-pipe time per (chunk, tap) instead of three (tools/ubench/mfma_f8_rate: 1.50x in a bare loop).  Epilogue = the STORE epilogue of the tap-GEMM (bias, residual,
// scale, second addend, activation, fp32 and / or fp16 hi/lo outputs with the fused pointwise Snake).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace svc {

namespace {

constexpr int CB_NW = 8, CB_NT = CB_NW * 64;      // BM (256; 128 | 64 for small grids) positions x BN (128 | 64) channels, 8 waves
constexpr int CB_SPAN = 64;                       // largest (k - 1) dil served
constexpr int cb_a_bytes(int bm) { return (bm + CB_SPAN) * 128; }   // one plane of the activation tile (64 channels = 128 B per row)
constexpr int CB_EPI_LD = 68;
// weight ring depth: the 256-row tile takes what the 160 KB of LDS leave beside the activation tile (5 x 16 KB | 8 x 8 KB;
// measured +1 % over 3 stages on the B = 64 bench); the small-grid tiles keep 3 stages so two workgroups share a CU
constexpr int cb_ns(int bn, int bm) { return bm == 128 ? 2 : (bm < 256 ? 3 : (bn == 128 ? 5 : 8)); }    // 128 / 64-row tiles: 80 KB, 2 workgroups per CU
constexpr int cb_lds(int bn, int bm) { return 2 * cb_a_bytes(bm) + cb_ns(bn, bm) * bn * 128; }      // BM 256: 80 KB + 80 | 64 KB

// chunk swizzle of the 128-byte-row LDS images: physical 16-byte slot = chunk ^ (row & 7).  A ds_read_b128 is served in
// four groups of 16 lanes -- {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS) -- on 64 banks = two 128-byte rows; with
// this key ANY 16 consecutive rows read conflict-free, which matters here because the tap-shifted activation windows
// start at arbitrary rows.  (The tap-GEMM's key, built for aligned 16-row groups and its permuted weight rows, costs
// 2.5x on unaligned windows: PMC showed SQ_LDS_BANK_CONFLICT = 40 % of SQ_LDS_IDX_ACTIVE in this kernel with it.)
__device__ __forceinline__ int cswz(int row) { return row & 7; }

template <int NSUB, int BN, int BM>
__global__ __launch_bounds__(CB_NT, BM < 256 ? 2 : 1) void kconv_kernel(const KConvParams p) {
    constexpr int CB_N = BN, CB_M = BM;
    constexpr int CB_NS = cb_ns(BN, BM);
    constexpr int CB_A_BYTES = cb_a_bytes(BM);
    constexpr int CB_W_BYTES = BN * 128;            // one weight tile
    constexpr int WROWS_M = BN == 128 ? BM / 4 : BM / 8;   // wave tile rows: (4 x 2 waves) x 64 columns | (8 x 1 waves) x 64 columns
    constexpr int TM = WROWS_M / 16;
    static_assert(TM >= 1, "tile");
    constexpr int WDPT = BN / 64;                   // weight-tile DMA instructions per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* a_hi = smem;
    char* a_lo = smem + CB_A_BYTES;
    char* w_ring = smem + 2 * CB_A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int n_mt = (p.Lout + CB_M - 1) / CB_M;
    const int n_nt = (p.N + CB_N - 1) / CB_N;
    // column tile fastest: the workgroups sharing one activation tile are neighbours IN THE SAME XCD's L2 -- blocks are
    // dealt round-robin over the 8 XCDs (bid % 8 labels the group), so logical tile ids are handed out in contiguous ranges
    // per XCD (as in the tap-GEMM; bijective for any grid size).  With the plain bid order the 2 .. 6 column tiles of a
    // 192 .. 768-channel layer sat on different XCDs and each fetched the activation tile on its own (PMC: 701 MB fetched
    // per launch against ~340 MB of operands).
    const int nblk = gridDim.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
    const int bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
    const int tile_n = bid % n_nt;
    const int rest = bid / n_nt;
    const int tile_m = rest % n_mt;
    const int b = rest / n_mt;
    const int p0 = tile_m * CB_M;                 // first output position of this workgroup
    const int n0 = tile_n * CB_N;
    const int span = (p.k - 1) * p.dil;
    const int R = CB_M + span;                    // tile rows in use
    const int nchunks = p.cin_pad / 64;
    const int total = nchunks * p.k * NSUB;       // weight tiles

    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const char* zero_ = reinterpret_cast<const char*>(p.zero_page);

    // ---- MFMA geometry
    const int wm0 = BN == 128 ? (wave >> 1) * WROWS_M : wave * WROWS_M, wn0 = BN == 128 ? (wave & 1) * 64 : 0;
    const int fr = lane & 15, fq = lane >> 4;
    float4v acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};

    // ---- staging geometry: a wave-level DMA instruction writes 8 rows x 128 B; the workgroup covers 64 rows per pass
    const int c8 = tid & 7, rr = tid >> 3;
    // weight tile source pointers (advance by tile): rows n0 + rr, n0 + rr + 64
    const half_t* wrow[WDPT];
#pragma unroll
    for (int i = 0; i < WDPT; ++i) {
        const int row = rr + 64 * i;
        wrow[i] = reinterpret_cast<const half_t*>(p.w) + (long)(n0 + row) * p.ldw + ((c8 ^ cswz(row)) << 3);
    }
    auto issue_w = [&](int it, int stage) {
        // tile `it` = (chunk c, tap t, sub s), it = (c * k + t) * NSUB + s ; weight column = (t * NSUB + s) * cin_pad + 64 c
        const int c = it / (p.k * NSUB);
        const int ts = it - c * (p.k * NSUB);
        const long col = (long)ts * p.cin_pad + 64L * c;
        char* dst = w_ring + stage * CB_W_BYTES + wave_u * 1024;
#pragma unroll
        for (int i = 0; i < WDPT; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(wrow[i] + col), (lptr_t)(dst + i * 64 * 128), 16, 0, 0);
    };
    auto issue_a = [&](int c) {
        for (int i = 0; i * 64 < R; ++i) {
            const int row = rr + 64 * i;
            const int pos = p0 - p.pad_left + row;
            const bool ok = (row < R) & (pos >= 0) & (pos < p.Lin);
            const long off = (((long)b * p.Lin + (ok ? pos : 0)) * p.cin_pad + 64L * c + ((c8 ^ cswz(row)) << 3)) * 2;
            const char* sh = ok ? reinterpret_cast<const char*>(p.a_hi) + off : zero_;
            __builtin_amdgcn_global_load_lds((gptr_t)sh, (lptr_t)(a_hi + (64 * i + wave_u * 8) * 128), 16, 0, 0);
            if constexpr (NSUB != 1) {
                const char* sl = ok ? reinterpret_cast<const char*>(p.a_lo) + off : zero_;
                __builtin_amdgcn_global_load_lds((gptr_t)sl, (lptr_t)(a_lo + (64 * i + wave_u * 8) * 128), 16, 0, 0);
            }
        }
    };

    // ---- main loop
#pragma unroll
    for (int s_ = 0; s_ < CB_NS - 1; ++s_)
        if (s_ < total) issue_w(s_, s_);
    int stage = 0, fill = CB_NS - 1, it = 0;
    for (int c = 0; c < nchunks; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");            // every wave is done with the previous chunk's tile
        issue_a(c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        for (int t = 0; t < p.k; ++t) {
            const int shift = t * p.dil;
#pragma unroll
            for (int s = 0; s < NSUB; ++s, ++it) {
                const int ahead = total - 1 - it;
                wait_tiles<CB_NS - 2, WDPT>(ahead);             // the next tiles' DMAs may stay in flight
                asm volatile("s_barrier" ::: "memory");
                if (it + CB_NS - 1 < total) issue_w(it + CB_NS - 1, fill);
                const char* at = ((NSUB == 3 && s == 2) || (NSUB == 2 && s == 1)) ? a_lo : a_hi;
                const char* wt = w_ring + stage * CB_W_BYTES;
                if constexpr (NSUB == 2) {
                    if (s == 1) {
                        // one K = 128 MFMA per tile pair: lane (fr, fq) supplies the 32 bytes of chunks fq and 4 + fq of its row
                        // for both operands (any K order works as long as it is the same on both sides; this one keeps the
                        // fp16 path's conflict-free read pattern); E8M0 scales: 2^-11 on the activation side, 2^-w8_exp on the weights
                        typedef int v8i __attribute__((ext_vector_type(8)));
                        v8i af8[TM], bf8[4];
#pragma unroll
                        for (int mt = 0; mt < TM; ++mt) {
                            const int row = wm0 + mt * 16 + fr + shift;
                            const u32x4 x0 = *reinterpret_cast<const u32x4*>(at + row * 128 + ((fq ^ cswz(row)) << 4));
                            const u32x4 x1 = *reinterpret_cast<const u32x4*>(at + row * 128 + (((4 + fq) ^ cswz(row)) << 4));
                            af8[mt] = (v8i){(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[2], (int)x1[3]};
                        }
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const int row = wn0 + nt * 16 + fr;
                            const u32x4 x0 = *reinterpret_cast<const u32x4*>(wt + row * 128 + ((fq ^ cswz(row)) << 4));
                            const u32x4 x1 = *reinterpret_cast<const u32x4*>(wt + row * 128 + (((4 + fq) ^ cswz(row)) << 4));
                            bf8[nt] = (v8i){(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[2], (int)x1[3]};
                        }
                        const int sc_a = 127 - 11, sc_b = 127 - p.w8_exp;
#pragma unroll
                        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af8[mt], bf8[nt], acc[mt][nt], 0, 0, 0, sc_a, 0, sc_b);
                        stage = stage + 1 == CB_NS ? 0 : stage + 1;
                        fill = fill + 1 == CB_NS ? 0 : fill + 1;
                        continue;
                    }
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4 af[TM], bf[4];
                    const int chunk = ks * 4 + fq;
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) {
                        const int row = wm0 + mt * 16 + fr + shift;
                        af[mt] = *reinterpret_cast<const u32x4*>(at + row * 128 + ((chunk ^ cswz(row)) << 4));
                    }
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int row = wn0 + nt * 16 + fr;
                        bf[nt] = *reinterpret_cast<const u32x4*>(wt + row * 128 + ((chunk ^ cswz(row)) << 4));
                    }
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, af[mt]),
                                                                                 __builtin_bit_cast(half8, bf[nt]), acc[mt][nt], 0, 0, 0);
                }
                stage = stage + 1 == CB_NS ? 0 : stage + 1;
                fill = fill + 1 == CB_NS ? 0 : fill + 1;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue: accumulators transposed through LDS (passes of RP = 32 | 16 rows per wave), 8 consecutive columns per lane
    constexpr int RP = TM >= 2 ? 32 : 16;
    float* ep = reinterpret_cast<float*>(smem) + wave * RP * CB_EPI_LD;
#pragma unroll
    for (int pass = 0; pass < TM * 16 / RP; ++pass) {
        if (pass > 0) __syncthreads();
#pragma unroll
        for (int mi = 0; mi < RP / 16; ++mi)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ep[(mi * 16 + fq * 4 + r) * CB_EPI_LD + nt * 16 + fr] = acc[pass * (RP / 16) + mi][nt][r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RP / 8; ++i) {            // RP rows x 8 chunks per wave pass
            const int ch = lane + 64 * i;
            const int row = ch >> 3, cc = ch & 7;
            const int pos = p0 + wm0 + pass * RP + row;
            const int n = n0 + wn0 + cc * 8;
            float v[8];
            {
                const float4v x0 = *reinterpret_cast<const float4v*>(ep + row * CB_EPI_LD + cc * 8);
                const float4v x1 = *reinterpret_cast<const float4v*>(ep + row * CB_EPI_LD + cc * 8 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = x0[j]; v[4 + j] = x1[j]; }
            }
            if (pos >= p.Lout || n >= p.N) continue;
            const long orow = (long)b * p.c_seq_rows + p.c_off + pos;
            if (p.bias) {
                const float4v b0 = *reinterpret_cast<const float4v*>(p.bias + n);
                const float4v b1 = *reinterpret_cast<const float4v*>(p.bias + n + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
            }
            if (p.act != KG_ACT_NONE) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = act_apply(v[j], p.act, p.act_slope);
            }
            if (p.res) {
                const float4v q0 = *reinterpret_cast<const float4v*>(p.res + orow * p.ldres + n);
                const float4v q1 = *reinterpret_cast<const float4v*>(p.res + orow * p.ldres + n + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] += q0[j]; v[4 + j] += q1[j]; }
            }
            if (p.out_scale != 0.f) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= p.out_scale;
            }
            if (p.res2) {
                const float4v q0 = *reinterpret_cast<const float4v*>(p.res2 + orow * p.ldres2 + n);
                const float4v q1 = *reinterpret_cast<const float4v*>(p.res2 + orow * p.ldres2 + n + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] += q0[j]; v[4 + j] += q1[j]; }
            }
            if (p.c32) {
                *reinterpret_cast<float4v*>(p.c32 + orow * p.ldc32 + n) = (float4v){v[0], v[1], v[2], v[3]};
                *reinterpret_cast<float4v*>(p.c32 + orow * p.ldc32 + n + 4) = (float4v){v[4], v[5], v[6], v[7]};
            }
            if (p.post_a) {
                float lo[8];
                float hf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int nn = n + j;
                    const float sv = nn < p.post_n ? v[j] + p.post_ib[nn] * sin_sq(p.post_a[nn] * v[j]) : 0.f;
                    const half_t hh = (half_t)sv;
                    v[j] = sv;
                    lo[j] = sv - (float)hh;
                    hf[j] = (float)hh;
                }
                if (p.c16_lo) {
                    if (p.c16_lo_fmt) {
                                const u32x4 pr = {lo_pair_p8(hf[0], lo[0]) | (lo_pair_p8(hf[1], lo[1]) << 16), lo_pair_p8(hf[2], lo[2]) | (lo_pair_p8(hf[3], lo[3]) << 16),
                                                  lo_pair_p8(hf[4], lo[4]) | (lo_pair_p8(hf[5], lo[5]) << 16), lo_pair_p8(hf[6], lo[6]) | (lo_pair_p8(hf[7], lo[7]) << 16)};
                                *reinterpret_cast<u32x4*>(p.c16_lo + orow * p.ldc16 + n) = pr;
                            }
                    else *reinterpret_cast<uint4*>(p.c16_lo + orow * p.ldc16 + n) = pack8(lo);
                }
            }
            if (p.c16) *reinterpret_cast<uint4*>(p.c16 + orow * p.ldc16 + n) = pack8(v);
        }
    }
}


// ---- weights of the fp16 + fp8-corrections mode: byte pairs (fp8(2^11 s w_lo), fp8(s w_hi)), s = 2^exp from the largest |w|
__global__ void absmax_kernel(const float* __restrict__ src, int n0, int n1, int n2, long s0, long s1, long s2,
                              const float* __restrict__ scale, float* __restrict__ out) {
    __shared__ float red[256];
    float m = 0.f;
    const long total = (long)n0 * n1 * n2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += 256L * gridDim.x) {
        const int i2 = (int)(i % n2);
        const long r = i / n2;
        const int i1 = (int)(r % n1), i0 = (int)(r / n1);
        float v = src[i0 * s0 + i1 * s1 + i2 * s2];
        if (scale) v *= scale[i0];
        m = fmaxf(m, fabsf(v));
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    // non-negative floats order like their bit patterns: one atomic per block, deterministic
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(red[0]));
}

__global__ void pack_p8_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int n0, int n1, int n2, long s0, long s1,
                               long s2, long d0, long d1, long d2, const float* __restrict__ scale, float sw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n0 * n1 * n2;
    if (i >= total) return;
    const int i2 = (int)(i % n2);
    const long r = i / n2;
    const int i1 = (int)(r % n1), i0 = (int)(r / n1);
    float v = src[i0 * s0 + i1 * s1 + i2 * s2];
    if (scale) v *= scale[i0];
    const float hi = (float)(half_t)v, lo = v - hi;
    // byte 0 pairs with the activation's fp8(hi) byte, byte 1 with its fp8(2^11 lo) byte
    const float b0 = fminf(fmaxf(lo * sw * 2048.f, -448.f), 448.f), b1 = fminf(fmaxf(hi * sw, -448.f), 448.f);
    dst[i0 * d0 + i1 * d1 + i2 * d2] = (unsigned short)((unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(b0, b1, 0, false) & 0xffffu);
}

}  // namespace

int kconv_pack_p8(const float* src, unsigned short* dst, int n0, int n1, int n2, long s0, long s1, long s2, long d0, long d1, long d2,
                  const float* scale, int* exp_out, hipStream_t st) {
    float* d_max = nullptr;
    SVC_CHECK_HIP(hipMalloc(&d_max, sizeof(float)));
    SVC_CHECK_HIP(hipMemsetAsync(d_max, 0, sizeof(float), st));
    hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, st, src, n0, n1, n2, s0, s1, s2, scale, d_max);
    float mx = 0.f;
    hipError_t e = hipMemcpyAsync(&mx, d_max, sizeof(float), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_max);
    SVC_CHECK_HIP(e);
    int ex = 0;
    if (mx > 0.f) {
        ex = (int)floorf(log2f(224.0f / mx));       // largest |s w| in [112, 224]: inside e4m3's 448
        ex = ex > 24 ? 24 : (ex < -24 ? -24 : ex);
    }
    *exp_out = ex;
    const long n = (long)n0 * n1 * n2;
    hipLaunchKernelGGL(pack_p8_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, src, dst, n0, n1, n2, s0, s1, s2, d0, d1, d2, scale,
                       ldexpf(1.0f, ex));
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

namespace {
}  // namespace

bool kconv_enabled() {
    static const bool off = [] { const char* e = getenv("SVC_KCONV"); return e && e[0] == '0'; }();
    return !off;
}

template <int NSUB, int BN, int BM>
int kconv_go(DeviceState* ds, const KConvParams& p, int grid, hipStream_t st) {
    // per device and instantiation: the attribute lives in the device's code object
    constexpr unsigned bit = 1u << ((NSUB == 3 ? 1 : (NSUB == 2 ? 2 : 0)) + 3 * ((BN == 64 ? 3 : 0) + (BM == 64 ? 0 : (BM == 128 ? 1 : 2))));
    if (!(ds->kconv_attr.load(std::memory_order_acquire) & bit)) {
        SVC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kconv_kernel<NSUB, BN, BM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          cb_lds(BN, BM)));
        ds->kconv_attr.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((kconv_kernel<NSUB, BN, BM>), dim3(grid), dim3(CB_NT), cb_lds(BN, BM), st, p);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

int kconv_launch(const KConvParams& p_in, hipStream_t st) {
    SVC_REQUIRE(p_in.k >= 1 && (p_in.k - 1) * p_in.dil <= CB_SPAN && p_in.cin_pad % 64 == 0 && p_in.N % 8 == 0, "kconv shape");
    DeviceState* ds = device_state();
    if (!ds) return 1;
    KConvParams p = p_in;
    p.zero_page = ds->zero_page;
    const int bn = p.N <= 64 ? 64 : 128;
    // Position tile (128-channel form).  128 rows with a 2-stage weight ring = 80 KB of LDS, so TWO workgroups share a CU
    // and one's activation-tile loads and epilogue run under the other's MFMAs: measured against the 256-row tile (one
    // workgroup per CU, 5-stage ring) the conv class of the default bench takes 66.7 vs 71.0 ms.  A single utterance gives
    // the early vocoder stages (L = 1.7 k .. 7 k positions) too few tiles even so: 64 rows there.  The summation order of
    // an output element (chunk, tap, sub-product) is the same in every form: bit-identical results.
    static const int bm_env = [] { const char* e = getenv("SVC_KCONV_BM"); return e ? atoi(e) : 0; }();
    const long g256 = (long)p.B * cdiv(p.Lout, 256) * cdiv(p.N, bn);
    int bm = 256;
    if (bn == 128) bm = g256 <= 96 ? 64 : 128;
    if (bn == 128 && (bm_env == 64 || bm_env == 128 || bm_env == 256)) bm = bm_env;
    const int grid = p.B * cdiv(p.Lout, bm) * cdiv(p.N, bn);
    if (grid <= 0) return 0;
    const bool prof = prof_enabled();
    if (prof) prof_begin(PROF_KGEMM_F16, st);
    int rc;
#define SVC_KCONV_GO(BN_, BM_) (p.nsub == 3 ? kconv_go<3, BN_, BM_>(ds, p, grid, st) : (p.nsub == 2 ? kconv_go<2, BN_, BM_>(ds, p, grid, st) : kconv_go<1, BN_, BM_>(ds, p, grid, st)))
    if (bn == 128) {
        if (bm == 64) rc = SVC_KCONV_GO(128, 64);
        else if (bm == 128) rc = SVC_KCONV_GO(128, 128);
        else rc = SVC_KCONV_GO(128, 256);
    } else {
        rc = SVC_KCONV_GO(64, 256);
    }
#undef SVC_KCONV_GO
    if (rc) return rc;
    if (prof) {
        const double M = (double)p.B * p.Lout, K = (double)p.k * p.cin_pad;
        double bytes = (M * p.cin_pad * (p.nsub != 1 ? 2 : 1) + (double)p.N * K * p.nsub) * 2.0;
        bytes += M * p.N * ((p.c32 ? 4 : 0) + (p.c16 ? 2 : 0) + (p.c16_lo ? 2 : 0) + (p.res ? 4 : 0) + (p.res2 ? 4 : 0));
        const unsigned long long tag = ((unsigned long long)(long)M << 40) | ((unsigned long long)(p.N & 0xFFFFF) << 20) |
                                       ((unsigned long long)((long)(K * p.nsub) & 0xFFFF) << 4) | 4u;
        prof_end(PROF_KGEMM_F16, 2.0 * M * p.N * K, bytes, st, tag);
    }
    return 0;
}

}  // namespace svc
