// Flash-style self-attention for the DiT blocks (head_dim 64, non-causal, key-padding by length).
//
// Layout contract (produced by the QKV tap-GEMM epilogue): q, k fp16 [rows][ld] with head h at
// column h*64, already rotated (RoPE) and q pre-scaled by log2(e)/sqrt(64); v transposed
// vt[seq][h*64 + d][key] so that both MFMA contractions read K-contiguous operands.
//
// Block = 4 waves = 128 queries of one (sequence, head); each wave owns 32 queries (two 16-wide
// MFMA column tiles).  Scores are computed transposed, S^T = K Q^T (keys on the accumulator rows),
// which makes the softmax row statistics lane-local up to two cross-lane maxima and lets the
// accumulator registers feed P^T straight into O^T += V^T P^T as the B operand (no LDS round trip).
// K and V^T tiles (64 keys) are double buffered in LDS with the same XOR-swizzled 128-byte-row image
// as the GEMM (conflict-free ds_read_b128 / ds_read_b64).
#include "common.h"

namespace svc {
namespace {

constexpr int KT = 64;        // keys per tile
constexpr int ROWB = 128;

__device__ __forceinline__ int lds_off(int row, int c16) { return row * ROWB + ((c16 ^ ((row >> 1) & 7)) << 4); }

__global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * KT * ROWB];   // [buf][K | Vt][64][128B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    // 1-D grid with an XCD-aware order: the query tiles of one (sequence, head) -- which all stream the same K / V^T --
    // are given to one XCD (bid % 8 labels the XCD group), so K/V are fetched into one L2 instead of up to 8.
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int nqt = (p.Tq + 127) / 128;
    const int qt_idx = lid % nqt;
    const int sh_idx = lid / nqt;
    const int h = sh_idx % p.H, seq = sh_idx / p.H;
    const int q0 = qt_idx * 128 + wave * 32;
    const long row_base = (long)seq * p.seq_rows;
    const int kv_len = p.kv_len ? p.kv_len[seq] : p.kv_len_const;
    const int n_kt = (kv_len + KT - 1) / KT;

    // Q fragments (B operand of S^T = K Q^T): lane holds Q[query fr][d = 32 ks + 8 fq ..]
    half8 qf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int qr = q0 + qt * 16 + fr;
        qr = qr < p.seq_rows ? qr : p.seq_rows - 1;
        const half_t* src = p.q + (row_base + qr) * p.ld_qk + h * 64 + fq * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = *reinterpret_cast<const half8*>(src + ks * 32);
    }

    float4v acc_o[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc_o[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-1e30f, -1e30f};
    float l_run[2] = {0.f, 0.f};

    // staging: 512 chunks per 64x128B tile -> 2 per thread, for K and for Vt
    const int sc = tid & 7, sr = tid >> 3;      // rows sr, sr + 32
    uint4 rk[2], rv[2];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = kt * KT + sr + 32 * i;
            rk[i] = key < p.seq_rows
                        ? *reinterpret_cast<const uint4*>(p.k + (row_base + key) * p.ld_qk + h * 64 + sc * 8)
                        : make_uint4(0, 0, 0, 0);
            const int d = sr + 32 * i;
            rv[i] = *reinterpret_cast<const uint4*>(p.vt + (long)seq * p.vt_seq_stride + (long)(h * 64 + d) * p.vt_ld +
                                                    kt * KT + sc * 8);
        }
    };
    auto store_tile = [&](int buf) {
        char* kb = smem + buf * 2 * KT * ROWB;
        char* vb = kb + KT * ROWB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<uint4*>(kb + lds_off(sr + 32 * i, sc)) = rk[i];
            *reinterpret_cast<uint4*>(vb + lds_off(sr + 32 * i, sc)) = rv[i];
        }
    };

    if (n_kt > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();

    for (int kt = 0; kt < n_kt; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < n_kt;
        if (more) load_tile(kt + 1);
        const char* kb = smem + buf * 2 * KT * ROWB;
        const char* vb = kb + KT * ROWB;

        // ---- S^T = K Q^T : acc_s[mt][qt][r] = S[key 16 mt + 4 fq + r][query 16 qt + fr]
        float4v acc_s[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc_s[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const half8 kf = *reinterpret_cast<const half8*>(kb + lds_off(mt * 16 + fr, ks * 4 + fq));
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    acc_s[mt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qt][ks], acc_s[mt][qt], 0, 0, 0);
            }
        }
        // ---- key-padding mask (only tiles that cross kv_len)
        if ((kt + 1) * KT > kv_len) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * KT + mt * 16 + fq * 4 + r;
                    if (key >= kv_len) {
                        acc_s[mt][0][r] = -1e30f;
                        acc_s[mt][1][r] = -1e30f;
                    }
                }
        }
        // ---- online softmax (base-2; q carries log2(e)/8)
        float alpha[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mx = -1e30f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc_s[mt][qt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[qt], mx);
            alpha[qt] = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
            m_run[qt] = m_new;
            float ls = 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(acc_s[mt][qt][r] - m_new);
                    acc_s[mt][qt][r] = e;
                    ls += e;
                }
            l_run[qt] = l_run[qt] * alpha[qt] + ls;       // lane-partial sum (reduced over fq at the end)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc_o[dt][qt] *= alpha[qt];
        }
        // ---- O^T += V^T P^T : k-slot (fq, e) <-> key 32 ks + 4 fq + e (e<4) | 32 ks + 16 + 4 fq + e-4
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 pf[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pf[qt][e] = (half_t)acc_s[2 * ks][qt][e];
                    pf[qt][4 + e] = (half_t)acc_s[2 * ks + 1][qt][e];
                }
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int d = dt * 16 + fr;
                const int base = lds_off(d, ks * 4 + (fq >> 1)) + (fq & 1) * 8;      // keys 32 ks + 4 fq
                const int base2 = lds_off(d, ks * 4 + 2 + (fq >> 1)) + (fq & 1) * 8; // keys 32 ks + 16 + 4 fq
                const half4 v0 = *reinterpret_cast<const half4*>(vb + base);
                const half4 v1 = *reinterpret_cast<const half4*>(vb + base2);
                half8 vf;
#pragma unroll
                for (int e = 0; e < 4; ++e) { vf[e] = v0[e]; vf[4 + e] = v1[e]; }
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    acc_o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qt], acc_o[dt][qt], 0, 0, 0);
            }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- finalize: O[query][d] = acc_o / l ; lane holds d = 16 dt + 4 fq + r for query 16 qt + fr
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float l = l_run[qt];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        const int qr = q0 + qt * 16 + fr;
        if (qr < p.Tq) {
            half_t* dst = p.out + (row_base + qr) * p.ld_out + h * 64 + fq * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                half4 o = {(half_t)(acc_o[dt][qt][0] * inv), (half_t)(acc_o[dt][qt][1] * inv),
                           (half_t)(acc_o[dt][qt][2] * inv), (half_t)(acc_o[dt][qt][3] * inv)};
                *reinterpret_cast<half4*>(dst + dt * 16) = o;
            }
        }
    }
}

}  // namespace

int attention_launch(const AttnParams& p, hipStream_t st) {
    SVC_REQUIRE(p.n_seq > 0 && p.H > 0 && p.Tq > 0, "attention shape");
    SVC_REQUIRE(p.vt_ld % 64 == 0 && p.ld_qk % 8 == 0 && p.ld_out % 4 == 0, "attention alignment");
    dim3 grid(cdiv(p.Tq, 128) * p.H * p.n_seq);
    const bool prof = prof_enabled();
    if (prof) prof_begin(PROF_ATTN, st);
    hipLaunchKernelGGL(attn_kernel, grid, dim3(256), 0, st, p);
    SVC_CHECK_HIP(hipGetLastError());
    if (prof) {
        // QK^T + PV = 4 * Tq * Tk * 64 flop per (seq, head); q,k,v read once, o written once (fp16)
        const double tk = p.kv_len ? p.seq_rows : p.kv_len_const;
        prof_end(PROF_ATTN, 4.0 * p.n_seq * p.H * (double)p.Tq * tk * 64.0, 2.0 * p.n_seq * p.H * 64.0 * (2.0 * p.Tq + 2.0 * tk), st);
    }
    return 0;
}

}  // namespace svc
