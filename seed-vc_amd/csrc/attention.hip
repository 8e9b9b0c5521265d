// Flash-style self-attention for the DiT blocks (head_dim 64, non-causal, key-padding by length).
//
// Layout contract (produced by the QKV tap-GEMM epilogue): q, k fp16 [rows][ld] with head h at
// column h*64, already rotated (RoPE) and q pre-scaled by log2(e)/sqrt(64); v transposed
// vt[seq][h*64 + d][key] so that both MFMA contractions read K-contiguous operands -- with the keys of every 32-group
// in vt_perm_pos() order (common.h) when AttnParams::vt_perm is set, which makes a V^T fragment one ds_read_b128.
//
// Block = 4 waves = 64 QT queries of one (sequence, head); each wave owns QT 16-wide MFMA column tiles of queries
// (QT = 2 for full grids; QT = 1 doubles the workgroup count of small launches -- a single utterance gives only
// 2 x H x 7 blocks of 128 queries for 256 CUs and is bound by the serial chain of one workgroup over the key tiles).
// Every per-query quantity, including the deferred-rescale decision, depends only on the query's own 16-wide tile,
// so both forms are bit-identical.  Scores are computed transposed, S^T = K Q^T (keys on the accumulator rows),
// which makes the softmax row statistics lane-local up to two cross-lane maxima and lets the
// accumulator registers feed P^T straight into O^T += V^T P^T as the B operand (no LDS round trip).
// K and V^T tiles (64 keys) stream through a 3-stage LDS-DMA ring (counted vmcnt + one raw s_barrier per tile, as in
// the tap-GEMM) with the same XOR-swizzled 128-byte-row image (conflict-free ds_read_b128 / ds_read_b64).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace svc {
namespace {

constexpr int KT = 64;        // keys per tile
constexpr int ROWB = 128;

__device__ __forceinline__ int lds_off(int row, int c16) { return row * ROWB + ((c16 ^ ((row >> 1) & 7)) << 4); }

template <int QT, bool VPERM>
__global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) {
    constexpr int BQ = 64 * QT;   // queries per block
    constexpr int NS = 3;         // ring stages: tiles kt + 1, kt + 2 in flight under the MFMAs of tile kt
    __shared__ __attribute__((aligned(16))) char smem[NS * 2 * KT * ROWB];   // [stage][K | Vt][64][128B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    // 1-D grid with an XCD-aware order: the query tiles of one (sequence, head) -- which all stream the same K / V^T --
    // are given to one XCD (bid % 8 labels the XCD group), so K/V are fetched into one L2 instead of up to 8.
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int nqt = (p.Tq - p.q_start + BQ - 1) / BQ;
    const int qt_idx = lid % nqt;
    const int sh_idx = lid / nqt;
    const int h = sh_idx % p.H, seq = sh_idx / p.H;
    const int q0 = p.q_start + qt_idx * BQ + wave * 16 * QT;
    const long row_base = (long)seq * p.seq_rows;
    const int kv_len = p.kv_len ? p.kv_len[seq] : p.kv_len_const;
    const int n_kt = (kv_len + KT - 1) / KT;

    // Q fragments (B operand of S^T = K Q^T): lane holds Q[query fr][d = 32 ks + 8 fq ..]
    half8 qf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int qr = q0 + qt * 16 + fr;
        qr = qr < p.seq_rows ? qr : p.seq_rows - 1;
        const half_t* src = p.q + (row_base + qr) * p.ld_qk + h * 64 + fq * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = *reinterpret_cast<const half8*>(src + ks * 32);
    }
    // "ones" A operand: row 0 of a 16-row tile is all ones -> one extra MFMA per P fragment yields the softmax row
    // sums in the accumulator (same fp16-rounded P as the PV product, no VALU adds)
    half8 ones_f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones_f[e] = fr == 0 ? (half_t)1.0f : (half_t)0.0f;

    float4v acc_o[4][QT], acc_l[QT];
#pragma unroll
    for (int j = 0; j < QT; ++j) {
        acc_l[j] = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) acc_o[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};
    }
    // running baseline (log2 units; q carries log2(e)/8).  It is subtracted inside the QK^T MFMA (accumulator
    // initialised to -m_run) and only moved when a tile's maximum exceeds it by more than THR (deferred rescale:
    // P <= 2^THR stays well inside fp16), so the common tile needs neither the subtraction nor the O rescale.
    constexpr float THR = 8.0f;
    float m_run[QT];
#pragma unroll
    for (int j = 0; j < QT; ++j) m_run[j] = 0.f;

    // staging: LDS-DMA (global_load_lds_dwordx4) into an NS-stage ring, no staging registers and no ds_write.  A wave
    // instruction writes 1 KiB = 8 tile rows linearly (LDS address = wave-uniform base + lane * 16), so lane l fetches,
    // for tile row r = 8 g + (l >> 3), the logical chunk (l & 7) ^ swz(r) (swizzle on the source side).  Each wave owns 2
    // of the 8 row groups of the K tile and 2 of the V^T tile: 4 DMA instructions per wave per key tile.  K rows past the
    // sequence are clamped to its last row (those keys are masked below); V^T columns are padded to a multiple of 64.
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int srow[2];
    const half_t* kcol[2];
    const half_t* vp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        srow[j] = (wave * 2 + j) * 8 + (lane >> 3);
        const int cs = (lane & 7) ^ ((srow[j] >> 1) & 7);
        kcol[j] = p.k + row_base * p.ld_qk + h * 64 + cs * 8;
        vp[j] = p.vt + (long)seq * p.vt_seq_stride + (long)(h * 64 + srow[j]) * p.vt_ld + cs * 8;
    }
    auto issue_tile = [&](int kt) {
        char* kb = smem + (kt % NS) * 2 * KT * ROWB + wave_u * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int kr = kt * KT + srow[j];
            kr = kr < p.seq_rows ? kr : p.seq_rows - 1;
            __builtin_amdgcn_global_load_lds((gptr_t)(kcol[j] + (long)kr * p.ld_qk), (lptr_t)(kb + j * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(vp[j] + kt * KT), (lptr_t)(kb + KT * ROWB + j * 1024), 16, 0, 0);
        }
    };
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_)
        if (s_ < n_kt) issue_tile(s_);

    // One key tile.  MASK (compile time) = the tile crosses kv_len: only the last tile can, so the loop below runs the
    // unmasked body and the masked one is a separate copy (left as a runtime test inside ONE body, hipcc if-converts the
    // test into 48 compares / selects that every tile executes).
    auto tile = [&](const int kt, auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        // tile kt has landed (this wave's DMAs: counted wait, the next tiles' may stay in flight); the barrier makes every
        // wave's part visible and guarantees that the stage refilled next (read during iteration kt - 1) is free
        wait_tiles<NS - 2, 4>(n_kt - 1 - kt);
        asm volatile("s_barrier" ::: "memory");
        if (kt + NS - 1 < n_kt) issue_tile(kt + NS - 1);
        const char* kb = smem + (kt % NS) * 2 * KT * ROWB;
        const char* vb = kb + KT * ROWB;

        // ---- S^T - m = K Q^T - m : acc_s[mt][qt][r] = S[key 16 mt + 4 fq + r][query 16 qt + fr] - m_run[qt]
        float4v acc_s[4][QT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < QT; ++j) acc_s[i][j] = (float4v){-m_run[j], -m_run[j], -m_run[j], -m_run[j]};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const half8 kf = *reinterpret_cast<const half8*>(kb + lds_off(mt * 16 + fr, ks * 4 + fq));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    acc_s[mt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qt][ks], acc_s[mt][qt], 0, 0, 0);
            }
        }
        // ---- key-padding mask
        if constexpr (MASK) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * KT + mt * 16 + fq * 4 + r;
                    if (key >= kv_len) {
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) acc_s[mt][qt][r] = -1e30f;
                    }
                }
        }
        // ---- baseline move (always on the first tile; afterwards only when some score of the 16-query tile overshoots the
        // baseline by more than THR -- decided per tile, so a query's arithmetic does not depend on which tiles share
        // its wave).  The decision only needs each lane's own maximum; the row maxima (two cross-lane exchanges) are
        // taken on the rare path that moves the baseline.
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float a = -1e30f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                a = fmaxf(a, fmaxf(fmaxf(acc_s[mt][qt][0], acc_s[mt][qt][1]), fmaxf(acc_s[mt][qt][2], acc_s[mt][qt][3])));
            if (__builtin_expect(kt == 0 || __any(a > THR), 0)) {
                a = fmaxf(a, __shfl_xor(a, 16));
                a = fmaxf(a, __shfl_xor(a, 32));
                const float delta = kt == 0 ? a : fmaxf(a, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_run[qt] += delta;
                acc_l[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) acc_o[dt][qt] *= alpha;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc_s[mt][qt] -= delta;
            }
        }
        // ---- P = 2^(S - m), packed to fp16 (round toward zero; the row sums below use the same rounded values)
        // ---- O^T += V^T P^T, l += 1^T P^T : k-slot (fq, e) <-> key 32 ks + 4 fq + e (e<4) | 32 ks + 16 + 4 fq + e-4
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 pf[QT];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                u32x4 u;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const float4v sv = acc_s[2 * ks + hh][qt];
                    const float e0 = __builtin_amdgcn_exp2f(sv[0]), e1 = __builtin_amdgcn_exp2f(sv[1]);
                    const float e2 = __builtin_amdgcn_exp2f(sv[2]), e3 = __builtin_amdgcn_exp2f(sv[3]);
                    u[2 * hh] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(e0, e1));
                    u[2 * hh + 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(e2, e3));
                }
                pf[qt] = __builtin_bit_cast(half8, u);
                acc_l[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones_f, pf[qt], acc_l[qt], 0, 0, 0);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int d = dt * 16 + fr;
                half8 vf;
                if constexpr (VPERM) {
                    // vt columns in vt_perm_pos() order: this lane's 8 keys are one 16-byte chunk
                    vf = *reinterpret_cast<const half8*>(vb + lds_off(d, ks * 4 + fq));
                } else {
                    const int base = lds_off(d, ks * 4 + (fq >> 1)) + (fq & 1) * 8;      // keys 32 ks + 4 fq
                    const int base2 = lds_off(d, ks * 4 + 2 + (fq >> 1)) + (fq & 1) * 8; // keys 32 ks + 16 + 4 fq
                    const half4 v0 = *reinterpret_cast<const half4*>(vb + base);
                    const half4 v1 = *reinterpret_cast<const half4*>(vb + base2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { vf[e] = v0[e]; vf[4 + e] = v1[e]; }
                }
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    acc_o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qt], acc_o[dt][qt], 0, 0, 0);
            }
        }
    };
    const bool ragged = (kv_len % KT) != 0;                 // the last tile holds keys past kv_len
    const int n_full = ragged ? n_kt - 1 : n_kt;
    for (int kt = 0; kt < n_full; ++kt) tile(kt, std::false_type{});
    if (ragged) tile(n_kt - 1, std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- finalize: O[query][d] = acc_o / l ; lane holds d = 16 dt + 4 fq + r for query 16 qt + fr;
    // the row sum of query fr sits in register 0 of lane fr (accumulator row 0 <-> fq = 0)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float l = __shfl(acc_l[qt][0], fr);
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
    SVC_REQUIRE(p.n_seq > 0 && p.H > 0 && p.Tq > 0 && p.q_start >= 0 && p.q_start < p.Tq, "attention shape");
    SVC_REQUIRE(p.vt_ld % 64 == 0 && p.ld_qk % 8 == 0 && p.ld_out % 4 == 0, "attention alignment");
    const int g128 = cdiv(p.Tq - p.q_start, 128) * p.H * p.n_seq;
    const bool prof = prof_enabled();
    if (prof) prof_begin(PROF_ATTN, st);
    static const int qt_env = [] { const char* e = getenv("SVC_ATTN_QT"); return e ? atoi(e) : 0; }();
    const dim3 g64(cdiv(p.Tq - p.q_start, 64) * p.H * p.n_seq);
    if (qt_env == 1 || (qt_env == 0 && g128 <= 256)) {
        if (p.vt_perm) hipLaunchKernelGGL((attn_kernel<1, true>), g64, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((attn_kernel<1, false>), g64, dim3(256), 0, st, p);
    } else {
        if (p.vt_perm) hipLaunchKernelGGL((attn_kernel<2, true>), dim3(g128), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((attn_kernel<2, false>), dim3(g128), dim3(256), 0, st, p);
    }
    SVC_CHECK_HIP(hipGetLastError());
    if (prof) {
        // QK^T + PV = 4 * Tq * Tk * 64 flop per (seq, head); q,k,v read once, o written once (fp16)
        const double tk = p.kv_len ? p.seq_rows : p.kv_len_const;
        prof_end(PROF_ATTN, 4.0 * p.n_seq * p.H * (double)(p.Tq - p.q_start) * tk * 64.0,
                 2.0 * p.n_seq * p.H * 64.0 * (2.0 * (p.Tq - p.q_start) + 2.0 * tk), st);
    }
    return 0;
}

}  // namespace svc
