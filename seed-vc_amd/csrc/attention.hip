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

// ---------------------------------------------------------------------------------------------------------------------
// The same attention on v_mfma_f32_32x32x16_f16: a wave owns ONE 32-query tile (block = 4 waves = 128 queries, as QT = 2).
// MEASURED SLOWER than the 16x16x32 kernel above and therefore NOT the default (SVC_ATTN32=1 selects it; the parity tests
// run both): in-model on the tiny B = 64 launch 700-736 TFLOP/s across its variants (ones-row MFMA or v_dot2 row sums,
// K fragments prefetched or not, 2- or 3-stage ring = 4 or 3 blocks per CU, refill DMAs in one burst or spread behind the
// S^T MFMAs) against 764 for attn_kernel<2> on the same box (profiles/r03 README).  The reasoning below holds for the
// issue slots, but the two independent 16-query tiles of attn_kernel<2> give one wave two dependency chains to overlap
// (S^T of one tile under the exp2 / pack of the other), which a single 32-query tile does not have.
// Why: an MFMA holds the SIMD's vector issue for 8 cycles whatever its shape (MI355X_MICROARCH.md, 'vector-instruction
// ISSUE cost'), i.e. 8 of the 16 cycles of a 16x16x32 but 8 of the 32 of a 32x32x16.  Per 64-key tile the 16x16 kernel
// issues 36 MFMAs (288 issue cycles) + ~540 cycles of exp / max / pack / LDS reads against 576 cycles of matrix pipe:
// issue-bound (profiles/r02_c_attention_ablation.txt: MFMA-only 69 us + VALU-only 66 us = 104 us, no overlap).  Here it is
// 20 MFMAs (160 issue cycles, the ones-row sums included) and ~520 of exp2 / max / pack / LDS reads against 640 cycles of
// matrix pipe.
// Fragment layouts (lane = (fr = lane % 32, fh = lane / 32)): A [32 x 16]: row fr, k = 8 fh .. 8 fh + 7;  B [16 x 32]: column
// fr, same k;  C [32 x 32]: column fr, register i <-> row 8 (i / 4) + 4 fh + (i % 4).
//   S^T block (32 keys x 32 queries) = K[32 keys][64 d] Q^T: A = K rows from LDS (one ds_read_b128: row, 16-byte chunk
//   2 kk + fh), B = Q in registers, 4 MFMAs over d.  The accumulators leave lane half fh with keys 8 a + 4 fh + r of every
//   16-key group, which IS the B operand of O^T += V^T P^T (16 keys per MFMA) once V^T's columns are stored in that order
//   (vt_perm_pos16, written by the QKV epilogues): a V^T fragment is then one ds_read_b128 as well.
// K / V^T staging, ring, masks and the deferred-rescale rule are those of attn_kernel; a query's arithmetic depends only
// on its own keys (the baseline moves per query; the per-tile test only decides WHETHER the move code runs), so results do
// not depend on which queries share a wave.  Not bit-identical to the 16x16 forms (another accumulation order in the MFMA).
typedef float float16v __attribute__((ext_vector_type(16)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// NW = waves per block: 4 (128 queries) for full grids, 2 (64 queries) for small ones (a single utterance gives only
// 2 x H x 7 blocks of 128 queries for 256 CUs); a wave's arithmetic is the same in both, so the forms are bit-identical.
template <int NW, int NS = 3, bool KPF = true, bool ONES = true>
__global__ __launch_bounds__(64 * NW) void attn32_kernel(const AttnParams p) {
    constexpr int BQ = 32 * NW;
    constexpr int RG = 8 / NW;    // 8-row groups of the K tile (and of the V^T tile) staged per wave
    __shared__ __attribute__((aligned(16))) char smem[NS * 2 * KT * ROWB];   // [stage][K | Vt][64][128B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int nqt = (p.Tq - p.q_start + BQ - 1) / BQ;
    const int qt_idx = lid % nqt;
    const int sh_idx = lid / nqt;
    const int h = sh_idx % p.H, seq = sh_idx / p.H;
    const int q0 = p.q_start + qt_idx * BQ + wave * 32;
    const long row_base = (long)seq * p.seq_rows;
    const int kv_len = p.kv_len ? p.kv_len[seq] : p.kv_len_const;
    const int n_kt = (kv_len + KT - 1) / KT;

    // Q fragments (B operand): lane holds Q[query fr][d = 16 kk + 8 fh ..]
    half8 qf[4];
    {
        int qr = q0 + fr;
        qr = qr < p.seq_rows ? qr : p.seq_rows - 1;
        const half_t* src = p.q + (row_base + qr) * p.ld_qk + h * 64 + fh * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const half8*>(src + kk * 16);
    }
    float16v acc_o[2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_o[hb][i] = 0.f;
    // row sums: one more MFMA per P fragment with an A operand whose row 0 is all ones (same fp16-rounded P as the PV
    // product).  The kernel is bound by vector ISSUE cycles (exp2 alone is 256 of ~700 per key tile), not by the matrix
    // pipe: an MFMA costs 8 issue cycles, the 4 v_dot2c it replaces ~40 and a serial dependency chain.
    half8 ones_f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones_f[e] = fr == 0 ? (half_t)1.0f : (half_t)0.0f;
    float16v acc_l;               // register 0 of lane (fr, fh = 0) = row sum of query fr
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_l[i] = 0.f;
    float dl[4] = {0.f, 0.f, 0.f, 0.f};     // !ONES: v_dot2 row sums, four independent chains
    constexpr float THR = 8.0f;
    float m_run = 0.f;

    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int srow[RG];
    const half_t* kcol[RG];
    const half_t* vp[RG];
#pragma unroll
    for (int j = 0; j < RG; ++j) {
        srow[j] = (wave * RG + j) * 8 + (lane >> 3);
        const int cs = (lane & 7) ^ ((srow[j] >> 1) & 7);
        kcol[j] = p.k + row_base * p.ld_qk + h * 64 + cs * 8;
        vp[j] = p.vt + (long)seq * p.vt_seq_stride + (long)(h * 64 + srow[j]) * p.vt_ld + cs * 8;
    }
    // one 1-KiB piece of a tile copy: piece 2 j = K row group j, 2 j + 1 = V^T row group j of this wave
    auto issue_piece = [&](int kt, int pc) {
        char* kb = smem + (kt % NS) * 2 * KT * ROWB + wave_u * (RG * 1024);
        const int j = pc >> 1;
        if (pc & 1) {
            __builtin_amdgcn_global_load_lds((gptr_t)(vp[j] + kt * KT), (lptr_t)(kb + KT * ROWB + j * 1024), 16, 0, 0);
        } else {
            int kr = kt * KT + srow[j];
            kr = kr < p.seq_rows ? kr : p.seq_rows - 1;
            __builtin_amdgcn_global_load_lds((gptr_t)(kcol[j] + (long)kr * p.ld_qk), (lptr_t)(kb + j * 1024), 16, 0, 0);
        }
    };
    auto issue_tile = [&](int kt) {
#pragma unroll
        for (int pc = 0; pc < 2 * RG; ++pc) issue_piece(kt, pc);
    };
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_)
        if (s_ < n_kt) issue_tile(s_);

    auto tile = [&](const int kt, auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        wait_tiles<NS - 2, 2 * RG>(n_kt - 1 - kt);
        asm volatile("s_barrier" ::: "memory");
        // The refill of the stage read during the previous tile (tile kt + 2) is NOT issued here in one burst: an LDS-DMA
        // instruction holds the wave for ~60-180 issue cycles (MI355X_MICROARCH.md, 'LDS-DMA piece issue cost'), so the
        // pieces go out one by one behind the S^T MFMAs, whose 32-cycle shadows they fill.
        const bool refill = kt + NS - 1 < n_kt;
        const char* kb = smem + (kt % NS) * 2 * KT * ROWB;
        const char* vb = kb + KT * ROWB;

        // ---- S^T - m : acc_s[kb32][i] = S[key 32 kb32 + 8 (i / 4) + 4 fh + (i % 4)][query fr] - m_run
        float16v acc_s[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_s[b][i] = -m_run;
        half8 kf[4][2];
        if constexpr (KPF) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int b = 0; b < 2; ++b) kf[kk][b] = *reinterpret_cast<const half8*>(kb + lds_off(b * 32 + fr, kk * 2 + fh));
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if constexpr (!KPF) {
#pragma unroll
                for (int b = 0; b < 2; ++b) kf[kk][b] = *reinterpret_cast<const half8*>(kb + lds_off(b * 32 + fr, kk * 2 + fh));
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) acc_s[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kk][b], qf[kk], acc_s[b], 0, 0, 0);
            if (refill) {
#pragma unroll
                for (int pc = kk * (2 * RG / 4); pc < (kk + 1) * (2 * RG / 4); ++pc) issue_piece(kt + NS - 1, pc);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MASK) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kt * KT + b * 32 + 8 * (i >> 2) + 4 * fh + (i & 3);
                    if (key >= kv_len) acc_s[b][i] = -1e30f;
                }
        }
        // ---- baseline move: always on the first tile, afterwards only when some score overshoots the baseline by more than THR
        {
            float a = -1e30f;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; i += 4)
                    a = fmaxf(a, fmaxf(fmaxf(acc_s[b][i], acc_s[b][i + 1]), fmaxf(acc_s[b][i + 2], acc_s[b][i + 3])));
            if (__builtin_expect(kt == 0 || __any(a > THR), 0)) {
                a = fmaxf(a, __shfl_xor(a, 32));             // the query's other 32 keys of this tile
                const float delta = kt == 0 ? a : fmaxf(a, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_run += delta;
                acc_l[0] *= alpha;
#pragma unroll
                for (int i = 0; i < 4; ++i) dl[i] *= alpha;
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) acc_o[hb] *= alpha;
#pragma unroll
                for (int b = 0; b < 2; ++b) acc_s[b] -= delta;
            }
        }
        // ---- P = 2^(S - m) packed to fp16 (round toward zero), row sums of the SAME rounded values by the ones-row MFMA,
        // ---- O^T += V^T P^T: PV step mm covers keys [16 mm, 16 mm + 16); its k slot (fh, j) is accumulator 8 (mm & 1) + j
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            u32x4 u;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float e0 = __builtin_amdgcn_exp2f(acc_s[mm >> 1][8 * (mm & 1) + 2 * jj]);
                const float e1 = __builtin_amdgcn_exp2f(acc_s[mm >> 1][8 * (mm & 1) + 2 * jj + 1]);
                const half2v pk = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(e0, e1));
                if constexpr (!ONES) dl[jj] = __builtin_amdgcn_fdot2(pk, (half2v){(_Float16)1.0f, (_Float16)1.0f}, dl[jj], false);
                u[jj] = __builtin_bit_cast(unsigned, pk);
            }
            const half8 pf = __builtin_bit_cast(half8, u);
            if constexpr (ONES) acc_l = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones_f, pf, acc_l, 0, 0, 0);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const half8 vf = *reinterpret_cast<const half8*>(vb + lds_off(hb * 32 + fr, mm * 2 + fh));
                acc_o[hb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, acc_o[hb], 0, 0, 0);
            }
        }
    };
    const bool ragged = (kv_len % KT) != 0;
    const int n_full = ragged ? n_kt - 1 : n_kt;
    for (int kt = 0; kt < n_full; ++kt) tile(kt, std::false_type{});
    if (ragged) tile(n_kt - 1, std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- finalize: lane (fr, fh) holds O[query fr][d = 32 hb + 8 (i / 4) + 4 fh + (i % 4)]
    float l;
    if constexpr (ONES) {
        l = __shfl(acc_l[0], fr);
    } else {
        const float t = (dl[0] + dl[1]) + (dl[2] + dl[3]);
        l = t + __shfl_xor(t, 32);
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    const int qr = q0 + fr;
    if (qr < p.Tq) {
        half_t* dst = p.out + (row_base + qr) * p.ld_out + h * 64 + fh * 4;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4 o = {(half_t)(acc_o[hb][4 * g] * inv), (half_t)(acc_o[hb][4 * g + 1] * inv),
                           (half_t)(acc_o[hb][4 * g + 2] * inv), (half_t)(acc_o[hb][4 * g + 3] * inv)};
                *reinterpret_cast<half4*>(dst + hb * 32 + g * 8) = o;
            }
    }
}

// natural-order V^T rows -> column order `mode`, in place: one thread per (row, 32-column group)
__global__ void vt_permute_kernel(half_t* __restrict__ vt, long rows, long vt_ld, int mode) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long groups = vt_ld / 32;
    if (t >= rows * groups) return;
    half_t* base = vt + (t / groups) * vt_ld + (t % groups) * 32;
    half_t v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = base[i];
#pragma unroll
    for (int i = 0; i < 32; ++i) base[vt_pos(i, mode)] = v[i];
}

}  // namespace

int attention_vt_mode(int n_seq, int H, int Tq) {
    (void)n_seq; (void)H; (void)Tq;                 // one kernel family for every grid size: results do not depend on the batch
    const char* e = getenv("SVC_ATTN32");
    return e && e[0] == '1' ? 2 : 1;
}

int attention_permute_vt(half_t* vt, long rows, long vt_ld, int mode, hipStream_t st) {
    SVC_REQUIRE(vt_ld % 32 == 0, "attention_permute_vt: vt_ld must be a multiple of 32");
    const long n = rows * (vt_ld / 32);
    hipLaunchKernelGGL(vt_permute_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, vt, rows, vt_ld, mode);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

int attention_launch(const AttnParams& p, hipStream_t st) {
    SVC_REQUIRE(p.n_seq > 0 && p.H > 0 && p.Tq > 0 && p.q_start >= 0 && p.q_start < p.Tq, "attention shape");
    SVC_REQUIRE(p.vt_ld % 64 == 0 && p.ld_qk % 8 == 0 && p.ld_out % 4 == 0, "attention alignment");
    const int g128 = cdiv(p.Tq - p.q_start, 128) * p.H * p.n_seq;
    const bool prof = prof_enabled();
    if (prof) prof_begin(PROF_ATTN, st);
    static const int qt_env = [] { const char* e = getenv("SVC_ATTN_QT"); return e ? atoi(e) : 0; }();
    const dim3 g64(cdiv(p.Tq - p.q_start, 64) * p.H * p.n_seq);
    if (p.vt_perm == 2) {
        if (g128 <= 256) hipLaunchKernelGGL((attn32_kernel<2>), g64, dim3(128), 0, st, p);
        else hipLaunchKernelGGL((attn32_kernel<4>), dim3(g128), dim3(256), 0, st, p);
    } else if (qt_env == 1 || (qt_env == 0 && g128 <= 256)) {
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
