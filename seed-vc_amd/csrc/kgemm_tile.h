// Tile kernel of the tap-GEMM (see kgemm.hip for the scheme).  A header because two translation units instantiate it:
// kgemm.hip (accumulators in VGPRs, -amdgpu-mfma-vgpr-form) and kgemm_big.hip (256 x 256 tiles whose 256 accumulator
// registers per lane live in AGPRs).
#pragma once
#include <stdlib.h>

#include "common.h"

namespace svc {

namespace {

// Tile geometry.
template <int BM, int BN, int RB, int NS, int NWV>
struct Geo {
    static constexpr int NW = NWV;                           // waves (BM / 32: 64x64 | 32x{64,32} wave tiles; BM / 64: 128x64)
    static constexpr int NT = NW * 64;                       // threads
    static constexpr int WAVES_N = (BN >= 128) ? 2 : 1;      // 256 x 256 with 4 waves: 2 x 2 wave tiles of 128 x 128
    static constexpr int WAVES_M = NW / WAVES_N;
    static constexpr int WTM = BM / WAVES_M;
    static constexpr int WTN = BN / WAVES_N;
    static constexpr int TM = WTM / 16;
    static constexpr int TN = WTN / 16;
    static constexpr int EPI_LD = WTN + 4;
    static constexpr int NSTAGE = NS;
    static constexpr int STAGE_BYTES = (BM + BN) * RB;
    static constexpr int LDS_AB = NSTAGE * STAGE_BYTES;
    // STORE epilogue (fp32 residual stream): accumulator tile transposed through LDS in EP row passes so that the
    // transposition region never exceeds the ring
    static constexpr int EPI_FULL = NW * WTM * EPI_LD * 4;
    static constexpr int EP = (EPI_FULL <= LDS_AB || TM == 1) ? 1 : ((EPI_FULL / 2 <= LDS_AB || TM == 2) ? 2 : 4);
    static constexpr int LDS_BYTES = LDS_AB;
    static constexpr int CPRW = RB / 16;                     // 16-byte chunks per tile row
    static constexpr int RPP = NT / CPRW;                    // tile rows covered by one staging pass of the block
    static constexpr int WROWS = 1024 / RB;                  // tile rows written by one wave-level DMA instruction
    static constexpr int A_ITERS = BM / RPP;
    static constexpr int B_ITERS = BN / RPP;
    static constexpr int DPT = A_ITERS + B_ITERS;            // LDS-DMA instructions per wave per tile
    static constexpr int KS = RB / 64;                       // MFMA k-steps (64 bytes of K each) per tile
};

// chunk swizzle of the LDS image: physical 16-byte slot = chunk ^ swz(row).  Conflict-free (for the lane groups a
// ds_read_b128 is serviced in) both for 16 consecutive rows (activation fragments) and for the permuted weight rows
// 4 TN g + 4 nt + (fr & 3), g = fr >> 2: the second term folds the two row bits that hold g (bits 4..5 for TN <= 4,
// bits 5..6 for the 128-column wave tiles, TN = 8) into the key.
template <int RB, int TN = 4>
__device__ __forceinline__ int swz_of(int row) {
    constexpr int GS = TN == 8 ? 5 : 4;
    if constexpr (RB == 128) return ((row >> 1) ^ (((row >> GS) & 3) << 1)) & 7;    // 2 rows per 256-byte bank line
    else if constexpr (TN == 8) return ((row >> 1) ^ (row >> 4)) & 3;               // (the pipelined loop's fragment bases rely on this form)
    else return (0 - (((row >> 2) ^ (row >> GS)) & 3)) & 3;                         // 4 rows per bank line
}

// ---- accumulators of the 128 x 128 wave tiles: 64 MFMA tiles = all 256 AGPRs of the lane, addressed by NAME from inline
// asm.  hipcc cannot be trusted with them: with every AGPR live across a loop body of several blocks its allocator copies
// and spills accumulator tuples around the MFMAs (measured in the ISA: 256 v_accvgpr moves + scratch traffic per tile).
// The compiler never sees an AGPR in these kernels (kgemm_big.hip is built with -amdgpu-spill-vgpr-to-agpr=0).  Indices
// are compile-time after unrolling, so each switch folds to its one case.
#define SVC_KG_ROW8(M, a, b, c, d, e, f, g, h) M(a) M(b) M(c) M(d) M(e) M(f) M(g) M(h)
#define SVC_KG_ALL64(M)                                                                                                \
    SVC_KG_ROW8(M, 0, 1, 2, 3, 4, 5, 6, 7) SVC_KG_ROW8(M, 8, 9, 10, 11, 12, 13, 14, 15)                                 \
    SVC_KG_ROW8(M, 16, 17, 18, 19, 20, 21, 22, 23) SVC_KG_ROW8(M, 24, 25, 26, 27, 28, 29, 30, 31)                       \
    SVC_KG_ROW8(M, 32, 33, 34, 35, 36, 37, 38, 39) SVC_KG_ROW8(M, 40, 41, 42, 43, 44, 45, 46, 47)                       \
    SVC_KG_ROW8(M, 48, 49, 50, 51, 52, 53, 54, 55) SVC_KG_ROW8(M, 56, 57, 58, 59, 60, 61, 62, 63)
#define SVC_KG_MFMA_CASE(i)                                                                                            \
    case i: asm volatile("v_mfma_f32_16x16x32_f16 a[4*" #i ":4*" #i "+3], %0, %1, a[4*" #i ":4*" #i "+3]" ::"v"(x), "v"(y)); break;
#define SVC_KG_READ_CASE(i)                                                                                            \
    case i: asm volatile("v_accvgpr_read_b32 %0, a[4*" #i "]\n\tv_accvgpr_read_b32 %1, a[4*" #i "+1]\n\t"              \
                         "v_accvgpr_read_b32 %2, a[4*" #i "+2]\n\tv_accvgpr_read_b32 %3, a[4*" #i "+3]"                 \
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
__device__ __forceinline__ void agpr_mfma_f16(int idx, const u32x4 x, const u32x4 y) {      // a[4 idx ..] += x * y
    switch (idx) { SVC_KG_ALL64(SVC_KG_MFMA_CASE) }
}
__device__ __forceinline__ float4v agpr_tile(int idx) {
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
    switch (idx) { SVC_KG_ALL64(SVC_KG_READ_CASE) }
    return (float4v){r0, r1, r2, r3};
}
__device__ __forceinline__ void agpr_zero_all() {
    asm volatile(".set svc_kg_i, 0\n.rept 256\n\tv_accvgpr_write_b32 a[svc_kg_i], 0\n.set svc_kg_i, svc_kg_i+1\n.endr" ::: "a255");
}
template <bool IN_AGPR>
__device__ __forceinline__ float4v acc_tile(const float4v& v, int idx) {
    if constexpr (IN_AGPR) return agpr_tile(idx);
    else return v;
}

template <typename T, int BM, int BN, int RB, int NS, int EPI, int NWV = BM / 32>
__global__ __launch_bounds__(NWV * 64, (NS * (BM + BN) * RB > 80 * 1024) ? 1 : 2) void kgemm_kernel(const KGemmParams p) {
    using G = Geo<BM, BN, RB, NS, NWV>;
    constexpr int EPC = 16 / sizeof(T);      // elements per 16-byte chunk
    constexpr int BKE = RB / sizeof(T);      // elements per k-tile
    constexpr int KT_MUL = 128 / RB;         // KGemmParams counts k-tiles of 128 bytes
    __shared__ __attribute__((aligned(16))) char smem[G::LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int n_tiles_n = (p.N + BN - 1) / BN;
    // XCD-aware tile order: blocks are dealt round-robin over the 8 XCDs (bid % 8 labels the XCD group), each
    // XCD has its own L2.  Give every XCD a contiguous range of tiles (column tile fastest) so the blocks that
    // re-read one A row panel share an L2 instead of fetching it 8 times.  Bijective for any grid size.
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    // within an XCD's range: column tiles in groups of p.group_n, row tile next, so that the ~64 workgroups an XCD runs at a
    // time share group_n weight slices AND ~64 / group_n activation panels in its 4 MB L2 (column-fastest over all of a wide N
    // re-fetches the whole weight matrix once per row panel: 14x the algorithmic reads on N = 4096, PMC)
    int tile_m, tile_n;
    if (p.group_n > 0 && p.group_n < n_tiles_n) {
        const int n_tiles_m = nblk / n_tiles_n;
        const int per_group = p.group_n * n_tiles_m;
        const int grp = lid / per_group, in_grp = lid - grp * per_group;
        const int first_n = grp * p.group_n;
        const int gsz = (n_tiles_n - first_n) < p.group_n ? (n_tiles_n - first_n) : p.group_n;
        tile_m = in_grp / gsz;
        tile_n = first_n + (in_grp - tile_m * gsz);
    } else {
        tile_m = lid / n_tiles_n;
        tile_n = lid - tile_m * n_tiles_n;
    }
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- staging geometry: thread -> (row r0 + RPP i, slot c); stage s: A rows at smem + s * STAGE_BYTES, B behind
    const int c = tid % G::CPRW;
    const int r0 = tid / G::CPRW;
    // logical chunk this lane fetches for staging row r0 + RPP i (swizzle on the source side; period 64 rows)
#define KG_CSRC(i) (c ^ swz_of<RB, G::TN>(r0 + G::RPP * (i)))

    int a_base[G::A_ITERS], a_pos[G::A_ITERS], a_len[G::A_ITERS];
    bool a_ok[G::A_ITERS];
#pragma unroll
    for (int i = 0; i < G::A_ITERS; ++i) {
        const int m = m0 + r0 + G::RPP * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        const int seq = mm / p.Lout;
        const int pos = mm - seq * p.Lout;
        a_base[i] = seq * p.a_seq_rows + p.a_off;
        a_pos[i] = pos * p.a_stride;
        a_len[i] = p.seq_len ? p.seq_len[seq] : p.a_len;
    }

    int total_kt = 0;
    for (int t = 0; t < p.n_taps; ++t) total_kt += p.a_ktiles[t] * KT_MUL;

    // ---- MFMA geometry
    const int wm0 = (wave / G::WAVES_N) * G::WTM;
    const int wn0 = (wave % G::WAVES_N) * G::WTN;
    const int fr = lane & 15;
    const int fq = lane >> 4;
    // 128 x 128 wave tiles with one k-step per tile: software-pipelined main loop, accumulators in named AGPRs (above)
    constexpr bool PIPE = sizeof(T) == 2 && G::TM == 8 && G::TN == 8 && G::KS == 1;
    float4v acc[G::TM][G::TN];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};
    if constexpr (PIPE) agpr_zero_all();
#define KG_ACC(mt, nt) acc_tile<PIPE>(acc[mt][nt], (mt) * G::TN + (nt))

    // ---- staging: LDS-DMA (global_load_lds_dwordx4).  One wave-instruction writes 1 KiB = WROWS tile rows linearly
    // (LDS address = wave-uniform base + lane * 16).  No staging VGPRs, no ds_write.  Loads are unconditional
    // (addresses clamped into the tensor); padded / out-of-range rows read a zero page.
    const char* zero_ = reinterpret_cast<const char*>(p.zero_page);
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (LDS-DMA base goes to M0)
    const int wrow = wave_u * G::WROWS;        // first tile row written by this wave within one staging pass
    int tap = 0, kin = 0;                      // cursor of the NEXT tile to load
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    // Per-lane source pointers.  The row mapping (sequence / padding / tap shift, 64-bit row * stride) is evaluated
    // once per TAP; inside a tap every k-tile just advances the pointers by one tile row of RB bytes (0 for rows
    // that read the zero page), so the per-tile address work is a handful of adds.
    const char* pa[G::A_ITERS];
    int pinc[G::A_ITERS];
    const char* pb[G::B_ITERS];
#pragma unroll
    for (int i = 0; i < G::B_ITERS; ++i)
        pb[i] = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.w) + (long)(n0 + r0 + G::RPP * i) * p.ldw + KG_CSRC(i) * EPC);

#define KG_TAP_SETUP()                                                                                        \
    do {                                                                                                      \
        const T* ap_ = reinterpret_cast<const T*>(p.a_ptr[tap]);                                              \
        const long lda_ = p.a_ld[tap];                                                                        \
        const int sh_ = p.a_shift[tap];                                                                       \
        _Pragma("unroll") for (int i = 0; i < G::A_ITERS; ++i) {                                              \
            const int q0_ = a_pos[i] + sh_;                                                                   \
            const int len = a_len[i];                                                                         \
            const bool oob = (q0_ < 0) | (q0_ >= len);                                                        \
            /* reflect inside max(len, reflect_min): sequences not longer than the padding are zero-extended */ \
            /* first (encodec.py pad1d), so a reflected index can land on an extension row = zero          */ \
            const int lenx_ = len > p.reflect_min ? len : p.reflect_min;                                      \
            const int qr_ = q0_ < 0 ? -q0_ : (q0_ >= lenx_ ? 2 * (lenx_ - 1) - q0_ : q0_);                    \
            const bool refl_ = p.pad_mode == KG_PAD_REFLECT;                                                  \
            int q = (oob & refl_) ? qr_ : q0_;                                                                \
            const bool zext_ = refl_ & ((q < 0) | (q >= len));                                                \
            q = q > len - 1 ? len - 1 : q;                                                                    \
            q = q < 0 ? 0 : q;                                                                                \
            const bool ok = a_ok[i] & !(oob & (p.pad_mode == KG_PAD_ZERO)) & !zext_;                          \
            const long row = (long)a_base[i] + q;                                                             \
            const unsigned long pr_ = (unsigned long)(ap_ + row * lda_ + KG_CSRC(i) * EPC);                        \
            const unsigned long mk_ = 0ul - (unsigned long)ok;             /* branch-free pointer select */   \
            pa[i] = reinterpret_cast<const char*>((pr_ & mk_) | ((unsigned long)zero_ & ~mk_));               \
            pinc[i] = ok ? RB : 0;                                                                            \
        }                                                                                                     \
    } while (0)

#define KG_DMA(BUF)                                                                                           \
    do {                                                                                                      \
        if (kin == 0) KG_TAP_SETUP();                                                                         \
        char* la_ = smem + (BUF) * G::STAGE_BYTES + wrow * RB;                                                \
        _Pragma("unroll") for (int i = 0; i < G::A_ITERS; ++i) {                                              \
            __builtin_amdgcn_global_load_lds((gptr_t)pa[i], (lptr_t)(la_ + G::RPP * i * RB), 16, 0, 0);      \
            pa[i] += pinc[i];                                                                                 \
        }                                                                                                     \
        char* lb_ = la_ + BM * RB;                                                                            \
        _Pragma("unroll") for (int i = 0; i < G::B_ITERS; ++i) {                                              \
            __builtin_amdgcn_global_load_lds((gptr_t)pb[i], (lptr_t)(lb_ + G::RPP * i * RB), 16, 0, 0);      \
            pb[i] += RB;                                                                                      \
        }                                                                                                     \
        if (++kin == p.a_ktiles[tap] * KT_MUL) { kin = 0; ++tap; }                                            \
    } while (0)

#define KG_COMPUTE(BUF, CT)                                                                                   \
    do {                                                                                                      \
        const char* a_ = smem + (BUF) * G::STAGE_BYTES;                                                       \
        const char* b_ = a_ + BM * RB;                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < G::KS; ++ks) {                                                \
            u32x4 af[G::TM], bf[G::TN];                                                                       \
            const int chunk = ks * 4 + fq;                                                                    \
            _Pragma("unroll") for (int mt = 0; mt < G::TM; ++mt) {                                            \
                const int row = wm0 + mt * 16 + fr;                                                           \
                af[mt] = *reinterpret_cast<const u32x4*>(a_ + row * RB + ((chunk ^ swz_of<RB, G::TN>(row)) << 4));   \
            }                                                                                                 \
            _Pragma("unroll") for (int nt = 0; nt < G::TN; ++nt) {                                            \
                const int row = DIRECT ? wn0 + CW * (fr >> 2) + 4 * nt + (fr & 3) : wn0 + nt * 16 + fr;       \
                bf[nt] = *reinterpret_cast<const u32x4*>(b_ + row * RB + ((chunk ^ swz_of<RB, G::TN>(row)) << 4));   \
            }                                                                                                 \
            _Pragma("unroll") for (int mt = 0; mt < G::TM; ++mt)                                              \
                _Pragma("unroll") for (int nt = 0; nt < G::TN; ++nt) {                                        \
                    /* CT: weights are the A operand -> accumulator = C^T tile (lane: 4 columns of 1 row) */  \
                    const u32x4 x_ = (CT) ? bf[nt] : af[mt];                                                  \
                    const u32x4 y_ = (CT) ? af[mt] : bf[nt];                                                  \
                    if constexpr (sizeof(T) == 2) {                                                           \
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                                 \
                            __builtin_bit_cast(half8, x_), __builtin_bit_cast(half8, y_), acc[mt][nt], 0, 0, 0); \
                    } else {                                                                                  \
                        /* lane group fq holds K = 4 fq + j of this 16-wide K group in element j (same */     \
                        /* permutation for A and B, so the contraction is exact). */                          \
                        const float4v fx = __builtin_bit_cast(float4v, x_);                                   \
                        const float4v fy = __builtin_bit_cast(float4v, y_);                                   \
                        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                         \
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fx[j], fy[j], acc[mt][nt], 0, 0, 0); \
                    }                                                                                         \
                }                                                                                             \
        }                                                                                                     \
    } while (0)

    // ---- LDS-DMA ring.  Each wave waits for its own DMAs of tile it with a COUNTED vmcnt (the DPT instructions of
    // each later tile already issued may stay outstanding), then a raw s_barrier makes every wave's part of tile it
    // visible and guarantees that the stage about to be refilled (read during iteration it-1) is no longer being
    // read.  __syncthreads() is avoided in the loop because it would drain the DMA queue (vmcnt(0)).
#define KG_MAINLOOP(CT)                                                                                       \
    for (int it = 0; it < total_kt; ++it) {                                                                   \
        const int ahead = total_kt - 1 - it;   /* tiles after `it` whose DMAs are issued: min(ahead, NS-2) */ \
        wait_tiles<G::NSTAGE - 2, G::DPT>(ahead);                                                             \
        asm volatile("s_barrier" ::: "memory");                                                               \
        if (it + G::NSTAGE - 1 < total_kt && !(p.debug & 1)) KG_DMA(fill);                                    \
        KG_COMPUTE(stage, CT);                                                                                \
        stage = stage + 1 == G::NSTAGE ? 0 : stage + 1;                                                       \
        fill = fill + 1 == G::NSTAGE ? 0 : fill + 1;                                                          \
    }

    // ---- software-pipelined form for the 128 x 128 wave tiles (one wave per SIMD, nothing else to hide behind).  The 64
    // MFMAs of a tile run as four quarters (A half x B half); a fragment half is re-read for the NEXT tile as soon as
    // its last quarter has issued, and consecutive tiles walk the quarters in mirrored orders so that what was freed first
    // is needed first: no second register set, and the MFMA pipe never waits for a barrier, a DMA issue or an LDS read.
    //   tile order (X, Y) = (0, 1) even / (1, 0) odd:  Q(X,0) Q(Y,0) | sync, refill, read B0' | Q(Y,1) | read AY' | Q(X,1) | read AX' B1'
    // A tile's stage is free once every wave holds all its fragments, which is the case at the mid-tile sync of the same
    // iteration: all NS stages carry tiles in flight (the prologue issues NS tiles).
    int fbase_a[4], fbase_bt[4], fbase_bn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // with swz = ((row >> 1) ^ (row >> 4)) & 3 (conflict-free in ds_read_b128's lane groups for 16 consecutive rows and
        // for the permuted rows 32 g + 4 nt + i alike) the 16-byte slot of a fragment row is a lane term XOR a compile-time
        // constant k: four lane bases per operand + immediate offsets address all 8 fragments
        const int g = fr >> 2, i = fr & 3;
        fbase_a[k] = (wm0 + fr) * RB + (((fq ^ (fr >> 1) ^ k) & 3) << 4);                  // rows wm0 + 16 mt + fr: k = mt & 3
        fbase_bn[k] = (BM + wn0 + fr) * RB + (((fq ^ (fr >> 1) ^ k) & 3) << 4);            // natural weight rows (STORE): k = nt & 3
        fbase_bt[k] = (BM + wn0 + 32 * g + i) * RB + (((fq ^ (i >> 1) ^ (2 * (g & 1)) ^ k) & 3) << 4);   // permuted rows 32 g + 4 nt + i
    }                                                                                      // (DIRECT): k = 2 (nt & 1) ^ (nt >> 2)
    u32x4 pf_a[G::TM], pf_b[G::TN];
#define KG_RD_A(H, BUF)                                                                                       \
    _Pragma("unroll") for (int mt = 4 * (H); mt < 4 * (H) + 4; ++mt)                                          \
        pf_a[mt] = *reinterpret_cast<const u32x4*>(smem + (BUF) * G::STAGE_BYTES + fbase_a[mt & 3] + mt * 16 * RB)
#define KG_RD_B(H, BUF, CT)                                                                                   \
    _Pragma("unroll") for (int nt = 4 * (H); nt < 4 * (H) + 4; ++nt)                                          \
        pf_b[nt] = DIRECT ? *reinterpret_cast<const u32x4*>(smem + (BUF) * G::STAGE_BYTES + fbase_bt[(2 * (nt & 1)) ^ (nt >> 2)] + nt * 4 * RB) \
                        : *reinterpret_cast<const u32x4*>(smem + (BUF) * G::STAGE_BYTES + fbase_bn[nt & 3] + nt * 16 * RB)
#define KG_Q(HA, HB, CT)                                                                                      \
    _Pragma("unroll") for (int mt = 4 * (HA); mt < 4 * (HA) + 4; ++mt)                                        \
        _Pragma("unroll") for (int nt = 4 * (HB); nt < 4 * (HB) + 4; ++nt)                                    \
            agpr_mfma_f16(mt * G::TN + nt, (CT) ? pf_b[nt] : pf_a[mt], (CT) ? pf_a[mt] : pf_b[nt])
#define KG_PIPE_STEP(X, Y, IT, CT)                                                                            \
    {                                                                                                         \
        const bool more_ = (IT) + 1 < total_kt;                                                               \
        const int nx_ = stage + 1 == G::NSTAGE ? 0 : stage + 1;                                               \
        KG_Q(X, 0, CT);                                                                                       \
        KG_Q(Y, 0, CT);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (more_) {                                                                                          \
            wait_tiles<G::NSTAGE - 2, G::DPT>(total_kt - 2 - (IT));      /* tile IT + 1 has landed */         \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  /* ... and everyone holds tile IT */ \
            if ((IT) + G::NSTAGE < total_kt && !(p.debug & 1)) KG_DMA(stage);                                                   \
            KG_RD_B(0, nx_, CT);                                                                              \
        }                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        KG_Q(Y, 1, CT);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (more_) KG_RD_A(Y, nx_);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        KG_Q(X, 1, CT);                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (more_) { KG_RD_A(X, nx_); KG_RD_B(1, nx_, CT); }                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        stage = nx_;                                                                                          \
    }
#define KG_PIPELOOP(CT)                                                                                       \
    {                                                                                                         \
        wait_tiles<G::NSTAGE - 1, G::DPT>(total_kt - 1);                                                      \
        asm volatile("s_barrier" ::: "memory");                                                               \
        KG_RD_A(0, 0); KG_RD_B(0, 0, CT); KG_RD_A(1, 0); KG_RD_B(1, 0, CT);                                   \
        for (int it = 0; it < total_kt; it += 2) {                                                            \
            KG_PIPE_STEP(0, 1, it, CT)                                                                        \
            if (it + 1 < total_kt) KG_PIPE_STEP(1, 0, it + 1, CT)                                             \
        }                                                                                                     \
        /* hipcc pads no hazards around inline asm: the last MFMAs retire before the epilogue reads the AGPRs */ \
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                                                   \
    }
    // Two epilogue styles.  DIRECT (SwiGLU / tanh-sigmoid / QKV+RoPE, fp16 outputs): C^T accumulators with permuted
    // weight rows, every lane owns CW consecutive columns of one row, no LDS.  STORE (fp32 residual stream, c32 + res
    // traffic dominates): C accumulators transposed through LDS so that 8 consecutive lanes cover 256 contiguous bytes
    // of a row -- 4x fewer memory requests per byte, which is what bounds that epilogue (measured: direct stores were
    // 10-20 % slower for wo / w2).
    constexpr bool DIRECT = EPI != KG_EPI_STORE;
    constexpr int CW = DIRECT ? 4 * G::TN : 8;         // consecutive output columns per lane
    // V blocks of the QKV GEMM (whole 128-column blocks: 2 D is a multiple of 128) keep the C orientation
    const bool v_blk = (EPI == KG_EPI_QKV_ROPE) && (n0 >= 2 * p.rope_D);

#pragma unroll
    for (int s_ = 0; s_ < G::NSTAGE - (PIPE ? 0 : 1); ++s_)
        if (s_ < total_kt) KG_DMA(s_);

    int stage = 0, fill = G::NSTAGE - 1;
    if constexpr (PIPE) {
        if constexpr (EPI == KG_EPI_QKV_ROPE) {
            if (v_blk) { KG_PIPELOOP(false) } else { KG_PIPELOOP(true) }
        } else if constexpr (DIRECT) {
            KG_PIPELOOP(true)
        } else {
            KG_PIPELOOP(false)
        }
    } else if constexpr (EPI == KG_EPI_QKV_ROPE) {
        if (v_blk) { KG_MAINLOOP(false) } else { KG_MAINLOOP(true) }
    } else if constexpr (DIRECT) {
        KG_MAINLOOP(true)
    } else {
        KG_MAINLOOP(false)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef KG_PIPELOOP
#undef KG_PIPE_STEP
#undef KG_Q
#undef KG_RD_A
#undef KG_RD_B
#undef KG_MAINLOOP
#undef KG_DMA
#undef KG_TAP_SETUP
#undef KG_COMPUTE
#undef KG_CSRC

    if (p.debug & 2) return;
    // ---- epilogue, straight from the accumulators.
    if constexpr (EPI == KG_EPI_QKV_ROPE) {
        if (v_blk) {
            // acc[mt][nt][r] = C[m0 + wm0 + 16 mt + 4 fq + r][col(nt, fr)]: 4 consecutive positions of one V column
            // -> 8-byte stores into vt[seq][d][vt_pos(pos, mode)] (the attention kernel's column order, common.h)
#pragma unroll
            for (int nt = 0; nt < G::TN; ++nt) {
                const int n = n0 + wn0 + CW * (fr >> 2) + 4 * nt + (fr & 3);
                if (n >= p.N) continue;
                const int d = n - 2 * p.rope_D;
#pragma unroll
                for (int mt = 0; mt < G::TM; ++mt) {
                    const int m = m0 + wm0 + mt * 16 + 4 * fq;
                    if (m >= p.M) continue;
                    const int seq = m / p.Lout;
                    const int pos = m - seq * p.Lout;
                    const float4v a4 = KG_ACC(mt, nt);
                    if (m + 4 <= p.M && (pos & 3) == 0 && pos + 4 <= p.Lout) {
                        const half4 h = {(half_t)a4[0], (half_t)a4[1], (half_t)a4[2], (half_t)a4[3]};
                        *reinterpret_cast<half4*>(p.vt + (long)seq * p.vt_seq_stride + (long)d * p.vt_ld + vt_pos(pos, p.vt_mode ? p.vt_mode : 1)) = h;
                    } else {
                        for (int j = 0; j < 4; ++j) {
                            const int mj = m + j;
                            if (mj >= p.M) break;
                            const int sj = mj / p.Lout;
                            const int pj = mj - sj * p.Lout;
                            p.vt[(long)sj * p.vt_seq_stride + (long)d * p.vt_ld + vt_pos(pj, p.vt_mode ? p.vt_mode : 1)] = (half_t)a4[j];
                        }
                    }
                }
            }
            return;
        }
    }

    if constexpr (!DIRECT) {
        __syncthreads();                                   // every wave is done reading the ring
    // ---- epilogue.  Accumulators go through a per-wave LDS region so that each lane ends up with CW consecutive
    // columns of one row (16/32-byte global accesses).  The wave
    // tile is processed in EP row passes; in each pass the chunk coordinates are computed first and the residual
    // rows are fetched BEFORE the LDS transposition, so their latency hides under it.
    constexpr int CW = 8;                              // consecutive columns handled by one lane
    constexpr int CPR = G::WTN / CW;                   // chunks per row
    constexpr int TMP = G::TM / G::EP;                 // m-tiles per pass
    constexpr int ROWS_P = TMP * 16;                   // wave-tile rows per pass
    constexpr int NCH = ROWS_P * CPR / 64;             // chunks per lane per pass
    static_assert(ROWS_P * CPR % 64 == 0, "epilogue chunking");
    float* ep = reinterpret_cast<float*>(smem) + wave * ROWS_P * G::EPI_LD;

#pragma unroll
    for (int pass = 0; pass < G::EP; ++pass) {
        const int prow0 = pass * ROWS_P;               // first wave-tile row of this pass
        long orow_[NCH];
        int seq_[NCH], pos_[NCH];
        bool ok_[NCH];
        float4v rs0[NCH], rs1[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + 64 * i;
            const int row = ch / CPR;
            const int cc = ch - row * CPR;
            const int m = m0 + wm0 + prow0 + row;
            const int n = n0 + wn0 + cc * CW;
            ok_[i] = (m < p.M) & (n < p.N);
            const int mm = ok_[i] ? m : 0;
            seq_[i] = mm / p.Lout;
            pos_[i] = mm - seq_[i] * p.Lout;
            orow_[i] = (long)seq_[i] * p.c_seq_rows + p.c_off + pos_[i];
            rs0[i] = (float4v){0.f, 0.f, 0.f, 0.f};
            rs1[i] = rs0[i];
            {
                if (p.res && p.vec_ok && ok_[i]) {
                    rs0[i] = *reinterpret_cast<const float4v*>(p.res + orow_[i] * p.ldres + n);
                    rs1[i] = *reinterpret_cast<const float4v*>(p.res + orow_[i] * p.ldres + n + 4);
                }
            }
        }
        if (pass > 0) __syncthreads();                 // the previous pass has been read out
#pragma unroll
        for (int mi = 0; mi < TMP; ++mi)
#pragma unroll
            for (int nt = 0; nt < G::TN; ++nt) {
                const float4v a4 = KG_ACC(pass * TMP + mi, nt);
#pragma unroll
                for (int r = 0; r < 4; ++r) ep[(mi * 16 + fq * 4 + r) * G::EPI_LD + nt * 16 + fr] = a4[r];
            }
        __syncthreads();

#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + 64 * i;
            const int row = ch / CPR;
            const int cc = ch - row * CPR;
            const int n = n0 + wn0 + cc * CW;
            float v[CW];
#pragma unroll
            for (int q4 = 0; q4 < CW / 4; ++q4) {
                const float4v x = *reinterpret_cast<const float4v*>(ep + row * G::EPI_LD + cc * CW + q4 * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[q4 * 4 + j] = x[j];
            }
            if (!ok_[i]) continue;
            const int seq = seq_[i];
            const int pos = pos_[i];
            const long orow = orow_[i];
            const int nv = (p.N - n) < CW ? (p.N - n) : CW;

            if (p.bias) {
                if (nv == CW) {
#pragma unroll
                    for (int q4 = 0; q4 < CW / 4; ++q4) {
                        const float4v b = *reinterpret_cast<const float4v*>(p.bias + n + q4 * 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[q4 * 4 + j] += b[j];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < CW; ++j) if (j < nv) v[j] += p.bias[n + j];
                }
            }
            if (p.rowvec) {
                const float* rv = p.rowvec + (long)seq * p.ld_rowvec + n;
#pragma unroll
                for (int j = 0; j < CW; ++j) if (j < nv) v[j] += rv[j];
            }

            {
                if (p.act != KG_ACT_NONE) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = act_apply(v[j], p.act, p.act_slope);
                }
                if (p.gate) {
                    const float* gv = p.gate + (long)seq * p.ld_gate + n;
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (j < nv) v[j] *= gv[j];
                }
                if (p.vec_ok) {
                    if (p.res) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v[j] += rs0[i][j]; v[4 + j] += rs1[i][j]; }
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
                    if (p.post_relu) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                    }
                    if (p.c32) {
                        *reinterpret_cast<float4v*>(p.c32 + orow * p.ldc32 + n) = (float4v){v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<float4v*>(p.c32 + orow * p.ldc32 + n + 4) = (float4v){v[4], v[5], v[6], v[7]};
                    }
                    if (p.post_a) {      // fused pointwise Snake on the way to the next conv's fp16 (hi / lo) operand planes
                        float lo[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int nn = n + j;
                            const float sv = nn < p.post_n ? v[j] + p.post_ib[nn] * sin_sq(p.post_a[nn] * v[j]) : 0.f;
                            const half_t h = (half_t)sv;
                            v[j] = sv;
                            lo[j] = sv - (float)h;
                        }
                        if (p.c16_lo) *reinterpret_cast<uint4*>(p.c16_lo + orow * p.ldc16 + n) = pack8(lo);
                    }
                    if (p.c16) *reinterpret_cast<uint4*>(p.c16 + orow * p.ldc16 + n) = pack8(v);
                } else {
                    for (int j = 0; j < nv; ++j) {
                        float o = v[j];
                        if (p.res) o += p.res[orow * p.ldres + n + j];
                        if (p.out_scale != 0.f) o *= p.out_scale;
                        if (p.res2) o += p.res2[orow * p.ldres2 + n + j];
                        if (p.post_relu) o = fmaxf(o, 0.f);
                        if (p.c32) p.c32[orow * p.ldc32 + n + j] = o;
                        if (p.post_a) {
                            const int nn = n + j;
                            o = nn < p.post_n ? o + p.post_ib[nn] * sin_sq(p.post_a[nn] * o) : 0.f;
                            if (p.c16_lo) p.c16_lo[orow * p.ldc16 + n + j] = (half_t)(o - (float)(half_t)o);
                        }
                        if (p.c16) p.c16[orow * p.ldc16 + n + j] = (half_t)o;
                    }
                }
            }
        }
    }
        return;
    }

    // acc[mt][nt][r] = C[m0 + wm0 + 16 mt + fr][n0 + wn0 + CW fq + 4 nt + r]: CW consecutive columns of one row
    const int n = n0 + wn0 + CW * fq;
    if (n >= p.N) return;
    const int nv = (p.N - n) < CW ? (p.N - n) : CW;
    const bool full = nv == CW && p.vec_ok;
    long orow_[G::TM];
    int seq_[G::TM], pos_[G::TM];
    bool ok_[G::TM];
#pragma unroll
    for (int mt = 0; mt < G::TM; ++mt) {
        const int m = m0 + wm0 + mt * 16 + fr;
        ok_[mt] = m < p.M;
        const int mm = ok_[mt] ? m : 0;
        seq_[mt] = mm / p.Lout;
        pos_[mt] = mm - seq_[mt] * p.Lout;
        orow_[mt] = (long)seq_[mt] * p.c_seq_rows + p.c_off + pos_[mt];
    }
    // column-only terms once per lane
    float bias_[CW];
#pragma unroll
    for (int j = 0; j < CW; ++j) bias_[j] = 0.f;
    if (p.bias) {
        if (nv == CW) {
#pragma unroll
            for (int q4 = 0; q4 < CW / 4; ++q4) {
                const float4v b = *reinterpret_cast<const float4v*>(p.bias + n + q4 * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) bias_[q4 * 4 + j] = b[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < CW; ++j) if (j < nv) bias_[j] = p.bias[n + j];
        }
    }
#pragma unroll
    for (int mt = 0; mt < G::TM; ++mt) {
        float v[CW];
#pragma unroll
        for (int nt = 0; nt < G::TN; ++nt) {
            const float4v a4 = KG_ACC(mt, nt);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * nt + r] = a4[r] + bias_[4 * nt + r];
        }
        if (!ok_[mt]) continue;
        const int seq = seq_[mt];
        const int pos = pos_[mt];
        const long orow = orow_[mt];

        if (p.rowvec) {
            const float* rv = p.rowvec + (long)seq * p.ld_rowvec + n;
#pragma unroll
            for (int j = 0; j < CW; ++j) if (j < nv) v[j] += rv[j];
        }

        if constexpr (EPI == KG_EPI_SWIGLU || EPI == KG_EPI_TANHSIG) {
            // interleaved (2j, 2j+1) weight rows -> lane-local pairs; exp2 / rcp on the transcendental unit
            constexpr float LOG2E = 1.4426950408889634f;
            float o[CW / 2];
#pragma unroll
            for (int j = 0; j < CW / 2; ++j) {
                const float a = v[2 * j], b = v[2 * j + 1];
                if constexpr (EPI == KG_EPI_SWIGLU) {
                    o[j] = a * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-a * LOG2E)) * b;
                } else {
                    const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.0f * LOG2E * a));
                    o[j] = th * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-b * LOG2E));
                }
            }
            if (p.c16) {
                if constexpr (CW >= 16) {
#pragma unroll
                    for (int q8 = 0; q8 < CW / 16; ++q8)
                        *reinterpret_cast<uint4*>(p.c16 + orow * p.ldc16 + (n >> 1) + q8 * 8) = pack8(o + q8 * 8);
                } else {
                    const half4 h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
                    *reinterpret_cast<half4*>(p.c16 + orow * p.ldc16 + (n >> 1)) = h;
                }
            }
            if (p.c32) {
#pragma unroll
                for (int q4 = 0; q4 < CW / 8; ++q4)
                    *reinterpret_cast<float4v*>(p.c32 + orow * p.ldc32 + (n >> 1) + q4 * 4) =
                        (float4v){o[q4 * 4], o[q4 * 4 + 1], o[q4 * 4 + 2], o[q4 * 4 + 3]};
            }
        } else if constexpr (EPI == KG_EPI_QKV_ROPE) {
            // q / k columns: rotate interleaved pairs with the position's (cos, sin); q also gets q_scale
            const int pair0 = (n & 63) >> 1;
            const float* tb = p.rope + ((long)pos * 32 + pair0) * 2;
            const float sc = n < p.rope_D ? p.q_scale : 1.0f;
            float o[CW];
#pragma unroll
            for (int q4 = 0; q4 < CW / 4; ++q4) {
                const float4v t = *reinterpret_cast<const float4v*>(tb + q4 * 4);      // (cos, sin) of 2 pairs
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const float cs = t[2 * h2], sn = t[2 * h2 + 1];
                    const float x0 = v[q4 * 4 + 2 * h2], x1 = v[q4 * 4 + 2 * h2 + 1];
                    o[q4 * 4 + 2 * h2] = (x0 * cs - x1 * sn) * sc;
                    o[q4 * 4 + 2 * h2 + 1] = (x1 * cs + x0 * sn) * sc;
                }
            }
#pragma unroll
            for (int q8 = 0; q8 < CW / 8; ++q8)
                *reinterpret_cast<uint4*>(p.c16 + orow * p.ldc16 + n + q8 * 8) = pack8(o + q8 * 8);
        }
    }
}

#undef KG_ACC

template <typename T, int BM, int BN, int RB, int NS, int EPI, int NWV = BM / 32>
int launch_one(const KGemmParams& p, hipStream_t st) {
    const int grid = cdiv(p.M, BM) * cdiv(p.N, BN);
    if (grid <= 0) return 0;
    hipLaunchKernelGGL((kgemm_kernel<T, BM, BN, RB, NS, EPI, NWV>), dim3(grid), dim3(NWV * 64), 0, st, p);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace

}  // namespace svc
