// Tap-GEMM for gfx950: C = epilogue(sum over taps of A_tap[rowmap] * W^T), MFMA 16x16 tiles.
//
// One kernel family serves every dense contraction on the path: DiT linears (1 tap), UViT / long-skip
// concatenations (2 taps = 2 source buffers), WaveNet and vocoder Conv1d (k taps = shifted rows of a
// channels-last activation, zero / reflect / replicate padding, stride), polyphase ConvTranspose1d
// (3 taps, N = stride * Cout), split-precision fp16x3 products (3 sub-taps per tap).  fp16 operands use
// v_mfma_f32_16x16x32_f16, fp32 operands use v_mfma_f32_16x16x4_f32 (exact fp32 fma chain); fp32 accumulate.
//
// Tile: BM x BN (128 x {128,64,32} with 4 waves, 256 x 128 with 8 waves), k-tile rows of RB = 128 or 64 bytes.
// Staging is LDS-DMA (global_load_lds_dwordx4) into an NS-stage ring: tiles it+1 .. it+NS-2 are in flight under
// the MFMAs of tile it, behind a counted vmcnt + raw s_barrier.  LDS image: [rows][RB] with the 16-byte chunk
// index XOR-swizzled by the row (applied on the DMA source side, the DMA writes linearly): conflict-free
// ds_read_b128 for the MFMA operand pattern.  The MFMAs compute C^T (weights as the A operand, with the weight rows
// of a wave tile read in a permuted order), so every lane ends up with 16 (or 8) CONSECUTIVE output columns of one
// output row in its accumulators: the epilogue needs no LDS transposition and no barrier, and SwiGLU / RoPE pairs
// are lane-local.  Only the V blocks of the QKV GEMM keep the C orientation (4 consecutive positions per lane) for
// their transposed store.
#include <stdlib.h>

#include "common.h"

namespace svc {

namespace {

// Tile geometry.
template <int BM, int BN, int RB, int NS, int NWV>
struct Geo {
    static constexpr int NW = NWV;                           // waves (BM / 32: 64x64 | 32x{64,32} wave tiles; BM / 64: 128x64)
    static constexpr int NT = NW * 64;                       // threads
    static constexpr int WAVES_N = (BN == 128) ? 2 : 1;
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
// 4 TN g + 4 nt + (fr & 3), g = fr >> 2: the second term folds row bits 4..5 into the key.
template <int RB>
__device__ __forceinline__ int swz_of(int row) {
    if constexpr (RB == 128) return ((row >> 1) ^ (((row >> 4) & 3) << 1)) & 7;     // 2 rows per 256-byte bank line
    else return (0 - (((row >> 2) ^ (row >> 4)) & 3)) & 3;                          // 4 rows per bank line
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
    const int tile_m = lid / n_tiles_n;
    const int tile_n = lid - tile_m * n_tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- staging geometry: thread -> (row r0 + RPP i, slot c); stage s: A rows at smem + s * STAGE_BYTES, B behind
    const int c = tid % G::CPRW;
    const int r0 = tid / G::CPRW;
    // logical chunk this lane fetches for staging row r0 + RPP i (swizzle on the source side; period 64 rows)
#define KG_CSRC(i) (c ^ swz_of<RB>(r0 + G::RPP * (i)))

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
    float4v acc[G::TM][G::TN];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};

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
                af[mt] = *reinterpret_cast<const u32x4*>(a_ + row * RB + ((chunk ^ swz_of<RB>(row)) << 4));   \
            }                                                                                                 \
            _Pragma("unroll") for (int nt = 0; nt < G::TN; ++nt) {                                            \
                const int row = DIRECT ? wn0 + CW * (fr >> 2) + 4 * nt + (fr & 3) : wn0 + nt * 16 + fr;       \
                bf[nt] = *reinterpret_cast<const u32x4*>(b_ + row * RB + ((chunk ^ swz_of<RB>(row)) << 4));   \
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
    for (int s_ = 0; s_ < G::NSTAGE - 1; ++s_)
        if (s_ < total_kt) KG_DMA(s_);

    int stage = 0, fill = G::NSTAGE - 1;
    if constexpr (EPI == KG_EPI_QKV_ROPE) {
        if (v_blk) { KG_MAINLOOP(false) } else { KG_MAINLOOP(true) }
    } else if constexpr (DIRECT) {
        KG_MAINLOOP(true)
    } else {
        KG_MAINLOOP(false)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
                    const float4v a4 = acc[mt][nt];
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
            for (int nt = 0; nt < G::TN; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ep[(mi * 16 + fq * 4 + r) * G::EPI_LD + nt * 16 + fr] = acc[pass * TMP + mi][nt][r];
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
        for (int nt = 0; nt < G::TN; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * nt + r] = acc[mt][nt][r] + bias_[4 * nt + r];
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
                if constexpr (CW == 16) *reinterpret_cast<uint4*>(p.c16 + orow * p.ldc16 + (n >> 1)) = pack8(o);
                else {
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

template <typename T, int BM, int BN, int RB, int NS, int EPI, int NWV = BM / 32>
int launch_one(const KGemmParams& p, hipStream_t st) {
    const int grid = cdiv(p.M, BM) * cdiv(p.N, BN);
    if (grid <= 0) return 0;
    hipLaunchKernelGGL((kgemm_kernel<T, BM, BN, RB, NS, EPI, NWV>), dim3(grid), dim3(NWV * 64), 0, st, p);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

// Variant selection for 128-column tiles.  debug bits 4..7 pick a variant explicitly (tuning harness):
//   0x10: 128x128, 64-byte rows, 3 stages   0x20: 128x128, 64-byte rows, 4 stages
//   0x40: 128x128, 128-byte rows, 2 stages  0x80: 256x128, 128-byte rows, 3 stages   (none: by reduction length)
template <typename T, int EPI>
int launch_wide(const KGemmParams& p, hipStream_t st) {
    const int v = p.debug & 0xF0;
    if (v == 0x10) return launch_one<T, 128, 128, 64, 3, EPI>(p, st);
    if (v == 0x20) return launch_one<T, 128, 128, 64, 4, EPI>(p, st);
    if (v == 0x80) return launch_one<T, 256, 128, 128, 3, EPI>(p, st);
    if (v == 0x40) return launch_one<T, 128, 128, 128, 2, EPI>(p, st);
    if (v == 0x50) return launch_one<T, 64, 64, 128, 2, EPI, 4>(p, st);       // small-M launches, see below
    if (v == 0x60) return launch_one<T, 64, 64, 128, 4, EPI, 4>(p, st);
    if (v == 0x70) return launch_one<T, 64, 128, 128, 2, EPI, 4>(p, st);
    // measured on MI355X (tools/gemm_bench.py, profiles/r01_c_gemm_variants.txt): short reductions (K <= 512 fp16)
    // run 5-15 % faster with 64-byte rows / 3 stages / 3 workgroups per CU; long reductions prefer 128-byte rows.
    long kt = 0;
    for (int t = 0; t < p.n_taps; ++t) kt += p.a_ktiles[t];
    // Small-M launches (single-utterance latency, M ~ 1.7 k rows): a 128x128 grid would leave most of the 256 CUs idle
    // and the launch is bound by the serial latency of ONE workgroup (barrier -> ds_read -> MFMA chain per k-tile, one
    // wave per SIMD, nothing to overlap with), not by memory: deep rings measured no gain, more and smaller wave tiles
    // do.  64x64 tiles with four 16x64 waves (tools/gemm_small_bench.py, us per launch, M = 1720): wo 12.9 -> 7.6,
    // QKV 13.6 -> 9.6, w2 20.8 -> 12.6, WaveNet in-layer (K = 2560) 27.9 -> 17.3 with a 4-stage ring; wide outputs
    // (w1/w3, N = 3072) take 64x128 tiles: 13.3 -> 11.2.  The K order of every output element is unchanged, so results
    // are bit-identical to the large tiles.
    const long gm64 = cdiv(p.M, 64);
    if (gm64 * cdiv(p.N, 64) <= 768) {
        if (kt >= 16) return launch_one<T, 64, 64, 128, 4, EPI, 4>(p, st);
        return launch_one<T, 64, 64, 128, 2, EPI, 4>(p, st);
    }
    if (gm64 * cdiv(p.N, 128) <= 768) return launch_one<T, 64, 128, 128, 2, EPI, 4>(p, st);
    if (kt <= 8) return launch_one<T, 128, 128, 64, 3, EPI>(p, st);      // a_ktiles counts 128-byte k-tiles
    return launch_one<T, 128, 128, 128, 2, EPI>(p, st);
}

template <typename T, int EPI>
int launch_bn(const KGemmParams& p, hipStream_t st) {
    // narrow outputs (vocoder tail, 80/18/1-channel heads) use narrower column tiles
    if (p.N <= 32) return launch_one<T, 128, 32, 128, 2, EPI>(p, st);
    if (p.N <= 64) return launch_one<T, 128, 64, 128, 2, EPI>(p, st);
    return launch_wide<T, EPI>(p, st);
}

}  // namespace

int kgemm_dispatch(const KGemmParams& p, int dtype, int epi, hipStream_t st);

int kgemm_launch(const KGemmParams& p_in, int dtype, int epi, hipStream_t st) {
    const KGemmParams& p0 = p_in;
    SVC_REQUIRE(p0.n_taps >= 1 && p0.n_taps <= KG_MAX_TAPS, "tap count");
    SVC_REQUIRE(p0.Lout > 0 && p0.M >= 0 && p0.N > 0, "shape");
    if (p0.M == 0) return 0;
    DeviceState* ds = device_state();
    if (!ds) return 1;
    KGemmParams p = p_in;
    p.zero_page = ds->zero_page;
    // tuning / test hook: SVC_KGEMM_VARIANT=16|32|128 forces a tile variant (see launch_wide) for every launch
    static const int env_variant = [] { const char* e = getenv("SVC_KGEMM_VARIANT"); return e ? atoi(e) & 0xF0 : 0; }();
    if (!(p.debug & 0xF0)) p.debug |= env_variant;
    const bool prof = prof_enabled();
    const int cls = dtype == 0 ? PROF_KGEMM_F16 : PROF_KGEMM_F32;
    if (prof) prof_begin(cls, st);
    const int rc = kgemm_dispatch(p, dtype, epi, st);
    if (prof) {
        long kt = 0;
        for (int t = 0; t < p.n_taps; ++t) kt += p.a_ktiles[t];
        const double K = (double)kt * (dtype == 0 ? 64 : 32);
        const double es = dtype == 0 ? 2 : 4;
        // algorithmic traffic: A once, W once, C once (fp32 and/or fp16), residual once
        double bytes = ((double)p.M * K / (p.n_taps > 1 && p.a_ptr[0] == p.a_ptr[p.n_taps - 1] ? p.n_taps : 1) + (double)p.N * K) * es;
        bytes += (double)p.M * p.N * ((p.c32 ? 4 : 0) + (p.c16 ? 2 : 0) + (p.res ? 4 : 0) + (p.res2 ? 4 : 0));
        const unsigned long long tag = ((unsigned long long)p.M << 40) | ((unsigned long long)(p.N & 0xFFFFF) << 20) |
                                       ((unsigned long long)((long)K & 0xFFFF) << 4) | (unsigned)(epi & 0xF);
        prof_end(cls, 2.0 * p.M * (double)p.N * K * (p.prof_flop_scale > 0.f ? p.prof_flop_scale : 1.0), bytes, st, tag);
    }
    return rc;
}

int kgemm_dispatch(const KGemmParams& p, int dtype, int epi, hipStream_t st) {
    if (dtype == 0) {
        switch (epi) {
            case KG_EPI_STORE: return launch_bn<half_t, KG_EPI_STORE>(p, st);
            case KG_EPI_SWIGLU: return launch_wide<half_t, KG_EPI_SWIGLU>(p, st);
            case KG_EPI_TANHSIG: return launch_wide<half_t, KG_EPI_TANHSIG>(p, st);
            case KG_EPI_QKV_ROPE: return launch_wide<half_t, KG_EPI_QKV_ROPE>(p, st);
        }
    } else {
        switch (epi) {
            case KG_EPI_STORE: return launch_bn<float, KG_EPI_STORE>(p, st);
            case KG_EPI_TANHSIG: return launch_wide<float, KG_EPI_TANHSIG>(p, st);
        }
    }
    set_error("kgemm: unsupported dtype/epilogue combination");
    return 1;
}

}  // namespace svc
