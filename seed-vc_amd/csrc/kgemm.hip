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
#include "kgemm_tile.h"

namespace svc {

namespace {

// Variant selection for 128-column tiles.  debug bits 4..7 pick a variant explicitly (tuning harness):
//   0x10: 128x128, 64-byte rows, 3 stages   0x20: 128x128, 64-byte rows, 4 stages
//   0x40: 128x128, 128-byte rows, 2 stages  0x80: 256x128, 128-byte rows, 3 stages   (none: by reduction length)
//   0xB0: 256x128 with four 128x64 waves, 64-byte rows, 3 stages (72 KB: two workgroups per CU)
//   0x90: 256x256 with four 128x128 waves, accumulators in AGPRs, pipelined (kgemm_big.hip; harness only)
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
    if (v == 0xB0) return launch_one<T, 256, 128, 64, 3, EPI, 4>(p, st);      // 128x64 wave tiles, two workgroups per CU
    if constexpr (sizeof(T) == 2) {
        // 256x256 (kgemm_big.hip): whole 256-column tiles only (weights are padded to 128 rows), V blocks from a tile edge
        const bool big_ok = p.N % 256 == 0 && (EPI != KG_EPI_QKV_ROPE || (2 * p.rope_D) % 256 == 0);
        if (v == 0x90 && big_ok) return kgemm_big_launch(p, EPI, st);
    }
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
    // wide SwiGLU projections on long reductions (the D = 768 model's w1 | w3: N = 4096, K = 768): 256 x 128 tiles with
    // 128 x 64 wave tiles issue 25 % fewer LDS-DMA and fragment reads per MFMA; 642 -> 729 TF on random operands
    // (M = 55 296), no gain or a loss for the other shapes (profiles/r03_j_gemm_tile_forms.txt)
    // ... and, in the model (two lanes), 9.07 -> 9.35 k frames/s with the SwiGLU launches, 9.28 -> 9.36 k with the QKV launches
    // as well (isolated: 605 -> 612 TF); the residual-stream GEMMs lose (bit 2).  SVC_KGEMM_TALL: bit mask, default 3.
    static const int tall = [] { const char* e = getenv("SVC_KGEMM_TALL"); return e ? atoi(e) : 3; }();
    if (sizeof(T) == 2 && kt >= 12 && p.N >= 768 && p.N % 128 == 0 && (long)cdiv(p.M, 256) * (p.N / 128) >= 512 &&   // >= 2 per CU
        ((EPI == KG_EPI_SWIGLU && (tall & 1) && p.N >= 2048) || (EPI == KG_EPI_QKV_ROPE && (tall & 2)) || (EPI == KG_EPI_STORE && (tall & 4))))
        return launch_one<T, 256, 128, 64, 3, EPI, 4>(p, st);
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
    static const int env_group = [] { const char* e = getenv("SVC_KGEMM_GROUP"); return e ? atoi(e) : 8; }();
    p.group_n = env_group;
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
