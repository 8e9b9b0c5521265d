// 256 x 256 tiles of the tap-GEMM: four waves, one per SIMD, each owning a 128 x 128 wave tile = 64 MFMA accumulators = all
// 256 AGPRs of a lane (named from inline asm, kgemm_tile.h), software-pipelined quarter by quarter.  Per MFMA it issues half
// the ds_read_b128 fragment reads and half the LDS-DMA requests of the 128 x 128 tile with 64 x 64 wave tiles.
// NOT on the product path (tuning-harness form 0x90 of launch_wide, held bit-identical to the other forms by
// tests/test_gpu_ops.py): its main loop alone reaches 1.03-1.2 PF on random operands (0.92 PF with the ring refills), but
// with ONE workgroup per CU nothing runs under its epilogue, and the epilogues of this model family are heavy (SwiGLU /
// RoPE / fp32 residual): 707 TF on the D = 768 w1|w3 projection against 729 for the 256 x 128 form with two workgroups per
// CU, and 320-540 TF on the residual-stream GEMMs against 545-680 (profiles/r03_j_gemm_tile_forms.txt).
#include "kgemm_tile.h"

namespace svc {

namespace {

template <int EPI>
int launch_big(const KGemmParams& p, hipStream_t st) {
    // 64-byte k-tile rows, 4-stage ring (128 KB), all four stages in flight (software-pipelined loop of kgemm_tile.h)
    return launch_one<half_t, 256, 256, 64, 4, EPI, 4>(p, st);
}

}  // namespace

int kgemm_big_launch(const KGemmParams& p, int epi, hipStream_t st) {
    // whole 256-column tiles only (weights are padded to 128 rows); V blocks of a QKV launch must start at a tile edge
    SVC_REQUIRE(p.N % 256 == 0 && (epi != KG_EPI_QKV_ROPE || (2 * p.rope_D) % 256 == 0), "256 x 256 form: N % 256");
    switch (epi) {
        case KG_EPI_STORE: return launch_big<KG_EPI_STORE>(p, st);
        case KG_EPI_SWIGLU: return launch_big<KG_EPI_SWIGLU>(p, st);
        case KG_EPI_TANHSIG: return launch_big<KG_EPI_TANHSIG>(p, st);
        case KG_EPI_QKV_ROPE: return launch_big<KG_EPI_QKV_ROPE>(p, st);
    }
    set_error("kgemm_big: unsupported epilogue");
    return 1;
}

}  // namespace svc
