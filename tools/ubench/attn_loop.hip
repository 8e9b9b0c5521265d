// Micro-benchmark 3: the DiT attention kernel (csrc/attention.hip, 128-query form) with parts switched off, to see
// what bounds it.  ABL bits: 1 no exp2 (a multiply instead), 2 no QK^T MFMAs, 4 no PV MFMAs, 8 no global loads / LDS
// stores / barrier (computes on one resident tile), 16 no baseline moves.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -o attn_loop attn_loop.hip && ./attn_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct AttnParams {
    const half_t* q; const half_t* k; long ld_qk;
    const half_t* vt; long vt_seq_stride; long vt_ld;
    half_t* out; long ld_out;
    int n_seq, H, seq_rows, Tq;
    int q_start;
    const int* kv_len; int kv_len_const;
};
namespace {
constexpr int KT = 64;        // keys per tile
constexpr int ROWB = 128;

__device__ __forceinline__ int lds_off(int row, int c16) { return row * ROWB + ((c16 ^ ((row >> 1) & 7)) << 4); }

template <int QT, int ABL>
__global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) {
    constexpr int BQ = 64 * QT;   // queries per block
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * KT * ROWB];   // [buf][K | Vt][64][128B]
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

    // staging: 512 chunks per 64x128B tile -> 2 per thread, for K and for Vt; pointers advance by one tile
    const int sc = tid & 7, sr = tid >> 3;      // rows sr, sr + 32
    const half_t* kp[2];
    const half_t* vp[2];
    int krow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        krow[i] = sr + 32 * i;
        kp[i] = p.k + (row_base + krow[i]) * p.ld_qk + h * 64 + sc * 8;
        vp[i] = p.vt + (long)seq * p.vt_seq_stride + (long)(h * 64 + sr + 32 * i) * p.vt_ld + sc * 8;
    }
    const long k_step = (long)KT * p.ld_qk;
    // Register staging.  The 64-query form (small grids: one or two blocks per CU, every key tile a dependent L2 /
    // Infinity-Cache round trip) keeps TWO tiles in flight in alternating register sets; the 128-query form has enough
    // blocks per CU to cover one tile's latency and keeps its registers for occupancy.
    constexpr int PD = QT == 1 ? 2 : 1;          // prefetch distance in key tiles
    u32x4 rk[PD][2], rv[PD][2];
    auto load_tile = [&](u32x4 (&rk)[2], u32x4 (&rv)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            rk[i] = krow[i] < p.seq_rows ? *reinterpret_cast<const u32x4*>(kp[i]) : z;
            rv[i] = *reinterpret_cast<const u32x4*>(vp[i]);
            kp[i] += k_step;
            vp[i] += KT;
            krow[i] += KT;
        }
    };
    auto store_tile = [&](int buf, const u32x4 (&rk)[2], const u32x4 (&rv)[2]) {
        char* kb = smem + buf * 2 * KT * ROWB;
        char* vb = kb + KT * ROWB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4*>(kb + lds_off(sr + 32 * i, sc)) = rk[i];
            *reinterpret_cast<u32x4*>(vb + lds_off(sr + 32 * i, sc)) = rv[i];
        }
    };

    if (n_kt > 0) {
        load_tile(rk[0], rv[0]);
        store_tile(0, rk[0], rv[0]);
        if (PD == 2 && n_kt > 1) load_tile(rk[PD - 1], rv[PD - 1]);      // tile 1 in flight in set 1
    }
    __syncthreads();

    // one key tile: `ld` = register set the tile kt + PD is requested into, `st` = set holding tile kt + 1
    auto step = [&](int kt, u32x4 (&ldk)[2], u32x4 (&ldv)[2], const u32x4 (&stk)[2], const u32x4 (&stv)[2]) {
        const int buf = kt & 1;
        const bool more = kt + 1 < n_kt;
        if (kt + PD < n_kt && !(ABL & 8)) load_tile(ldk, ldv);
        const char* kb = smem + buf * 2 * KT * ROWB;
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
                    if (!(ABL & 2)) acc_s[mt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qt][ks], acc_s[mt][qt], 0, 0, 0); else acc_s[mt][qt][0] += (float)kf[0];
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
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) acc_s[mt][qt][r] = -1e30f;
                    }
                }
        }
        // ---- tile maxima relative to the baseline
        float mx[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float a = -1e30f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                a = fmaxf(a, fmaxf(fmaxf(acc_s[mt][qt][0], acc_s[mt][qt][1]), fmaxf(acc_s[mt][qt][2], acc_s[mt][qt][3])));
            a = fmaxf(a, __shfl_xor(a, 16));
            a = fmaxf(a, __shfl_xor(a, 32));
            mx[qt] = a;
        }
        // ---- baseline move (always on the first tile; afterwards only when a row of the 16-query tile overshoots by
        // more than THR -- decided per tile, so a query's arithmetic does not depend on which tiles share its wave)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            if (!(ABL & 16) && (kt == 0 || __any(mx[qt] > THR))) {
                const float delta = kt == 0 ? mx[qt] : fmaxf(mx[qt], 0.f);
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
                    const float e0 = (ABL & 1) ? sv[0] * 0.01f : __builtin_amdgcn_exp2f(sv[0]), e1 = (ABL & 1) ? sv[1] * 0.01f : __builtin_amdgcn_exp2f(sv[1]);
                    const float e2 = (ABL & 1) ? sv[2] * 0.01f : __builtin_amdgcn_exp2f(sv[2]), e3 = (ABL & 1) ? sv[3] * 0.01f : __builtin_amdgcn_exp2f(sv[3]);
                    u[2 * hh] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(e0, e1));
                    u[2 * hh + 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(e2, e3));
                }
                pf[qt] = __builtin_bit_cast(half8, u);
                acc_l[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones_f, pf[qt], acc_l[qt], 0, 0, 0);
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
                for (int qt = 0; qt < QT; ++qt)
                    if (!(ABL & 4)) acc_o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qt], acc_o[dt][qt], 0, 0, 0); else acc_o[dt][qt][0] += (float)vf[0] * (float)pf[qt][0];
            }
        }
        if (more && !(ABL & 8)) store_tile(buf ^ 1, stk, stv);
        if (!(ABL & 8)) __syncthreads();
    };
    if constexpr (PD == 1) {
        for (int kt = 0; kt < n_kt; ++kt) step(kt, rk[0], rv[0], rk[0], rv[0]);
    } else {
        for (int kt = 0; kt < n_kt; kt += 2) {
            step(kt, rk[0], rv[0], rk[1], rv[1]);                         // set 0 (tile kt, stored) is free; set 1 holds kt + 1
            if (kt + 1 < n_kt) step(kt + 1, rk[1], rv[1], rk[0], rv[0]);
        }
    }

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

template <int ABL>
void run(const AttnParams& p, const char* name) {
    const int grid = ((p.Tq + 127) / 128) * p.H * p.n_seq;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((attn_kernel<2, ABL>), dim3(grid), dim3(256), 0, 0, p);
    hipEventRecord(e0, 0);
    const int it = 20;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((attn_kernel<2, ABL>), dim3(grid), dim3(256), 0, 0, p);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= it;
    const double fl = 4.0 * p.n_seq * p.H * (double)p.Tq * p.seq_rows * 64.0;
    printf("%-44s %8.1f us  %7.1f TF\n", name, ms * 1e3, fl / ms / 1e9);
}

int main() {
    const int N = 64, H = 6, T = 864, D = H * 64, vt_ld = 896;
    std::vector<half_t> hq((size_t)N * T * 2 * D), hv((size_t)N * D * vt_ld);
    srand(1);
    for (auto& x : hq) x = (half_t)((rand() % 2001 - 1000) * 0.0012f);
    for (auto& x : hv) x = (half_t)((rand() % 2001 - 1000) * 0.001f);
    half_t *qk, *vt, *out;
    hipMalloc(&qk, hq.size() * 2); hipMalloc(&vt, hv.size() * 2); hipMalloc(&out, (size_t)N * T * D * 2);
    hipMemcpy(qk, hq.data(), hq.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(vt, hv.data(), hv.size() * 2, hipMemcpyHostToDevice);
    AttnParams p{};
    p.q = qk; p.k = qk + D; p.ld_qk = 2 * D; p.vt = vt; p.vt_seq_stride = (long)D * vt_ld; p.vt_ld = vt_ld;
    p.out = out; p.ld_out = D; p.n_seq = N; p.H = H; p.seq_rows = T; p.Tq = T; p.q_start = 0; p.kv_len = nullptr; p.kv_len_const = T;
    run<0>(p, "full kernel");
    run<1>(p, "no exp2");
    run<16>(p, "no baseline moves");
    run<17>(p, "no exp2, no baseline moves");
    run<8>(p, "no loads / LDS stores / barrier");
    run<9>(p, "no loads, no exp2");
    run<6>(p, "no MFMAs (VALU + staging only)");
    run<14>(p, "no MFMAs, no loads (VALU only)");
    run<31>(p, "nothing but the loop skeleton");
    run<25>(p, "MFMAs + max/cvt only (no exp, loads, moves)");
    return 0;
}
