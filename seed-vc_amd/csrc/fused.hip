// Fused DiT row-panel kernel for gfx950: everything between two attention calls in ONE launch.
//
// Reference op chain covered (modules/diffusion_transformer.py:263-271 TransformerBlock.forward, :173-191 Attention
// projections, :30-48 AdaptiveLayerNorm, :222-235 FeedForward, :129-136 UViT skip_in_linear; v2: modules/v2/dit_model.py
// :136-143 with the AdaLN-zero gates):
//     x  = x + [gate_a *] wo(attn_out)                               P1
//     x  = x + [gate_f *] w2(silu(w1(n)) * w3(n)),  n = ffn_norm(x)  P2-P4
//     x  = skip_in_linear(cat[x, skip])                               P5   (UViT receiver layers)
//     q, k, v = wqkv(attention_norm(x)) ; RoPE(q, k)                  P6-P7 (of the NEXT layer)
// All of it is row-local, so a workgroup owns 128 rows (4 waves x 32 rows, ONE wave per SIMD with the whole 512-entry
// register file) and never exchanges activations with another workgroup: the fp32 residual rows sit in the MFMA
// accumulators (C^T form: lane = row, registers = 4 consecutive columns per 16-column tile), the fp16 GEMM operands are
// built from them in registers (an accumulator tile pair IS the next MFMA's B operand once the weights' K order is
// permuted to match, which is done at pack time), and RMSNorm row sums are two wave shuffles.  Only weights move: the
// layer's matrices are pre-packed as a linear stream of 1-KiB MFMA A-operand fragments in exactly the order the kernel
// consumes them, and streamed L2 -> LDS by LDS-DMA into a ring of D/16-fragment slots (24 / 32 KiB; 5 / 4 slots) behind
// counted vmcnt waits and one raw s_barrier per slot (48 / 64 MFMAs per wave per barrier).  Fragments are read back
// with the lane-linear ds_read_b128 (address = slot + fragment * 1 KiB + lane * 16: conflict-free by construction, no
// swizzle anywhere).  HBM / L2 traffic per layer: x once in, once out (fp32), attn_out in, q/k/v^T out, weights once per
// 128 rows (128 flop per weight byte instead of the tap-GEMM's 64).
//
// Supported widths: D = 384 (tiny) and D = 512 (small, v2); D = 768 does not fit the register file in this form and stays
// on the tap-GEMM path.  Built WITHOUT -amdgpu-mfma-vgpr-form: the accumulators must be allowed to live in AGPRs.
#include <stdlib.h>

#include <utility>

#include "common.h"

namespace svc {

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int D>
struct PG {
    static constexpr int NT = D / 16;                 // 16-column tiles of a D-wide row
    static constexpr int KC = D / 32;                 // 32-deep k chunks of a D-long reduction
    static constexpr int SLOT_BYTES = NT * 1024;      // one slot = NT fragments
    static constexpr int NS = D == 384 ? 5 : 4;       // ring slots
    static constexpr int RING = NS * SLOT_BYTES;
    static constexpr int PF = D == 384 ? 6 : 4;       // fragment reads in flight ahead of their MFMAs
    static constexpr int NVEC = 9;                    // per-column vectors kept in LDS
    static constexpr int LDS = RING + NVEC * D * 4;
};

// Walks the NF fragments of an acquired slot with the LDS reads running PF fragments ahead of their MFMAs (rotating
// register queue, statically indexed after unrolling): hipcc alone issues a read, waits lgkmcnt(0) and then computes.
// The reads are inline-asm ds_read_b128 with hand-counted waits: left to hipcc, the reads of a rotating queue get ONE
// s_waitcnt lgkmcnt(0) per PF fragments, i.e. every PF-th fragment waits for the read issued two MFMAs earlier -- a full
// LDS latency exposed per PF fragments (SQ_WAIT_ANY 24-39 % of the wave cycles).  Here fragment f waits with
// lgkmcnt(PF - 1): only for its own read, issued PF fragments earlier; the wait names the register ("+v"), so its MFMAs
// cannot be scheduled above it, and sched_barrier(0) after every fragment keeps the issue order as written.  Scalar
// loads share the counter but can only make the counted wait conservative (they complete out of order: fewer LDS reads
// than N may then be pending, never more).
__device__ __forceinline__ half8 frag_read(unsigned lds_addr, int off) {
    half8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(0) : "memory");
    (void)off;
    return r;
}
template <int OFF>
__device__ __forceinline__ void frag_read_to(half8& r, unsigned lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void frag_wait(half8& r) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(N));
}

template <int NF, int PF, int TM, int RFI, typename F, typename R>
__device__ __forceinline__ void stream_frags(const char* base, F&& body, R&& refill) {
    static_assert(NF * 1024 <= 65536 && PF >= 1 && PF <= 8, "ds_read offset field");
    const unsigned la = (unsigned)(uintptr_t)(lptr_t)const_cast<char*>(base);
    half8 q[PF];
    [&]<int... I>(std::integer_sequence<int, I...>) { (frag_read_to<I * 1024>(q[I], la), ...); }(std::make_integer_sequence<int, PF>{});
    [&]<int... Fi>(std::integer_sequence<int, Fi...>) {
        ([&] {
            constexpr int f = Fi;
            constexpr int pending = (NF - f < PF ? NF - f : PF) - 1;      // reads younger than fragment f's
            frag_wait<pending>(q[f % PF]);
            const half8 w = q[f % PF];
            body(f, w);
            if constexpr (f + PF < NF) frag_read_to<(f + PF) * 1024>(q[f % PF], la);
            if constexpr (f % RFI == RFI - 1) refill(f / RFI);            // one 1-KiB piece of the ring refill per RFI fragments
            __builtin_amdgcn_sched_barrier(0);
        }(), ...);
    }(std::make_integer_sequence<int, NF>{});
}

// D = 512 variant for the 32-register side accumulators (SwiGLU inputs, q/k/v tiles).  With the 256 residual accumulators
// filling the AGPR half of the register file, hipcc still places these MFMA results in AGPRs and shuffles 32 residual
// registers through VGPRs every trip (and serialises the LDS reads for lack of registers).  Inline-asm MFMAs with "v"
// constraints pin them in VGPRs.  Reads run one 4-fragment group ahead, fenced by sched_barrier(0).  hipcc inserts no
// wait states around inline asm: mfma_guard() pads the MFMA -> VALU read hazard before the results are consumed.
__device__ __forceinline__ void mfma_v(float4v& c, const half8 a, const half8 b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_guard(float4v (&a)[4][2]) {
    asm volatile("s_nop 15\n\ts_nop 3"
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]), "+v"(a[3][1]));
}
__device__ __forceinline__ void mfma_guard(float4v (&a)[4][1]) {}      // TM = 1 never takes the inline-asm path
template <int NF, typename F, typename R>
__device__ __forceinline__ void stream_frags_asm(const char* base, F&& body, R&& refill) {
    static_assert(NF % 4 == 0, "fragment groups of 4");
    half8 q[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[0][i] = *reinterpret_cast<const half8*>(base + i * 1024);
#pragma unroll
    for (int g = 0; g < NF / 4; ++g) {
        if (g + 1 < NF / 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) q[(g + 1) & 1][i] = *reinterpret_cast<const half8*>(base + ((g + 1) * 4 + i) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) body(g * 4 + i, q[g & 1][i]);
        refill(g);
        __builtin_amdgcn_sched_barrier(0);
    }
}

enum { V_AF = 0, V_BF = 1, V_GA = 2, V_GF = 3, V_AA = 4, V_BA = 5, V_BS = 6, V_AN = 7, V_BN = 8 };

// TM = 16-row tiles per wave: 2 -> 4 waves x 32 rows, one wave per SIMD with the whole register file; 1 -> 8 waves x 16
// rows, two waves per SIMD with 256 registers each.  With one wave per SIMD every LDS-DMA issue (~60-100 cycles each),
// every LDS read latency at the start of a slot and every barrier is time the SIMD's matrix pipe idles (measured: 45 % of
// a panel's time in MFMAs); with two, the partner wave computes meanwhile, at the price of every fragment being read from
// LDS by twice as many waves (the LDS read port then runs at the MFMA rate).
template <int D, bool GATED, int TM>
__global__ __launch_bounds__(128 / (16 * TM) * 64, 128 / (16 * TM) / 4) void dit_panel_kernel(const PanelParams p) {
    using G = PG<D>;
    constexpr int NT = G::NT, KC = G::KC, NS = G::NS, PF = G::PF;
    constexpr int NW = 128 / (16 * TM), NTHR = NW * 64, DPS = NT / NW, RFI = NT / DPS;
    constexpr bool ASM_SIDE = D == 512 && TM == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* vec = reinterpret_cast<float*>(smem + G::RING);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * 128 + wave * (16 * TM);

    // ---- per-column vectors -> LDS (plain loads; before any LDS-DMA is in flight)
    for (int c = tid; c < D; c += NTHR) {
        const float one = p.add_one ? 1.f : 0.f;
        vec[V_AF * D + c] = p.g_ffn ? p.g_ffn[c] * (p.w_f ? one + p.w_f[c] : 1.f) : 0.f;
        vec[V_BF * D + c] = p.b_f ? p.b_f[c] : 0.f;
        vec[V_GA * D + c] = p.gate_a ? p.gate_a[c] : 1.f;
        vec[V_GF * D + c] = p.gate_f ? p.gate_f[c] : 1.f;
        vec[V_AA * D + c] = p.g_attn ? p.g_attn[c] * (p.w_a ? one + p.w_a[c] : 1.f) : 0.f;
        vec[V_BA * D + c] = p.b_a ? p.b_a[c] : 0.f;
        vec[V_BS * D + c] = p.bskip ? p.bskip[c] : 0.f;
        vec[V_AN * D + c] = p.g_fin ? p.g_fin[c] * (p.w_fin ? one + p.w_fin[c] : 1.f) : 0.f;
        vec[V_BN * D + c] = p.b_fin ? p.b_fin[c] : 0.f;
    }
    __syncthreads();

    // ---- weight-fragment ring
    const char* wsrc = reinterpret_cast<const char*>(p.wstream) + wave * (G::SLOT_BYTES / NW) + lane * 16;
    int s_next = 0, stage = 0, fill = NS - 1, s_issued = 0;
    // One 1-KiB piece of a slot copy: the instruction's immediate offset advances the global and the LDS address alike,
    // so 4 pieces share one address pair and one M0 value.
    auto piece = [&](const char* src, char* dst, int i) {
        const char* s4 = src + (i >> 2) * 4096;
        char* d4 = dst + (i >> 2) * 4096;
        switch (i & 3) {
            case 0: __builtin_amdgcn_global_load_lds((gptr_t)s4, (lptr_t)d4, 16, 0, 0); break;
            case 1: __builtin_amdgcn_global_load_lds((gptr_t)s4, (lptr_t)d4, 16, 1024, 0); break;
            case 2: __builtin_amdgcn_global_load_lds((gptr_t)s4, (lptr_t)d4, 16, 2048, 0); break;
            default: __builtin_amdgcn_global_load_lds((gptr_t)s4, (lptr_t)d4, 16, 3072, 0); break;
        }
    };
    // Wait for slot s_next (own DMAs by counted vmcnt, everyone else's by the barrier) and return its LDS address.  The
    // ring slot every wave has just finished reading is refilled piecewise between the MFMAs of the acquired slot
    // (refill(i), i < DPS, issued by the fragment walkers): an LDS-DMA instruction costs ~60-100 issue cycles, and with
    // one wave per SIMD nothing else can issue meanwhile, so they are spread out instead of stacked behind the barrier.
    // Past the end of the stream the refill re-copies the last slot into a ring slot nobody reads again (branch-free
    // loop body); extra or foreign vector-memory operations only make the counted waits conservative.
    const char* rf_src = nullptr;
    char* rf_dst = nullptr;
    auto acquire = [&]() -> const char* {
        const int ahead = p.n_slots - 1 - s_next;
        if constexpr (NS == 5) {
            if (ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * DPS) : "memory");
            else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPS) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPS) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
        const int src_slot = s_issued < p.n_slots ? s_issued : p.n_slots - 1;
        rf_src = wsrc + (long)src_slot * G::SLOT_BYTES;
        rf_dst = smem + fill * G::SLOT_BYTES + wave * (G::SLOT_BYTES / NW);
        ++s_issued;
        const char* ptr = smem + stage * G::SLOT_BYTES + lane * 16;
        stage = stage + 1 == NS ? 0 : stage + 1;
        fill = fill + 1 == NS ? 0 : fill + 1;
        ++s_next;
        return ptr;
    };
    auto refill = [&](int i) { piece(rf_src, rf_dst, i); };
#define FRAG(base, idx) (*reinterpret_cast<const half8*>((base) + (idx) * 1024))
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_) {
        const int src_slot = s_ < p.n_slots ? s_ : p.n_slots - 1;
#pragma unroll
        for (int i = 0; i < DPS; ++i)
            piece(wsrc + (long)src_slot * G::SLOT_BYTES, smem + s_ * G::SLOT_BYTES + wave * (G::SLOT_BYTES / NW), i);
        ++s_issued;
    }

    // ---- rows of this wave in the C^T form: lane (fr, fq) of m-tile mt = row m0 + 16 mt + fr, columns 16 t + 4 fq + r
    long grow[TM];
    bool rok[TM];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const int m = m0 + 16 * mt + fr;
        rok[mt] = m < p.M;
        const int mm = rok[mt] ? m : p.M - 1;
        const int seq = mm / p.Lout;
        grow[mt] = (long)seq * p.seq_rows + p.row_off + (mm - seq * p.Lout);
    }

    float4v acc[NT][TM];          // the residual rows (fp32), later the MLP / skip accumulators
    half8 xn[KC][TM];             // fp16 B-operand fragments: k slot (fq, j) of chunk c <-> column 32 c + 16 (j >> 2) + 4 fq + (j & 3)

    // RMSNorm (+ adaptive modulation) of the rows held in acc -> xn;  A = gamma * (add_one + w), B = b
    auto norm_to_xn = [&](const float* va, const float* vb) {
        float rs[TM];
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            float ss = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) ss += acc[t][mt][r] * acc[t][mt][r];
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            rs[mt] = rsqrtf(ss / (float)D + p.eps);
        }
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const float4v a0 = *reinterpret_cast<const float4v*>(va + 32 * c + 4 * fq);
            const float4v a1 = *reinterpret_cast<const float4v*>(va + 32 * c + 16 + 4 * fq);
            const float4v b0 = *reinterpret_cast<const float4v*>(vb + 32 * c + 4 * fq);
            const float4v b1 = *reinterpret_cast<const float4v*>(vb + 32 * c + 16 + 4 * fq);
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                half8 h;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h[j] = (half_t)(acc[2 * c][mt][j] * rs[mt] * a0[j] + b0[j]);
                    h[4 + j] = (half_t)(acc[2 * c + 1][mt][j] * rs[mt] * a1[j] + b1[j]);
                }
                xn[c][mt] = h;
            }
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
            if (rok[mt]) {
#pragma unroll
                for (int t = 0; t < NT; ++t) *reinterpret_cast<float4v*>(p.x + grow[mt] * D + 16 * t + 4 * fq) = acc[t][mt];
            }
    };

    if (p.do_post) {
        // ---------------------------------------------------------------- P1: x1 = x + [gate_a *] ao Wo^T
        {
            half8 af[KC][TM];
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
                for (int kc = 0; kc < KC; ++kc)
                    af[kc][mt] = *reinterpret_cast<const half8*>(p.ao + grow[mt] * D + 32 * kc + 8 * fq);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t][mt] = GATED ? (float4v){0.f, 0.f, 0.f, 0.f}
                                       : *reinterpret_cast<const float4v*>(p.x + grow[mt] * D + 16 * t + 4 * fq);
            }
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const char* base = acquire();
                stream_frags<NT, PF, TM, RFI>(base, [&](int t, half8 w) {
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) acc[t][mt] = MFMA(w, af[kc][mt], acc[t][mt]);
                }, refill);
            }
        }
        if constexpr (GATED) {
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4v xi = *reinterpret_cast<const float4v*>(p.x + grow[mt] * D + 16 * t + 4 * fq);
                    const float4v ga = *reinterpret_cast<const float4v*>(vec + V_GA * D + 16 * t + 4 * fq);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][mt][r] = xi[r] + ga[r] * acc[t][mt][r];
                }
            store_x();          // x1 is re-read after the MLP (the registers are needed for the MLP accumulators)
        }
        // ---------------------------------------------------------------- P2: n = ffn_norm(x1)
        norm_to_xn(vec + V_AF * D, vec + V_BF * D);
        if constexpr (GATED) {
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t][mt] = (float4v){0.f, 0.f, 0.f, 0.f};
        }
        // ---------------------------------------------------------------- P3: acc += W2 swiglu(W13 n), 32 hidden units per trip
        const int n_chunks = p.I / 32;
        for (int c = 0; c < n_chunks; ++c) {
            float4v a1[4][TM];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) a1[t][mt] = (float4v){0.f, 0.f, 0.f, 0.f};
            {
                const char* base = acquire();
                if constexpr (ASM_SIDE) {
                    stream_frags_asm<NT>(base, [&](int f, half8 w) {
                        _Pragma("unroll") for (int mt = 0; mt < TM; ++mt) mfma_v(a1[f & 3][mt], w, xn[f >> 2][mt]);
                    }, refill);
                } else {
                    stream_frags<NT, PF, TM, RFI>(base, [&](int f, half8 w) {
                        _Pragma("unroll") for (int mt = 0; mt < TM; ++mt) a1[f & 3][mt] = MFMA(w, xn[f >> 2][mt], a1[f & 3][mt]);
                    }, refill);
                }
            }
            {
                const char* base = acquire();
                if constexpr (ASM_SIDE) {
                    stream_frags_asm<NT>(base, [&](int f, half8 w) {
                        _Pragma("unroll") for (int mt = 0; mt < TM; ++mt) mfma_v(a1[f & 3][mt], w, xn[KC / 2 + (f >> 2)][mt]);
                    }, refill);
                    mfma_guard(a1);
                } else {
                    stream_frags<NT, PF, TM, RFI>(base, [&](int f, half8 w) {
                        _Pragma("unroll") for (int mt = 0; mt < TM; ++mt) a1[f & 3][mt] = MFMA(w, xn[KC / 2 + (f >> 2)][mt], a1[f & 3][mt]);
                    }, refill);
                }
            }
            // rows (2i, 2i+1) of a tile = (w1, w3) of one hidden unit: lane-local SwiGLU; unit 8 fq + 2 t + i -> k slot j = 2 t + i
            half8 hf[TM];
            constexpr float LOG2E = 1.4426950408889634f;
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float a = a1[t][mt][2 * i], b = a1[t][mt][2 * i + 1];
                        hf[mt][2 * t + i] = (half_t)(a * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-a * LOG2E)) * b);
                    }
            {
                const char* base = acquire();
                stream_frags<NT, PF, TM, RFI>(base, [&](int t, half8 w) {
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) acc[t][mt] = MFMA(w, hf[mt], acc[t][mt]);
                }, refill);
            }
        }
        if constexpr (GATED) {
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4v xi = *reinterpret_cast<const float4v*>(p.x + grow[mt] * D + 16 * t + 4 * fq);
                    const float4v gf = *reinterpret_cast<const float4v*>(vec + V_GF * D + 16 * t + 4 * fq);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][mt][r] = xi[r] + gf[r] * acc[t][mt][r];
                }
        }
        // ---------------------------------------------------------------- P4: layer output
        if (!p.do_skip) store_x();
        if (p.c16) {
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
                if (rok[mt]) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const half4 h = {(half_t)acc[t][mt][0], (half_t)acc[t][mt][1], (half_t)acc[t][mt][2], (half_t)acc[t][mt][3]};
                        *reinterpret_cast<half4*>(p.c16 + grow[mt] * D + 16 * t + 4 * fq) = h;
                    }
                }
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][mt] = *reinterpret_cast<const float4v*>(p.x + grow[mt] * D + 16 * t + 4 * fq);
    }

    if (p.do_skip) {
        // ---------------------------------------------------------------- P5: x = Wskip [x | skip] + b (UViT receiver)
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                half8 h;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h[j] = (half_t)acc[2 * c][mt][j];
                    h[4 + j] = (half_t)acc[2 * c + 1][mt][j];
                }
                xn[c][mt] = h;
            }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) acc[t][mt] = *reinterpret_cast<const float4v*>(vec + V_BS * D + 16 * t + 4 * fq);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const char* base = acquire();
            stream_frags<NT, PF, TM, RFI>(base, [&](int t, half8 w) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) acc[t][mt] = MFMA(w, xn[kc][mt], acc[t][mt]);
            }, refill);
        }
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
                xn[kc][mt] = *reinterpret_cast<const half8*>(p.skip_in + grow[mt] * D + 32 * kc + 8 * fq);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const char* base = acquire();
            stream_frags<NT, PF, TM, RFI>(base, [&](int t, half8 w) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) acc[t][mt] = MFMA(w, xn[kc][mt], acc[t][mt]);
            }, refill);
        }
        store_x();
    }

    if (p.do_final) {
        // ---------------------------------------------------------------- final adaptive norm -> fp16 rows for the head
        float rs[TM];
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            float ss = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) ss += acc[t][mt][r] * acc[t][mt][r];
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            rs[mt] = rsqrtf(ss / (float)D + p.eps);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4v a = *reinterpret_cast<const float4v*>(vec + V_AN * D + 16 * t + 4 * fq);
            const float4v b = *reinterpret_cast<const float4v*>(vec + V_BN * D + 16 * t + 4 * fq);
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
                if (rok[mt]) {
                    half4 h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = (half_t)(acc[t][mt][r] * rs[mt] * a[r] + b[r]);
                    *reinterpret_cast<half4*>(p.n16 + grow[mt] * D + 16 * t + 4 * fq) = h;
                }
        }
    }

    if (p.do_qkv) {
        // ---------------------------------------------------------------- P6: n = attention_norm(x) of the next layer
        norm_to_xn(vec + V_AA * D, vec + V_BA * D);
        // ---------------------------------------------------------------- P7: q, k (RoPE) and v^T, 64 columns (one head) per trip
        float rp[TM][16];          // (cos, sin) of pairs 8 fq .. 8 fq + 7 at this lane's positions
        int posl[TM];
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int m = m0 + 16 * mt + fr;
            const int mm = m < p.M ? m : p.M - 1;
            posl[mt] = p.row_off + mm % p.Lout;
            const float* tb = p.rope + ((long)posl[mt] * 32 + 8 * fq) * 2;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4v t4 = *reinterpret_cast<const float4v*>(tb + 4 * q4);
#pragma unroll
                for (int j = 0; j < 4; ++j) rp[mt][4 * q4 + j] = t4[j];
            }
        }
        const int n_groups = 3 * D / 64, n_qk = 2 * D / 64;
        for (int grp = 0; grp < n_groups; ++grp) {
            float4v a1[4][TM];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) a1[t][mt] = (float4v){0.f, 0.f, 0.f, 0.f};
            if (grp < n_qk) {
#pragma unroll
                for (int hs = 0; hs < 2; ++hs) {
                    const char* base = acquire();
                    if constexpr (ASM_SIDE) {
                        stream_frags_asm<NT>(base, [&](int f, half8 w) {
                            _Pragma("unroll") for (int mt = 0; mt < TM; ++mt) mfma_v(a1[f & 3][mt], w, xn[hs * (KC / 2) + (f >> 2)][mt]);
                        }, refill);
                        if (hs == 1) mfma_guard(a1);
                    } else {
                        stream_frags<NT, PF, TM, RFI>(base, [&](int f, half8 w) {
                            _Pragma("unroll") for (int mt = 0; mt < TM; ++mt) a1[f & 3][mt] = MFMA(w, xn[hs * (KC / 2) + (f >> 2)][mt], a1[f & 3][mt]);
                        }, refill);
                    }
                }
                // lane (row fr, fq): columns 64 grp + 16 fq + 4 t + r  (weight rows permuted at pack time)
                const float sc = grp < D / 64 ? p.q_scale : 1.0f;
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    float o[16];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float x0 = a1[i >> 1][mt][2 * (i & 1)], x1 = a1[i >> 1][mt][2 * (i & 1) + 1];
                        const float cs = rp[mt][2 * i], sn = rp[mt][2 * i + 1];
                        o[2 * i] = (x0 * cs - x1 * sn) * sc;
                        o[2 * i + 1] = (x1 * cs + x0 * sn) * sc;
                    }
                    if (rok[mt]) {
                        half_t* dst = p.qk + grow[mt] * (2 * D) + 64 * grp + 16 * fq;
                        *reinterpret_cast<uint4*>(dst) = pack8(o);
                        *reinterpret_cast<uint4*>(dst + 8) = pack8(o + 8);
                    }
                }
            } else {
                // v keeps the C orientation (activations as the A operand): lane = column 16 t + fr, registers = 4 consecutive rows
#pragma unroll
                for (int hs = 0; hs < 2; ++hs) {
                    const char* base = acquire();
                    if constexpr (ASM_SIDE) {
                        stream_frags_asm<NT>(base, [&](int f, half8 w) {
                            _Pragma("unroll") for (int mt = 0; mt < TM; ++mt) mfma_v(a1[f & 3][mt], xn[hs * (KC / 2) + (f >> 2)][mt], w);
                        }, refill);
                        if (hs == 1) mfma_guard(a1);
                    } else {
                        stream_frags<NT, PF, TM, RFI>(base, [&](int f, half8 w) {
                            _Pragma("unroll") for (int mt = 0; mt < TM; ++mt) a1[f & 3][mt] = MFMA(xn[hs * (KC / 2) + (f >> 2)][mt], w, a1[f & 3][mt]);
                        }, refill);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    const int m = m0 + 16 * mt + 4 * fq;
                    if (m < p.M) {
                        const int seq = m / p.Lout;
                        const int pos = p.row_off + (m - seq * p.Lout);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int d = 64 * (grp - n_qk) + 16 * t + fr;
                            const half4 h = {(half_t)a1[t][mt][0], (half_t)a1[t][mt][1], (half_t)a1[t][mt][2], (half_t)a1[t][mt][3]};
                            *reinterpret_cast<half4*>(p.vt + (long)seq * p.vt_seq_stride + (long)d * p.vt_ld + vt_pos(pos, p.vt_mode ? p.vt_mode : 1)) = h;
                        }
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef FRAG
#undef MFMA
}

// ---------------------------------------------------------------------------------------------- fragment packing
// One thread per fp16 element of the stream.  mode: 0 wo, 1 mlp, 2 skip, 3 qkv (see the fragment definitions in DESIGN.md 5)
__global__ void pack_stream_kernel(half_t* __restrict__ dst, long n_elems, int mode, int D, int I, const float* __restrict__ w0,
                                   const float* __restrict__ w1, const float* __restrict__ w2) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elems) return;
    const int NT = D / 16, KC = D / 32;
    const long f = e >> 9;                 // fragment
    const int l = (int)((e >> 3) & 63), j = (int)(e & 7);
    const int fr = l & 15, g = l >> 4;
    const int kperm = 16 * (j >> 2) + 4 * g + (j & 3);      // k slot (g, j) of an operand built from C^T accumulators
    const int knat = 8 * g + j;
    float v = 0.f;
    if (mode == 0) {                       // wo: f = kc * NT + t
        const int kc = (int)(f / NT), t = (int)(f % NT);
        v = w0[(long)(16 * t + fr) * D + 32 * kc + knat];
    } else if (mode == 1) {                // mlp: per chunk [S1: kc * 4 + t (2 slots)] [S2: t (1 slot)]
        const long per = 3L * NT;
        const int c = (int)(f / per), q = (int)(f % per);
        if (q < 2 * NT) {
            const int kc = q / 4, t = q % 4;
            const int h = 32 * c + 8 * (fr >> 2) + 2 * t + ((fr & 3) >> 1);
            const float* src = (fr & 1) ? w1 : w0;          // w0 = w1 weights, w1 = w3 weights
            v = src[(long)h * D + 32 * kc + kperm];
        } else {
            const int t = q - 2 * NT;
            v = w2[(long)(16 * t + fr) * I + 32 * c + knat];
        }
    } else if (mode == 2) {                // skip: f = kc * NT + t, kc < 2 KC ; weight [D][2 D]
        const int kc = (int)(f / NT), t = (int)(f % NT);
        const int k = kc < KC ? 32 * kc + kperm : D + 32 * (kc - KC) + knat;
        v = w0[(long)(16 * t + fr) * (2 * D) + k];
    } else {                               // qkv: f = (grp * KC + kc) * 4 + t ; weight [3 D][D]
        const int grp = (int)(f / (4 * KC)), q = (int)(f % (4 * KC));
        const int kc = q / 4, t = q % 4;
        const int n = grp < 2 * D / 64 ? 64 * grp + 16 * (fr >> 2) + 4 * t + (fr & 3) : 64 * grp + 16 * t + fr;
        v = w0[(long)n * D + 32 * kc + kperm];
    }
    dst[e] = (half_t)v;
}

}  // namespace

bool fused_supported(int D, int I) {
    static const bool off = [] { const char* e = getenv("SVC_FUSED"); return e && e[0] == '0'; }();
    return !off && (D == 384 || D == 512) && I % 32 == 0;
}

static long slots_post(int D, int I) { return D / 32 + 3L * (I / 32); }
static long slots_skip(int D) { return 2L * (D / 32); }
static long slots_qkv(int D) { return 2L * (3 * D / 64); }

long fused_stream_halfs(int D, int I, bool post, bool skip, bool qkv) {
    const long slot_halfs = (long)(D / 16) * 512;
    return ((post ? slots_post(D, I) : 0) + (skip ? slots_skip(D) : 0) + (qkv ? slots_qkv(D) : 0)) * slot_halfs;
}

long fused_pack_stream(half_t* dst, int D, int I, const float* wo, const float* w1, const float* w3, const float* w2,
                       const float* wskip, const float* wqkv, hipStream_t st) {
    const long slot_halfs = (long)(D / 16) * 512;
    long off = 0;
    auto run = [&](int mode, long slots, const float* a, const float* b, const float* c) -> bool {
        const long n = slots * slot_halfs;
        hipLaunchKernelGGL(pack_stream_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, dst + off, n, mode, D, I, a, b, c);
        off += n;
        return hipGetLastError() == hipSuccess;
    };
    if (wo) {
        if (!run(0, D / 32, wo, nullptr, nullptr)) return 0;
        if (!run(1, 3L * (I / 32), w1, w3, w2)) return 0;
    }
    if (wskip && !run(2, slots_skip(D), wskip, nullptr, nullptr)) return 0;
    if (wqkv && !run(3, slots_qkv(D), wqkv, nullptr, nullptr)) return 0;
    return off;
}

int fused_panel_launch(const PanelParams& p, int D, bool gated, hipStream_t st) {
    SVC_REQUIRE(D == 384 || D == 512, "fused panel kernel: D must be 384 or 512");
    SVC_REQUIRE(p.M > 0 && p.Lout > 0 && p.I % 32 == 0, "fused panel kernel: shape");
    SVC_REQUIRE(!p.do_qkv || (p.M % 4 == 0 && p.Lout % 4 == 0 && p.row_off % 4 == 0), "fused panel kernel: v^T stores need 4-row groups");
    DeviceState* ds = device_state();
    if (!ds) return 1;
    if (!ds->fused_attr.load(std::memory_order_acquire)) {
#define SVC_PANEL_ATTR(DD, GG, TT) SVC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dit_panel_kernel<DD, GG, TT>), hipFuncAttributeMaxDynamicSharedMemorySize, PG<DD>::LDS))
        SVC_PANEL_ATTR(384, false, 1); SVC_PANEL_ATTR(384, true, 1); SVC_PANEL_ATTR(512, false, 1); SVC_PANEL_ATTR(512, true, 1);
        SVC_PANEL_ATTR(384, false, 2); SVC_PANEL_ATTR(384, true, 2); SVC_PANEL_ATTR(512, false, 2); SVC_PANEL_ATTR(512, true, 2);
#undef SVC_PANEL_ATTR
        ds->fused_attr.store(true, std::memory_order_release);
    }
    const long want = (p.do_post ? slots_post(D, p.I) : 0) + (p.do_skip ? slots_skip(D) : 0) + (p.do_qkv ? slots_qkv(D) : 0);
    SVC_REQUIRE(want == p.n_slots, "fused panel kernel: slot count does not match the enabled phases");
    const int grid = cdiv(p.M, 128);
    const bool prof = prof_enabled();
    if (prof) prof_begin(PROF_FUSED, st);
    // waves per SIMD, measured on MI355X: D = 384 is fastest with one wave per SIMD owning 32 rows
    // (811 vs 806 TFLOP/s), D = 512 with two waves per SIMD owning 16 rows each (845 vs 751: with 32 rows its 256
    // residual accumulators leave no room and the side accumulators need the inline-asm path).  SVC_FUSED_TM overrides.
    static const int tm_env = [] { const char* e = getenv("SVC_FUSED_TM"); return e ? atoi(e) : 0; }();
    const int tm = (tm_env == 1 || tm_env == 2) ? tm_env : (D == 384 ? 2 : 1);
#define SVC_PANEL_LAUNCH(DD, GG, TT) hipLaunchKernelGGL((dit_panel_kernel<DD, GG, TT>), dim3(grid), dim3(128 / (16 * TT) * 64), PG<DD>::LDS, st, p)
    if (tm == 2) {
        if (D == 384) { if (gated) SVC_PANEL_LAUNCH(384, true, 2); else SVC_PANEL_LAUNCH(384, false, 2); }
        else { if (gated) SVC_PANEL_LAUNCH(512, true, 2); else SVC_PANEL_LAUNCH(512, false, 2); }
    } else {
        if (D == 384) { if (gated) SVC_PANEL_LAUNCH(384, true, 1); else SVC_PANEL_LAUNCH(384, false, 1); }
        else { if (gated) SVC_PANEL_LAUNCH(512, true, 1); else SVC_PANEL_LAUNCH(512, false, 1); }
    }
#undef SVC_PANEL_LAUNCH
    SVC_CHECK_HIP(hipGetLastError());
    if (prof) {
        const double M = p.M, Dd = D;
        double macs = 0;      // per row
        if (p.do_post) macs += Dd * Dd + 3.0 * Dd * p.I;
        if (p.do_skip) macs += 2.0 * Dd * Dd;
        if (p.do_qkv) macs += 3.0 * Dd * Dd;
        // algorithmic bytes: x in/out fp32, attn_out in, q/k/v out, skip in, fp16 copies out, weights once
        double bytes = 2.0 * macs;
        bytes += M * Dd * ((p.do_post ? 4 + 2 : 4) + ((p.do_post && !p.do_skip) || p.do_skip ? 4 : 0) + (p.c16 ? 2 : 0) +
                           (p.do_skip ? 2 : 0) + (p.do_qkv ? 6 : 0) + (p.do_final ? 2 : 0));
        const unsigned long long tag = ((unsigned long long)p.M << 40) | ((unsigned long long)(D & 0xFFFFF) << 20) |
                                       ((unsigned long long)(p.I & 0xFFFF) << 4) |
                                       (unsigned)((p.do_post ? 1 : 0) | (p.do_skip ? 2 : 0) | (p.do_qkv ? 4 : 0) | (p.do_final ? 8 : 0));
        prof_end(PROF_FUSED, 2.0 * M * macs, bytes, st, tag);
    }
    return 0;
}

}  // namespace svc
