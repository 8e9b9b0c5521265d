// Shared declarations for the gfx950 kernels of the seed-vc hot path.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

namespace svc {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

void set_error(const std::string& msg);
const char* get_error();

#define SVC_CHECK_HIP(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            svc::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " at " + __FILE__ + \
                           ":" + std::to_string(__LINE__));                                  \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

#define SVC_REQUIRE(cond, msg)                                                               \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            svc::set_error(std::string("requirement failed: ") + #cond + " -- " + (msg));    \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

// sin^2(x) for the snake activations: 3-term Cody-Waite reduction by pi and a degree-9 near-minimax polynomial of sin
// on [-pi/2, pi/2] (|error| < 1.6e-7 for |x| < 60, checked in float32 emulation; the sign of sin is irrelevant once
// squared).  ~12 VALU ops instead of libm sinf's ~45: the channels-last activation kernels are VALU-issue bound.
__device__ __forceinline__ float sin_sq(float x) {
    const float n = rintf(x * 0.318309886183790672f);
    float r = fmaf(n, -3.140625f, x);
    r = fmaf(n, -9.67502593994140625e-4f, r);
    r = fmaf(n, -1.509957990978376e-7f, r);
    const float r2 = r * r;
    float p = fmaf(r2, 2.6348154733568663e-06f, -0.00019822761532850564f);
    p = fmaf(p, r2, 0.008333242498338223f);
    p = fmaf(p, r2, -0.1666666567325592f);
    const float sn = fmaf(p * r2, r, r);
    return sn * sn;
}

// ------------------------------------------------------------------ per-device lazily created state (device.hip)
// One record per HIP device, created under a mutex on the first launch with that device current: the 256-byte zero page
// that padded / out-of-range tile rows are DMA'd from, and the one-time kernel attribute setup of that device.
struct DeviceState {
    int device = -1;
    void* zero_page = nullptr;
    // handles on different threads share this state (seedvc_hip.h: different handles may run concurrently): the flags are
    // atomics; setting an attribute twice is harmless (hipFuncSetAttribute is idempotent)
    std::atomic<unsigned> kconv_attr{0};      // bit per kconv_kernel instantiation whose LDS attribute is set on this device
    std::atomic<bool> fused_attr{false};
};
DeviceState* device_state();      // state of the CURRENT device; nullptr on failure (error set)
int current_device();             // hipGetDevice, -1 on failure

// ------------------------------------------------------------------ optional launch timing (prof.hip)
// When enabled, every tap-GEMM / attention launch is bracketed by HIP events on its own stream and its
// algorithmic FLOPs / bytes are recorded per kernel class; used by bench.py for the roofline object.
enum { PROF_KGEMM_F16 = 0, PROF_KGEMM_F32 = 1, PROF_ATTN = 2, PROF_FUSED = 3, PROF_N = 4 };
bool prof_enabled();
void prof_begin(int cls, hipStream_t st);
void prof_end(int cls, double flops, double bytes, hipStream_t st, unsigned long long tag = 0);

// epilogue activations shared by the tap-GEMM and the resident-tile conv (enum KG_ACT_* below)
// s_waitcnt vmcnt(min(ahead, MAXA) * DPT): the DMAs of up to MAXA later tiles may stay outstanding (immediate operand,
// so one compare chain over the possible counts)
template <int MAXA, int DPT>
__device__ __forceinline__ void wait_tiles(int ahead) {
    static_assert(MAXA * DPT <= 63, "vmcnt is a 6-bit counter");
    if constexpr (MAXA <= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        if (ahead >= MAXA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXA * DPT) : "memory");
        else wait_tiles<MAXA - 1, DPT>(ahead);
    }
}

__device__ __forceinline__ float act_apply(float v, int act, float slope);
__device__ __forceinline__ uint4 pack8(const float* v) {
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (half_t)v[j];
    return *reinterpret_cast<uint4*>(&h);
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline long round_up(long a, long b) { return (a + b - 1) / b * b; }

// ------------------------------------------------------------------ tap-GEMM (kgemm.hip)
// C[m][n] = epilogue( sum_taps sum_k A_tap[rowmap(m, tap)][k] * W[n][koff(tap) + k] )
// A: row-major, K-contiguous, fp16 or fp32.  W: [Npad][Ktot] K-contiguous, same dtype as A.
// The M index space is (sequence, position): m = seq * Lout + pos.  A rows are addressed as
//   a_row = seq * a_seq_rows + a_off + pad(pos * a_stride + shift[tap]),  pad over [0, len(seq)).
constexpr int KG_MAX_TAPS = 48;   // 16-tap conv x 3 split-precision products

enum { KG_PAD_ZERO = 0, KG_PAD_REFLECT = 1, KG_PAD_CLAMP = 2 };
enum { KG_ACT_NONE = 0, KG_ACT_SILU = 1, KG_ACT_ELU = 2, KG_ACT_LRELU = 3, KG_ACT_TANH = 4, KG_ACT_ABS = 5,
       KG_ACT_CLAMP = 6 /* clamp to +-act_slope */, KG_ACT_SIGMOID = 7 };
enum {
    KG_EPI_STORE = 0,     // bias / per-seq rowvec / activation / gate / residual; fp32 and/or fp16 out
    KG_EPI_SWIGLU = 1,    // columns (2j, 2j+1) = (w1_j, w3_j): out[j] = silu(a) * b           (fp16 out)
    KG_EPI_TANHSIG = 2,   // columns (2j, 2j+1) = (t_j, s_j):   out[j] = tanh(a) * sigmoid(b)  (+bias/rowvec)
    KG_EPI_QKV_ROPE = 3,  // q,k: interleaved-pair RoPE (+q scale) -> fp16 ; v: transposed store
};

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    switch (act) {
        case KG_ACT_SILU: return v / (1.0f + __expf(-v));
        case KG_ACT_ELU: return v > 0.f ? v : (expm1f(v));
        case KG_ACT_LRELU: return v > 0.f ? v : v * slope;
        case KG_ACT_TANH: return tanhf(v);
        case KG_ACT_ABS: return fabsf(v);
        case KG_ACT_CLAMP: return fminf(fmaxf(v, -slope), slope);
        case KG_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        default: return v;
    }
}

struct KGemmParams {
    // A
    const void* a_ptr[KG_MAX_TAPS];
    long a_ld[KG_MAX_TAPS];     // row stride (elements)
    int a_shift[KG_MAX_TAPS];
    int a_ktiles[KG_MAX_TAPS];  // k-tiles (of 128 bytes) contributed by this tap
    int n_taps;
    int Lout, a_seq_rows, a_off, a_stride, a_len, pad_mode;
    int reflect_min;            // KG_PAD_REFLECT on a sequence shorter than this reflects inside the sequence zero-extended to
                                // this length (encodec.py pad1d: extra zeros when length <= max_pad); 0 = plain reflect
    const int* seq_len;         // optional per-sequence valid length (positions), overrides a_len
    const void* zero_page;      // filled in by kgemm_launch
    float prof_flop_scale;      // launch-timing bookkeeping: algorithmic / issued FLOPs (1/3 for split-precision taps); 0 = 1
    int debug;                  // diagnostics only: bit0 skip tile loads after the first, bit1 skip the epilogue
    int group_n;                // tile order: column tiles walked in groups of this many (0 = all of N); set by kgemm_launch
    // W
    const void* w;
    long ldw;
    // C
    int M, N;
    int c_seq_rows, c_off;
    float* c32; long ldc32;
    half_t* c16; long ldc16;
    // optional pointwise Snake applied to the stored value on its way to the fp16 planes (HiFT: the NEXT conv's
    // operand): c16 / c16_lo = hi / lo parts of v + post_ib[n] * sin^2(post_a[n] * v); c32 keeps v itself
    const float* post_a; const float* post_ib; int post_n; half_t* c16_lo;
    const float* bias;                       // [N]
    const float* rowvec; long ld_rowvec;     // [nseq][N] per-sequence additive vector (ld may be 0)
    const float* gate;   long ld_gate;       // [nseq][N] multiplicative gate applied before the residual
    const float* res;    long ldres;         // residual, addressed like C
    const float* res2;   long ldres2;        // second addend applied after the scale: (v + res) * out_scale + res2
    float out_scale;                         // 0 means 1
    int act; float act_slope;
    int post_relu;                           // ReLU applied AFTER the residual / scale / second addend (res blocks: relu(y + x))
    int vec_ok;                              // all ld % 8 == 0 && N % 8 == 0 -> 16-byte epilogue path
    // QKV_ROPE
    const float* rope;                       // [pos][32][2] cos/sin
    int rope_D;                              // model dim D (q: [0,D), k: [D,2D), v: [2D,3D))
    float q_scale;
    half_t* vt; long vt_seq_stride; long vt_ld; // V^T buffer [seq][D][vt_ld]
    int vt_mode;                                // column order the consumer wants (vt_pos); 0 is treated as 1
};

int kgemm_launch(const KGemmParams& p, int dtype /*0=f16,1=f32*/, int epi, hipStream_t st);
// 256 x 256 tiles (kgemm_big.hip), fp16 only, N % 256 == 0: tuning-harness form 0x90 of the launch (not a default)
int kgemm_big_launch(const KGemmParams& p, int epi, hipStream_t st);

// ------------------------------------------------------------------ resident-tile Conv1d (kconv.hip)
// Stride-1, zero-padded Conv1d on channels-last fp16 activations [B][Lin][cin_pad] (hi, and lo in split precision);
// weights as packed for the tap-GEMM: [Npad][k][nsub][cin_pad].  Output / epilogue fields as in KGemmParams.
struct KConvParams {
    const void* a_hi; const void* a_lo;
    const void* w; long ldw;
    const float* bias;
    int B, Lin, Lout, N, cin_pad, k, dil, pad_left, nsub;
    int c_seq_rows, c_off;
    float* c32; long ldc32;
    half_t* c16; half_t* c16_lo; long ldc16;
    const float* post_a; const float* post_ib; int post_n;
    const float* res; long ldres;
    const float* res2; long ldres2;
    float out_scale;
    int act; float act_slope;
    const void* zero_page;
    int w8_exp;        // nsub == 2 ("fp16 + fp8 corrections"): log2 of the power-of-two scale the fp8 weight bytes carry
    int c16_lo_fmt;    // format of the c16_lo plane written by the epilogue: 0 = fp16 residual, 1 = fp8 pair (lo_pair_p8)
};
bool kconv_enabled();
int kconv_launch(const KConvParams& p, hipStream_t st);
// The "lo" plane of a conv operand in the fp16 + fp8-corrections mode: per element ONE 16-bit word holding two OCP e4m3
// bytes, byte 0 = fp8(hi) (the fp16 hi value again, at 3 mantissa bits) and byte 1 = fp8(2^11 (v - hi)).  Against weight
// rows of byte pairs (fp8(2^11 s w_lo), fp8(s w_hi)) one K = 128 block-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4,
// twice the K of two fp16 MFMAs in the same cycles) adds hi * w_lo + lo * w_hi for 64 channels, scaled back by 2^-11 / s
// through the instruction's E8M0 scale operands: the two correction products of the split-precision scheme cost what ONE
// fp16 product costs.  Their relative error (2^-4) sits on terms that are 2^-11 of the result.
__device__ __forceinline__ unsigned lo_pair_p8(float hi_as_float, float lo) {
    const float a = __builtin_amdgcn_fmed3f(hi_as_float, -448.f, 448.f), b = __builtin_amdgcn_fmed3f(lo * 2048.f, -448.f, 448.f);
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
}
// packs conv weights [n0 = Cout][n1 = k][n2 = Cin] (strides s*, optional per-Cout scale) as such byte pairs; *exp_out =
// log2 of the scale chosen from the largest |w|.  Synchronises the stream (pack time only).
int kconv_pack_p8(const float* src, unsigned short* dst, int n0, int n1, int n2, long s0, long s1, long s2, long d0, long d1, long d2,
                  const float* scale, int* exp_out, hipStream_t st);

// ------------------------------------------------------------------ fused DiT row-panel kernel (fused.hip)
// Everything between two attention calls of the DiT is row-local: wo + residual, ffn-norm, w1/w3 + SwiGLU, w2 + residual,
// (UViT skip linear,) the next layer's attention-norm, QKV projection and RoPE.  One workgroup owns 128 rows (4 waves x 32
// rows, the activations live in registers as MFMA operands) and streams the layer's pre-packed weight fragments through
// an LDS ring; the residual stream crosses HBM once per layer.
struct PanelParams {
    int M;                       // rows of this launch = n_seq * Lout
    int Lout, seq_rows, row_off; // local row m -> sequence m / Lout, position row_off + m % Lout; buffer row = seq * seq_rows + position
    const half_t* wstream;       // packed fragment stream (1 KiB fragments, consumption order), see fused.hip
    int n_slots;                 // slots (D/16 fragments each) in the stream for the enabled phases
    float eps;
    // post-attention half: x += [gate_a *] ao Wo^T ; x += [gate_f *] W2 swiglu(W13 norm(x))
    int do_post;
    const half_t* ao;            // [rows][D] attention output
    float* x;                    // [rows][D] fp32 residual stream (read and written)
    const float* gate_a; const float* gate_f;            // [D] or null (v2 AdaLN-zero gates)
    const float* g_ffn; const float* w_f; const float* b_f; int add_one;
    int I;                       // FFN inner width
    half_t* c16;                 // optional fp16 copy of the layer output (UViT skip emitters)
    // UViT skip linear of the NEXT layer: x = Wskip [x | skip_in] + bskip
    int do_skip;
    const half_t* skip_in; const float* bskip;
    // pre-attention half of the NEXT layer: q, k (RoPE, q scaled) and v^T
    int do_qkv;
    const float* g_attn; const float* w_a; const float* b_a;
    const float* rope; float q_scale;
    half_t* qk; half_t* vt; long vt_seq_stride; long vt_ld; int vt_mode;   // vt_mode: see KGemmParams
    // after the last layer: final adaptive norm -> fp16 rows for the head
    int do_final;
    const float* g_fin; const float* w_fin; const float* b_fin; half_t* n16;
};
bool fused_supported(int D, int I);
// bytes of the fragment stream of one layer: [wo | mlp | skip? | qkv?]
long fused_stream_halfs(int D, int I, bool post, bool skip, bool qkv);
// packs nn.Linear weights (fp32, [out][in]) into the stream at `dst`; returns the number of halfs written (0 on error)
long fused_pack_stream(half_t* dst, int D, int I, const float* wo, const float* w1, const float* w3, const float* w2,
                       const float* wskip, const float* wqkv, hipStream_t st);
int fused_panel_launch(const PanelParams& p, int D, bool gated, hipStream_t st);

// ------------------------------------------------------------------ attention (attention.hip)
struct AttnParams {
    const half_t* q; const half_t* k; long ld_qk;   // rows = seq * seq_rows + pos ; head h at column h*64
    const half_t* vt; long vt_seq_stride; long vt_ld;  // [seq][H*64][vt_ld]
    half_t* out; long ld_out;                        // [rows][H*64]
    int n_seq, H, seq_rows, Tq;                      // queries [q_start, Tq) of every sequence are computed
    int q_start;
    const int* kv_len; int kv_len_const;             // keys [0, len) attended
    int vt_perm;                                     // column order of vt: 0 natural, 1 vt_perm_pos(), 2 vt_perm_pos16() (vt_pos)
};
// Column order of V^T inside every group of 32 keys: key 16 h + 4 f + r  ->  position 8 f + 4 h + r.  The P^T operand
// of the PV MFMA holds, in lane group f, the keys {4 f .. 4 f + 3} and {16 + 4 f .. 16 + 4 f + 3} of a 32-key step (that is
// where the S^T accumulators leave them); with this order those 8 values are contiguous, so a V^T fragment is ONE
// conflict-free ds_read_b128 instead of two ds_read_b64 (which bank-conflict: 36 % of the LDS cycles, PMC).
__host__ __device__ inline int vt_perm_pos(int pos) {
    return (pos & ~31) | (((pos >> 2) & 3) << 3) | (((pos >> 4) & 1) << 2) | (pos & 3);
}
// Mode 2 = the column order of the 32x32x16 attention kernel (attn32_kernel): inside every group of 16 keys, key
// 8 a + 4 f + r -> position 8 f + 4 a + r (bits 2 and 3 swapped).  Its S^T accumulators leave lane half f with the keys
// {4 f .. 4 f + 3} and {8 + 4 f .. 8 + 4 f + 3} of a 16-key PV step.  Both orders keep runs of 4 keys together (the
// producers store 4 positions at a time).  mode: 0 natural, 1 vt_perm_pos, 2 this one.
__host__ __device__ inline int vt_perm_pos16(int pos) {
    return (pos & ~12) | ((pos & 4) << 1) | ((pos & 8) >> 1);
}
__host__ __device__ inline int vt_pos(int pos, int mode) {
    return mode == 2 ? vt_perm_pos16(pos) : (mode == 1 ? vt_perm_pos(pos) : pos);
}
// Which V^T order the attention launch wants from its producer (the QKV epilogues): 1 = the 16x16x32 kernels (default);
// SVC_ATTN32=1 (read per call) selects 2 = the 32x32x16 kernels (measured slower, kept tested: attention.hip).
int attention_vt_mode(int n_seq, int H, int Tq);
int attention_launch(const AttnParams& p, hipStream_t st);
// in-place column permutation of a natural-order V^T buffer [rows][vt_ld] into `mode` (op-level entry point only)
int attention_permute_vt(half_t* vt, long rows, long vt_ld, int mode, hipStream_t st);

// ------------------------------------------------------------------ elementwise / norm kernels (elementwise.hip)
// y16[row] = (rmsnorm(x[row]) * gamma) * (mul_add1 + w[seq]) + b[seq]      (w/b may be null)
int rmsnorm_mod_launch(const float* x, long ldx, half_t* y, long ldy, const float* gamma,
                       const float* w, const float* b, long ld_wb, int add_one,
                       int rows, int D, int seq_rows, float eps, hipStream_t st);
// LayerNorm (no affine) * (1 + scale) + shift
int layernorm_mod_launch(const float* x, long ldx, half_t* y, long ldy, const float* scale,
                         const float* shift, int rows, int D, float eps, hipStream_t st);

}  // namespace svc
