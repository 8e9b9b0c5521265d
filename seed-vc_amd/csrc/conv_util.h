// Conv1d / ConvTranspose1d on the tap-GEMM: weight packing and launch helpers shared by the vocoders and the
// length regulator (channels-last activations [B][L][Cpad]).
#pragma once
#include <stdlib.h>
#include <string.h>

#include <string>

#include "model_util.h"

namespace svc {

struct ConvW {
    void* w = nullptr;
    float* bias = nullptr;
    long ldw = 0;
    int k = 0, cin_pad = 0, cout = 0, cout_pad = 0, dtype = 0, vd = 0;
    // vd == 3 and a shape the resident-tile conv kernel takes: second packing [w_hi | fp8 byte pairs] per tap (kconv.hip)
    void* w8 = nullptr;
    long ldw8 = 0;
    int w8_exp = 0;
    // ConvTranspose only
    int stride = 0;
};

// vocoder operand mode `vd`: 0 = fp16, 1 = fp32, 2 = fp16x3 (split hi/lo planes on the fp16 MFMA), 3 = fp16 + fp8 corrections:
// the convs the resident-tile kernel takes compute hi*hi on the fp16 MFMA and both correction products in one block-scaled
// fp8 MFMA (kconv.hip); every other conv of the model runs as in mode 2
inline int gdt(int vd) { return vd == 1 ? 1 : 0; }          // tap-GEMM operand dtype
inline bool is_split(int vd) { return vd == 2 || vd == 3; }
inline int cpad(int c, int vd) { return (int)round_up(c, ktile_elems(gdt(vd))); }
inline size_t vesize(int vd) { return esize(gdt(vd)); }

struct ActBuf {   // conv operand: one plane, or hi + lo planes in split mode
    const void* hi = nullptr;
    const void* lo = nullptr;
};

// Conv1d weight [Cout][Cin][k] (optionally weight-normed) -> [Npad][k * Cin_pad]
inline int pack_conv1d(const StateDict& sd, const std::string& prefix, int cout, int cin, int k, bool has_bias, int dtype,
                Arena& ar, hipStream_t st, ConvW* out) {
    WeightSrc ws;
    if (resolve_weight(sd, prefix, ar, st, &ws)) return 1;
    if (require_shape(ws.desc, prefix + ".weight", {cout, cin, k})) return 1;
    const int vd = dtype;
    const int nsub = is_split(vd) ? 3 : 1;
    out->vd = vd;
    out->dtype = gdt(vd);
    out->k = k;
    out->cin_pad = cpad(cin, vd);
    out->cout = cout;
    out->cout_pad = cpad(cout, vd);
    out->ldw = (long)k * nsub * out->cin_pad;
    out->w = ar.alloc((size_t)round_up(out->cout_pad, 128) * out->ldw * vesize(vd), st);
    out->bias = ar.alloc_n<float>(round_up(out->cout_pad, 8), st);
    if (!out->w || !out->bias) return 1;
    if (!is_split(vd)) {
        if (pack_any(out->dtype, ws.v, out->w, 0, cout, k, cin, (long)cin * k, 1, k, out->ldw, out->cin_pad, 1, ws.scale, st)) return 1;
    } else {
        // per tap: [w_hi | w_lo | w_hi] against the operand sub-taps [a_hi | a_hi | a_lo]
        half_t* w16 = reinterpret_cast<half_t*>(out->w);
        const long tap_ld = 3L * out->cin_pad;
        for (int sub = 0; sub < 3; ++sub) {
            half_t* dst = w16 + (long)sub * out->cin_pad;
            if (sub == 1) {
                if (pack_f16_lo_launch(ws.v, dst, cout, k, cin, (long)cin * k, 1, k, out->ldw, tap_ld, 1, ws.scale, st)) return 1;
            } else {
                if (pack_f16_launch(ws.v, dst, cout, k, cin, (long)cin * k, 1, k, out->ldw, tap_ld, 1, ws.scale, st)) return 1;
            }
        }
    }
    if (vd == 3 && k >= 3 && out->cin_pad >= 64 && out->cout_pad >= 64) {
        // [w_hi | byte pairs] per tap; a 64-channel chunk is 128 bytes in both halves (fp16 / two fp8 bytes per channel)
        out->ldw8 = (long)k * 2 * out->cin_pad;
        out->w8 = ar.alloc((size_t)round_up(out->cout_pad, 128) * out->ldw8 * 2, st);
        if (!out->w8) return 1;
        half_t* w16 = reinterpret_cast<half_t*>(out->w8);
        const long tap_ld = 2L * out->cin_pad;
        if (pack_f16_launch(ws.v, w16, cout, k, cin, (long)cin * k, 1, k, out->ldw8, tap_ld, 1, ws.scale, st)) return 1;
        if (kconv_pack_p8(ws.v, reinterpret_cast<unsigned short*>(w16 + out->cin_pad), cout, k, cin, (long)cin * k, 1, k, out->ldw8, tap_ld, 1,
                          ws.scale, &out->w8_exp, st)) return 1;
    }
    if (has_bias) {
        const auto* b = sd.get(prefix + ".bias");
        if (require_shape(b, prefix + ".bias", {cout})) return 1;
        SVC_CHECK_HIP(hipMemcpyAsync(out->bias, b->data, cout * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    return 0;
}

// ConvTranspose1d weight [Cin][Cout][k], k = 2 s, padding s/2 -> [s * Cout_pad][3 * Cin_pad]
inline int pack_convT(const StateDict& sd, const std::string& prefix, int cin, int cout, int k, int s, int dtype, Arena& ar,
               hipStream_t st, ConvW* out) {
    if (k != 2 * s || (s % 2) != 0) {
        set_error("ConvTranspose1d: only kernel = 2*stride, padding = stride/2 is supported (" + prefix + ")");
        return 1;
    }
    WeightSrc ws;
    if (resolve_weight(sd, prefix, ar, st, &ws)) return 1;
    if (require_shape(ws.desc, prefix + ".weight", {cin, cout, k})) return 1;
    const auto* b = sd.get(prefix + ".bias");
    if (require_shape(b, prefix + ".bias", {cout})) return 1;
    const int vd = dtype;
    const int nsub = is_split(vd) ? 3 : 1;
    out->vd = vd;
    out->dtype = gdt(vd);
    out->k = k;
    out->stride = s;
    out->cin_pad = cpad(cin, vd);
    out->cout = cout;
    out->cout_pad = cpad(cout, vd);
    out->ldw = 3L * nsub * out->cin_pad;
    const long N = (long)s * out->cout_pad;
    out->w = ar.alloc((size_t)round_up(N, 128) * out->ldw * vesize(vd), st);
    out->bias = ar.alloc_n<float>(N, st);
    if (!out->w || !out->bias) return 1;
    const int p = s / 2;
    for (int r = 0; r < s; ++r) {
        for (int j = 0; j < 3; ++j) {
            const int kk = r + p + s - s * j;       // y[s q + r] += x[q - 1 + j] * w[kk]
            if (kk < 0 || kk >= k) continue;
            // index space (ci, co): src [ci][co][kk], dst row r*Cout_pad + co, column (j * nsub + sub) * Cin_pad + ci
            for (int sub = 0; sub < nsub; ++sub) {
                const long off = (long)r * out->cout_pad * out->ldw + ((long)j * nsub + sub) * out->cin_pad;
                if (!is_split(vd)) {
                    if (pack_any(out->dtype, ws.v + kk, out->w, off, cin, cout, 1, (long)cout * k, k, 0, 1, out->ldw, 0, ws.scale, st)) return 1;
                } else if (sub == 1) {
                    if (pack_f16_lo_launch(ws.v + kk, reinterpret_cast<half_t*>(out->w) + off, cin, cout, 1, (long)cout * k, k, 0, 1,
                                           out->ldw, 0, ws.scale, st)) return 1;
                } else {
                    if (pack_f16_launch(ws.v + kk, reinterpret_cast<half_t*>(out->w) + off, cin, cout, 1, (long)cout * k, k, 0, 1,
                                        out->ldw, 0, ws.scale, st)) return 1;
                }
            }
        }
        SVC_CHECK_HIP(hipMemcpyAsync(out->bias + (long)r * out->cout_pad, b->data, cout * sizeof(float),
                                     hipMemcpyDeviceToDevice, st));
    }
    return 0;
}

struct ConvRun {
    ActBuf a;                  // [B][Lin][cin_pad] (hi / lo planes in split mode)
    int B = 0, Lin = 0, Lout = 0;
    int dilation = 1, stride = 1, pad_left = 0, pad_mode = KG_PAD_ZERO;
    float* c32 = nullptr; long ldc32 = 0;
    half_t* c16 = nullptr; long ldc16 = 0;
    int c_rows = 0, c_off = 0;            // output rows per sequence / row offset (0 -> Lout, 0)
    const float* res = nullptr; long ldres = 0;
    const float* res2 = nullptr; long ldres2 = 0;
    float out_scale = 0.f;
    int act = KG_ACT_NONE; float act_slope = 0.f;
    int n_override = 0;                   // write only the first n columns (e.g. 1-channel output)
    const int* seq_len = nullptr;         // device [B]: valid input rows per sequence (ragged batches), else Lin
    // fused pointwise Snake towards the next conv's operand planes (c16 = hi, c16_lo = lo); c32 keeps the raw value
    const float* post_a = nullptr; const float* post_ib = nullptr; int post_n = 0; half_t* c16_lo = nullptr;
    int c16_lo_fmt = 0;                   // format of the c16_lo plane: 0 fp16 residual, 1 fp8 byte pairs (the NEXT conv runs p8)
    bool p8 = false;                      // THIS conv runs as fp16 + fp8 corrections: a.lo holds byte pairs (conv_p8_ok must hold)
};

// smallest output width sent to the resident-tile kernel (A/B hook SVC_KCONV_MIN_N; 64-column layers use its 64-wide tile)
inline int kconv_min_n() {
    static const int v = [] { const char* e = getenv("SVC_KCONV_MIN_N"); return e ? atoi(e) : 64; }();
    return v;
}

// A conv of a vd == 3 model runs as fp16 + fp8 corrections iff it has the second packing and its call takes the
// resident-tile kernel: a layer / length property, never the batch's.  The producer of its operand planes asks the same
// question to choose the lo-plane format.
inline bool conv_p8_ok(const ConvW& w, int Lout, int dilation) {
    return w.w8 != nullptr && kconv_enabled() && (w.k - 1) * dilation <= 64 && Lout >= 192 && w.cout_pad >= kconv_min_n() &&
           w.cout_pad % 8 == 0;
}

inline int conv1d_run(const ConvW& w, const ConvRun& r, hipStream_t st) {
    KGemmParams p;
    memset(&p, 0, sizeof(p));
    const int nsub = is_split(w.vd) ? 3 : 1;
    SVC_REQUIRE(w.k * nsub <= KG_MAX_TAPS, "conv kernel size exceeds the tap limit");
    p.M = r.B * r.Lout;
    p.N = r.n_override ? r.n_override : w.cout_pad;
    p.Lout = r.Lout;
    p.a_seq_rows = r.Lin;
    p.a_len = r.Lin;
    p.seq_len = r.seq_len;
    p.a_stride = r.stride;
    p.pad_mode = r.pad_mode;
    p.n_taps = w.k * nsub;
    const int kt = w.cin_pad / ktile_elems(w.dtype);
    for (int t = 0; t < w.k; ++t)
        for (int sub = 0; sub < nsub; ++sub) {
            const int i = t * nsub + sub;
            p.a_ptr[i] = sub == 2 ? r.a.lo : r.a.hi;     // sub-taps [a_hi | a_hi | a_lo] x weights [w_hi | w_lo | w_hi]
            p.a_ld[i] = w.cin_pad;
            p.a_ktiles[i] = kt;
            p.a_shift[i] = t * r.dilation - r.pad_left;
        }
    p.w = w.w;
    p.ldw = w.ldw;
    p.bias = w.bias;
    p.c_seq_rows = r.c_rows ? r.c_rows : r.Lout;
    p.c_off = r.c_off;
    p.c32 = r.c32; p.ldc32 = r.ldc32;
    p.c16 = r.c16; p.ldc16 = r.ldc16;
    p.res = r.res; p.ldres = r.ldres;
    p.res2 = r.res2; p.ldres2 = r.ldres2;
    p.out_scale = r.out_scale;
    p.act = r.act; p.act_slope = r.act_slope;
    p.post_a = r.post_a; p.post_ib = r.post_ib; p.post_n = r.post_n; p.c16_lo = r.c16_lo;
    p.prof_flop_scale = 1.0f / nsub;
    p.vec_ok = (p.N % 8 == 0) && (r.ldc32 % 8 == 0) && (r.ldc16 % 8 == 0) && (r.ldres % 8 == 0) && (r.ldres2 % 8 == 0);
    // Long stride-1 convs with several taps keep their activation tile resident in LDS (kconv.hip).  The choice depends
    // on the layer and the sequence length only, never on the batch size, so batched and single runs stay bit-identical.
    const bool kconv_ok = kconv_enabled() && w.dtype == 0 && r.stride == 1 && w.k >= 3 && (w.k - 1) * r.dilation <= 64 &&
                          r.pad_mode == KG_PAD_ZERO && !r.seq_len && !r.n_override && p.vec_ok && r.Lout >= 192 && w.cin_pad >= 64 &&
                          p.N >= kconv_min_n();
    // (a conv and the conv it feeds inside a residual stack have the same shape, so both take the same kernel: the tap-GEMM
    // epilogue never has to write byte pairs)
    SVC_REQUIRE((!r.p8 && !r.c16_lo_fmt) || (kconv_ok && (!r.p8 || conv_p8_ok(w, r.Lout, r.dilation))),
                "conv1d_run: the fp8-pair operand format needs the resident-tile kernel");
    if (kconv_ok) {
        KConvParams q;
        memset(&q, 0, sizeof(q));
        q.a_hi = r.a.hi; q.a_lo = r.a.lo; q.w = r.p8 ? w.w8 : w.w; q.ldw = r.p8 ? w.ldw8 : w.ldw; q.bias = w.bias;
        q.B = r.B; q.Lin = r.Lin; q.Lout = r.Lout; q.N = p.N; q.cin_pad = w.cin_pad; q.k = w.k; q.dil = r.dilation;
        q.pad_left = r.pad_left; q.nsub = r.p8 ? 2 : nsub; q.w8_exp = w.w8_exp; q.c16_lo_fmt = r.c16_lo_fmt;
        q.c_seq_rows = p.c_seq_rows; q.c_off = p.c_off;
        q.c32 = r.c32; q.ldc32 = r.ldc32; q.c16 = r.c16; q.c16_lo = r.c16_lo; q.ldc16 = r.ldc16;
        q.post_a = r.post_a; q.post_ib = r.post_ib; q.post_n = r.post_n;
        q.res = r.res; q.ldres = r.ldres; q.res2 = r.res2; q.ldres2 = r.ldres2;
        q.out_scale = r.out_scale; q.act = r.act; q.act_slope = r.act_slope;
        return kconv_launch(q, st);
    }
    return kgemm_launch(p, w.dtype, KG_EPI_STORE, st);
}

// x [B][L][cin_pad] -> y [B][L*s][cout_pad] (channels-last), written at row offset c_off of c_rows-row sequences
inline int convT_run(const ConvW& w, ActBuf a, int B, int L, float* c32, int c_rows_q, int c_off_q, hipStream_t st) {
    KGemmParams p;
    memset(&p, 0, sizeof(p));
    const int nsub = is_split(w.vd) ? 3 : 1;
    p.M = B * L;
    p.N = w.stride * w.cout_pad;
    p.Lout = L;
    p.a_seq_rows = L;
    p.a_len = L;
    p.a_stride = 1;
    p.pad_mode = KG_PAD_ZERO;
    p.n_taps = 3 * nsub;
    const int kt = w.cin_pad / ktile_elems(w.dtype);
    for (int t = 0; t < 3; ++t)
        for (int sub = 0; sub < nsub; ++sub) {
            const int i = t * nsub + sub;
            p.a_ptr[i] = sub == 2 ? a.lo : a.hi;
            p.a_ld[i] = w.cin_pad;
            p.a_ktiles[i] = kt;
            p.a_shift[i] = t - 1;
        }
    p.w = w.w;
    p.ldw = w.ldw;
    p.bias = w.bias;
    p.c_seq_rows = c_rows_q ? c_rows_q : L;
    p.c_off = c_off_q;
    p.c32 = c32;
    p.ldc32 = p.N;
    p.vec_ok = 1;
    p.prof_flop_scale = 1.0f / nsub;
    return kgemm_launch(p, w.dtype, KG_EPI_STORE, st);
}

}  // namespace svc
