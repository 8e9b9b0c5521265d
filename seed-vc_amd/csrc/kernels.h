// Launch wrappers of the elementwise / layout / packing kernels (elementwise.hip, vocoder_ops.hip).
#pragma once
#include "common.h"

namespace svc {

int bct_to_btc_launch(const float* src, int B, int C, int T_src, float* dst32, long ld32, half_t* dst16, long ld16,
                      int seq_rows, int t_valid, float scale, hipStream_t st);
int btc_to_bct_launch(const float* src, long ld, int seq_rows, float* dst, int B, int C, int T, hipStream_t st);
int cast_rows_launch(const float* src, long lds_, half_t* dst, long ldd, int rows, int cols, hipStream_t st);
int timestep_feat_launch(const float* t, const float* freqs, float* out, int n, hipStream_t st);
int silu_launch(const float* x, float* y, long n, hipStream_t st);
int prefix_rows_launch(float* xin, int n_seq, int seq_rows, int D, int n_prefix, int t_rows, const float* tok_time,
                       const float* tok_style, int time_first, hipStream_t st);
int euler_cfg_launch(float* x, long ldx, half_t* x16, long ldx16, int x_rows, const float* v, long ldv,
                     long v_stream_stride, int v_rows, int B, int T, int C, const int* prompt_len, float dt, float c0,
                     float ca, float cb, int stream_a, int stream_b, hipStream_t st);
int pack_f16_launch(const float* src, half_t* dst, int n0, int n1, int n2, long s0, long s1, long s2, long d0, long d1,
                    long d2, const float* scale, hipStream_t st);
int pack_f16_lo_launch(const float* src, half_t* dst, int n0, int n1, int n2, long s0, long s1, long s2, long d0, long d1,
                       long d2, const float* scale, hipStream_t st);
int pack_f32_launch(const float* src, float* dst, int n0, int n1, int n2, long s0, long s1, long s2, long d0, long d1,
                    long d2, const float* scale, hipStream_t st);
int wn_scale_launch(const float* g, const float* v, int rows, long row_elems, float* out, hipStream_t st);
int small_linear_launch(const float* in, long ld_in, const float* W, long ldw, const float* bias, float* out, long ld_out,
                        int R, int N, int K, int act, hipStream_t st);
int add_rowvec_launch(float* dst, const float* a, long lda, const float* bvec, int n_seq, int n, hipStream_t st);

}  // namespace svc

namespace svc {
// aa_act.hip
int aa_act_rows_launch(const void* x, void* y, const float* up12_dev, const float* dn12_dev, const float* log_alpha,
                       const float* log_beta, int B, int C, int L, int dtype, hipStream_t st);
// channels-last activation: mode 0 anti-aliased snake, 1 plain snake, 2 leaky relu
// y_lo != null (fp16 output only): also write the residual plane x - float(half(x)) (split-precision operands)
int act_cl_launch(const float* x, long ldx, void* y, void* y_lo, long ldy, int out_f16, const float* taps12_host, const float* a,
                  const float* inv_b, int B, int C, int L, int mode, float slope, hipStream_t st, int lo_fmt = 0);
}  // namespace svc
