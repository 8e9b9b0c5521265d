// Log-mel front-end (first half): `mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size,
// win_size, fmin, fmax, center=False)` of modules/audio.py:45-82 -- reflect pad (n_fft - hop)/2, STFT (Hann window,
// onesided), sqrt(re^2 + im^2 + 1e-9), mel matmul, log(clamp(., 1e-5)).
//
// The STFT is ONE tap-GEMM on the fp32 MFMA: the padded signal is read as an [n_frames][n_fft] matrix whose rows
// overlap in memory (row stride = hop), against the window-weighted DFT basis [2 (n_fft/2 + 1)][n_fft]; a second GEMM
// applies the mel filterbank.  Runs once per source / reference clip.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "model_util.h"

using namespace svc;

namespace {

// reflect padding without edge repeat (F.pad mode="reflect"): dst [B][stride], valid [0, L + 2 pad)
__global__ void mel_pad_kernel(const float* __restrict__ y, int L, float* __restrict__ dst, long stride, int pad) {
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= stride) return;
    float v = 0.f;
    if (i < L + 2 * pad) {
        long q = i - pad;
        q = q < 0 ? -q : (q >= L ? 2L * (L - 1) - q : q);
        v = y[(long)b * L + q];
    }
    dst[(long)b * stride + i] = v;
}

// spec [M][ld_s] = (re[0..nb) | im[0..nb)) -> mag [M][ld_m] = sqrt(re^2 + im^2 + 1e-9), pad columns zero
__global__ void mel_mag_kernel(const float* __restrict__ spec, long ld_s, float* __restrict__ mag, long ld_m, int nb, long M) {
    const long m = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ld_m) return;
    float v = 0.f;
    if (k < nb) {
        const float re = spec[m * ld_s + k], im = spec[m * ld_s + nb + k];
        v = sqrtf(re * re + im * im + 1e-9f);
    }
    mag[m * ld_m + k] = v;
}

// out[b][mel][frame] = log(max(c[b * frames + frame][mel], 1e-5))
__global__ void mel_log_kernel(const float* __restrict__ c, long ldc, float* __restrict__ out, int n_mels, int frames) {
    const int b = blockIdx.z, f = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (f >= frames) return;
    out[((long)b * n_mels + j) * frames + f] = logf(fmaxf(c[((long)b * frames + f) * ldc + j], 1e-5f));
}

}  // namespace

struct svc_mel {
    int n_fft, hop, n_mels, nb;          // nb = n_fft / 2 + 1
    Arena weights, work;
    float* dft = nullptr;  long ld_dft = 0;  int n_dft_pad = 0;     // [2 nb -> pad 128][n_fft]
    float* fb = nullptr;   long ld_fb = 0;                           // [n_mels -> pad 128][nb -> pad 32]
    int cap_B = 0, cap_L = 0;
    float *padded = nullptr, *spec = nullptr, *mag = nullptr, *melc = nullptr;
    long stride = 0;
};

extern "C" {

int svc_mel_create(int n_fft, int hop, int win, int n_mels, const float* window, const float* mel_basis, void* stream,
                   svc_mel_t** out) {
    SVC_REQUIRE(out && window && mel_basis, "null argument");
    SVC_REQUIRE(n_fft >= 64 && n_fft % 32 == 0 && win == n_fft && hop >= 4 && hop % 4 == 0 && (n_fft - hop) % 2 == 0 && n_mels >= 1,
                "mel front-end: n_fft multiple of 32, win_size == n_fft, hop multiple of 4");
    hipStream_t st = (hipStream_t)stream;
    auto* m = new svc_mel();
    m->n_fft = n_fft; m->hop = hop; m->n_mels = n_mels; m->nb = n_fft / 2 + 1;
    auto fail = [&]() { delete m; return 1; };
    // window-weighted DFT basis, float64 trigonometry -> fp32 (rows: cos block, then -sin block)
    std::vector<float> hw(n_fft);
    if (hipMemcpy(hw.data(), window, n_fft * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { set_error("window copy failed"); return fail(); }
    const int nrows = 2 * m->nb;
    m->n_dft_pad = (int)round_up(nrows, 128);
    m->ld_dft = n_fft;
    std::vector<float> basis((size_t)m->n_dft_pad * n_fft, 0.f);
    for (int k = 0; k < m->nb; ++k)
        for (int n = 0; n < n_fft; ++n) {
            const double ang = 2.0 * M_PI * (double)(((long)k * n) % n_fft) / (double)n_fft;
            basis[(size_t)k * n_fft + n] = (float)(cos(ang) * (double)hw[n]);
            basis[(size_t)(m->nb + k) * n_fft + n] = (float)(-sin(ang) * (double)hw[n]);
        }
    m->dft = m->weights.alloc_n<float>(basis.size(), st);
    const int nbp = (int)round_up(m->nb, 32);
    m->ld_fb = nbp;
    m->fb = m->weights.alloc_n<float>((size_t)round_up(n_mels, 128) * nbp, st);
    if (!m->dft || !m->fb) return fail();
    if (hipMemcpyAsync(m->dft, basis.data(), basis.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { set_error("basis upload failed"); return fail(); }
    if (pack_f32_launch(mel_basis, m->fb, n_mels, 1, m->nb, m->nb, 0, 1, nbp, 0, 1, nullptr, st)) return fail();
    if (hipStreamSynchronize(st) != hipSuccess) { set_error("sync failed"); return fail(); }
    *out = m;
    return 0;
}

void svc_mel_destroy(svc_mel_t* m) { delete m; }

int svc_mel_frames(const svc_mel_t* m, int L) { return m && L >= m->hop ? 1 + (L - m->hop) / m->hop : 0; }

int svc_mel_forward(svc_mel_t* m, const float* y, int B, int L, float* out, void* stream) {
    SVC_REQUIRE(m && y && out && B >= 1, "bad argument");
    const int pad = (m->n_fft - m->hop) / 2;
    SVC_REQUIRE(L > pad, "signal shorter than the reflect padding");
    hipStream_t st = (hipStream_t)stream;
    const int Lp = L + 2 * pad;
    const int frames = 1 + (Lp - m->n_fft) / m->hop;
    const int nbp = (int)m->ld_fb, ld_s = (int)round_up(2 * m->nb, 8), ld_c = (int)round_up(m->n_mels, 32);
    if (B > m->cap_B || L > m->cap_L) {
        SVC_CHECK_HIP(hipStreamSynchronize(st));
        m->work.release();
        m->cap_B = std::max(B, m->cap_B); m->cap_L = std::max(L, m->cap_L);
        const long Lpc = m->cap_L + 2L * pad;
        m->stride = round_up(Lpc + m->n_fft, m->hop);              // rows of n_fft read at stride hop stay inside
        const long fr = 1 + (Lpc - m->n_fft) / m->hop;
        m->padded = m->work.alloc_n<float>((size_t)m->cap_B * m->stride + m->n_fft, st);
        m->spec = m->work.alloc_n<float>((size_t)m->cap_B * fr * ld_s, st);
        m->mag = m->work.alloc_n<float>((size_t)m->cap_B * fr * nbp, st);
        m->melc = m->work.alloc_n<float>((size_t)m->cap_B * fr * ld_c, st);
        if (!m->padded || !m->spec || !m->mag || !m->melc) return 1;
    }
    hipLaunchKernelGGL(mel_pad_kernel, dim3(cdiv(m->stride, 256), B), dim3(256), 0, st, y, L, m->padded, m->stride, pad);
    SVC_CHECK_HIP(hipGetLastError());
    // STFT: rows = frames (stride hop inside each padded sequence), K = n_fft
    KGemmParams p;
    memset(&p, 0, sizeof(p));
    p.M = B * frames; p.N = 2 * m->nb; p.Lout = frames;
    p.a_seq_rows = (int)(m->stride / m->hop); p.a_len = p.a_seq_rows; p.a_stride = 1; p.n_taps = 1;
    p.a_ptr[0] = m->padded; p.a_ld[0] = m->hop; p.a_ktiles[0] = m->n_fft / 32;
    p.w = m->dft; p.ldw = m->ld_dft;
    p.c_seq_rows = frames; p.c32 = m->spec; p.ldc32 = ld_s; p.vec_ok = 1;      // pad columns (zero weight rows) land in the ld padding
    if (kgemm_launch(p, 1, KG_EPI_STORE, st)) return 1;
    hipLaunchKernelGGL(mel_mag_kernel, dim3(cdiv(nbp, 128), B * frames), dim3(128), 0, st, m->spec, (long)ld_s, m->mag, (long)nbp,
                       m->nb, (long)B * frames);
    SVC_CHECK_HIP(hipGetLastError());
    memset(&p, 0, sizeof(p));
    p.M = B * frames; p.N = m->n_mels; p.Lout = frames;
    p.a_seq_rows = frames; p.a_len = frames; p.a_stride = 1; p.n_taps = 1;
    p.a_ptr[0] = m->mag; p.a_ld[0] = nbp; p.a_ktiles[0] = nbp / 32;
    p.w = m->fb; p.ldw = m->ld_fb;
    p.c_seq_rows = frames; p.c32 = m->melc; p.ldc32 = ld_c; p.vec_ok = 1;
    if (kgemm_launch(p, 1, KG_EPI_STORE, st)) return 1;
    hipLaunchKernelGGL(mel_log_kernel, dim3(cdiv(frames, 128), m->n_mels, B), dim3(128), 0, st, m->melc, (long)ld_c, out, m->n_mels, frames);
    SVC_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
