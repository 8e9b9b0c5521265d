"""Host-side mirror of the log-mel front-end over the C ABI (first half).

`MelSpectrogram(**mel_fn_args)(y)` has the call surface of `to_mel = lambda x: mel_spectrogram(x, **mel_fn_args)`
(inference.py:315-327, modules/audio.py:45-82): y (B, L) in [-1, 1] -> (B, num_mels, frames).
The mel filterbank is `librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` in the reference; pass the reference's own
cached tensor as `mel_basis=` when it is available, otherwise `slaney_mel_basis` restates librosa's default
(Slaney scale, area normalisation) -- parity of that restatement is unpinned (librosa is absent from the build image).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def slaney_mel_basis(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """librosa.filters.mel defaults (htk=False, norm='slaney'), float32 [n_mels][n_fft // 2 + 1]."""
    fmax = sr / 2.0 if fmax in (None, "None") else float(fmax)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-12) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return torch.from_numpy(w.astype(np.float32))


class MelSpectrogram:
    def __init__(self, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin=0, fmax=None, center=False, mel_basis=None,
                 device="cuda:0"):
        assert not center, "the reference drivers call mel_spectrogram with center=False"
        self.device = torch.device(device)
        self.n_fft, self.hop, self.n_mels = int(n_fft), int(hop_size), int(num_mels)
        if mel_basis is None:
            mel_basis = slaney_mel_basis(sampling_rate, n_fft, num_mels, fmin, fmax)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            win = torch.hann_window(int(win_size)).to(self.device)
            mb = _lib.f32c(mel_basis, self.device)
            _lib.check(_lib.lib().svc_mel_create(self.n_fft, self.hop, int(win_size), self.n_mels, _lib.ptr(win), _lib.ptr(mb),
                                                 _lib.stream_ptr(), C.byref(self._h)))

    @torch.inference_mode()
    def __call__(self, y):
        with torch.cuda.device(self.device):
            yy = _lib.f32c(y, self.device)
            B, L = yy.shape
            frames = 1 + (L - self.hop) // self.hop
            out = torch.empty(B, self.n_mels, frames, device=self.device)
            _lib.check(_lib.lib().svc_mel_forward(self._h, _lib.ptr(yy), B, L, _lib.ptr(out), _lib.stream_ptr()))
        return out

    def __del__(self):
        try:
            if self._h:
                _lib.lib().svc_mel_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass
