"""Host-side mirror of the CAMPPlus style encoder and the Kaldi fbank feeding it (second half).

`CAMPPlus(cfg, state_dict)(feat)` has the call surface of `campplus_model(feat2.unsqueeze(0))` (inference.py:430,
modules/campplus/DTDNN.py:132-137): feat (B, T, 80) -> (B, 192).  `.fbank(wave_16k)` replaces
`torchaudio.compliance.kaldi.fbank(wave, num_mel_bins=80, dither=0, sample_frequency=16000)` (inference.py:418-428);
`.style(wave_16k)` is the drivers' three lines together: fbank, mean normalisation over time, embedding.
"""
import ctypes as C

import torch

from . import _lib


class CAMPPlus:
    def __init__(self, cfg, state_dict, device="cuda:0"):
        self.cfg = cfg
        self.device = torch.device(device)
        c = _lib.CampplusConfig()
        for k in ("feat_dim", "embedding_size", "growth_rate", "bn_size", "init_channels", "m_channels", "seg_len"):
            setattr(c, k, int(cfg[k]))
        c.n_blocks = len(cfg["block_layers"])
        for i, (n, k, d) in enumerate(zip(cfg["block_layers"], cfg["block_kernel"], cfg["block_dilation"])):
            c.block_layers[i], c.block_kernel[i], c.block_dilation[i] = int(n), int(k), int(d)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            descs, n, keep = _lib.make_descs(state_dict, self.device)
            _lib.check(_lib.lib().svc_campplus_create(C.byref(c), descs, n, _lib.stream_ptr(), C.byref(self._h)))
            torch.cuda.current_stream().synchronize()
        del keep

    def eval(self):
        return self

    @torch.inference_mode()
    def __call__(self, x, x_lens=None):
        if x_lens is not None:
            raise NotImplementedError("masked statistics pooling (x_lens) is not on the inference path of the drivers")
        B, T, F = x.shape
        with torch.cuda.device(self.device):
            xx = _lib.f32c(x, self.device)
            out = torch.empty(B, self.cfg["embedding_size"], device=self.device)
            _lib.check(_lib.lib().svc_campplus_forward(self._h, _lib.ptr(xx), B, T, _lib.ptr(out), _lib.stream_ptr()))
        return out

    forward = __call__

    @torch.inference_mode()
    def fbank(self, wave):
        """wave (1, L) or (L,) at 16 kHz -> (frames, feat_dim)."""
        with torch.cuda.device(self.device):
            w = _lib.f32c(wave, self.device).reshape(-1)
            n = _lib.lib().svc_kaldi_fbank_frames(int(w.numel()))
            out = torch.empty(n, self.cfg["feat_dim"], device=self.device)
            _lib.check(_lib.lib().svc_kaldi_fbank(self._h, _lib.ptr(w), int(w.numel()), _lib.ptr(out), _lib.stream_ptr()))
        return out

    def style(self, wave_16k):
        feat = self.fbank(wave_16k)
        feat = feat - feat.mean(dim=0, keepdim=True)          # inference.py:429
        return self(feat.unsqueeze(0))

    def close(self):
        if self._h:
            _lib.lib().svc_campplus_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
