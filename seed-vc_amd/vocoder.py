"""Host-side mirrors of the reference vocoders over the C ABI.

`BigVGAN(h, state_dict)(mel) -> (B, 1, L)` mirrors `BigVGAN.forward` (modules/bigvgan/bigvgan.py:360-386; the
drivers call it as `vocoder_fn(vc_target.float())`, inference.py:506, and `.squeeze()` the result).
`HiFT(cfg, state_dict)(mel) -> (B, L)` mirrors `HiFTGenerator.forward` / `.inference`
(modules/hifigan/generator.py:400-436,452-454).  `precision`: "fp32" = exact fp32 MFMA, "fp16x3" = split hi/lo fp16
operands with three MFMA products per step (fp32-class accuracy, faster), "fp16p8" = the same split with the two
correction products of the long stride-1 convs in ONE block-scaled fp8 MFMA (waveform RMS ~1e-5: inside the 1e-4 bound
with margin, faster again), "fp16" = plain fp16 operands (2.4e-4: outside the bound, reported only).  HiFT's random draws (SineGen phases and noise,
generator.py:208-222) are drawn here with torch when the caller does not pass them.
"""
import ctypes as C
import math

import torch

from . import _lib
from .specs import bigvgan_total_upsample, hift_total_upsample


PRECISIONS = {"fp32": 0, "fp16": 1, "fp16x3": 2, "fp16p8": 3}


class BigVGAN:
    def __init__(self, h, state_dict, device="cuda:0", precision="fp16p8"):
        self.h = h
        self.device = torch.device(device)
        self.total_up = bigvgan_total_upsample(h)
        c = _lib.BigVGANConfig()
        c.num_mels = h["num_mels"]
        c.upsample_initial_channel = h["upsample_initial_channel"]
        c.num_upsamples = len(h["upsample_rates"])
        c.num_kernels = len(h["resblock_kernel_sizes"])
        for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
            c.upsample_rates[i], c.upsample_kernel_sizes[i] = u, k
        for j, (k, d) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            c.resblock_kernel_sizes[j] = k
            for e in range(3):
                c.resblock_dilation_sizes[j][e] = d[e]
        c.use_tanh_at_final = int(h.get("use_tanh_at_final", True))
        c.use_bias_at_final = int(h.get("use_bias_at_final", True))
        c.snake_logscale = int(h["snake_logscale"])
        c.snakebeta = int(h["activation"] == "snakebeta")
        c.precision = PRECISIONS[precision]
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            descs, n, keep = _lib.make_descs(state_dict, self.device)
            _lib.check(_lib.lib().svc_bigvgan_create(C.byref(c), descs, n, _lib.stream_ptr(), C.byref(self._h)))
            torch.cuda.current_stream().synchronize()
        del keep

    def set_microbatch(self, n):
        """Utterances per internal pass (0 = library default); the output does not depend on it."""
        _lib.check(_lib.lib().svc_bigvgan_set_microbatch(self._h, int(n)))

    @torch.inference_mode()
    def __call__(self, mel):
        B, _, S = mel.shape
        with torch.cuda.device(self.device):
            mel = _lib.f32c(mel, self.device)
            out = torch.empty(B, 1, S * self.total_up, device=self.device, dtype=torch.float32)
            _lib.check(_lib.lib().svc_bigvgan_forward(self._h, _lib.ptr(mel), B, S, _lib.ptr(out), _lib.stream_ptr()))
        return out

    forward = __call__

    def close(self):
        if self._h:
            _lib.lib().svc_bigvgan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HiFT:
    def __init__(self, cfg, state_dict, device="cuda:0", precision="fp16p8"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.total_up = hift_total_upsample(cfg)
        c = _lib.HiftConfig()
        c.in_channels, c.base_channels = cfg["in_channels"], cfg["base_channels"]
        c.nb_harmonics, c.sampling_rate = cfg["nb_harmonics"], cfg["sampling_rate"]
        c.nsf_alpha, c.nsf_sigma = cfg["nsf_alpha"], cfg["nsf_sigma"]
        c.nsf_voiced_threshold = cfg["nsf_voiced_threshold"]
        c.num_upsamples = len(cfg["upsample_rates"])
        for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
            c.upsample_rates[i], c.upsample_kernel_sizes[i] = u, k
        c.istft_n_fft, c.istft_hop = cfg["istft_n_fft"], cfg["istft_hop"]
        c.num_kernels = len(cfg["resblock_kernel_sizes"])
        for j, (k, d) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            c.resblock_kernel_sizes[j] = k
            for e in range(3):
                c.resblock_dilation_sizes[j][e] = d[e]
        for j, (k, d) in enumerate(zip(cfg["source_resblock_kernel_sizes"], cfg["source_resblock_dilation_sizes"])):
            c.source_resblock_kernel_sizes[j] = k
            for e in range(3):
                c.source_resblock_dilation_sizes[j][e] = d[e]
        c.lrelu_slope, c.audio_limit = cfg["lrelu_slope"], cfg["audio_limit"]
        c.f0_cond_channels = cfg["f0_cond_channels"]
        c.precision = PRECISIONS[precision]
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            descs, n, keep = _lib.make_descs(state_dict, self.device)
            _lib.check(_lib.lib().svc_hift_create(C.byref(c), descs, n, _lib.stream_ptr(), C.byref(self._h)))
            torch.cuda.current_stream().synchronize()
        del keep

    def set_microbatch(self, n):
        """Utterances per internal pass (0 = library default); the output does not depend on it."""
        _lib.check(_lib.lib().svc_hift_set_microbatch(self._h, int(n)))

    @torch.inference_mode()
    def __call__(self, x, f0=None, phase0=None, noise=None, return_f0=False):
        B, _, S = x.shape
        nh = self.cfg["nb_harmonics"] + 1
        Lw = S * self.total_up
        dev = self.device
        with torch.cuda.device(dev):
            mel = _lib.f32c(x, dev)
            if phase0 is None:     # Uniform(-pi, pi).sample((B, nh, 1)): generator.py:208-209
                phase0 = (torch.rand(B, nh, 1, device=dev) * 2 - 1) * math.pi
            if noise is None:      # torch.randn_like(sine_waves): generator.py:222
                noise = torch.randn(B, nh, Lw, device=dev)
            phase0, noise = _lib.f32c(phase0, dev), _lib.f32c(noise, dev)
            f0t = _lib.f32c(f0, dev) if f0 is not None else None
            out = torch.empty(B, Lw, device=dev, dtype=torch.float32)
            f0_out = torch.empty(B, S, device=dev, dtype=torch.float32) if return_f0 else None
            _lib.check(_lib.lib().svc_hift_forward(self._h, _lib.ptr(mel), _lib.ptr(f0t), _lib.ptr(phase0), _lib.ptr(noise),
                                                   B, S, _lib.ptr(out), _lib.ptr(f0_out), _lib.stream_ptr()))
        return (out, f0_out) if return_f0 else out

    forward = __call__
    inference = __call__

    def close(self):
        if self._h:
            _lib.lib().svc_hift_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
