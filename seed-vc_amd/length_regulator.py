"""Host-side mirror of `InterpolateRegulator` over the C ABI.

Same call surface as the reference module: v1 `length_regulator(x, ylens=None, n_quantizers=None, f0=None)` returns
`(out * mask, olens, None, None, None)` (modules/length_regulator.py:90-141); v2 `(x, ylens=None, f0=None)` returns
`(out, olens)` (modules/v2/length_regulator.py:74-105).  A batch is B independent utterances: `in_lens` / `f0_lens`
give each one's own input lengths (default: the full padded length, which is the reference's B = 1 behaviour).
"""
import ctypes as C

import torch

from . import _lib, specs


class InterpolateRegulator:
    def __init__(self, cfg, state_dict, device="cuda:0"):
        self.cfg = cfg
        self.device = torch.device(device)
        c = _lib.LrConfig()
        c.channels, c.in_channels, c.out_channels = int(cfg["channels"]), int(cfg.get("in_channels") or 0), int(cfg["out_channels"])
        c.is_discrete, c.codebook_size = int(bool(cfg["is_discrete"])), int(cfg["codebook_size"])
        c.n_convs = int(cfg["n_convs"])
        c.interpolate = int(cfg["n_convs"] > 0)
        c.has_final_conv = int(specs.lr_has_final_conv(cfg))
        c.f0_condition, c.n_f0_bins = int(bool(cfg["f0_condition"])), int(cfg["n_f0_bins"])
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            descs, n, keep = _lib.make_descs(state_dict, self.device)
            _lib.check(_lib.lib().svc_lr_create(C.byref(c), descs, n, _lib.stream_ptr(), C.byref(self._h)))
            torch.cuda.current_stream().synchronize()
        del keep

    @torch.inference_mode()
    def __call__(self, x, ylens=None, n_quantizers=None, f0=None, in_lens=None, f0_lens=None):
        cfg = self.cfg
        B, Tin = x.shape[0], x.shape[1]
        if cfg["is_discrete"] and x.dim() == 3:
            x = x[:, 0]                                         # length_regulator.py:117-121 (single codebook)
            Tin = x.shape[1]
        in_lens = [Tin] * B if in_lens is None else [int(v) for v in in_lens]
        if cfg["n_convs"] > 0:
            yl = [int(v) for v in ylens.tolist()]
        else:                                                   # no interpolation: ylens clamp (length_regulator.py:127)
            yl = [min(int(v), Tin) for v in ylens.tolist()] if ylens is not None else list(in_lens)
        Tout = max(yl) if cfg["n_convs"] > 0 else Tin
        with torch.cuda.device(self.device):
            out = torch.empty(B, Tout, cfg["out_channels"], device=self.device)
            xp = tp = None
            if cfg["is_discrete"]:
                tok = x.to(device=self.device, dtype=torch.int64).contiguous()
                tp = C.c_void_p(tok.data_ptr())
            else:
                xx = _lib.f32c(x, self.device)
                xp = _lib.ptr(xx)
            fp, fl, Tf = None, None, 0
            if f0 is not None:
                ff = _lib.f32c(f0, self.device)
                Tf = ff.shape[1]
                fp = _lib.ptr(ff)
                fl = (C.c_int32 * B)(*([Tf] * B if f0_lens is None else [int(v) for v in f0_lens]))
            _lib.check(_lib.lib().svc_lr_forward(self._h, xp, tp, (C.c_int32 * B)(*in_lens), B, Tin, (C.c_int32 * B)(*yl),
                                                 Tout, fp, fl, Tf, _lib.ptr(out), _lib.stream_ptr()))
        olens = torch.tensor(yl, dtype=torch.long, device=ylens.device if ylens is not None else "cpu")
        if cfg["version"] == 1:
            return out, olens, None, None, None
        return out, olens

    forward = __call__

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().svc_lr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
