"""Host-side mirror of the reference's sampler / estimator interface over the C ABI.

`CFM.inference(mu, x_lens, prompt, style, f0, n_timesteps, temperature=1.0, inference_cfg_rate=0.5)`
keeps the reference signature (modules/flow_matching.py:30-31; v2: modules/v2/cfm.py:16-25), so a
driver that calls `model.cfm.inference(...)` (inference.py:483-505, seed_vc_wrapper.py:575-603) runs
unchanged.  Differences that are additions, not changes: `z=` lets the caller supply the noise
(otherwise torch.randn like the reference), and B > 1 is accepted and means B independent B=1 runs
(the reference sampler raises for B > 1).
"""
import ctypes as C

import torch

from . import _lib
from .specs import dit_config


def _cfg_struct(cfg):
    c = _lib.DitConfig()
    c.version = cfg["version"]
    c.hidden_dim, c.num_heads, c.depth = cfg["D"], cfg["H"], cfg["L"]
    c.in_channels, c.content_dim, c.style_dim = cfg["C"], cfg["Dc"], cfg["style_dim"]
    c.final_layer_type = 1 if cfg["head"] == "wavenet" else 0
    c.time_as_token, c.style_as_token = int(cfg["time_as_token"]), int(cfg["style_as_token"])
    c.uvit_skip_connection, c.long_skip_connection = int(cfg["uvit"]), int(cfg["long_skip"])
    c.style_condition = int(cfg["style_condition"])
    c.wn_hidden_dim = cfg.get("wn_dim", 0)
    c.wn_num_layers = cfg.get("wn_layers", 0)
    c.wn_kernel_size = cfg.get("wn_kernel", 0)
    c.wn_dilation_rate = cfg.get("wn_dilation", 1)
    return c


class Estimator:
    """Packed DiT on one device (mirror of `CFM.estimator`)."""

    def __init__(self, cfg, state_dict, device="cuda:0"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.in_channels = cfg["C"]
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            descs, n, keep = _lib.make_descs(state_dict, self.device)
            cs = _cfg_struct(cfg)
            _lib.check(_lib.lib().svc_dit_create(C.byref(cs), descs, n, _lib.stream_ptr(), C.byref(self._h)))
            torch.cuda.current_stream().synchronize()
        del keep

    def setup_caches(self, max_batch_size=1, max_seq_length=8192):
        """No-op kept for call compatibility (inference.py:90): RoPE table and skip lists are built at pack time."""
        return None

    def set_microbatch(self, n):
        _lib.check(_lib.lib().svc_dit_set_microbatch(self._h, int(n)))

    def set_fused_min_rows(self, rows):
        """Token rows per launch from which the transformer layers use the fused row-panel kernel (-1: default)."""
        _lib.check(_lib.lib().svc_dit_set_fused_min_rows(self._h, C.c_long(int(rows))))

    def set_graphs(self, on):
        """hipGraph capture / replay of the sampler's Euler loop (off by default; bit-identical to eager launches)."""
        _lib.check(_lib.lib().svc_dit_set_graphs(self._h, int(bool(on))))

    @property
    def fused_available(self):
        return bool(_lib.lib().svc_dit_fused_available(self._h))

    def __call__(self, x, prompt_x, x_lens, t, style, cond, mask_content=False):
        """estimator(x, prompt_x, x_lens, t, style, mu) -> (N, C, T); reference: diffusion_transformer.py:486."""
        N, Cc, T = x.shape
        with torch.cuda.device(self.device):
            x, prompt_x, style, cond = (_lib.f32c(a, self.device) for a in (x, prompt_x, style, cond))
            out = torch.empty_like(x)
            tval = float(t.reshape(-1)[0]) if torch.is_tensor(t) else float(t)
            lens = None
            if x_lens is not None:
                xl = [int(v) for v in (x_lens.tolist() if torch.is_tensor(x_lens) else x_lens)]
                if len(xl) == 1 and N > 1:
                    xl = xl * N
                lens = _lib.i64_host(xl)
            _lib.check(_lib.lib().svc_dit_forward(self._h, N, T, _lib.ptr(x), _lib.ptr(prompt_x), lens,
                                                  C.c_float(tval), _lib.ptr(style), _lib.ptr(cond), _lib.ptr(out),
                                                  _lib.stream_ptr()))
        return out

    def close(self):
        if self._h:
            _lib.lib().svc_dit_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CFM:
    """Mirror of modules.flow_matching.CFM / modules.v2.cfm.CFM for inference."""

    def __init__(self, cfg, state_dict, device="cuda:0"):
        self.cfg = cfg
        self.estimator = Estimator(cfg, state_dict, device)
        self.in_channels = cfg["C"]
        self.device = self.estimator.device

    @classmethod
    def from_preset(cls, name, state_dict, device="cuda:0", **overrides):
        return cls(dit_config(name, **overrides), state_dict, device)

    @torch.inference_mode()
    def inference(self, mu, x_lens, prompt, style, f0=None, n_timesteps=10, temperature=1.0,
                  inference_cfg_rate=0.5, random_voice=False, z=None, prompt_lens=None):
        if self.cfg["version"] == 2 and not isinstance(f0, (type(None), torch.Tensor)):
            # v2 call form: inference(mu, x_lens, prompt, style, n_timesteps, temperature, inference_cfg_rate, ...)
            n_timesteps, f0 = f0, None
        B, T = mu.size(0), mu.size(1)
        dev = self.device
        with torch.cuda.device(dev):
            mu, prompt, style = (_lib.f32c(a, dev) for a in (mu, prompt, style))
            if z is None:
                z = torch.randn([B, self.in_channels, T], device=dev)      # flow_matching.py:50
            z = _lib.f32c(z, dev)
            out = torch.empty(B, self.in_channels, T, device=dev, dtype=torch.float32)
            a = _lib.CfmArgs()
            a.B, a.T, a.P = B, T, prompt.size(-1)
            a.mu, a.prompt, a.style, a.z, a.out = (t.data_ptr() for t in (mu, prompt, style, z, out))
            xl = None
            if x_lens is not None:
                xl = [int(v) for v in (x_lens.tolist() if torch.is_tensor(x_lens) else x_lens)]
                if len(xl) == 1 and B > 1:
                    xl = xl * B
            lens_keep = _lib.i64_host(xl)
            plens_keep = _lib.i64_host(prompt_lens)
            a.x_lens = lens_keep
            a.prompt_lens = plens_keep
            a.n_timesteps = int(n_timesteps)
            a.temperature = float(temperature)
            if isinstance(inference_cfg_rate, (list, tuple)):
                a.cfg_rate[0], a.cfg_rate[1] = float(inference_cfg_rate[0]), float(inference_cfg_rate[1])
            else:
                a.cfg_rate[0], a.cfg_rate[1] = float(inference_cfg_rate), float(inference_cfg_rate)
            a.random_voice = int(bool(random_voice))
            _lib.check(_lib.lib().svc_cfm_sample(self.estimator._h, C.byref(a), _lib.stream_ptr()))
        return out
