"""Host-side mirror of the v2 AR model's generation step over the C ABI (modules/v2/ar.py).

`ARModel.forward_generate(x, input_pos, kv_pos)` mirrors `NaiveTransformer.forward_generate` (B = 1) and returns the
last token's logits; `decode_step` is the hipGraph-captured one-token step (the reference's `compiled_decode_fn`);
`sample(...)` mirrors `sample()/logits_to_probs()` with the Exp(1) noise drawn here unless supplied.
"""
import ctypes as C

import torch

from . import _lib


class ARModel:
    def __init__(self, cfg, state_dict, device="cuda:0"):
        self.cfg = cfg
        self.device = torch.device(device)
        c = _lib.ArConfig()
        for k in ("dim", "n_head", "n_local_heads", "head_dim", "n_layer", "intermediate_size", "vocab_size", "max_seq_len"):
            setattr(c, k, int(cfg[k]))
        c.rope_base, c.norm_eps = float(cfg["rope_base"]), float(cfg["norm_eps"])
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            descs, n, keep = _lib.make_descs(state_dict, self.device)
            _lib.check(_lib.lib().svc_ar_create(C.byref(c), descs, n, _lib.stream_ptr(), C.byref(self._h)))
            torch.cuda.current_stream().synchronize()
        del keep
        # host-side pieces of NaiveWrapper.generate's prompt assembly (ar.py:390-396)
        self._sep = state_dict["sep_token_emb"].detach().to(self.device, torch.float32) if "sep_token_emb" in state_dict else None
        self._emb = (state_dict["model.embeddings.weight"].detach().to(self.device, torch.float32)
                     if "model.embeddings.weight" in state_dict else None)

    def setup_caches(self, max_batch_size=1, max_seq_len=None, dtype=None, device=None):
        """Kept for call compatibility (vc_wrapper.py:328-329); the cache lives in the handle.  Resets it."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().svc_ar_reset(self._h, _lib.stream_ptr()))

    @torch.inference_mode()
    def forward_generate(self, x, input_pos, kv_pos):
        """x (1, S, dim) -> logits (1, 1, vocab) of the last token."""
        S = x.shape[1]
        with torch.cuda.device(self.device):
            xx = _lib.f32c(x, self.device).reshape(S, -1)
            out = torch.empty(self.cfg["vocab_size"], device=self.device)
            ip, kp = _lib.i64_host(input_pos.tolist()), _lib.i64_host(kv_pos.tolist())
            _lib.check(_lib.lib().svc_ar_forward_generate(self._h, _lib.ptr(xx), S, ip, kp, _lib.ptr(out), _lib.stream_ptr()))
        return out.reshape(1, 1, -1)

    @torch.inference_mode()
    def decode_step(self, x, input_pos=None, kv_pos=None):
        """One-token step from the captured hipGraph; pass positions on the first step only (then they auto-advance)."""
        with torch.cuda.device(self.device):
            xx = _lib.f32c(x, self.device).reshape(-1)
            out = torch.empty(self.cfg["vocab_size"], device=self.device)
            set_pos = int(input_pos is not None)
            _lib.check(_lib.lib().svc_ar_decode_step(self._h, _lib.ptr(xx), set_pos, C.c_int64(int(input_pos or 0)),
                                                     C.c_int64(int(kv_pos or 0)), _lib.ptr(out), _lib.stream_ptr()))
        return out.reshape(1, 1, -1)

    @torch.inference_mode()
    def sample(self, logits, previous_tokens=None, suppress_tokens=None, temperature=0.7, top_p=0.7, repetition_penalty=1.5,
               exp_noise=None, return_probs=False):
        V = self.cfg["vocab_size"]
        with torch.cuda.device(self.device):
            lg = _lib.f32c(logits, self.device).reshape(-1)[-V:].contiguous()
            if exp_noise is None:
                exp_noise = torch.empty(V, device=self.device).exponential_(1)      # ar.py:726
            q = _lib.f32c(exp_noise, self.device)
            prev = previous_tokens.to(self.device, torch.int32).contiguous() if previous_tokens is not None else None
            sup = int(suppress_tokens[0]) if suppress_tokens else -1
            idx = torch.empty(1, device=self.device, dtype=torch.int32)
            probs = torch.empty(V, device=self.device) if return_probs else None
            _lib.check(_lib.lib().svc_ar_sample(self._h, _lib.ptr(lg), _lib.ptr(prev), 0 if prev is None else prev.numel(), sup,
                                                C.c_float(temperature), C.c_float(top_p), C.c_float(repetition_penalty),
                                                _lib.ptr(q), _lib.ptr(idx), _lib.ptr(probs), _lib.stream_ptr()))
        return (idx, probs) if return_probs else idx

    @torch.inference_mode()
    def generate(self, prompt_text, prompt_target, compiled_decode_fn=None, top_p=0.7, temperature=0.7,
                 repetition_penalty=1.5, exp_noise=None, max_new=4001, check_every=16):
        """`NaiveWrapper.generate` (modules/v2/ar.py:382-422): prompt_text (1, Tt, dim) condition embeddings,
        prompt_target (1, Tp) tokens -> (1, n) generated tokens.  The token loop runs on the device
        (`svc_ar_generate`); exp_noise (max_new, vocab) pins the Exp(1) draws (drawn here when None).
        compiled_decode_fn is accepted and ignored: the captured hipGraph step is always used."""
        V, D = self.cfg["vocab_size"], self.cfg["dim"]
        with torch.cuda.device(self.device):
            text = _lib.f32c(prompt_text, self.device)
            sep = self._sep.reshape(1, 1, D)
            tgt = prompt_target.to(self.device).long()
            tgt_emb = self._emb[tgt[0]][None] if tgt.numel() else torch.zeros(1, 0, D, device=self.device)
            emb_seq = torch.cat([sep, text, sep, tgt_emb], dim=1)[0].contiguous()
            S = emb_seq.shape[0]
            input_pos = list(range(text.size(1) + 1)) + [0] + [i + 1 for i in range(tgt_emb.size(1))]
            kv_pos = list(range(S))
            max_new = min(int(max_new), self.cfg["max_seq_len"] - S + 1)
            if exp_noise is None:
                exp_noise = torch.empty(max_new, V, device=self.device).exponential_(1)
            q = _lib.f32c(exp_noise, self.device)
            assert q.shape[0] >= max_new and q.shape[1] == V
            toks = torch.zeros(max_new, device=self.device, dtype=torch.int32)
            n = C.c_int32(0)
            self.setup_caches()
            _lib.check(_lib.lib().svc_ar_generate(self._h, _lib.ptr(emb_seq), S, _lib.i64_host(input_pos), _lib.i64_host(kv_pos),
                                                  _lib.ptr(q), max_new, 10, C.c_float(temperature), C.c_float(top_p),
                                                  C.c_float(repetition_penalty), int(check_every), _lib.ptr(toks), C.byref(n),
                                                  _lib.stream_ptr()))
        return toks[:n.value].long()[None, :]

    def close(self):
        if self._h:
            _lib.lib().svc_ar_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
