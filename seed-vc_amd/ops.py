"""Op-level wrappers over the C ABI (parity tests and the anti-aliased activation seam)."""
import torch

from . import _lib


def anti_alias_activation_forward(inputs, up_ftr, down_ftr, alpha, beta):
    """Drop-in for `anti_alias_activation_cuda.forward(inputs, up_ftr, down_ftr, alpha, beta)`
    (reference: modules/bigvgan/alias_free_activation/cuda/activation1d.py:23-25): (B,C,L) in -> (B,C,L) out,
    same dtype (fp32 / fp16 / bf16); alpha/beta are log-scale, exp() is applied in the kernel."""
    assert inputs.is_cuda and inputs.dim() == 3
    x = inputs.contiguous()
    dt = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[x.dtype]
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        up = _lib.f32c(up_ftr, x.device).reshape(-1)
        dn = _lib.f32c(down_ftr, x.device).reshape(-1)
        al = _lib.f32c(alpha, x.device).reshape(-1)
        be = _lib.f32c(beta, x.device).reshape(-1)
        B, Cc, L = x.shape
        _lib.check(_lib.lib().svc_anti_alias_act_fwd(_lib.ptr(x), _lib.ptr(y), _lib.ptr(up), _lib.ptr(dn), _lib.ptr(al),
                                                     _lib.ptr(be), B, Cc, L, dt, _lib.stream_ptr()))
    return y


def linear(a, w, bias=None, dtype="f16", act=0):
    M, K = a.shape
    N = w.shape[0]
    dev = a.device
    with torch.cuda.device(dev):
        a, w = _lib.f32c(a), _lib.f32c(w)
        b = _lib.f32c(bias) if bias is not None else None
        c = torch.empty(M, N, device=dev)
        _lib.check(_lib.lib().svc_op_linear(_lib.ptr(a), _lib.ptr(w), _lib.ptr(b), _lib.ptr(c), M, N, K,
                                            0 if dtype == "f16" else 1, act, _lib.stream_ptr()))
    return c


def conv1d_cl(x, w, bias, dilation=1, stride=1, pad_left=0, Lout=None, pad_mode=0, dtype="f32"):
    """x (B, L, Cin) channels-last, w (Cout, Cin, k) torch layout -> (B, Lout, Cout)."""
    B, L, Cin = x.shape
    Cout, _, k = w.shape
    dev = x.device
    if Lout is None:
        Lout = L
    with torch.cuda.device(dev):
        x, w = _lib.f32c(x), _lib.f32c(w)
        b = _lib.f32c(bias) if bias is not None else None
        y = torch.empty(B, Lout, Cout, device=dev)
        _lib.check(_lib.lib().svc_op_conv1d(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), B, L, Cin, Cout, k,
                                            dilation, stride, pad_left, Lout, pad_mode, {"f16": 0, "f32": 1, "f16x3": 2}[dtype],
                                            _lib.stream_ptr()))
    return y


def conv_transpose1d_cl(x, w, bias, stride, dtype="f32"):
    """x (B, L, Cin), w (Cin, Cout, k = 2*stride) -> (B, L*stride, Cout); padding = stride // 2."""
    B, L, Cin = x.shape
    _, Cout, k = w.shape
    dev = x.device
    with torch.cuda.device(dev):
        x, w = _lib.f32c(x), _lib.f32c(w)
        b = _lib.f32c(bias) if bias is not None else None
        y = torch.empty(B, L * stride, Cout, device=dev)
        _lib.check(_lib.lib().svc_op_conv_transpose1d(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), B, L, Cin, Cout,
                                                      k, stride, {"f16": 0, "f32": 1, "f16x3": 2}[dtype], _lib.stream_ptr()))
    return y


def attention(q, k, v, kv_lens=None):
    """q,k,v (N, T, H, 64) fp32 -> softmax(q k^T / 8, keys < kv_len) v."""
    N, T, H, hd = q.shape
    assert hd == 64
    dev = q.device
    with torch.cuda.device(dev):
        q, k, v = _lib.f32c(q), _lib.f32c(k), _lib.f32c(v)
        out = torch.empty_like(q)
        lens = _lib.i64_host(kv_lens)
        _lib.check(_lib.lib().svc_op_attention(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(out), N, T, H, lens,
                                               _lib.stream_ptr()))
    return out


def rmsnorm(x, gamma, w=None, b=None, add_one=False):
    rows, D = x.shape
    dev = x.device
    with torch.cuda.device(dev):
        x, gamma = _lib.f32c(x), _lib.f32c(gamma)
        w = _lib.f32c(w) if w is not None else None
        b = _lib.f32c(b) if b is not None else None
        y = torch.empty_like(x)
        _lib.check(_lib.lib().svc_op_rmsnorm(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(w), _lib.ptr(b), int(add_one),
                                             _lib.ptr(y), rows, D, _lib.stream_ptr()))
    return y
