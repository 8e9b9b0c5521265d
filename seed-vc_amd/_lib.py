"""ctypes binding of libseedvc_hip.so (C ABI in include/seedvc_hip.h).

PyTorch is used only for device memory and streams: tensors are passed as raw device pointers and
the current HIP stream handle.  There is NO CPU fallback: a missing library raises ImportError here,
and every entry point raises RuntimeError with the library's message on failure.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libseedvc_hip.so")

EXPORTS = [
    "svc_abi_version", "svc_last_error",
    "svc_dit_create", "svc_dit_destroy", "svc_dit_set_microbatch", "svc_dit_set_fused_min_rows", "svc_dit_fused_available", "svc_dit_set_graphs", "svc_cfm_sample", "svc_dit_forward",
    "svc_bigvgan_create", "svc_bigvgan_destroy", "svc_bigvgan_forward", "svc_bigvgan_set_microbatch",
    "svc_hift_create", "svc_hift_destroy", "svc_hift_forward", "svc_hift_set_microbatch",
    "svc_anti_alias_act_fwd",
    "svc_ar_create", "svc_ar_destroy", "svc_ar_reset", "svc_ar_forward_generate", "svc_ar_decode_step", "svc_ar_sample", "svc_ar_generate",
    "svc_lr_create", "svc_lr_destroy", "svc_lr_forward", "svc_crossfade",
    "svc_campplus_create", "svc_campplus_destroy", "svc_campplus_forward", "svc_kaldi_fbank_frames", "svc_kaldi_fbank",
    "svc_mel_create", "svc_mel_destroy", "svc_mel_frames", "svc_mel_forward",
    "svc_prof_enable", "svc_prof_collect",
    "svc_op_linear", "svc_op_conv1d", "svc_op_conv_transpose1d", "svc_op_attention", "svc_op_rmsnorm",
]


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int), ("shape", C.c_int64 * 4)]


class DitConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "version", "hidden_dim", "num_heads", "depth", "in_channels", "content_dim", "style_dim",
        "final_layer_type", "time_as_token", "style_as_token", "uvit_skip_connection", "long_skip_connection",
        "style_condition", "wn_hidden_dim", "wn_num_layers", "wn_kernel_size", "wn_dilation_rate")]


class CfmArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("P", C.c_int),
                ("mu", C.c_void_p), ("prompt", C.c_void_p), ("style", C.c_void_p), ("z", C.c_void_p),
                ("x_lens", C.POINTER(C.c_int64)), ("prompt_lens", C.POINTER(C.c_int64)),
                ("n_timesteps", C.c_int), ("temperature", C.c_float), ("cfg_rate", C.c_float * 2),
                ("random_voice", C.c_int), ("out", C.c_void_p)]


class BigVGANConfig(C.Structure):
    _fields_ = [("num_mels", C.c_int), ("upsample_initial_channel", C.c_int), ("num_upsamples", C.c_int),
                ("num_kernels", C.c_int), ("upsample_rates", C.c_int * 8), ("upsample_kernel_sizes", C.c_int * 8),
                ("resblock_kernel_sizes", C.c_int * 4), ("resblock_dilation_sizes", (C.c_int * 3) * 4),
                ("use_tanh_at_final", C.c_int), ("use_bias_at_final", C.c_int), ("snake_logscale", C.c_int),
                ("snakebeta", C.c_int), ("precision", C.c_int)]


class HiftConfig(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("base_channels", C.c_int), ("nb_harmonics", C.c_int),
                ("sampling_rate", C.c_int), ("nsf_alpha", C.c_float), ("nsf_sigma", C.c_float),
                ("nsf_voiced_threshold", C.c_float), ("num_upsamples", C.c_int), ("upsample_rates", C.c_int * 4),
                ("upsample_kernel_sizes", C.c_int * 4), ("istft_n_fft", C.c_int), ("istft_hop", C.c_int),
                ("num_kernels", C.c_int), ("resblock_kernel_sizes", C.c_int * 4),
                ("resblock_dilation_sizes", (C.c_int * 3) * 4), ("source_resblock_kernel_sizes", C.c_int * 4),
                ("source_resblock_dilation_sizes", (C.c_int * 3) * 4), ("lrelu_slope", C.c_float),
                ("audio_limit", C.c_float), ("f0_cond_channels", C.c_int), ("precision", C.c_int)]


class ArConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("dim", "n_head", "n_local_heads", "head_dim", "n_layer", "intermediate_size",
                                       "vocab_size", "max_seq_len")] + [("rope_base", C.c_float), ("norm_eps", C.c_float)]


class CampplusConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("feat_dim", "embedding_size", "growth_rate", "bn_size", "init_channels", "m_channels", "n_blocks")] + \
               [("block_layers", C.c_int * 4), ("block_kernel", C.c_int * 4), ("block_dilation", C.c_int * 4), ("seg_len", C.c_int)]


class LrConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("channels", "in_channels", "out_channels", "is_discrete", "codebook_size", "n_convs",
                                       "interpolate", "has_final_conv", "f0_condition", "n_f0_bins")]


_lib = None


def lib():
    """Loads the shared library (once).  No fallback: the HIP path is the product."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              f"(hipcc --offload-arch=gfx950); there is no CPU fallback")
        l = C.CDLL(LIB_PATH)
        l.svc_last_error.restype = C.c_char_p
        if l.svc_abi_version() != 1:
            raise ImportError("libseedvc_hip.so ABI version mismatch")
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("seedvc_hip: " + (lib().svc_last_error() or b"unknown error").decode())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def f32c(t, device=None):
    """contiguous fp32 device tensor"""
    t = t.detach()
    if device is not None:
        t = t.to(device)
    return t.to(torch.float32).contiguous()


def make_descs(state_dict, device):
    """state_dict -> (TensorDesc array, keep-alive list). Non-float entries (masks, index buffers) are skipped."""
    keep, items = [], []
    for k, v in state_dict.items():
        if not torch.is_tensor(v) or not (v.is_floating_point()):
            continue
        if v.dim() > 4:
            continue
        t = f32c(v, device)
        keep.append(t)
        items.append((k, t))
    arr = (TensorDesc * len(items))()
    for i, (k, t) in enumerate(items):
        kb = k.encode()
        keep.append(kb)
        arr[i].name = kb
        arr[i].data = t.data_ptr()
        arr[i].ndim = t.dim()
        for j in range(4):
            arr[i].shape[j] = t.shape[j] if j < t.dim() else 1
    return arr, len(items), keep


def i64_host(values):
    if values is None:
        return None
    vals = [int(v) for v in values]
    return (C.c_int64 * len(vals))(*vals)
