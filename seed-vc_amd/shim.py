"""Drop-in shim: routes an already-loaded reference model through the HIP path without editing the reference.

Usage inside the reference tree, after `load_models()` (inference.py:40-337):

    import seedvc_amd.shim as shim
    shim.patch_cfm(model.cfm, config["model_params"])        # model.cfm.inference(...) now runs on libseedvc_hip.so
    vocoder_fn = shim.wrap_vocoder(vocoder_fn)               # BigVGAN / HiFTGenerator module -> HIP vocoder
    shim.patch_activation1d()                                # optional: the reference's own CUDA-extension seam
    shim.patch_length_regulator(model.length_regulator)      # content -> mu on the HIP path as well (8f row 1)
    campplus_model = shim.wrap_campplus(campplus_model)      # style encoder (+ .fbank / .style for the Kaldi front-end)

Checkpoint loading stays in the reference (`build_model` + `load_checkpoint`, `BigVGAN.from_pretrained`,
`hift_gen.load_state_dict`); the shim only reads `module.state_dict()` and hyper-parameters.
The call signatures are unchanged; `setup_caches` stays callable.
"""
import types

import torch

from . import specs
from .cfm import CFM
from .vocoder import BigVGAN, HiFT


def dit_cfg_from_reference_args(model_params):
    """Munch / dict from the preset YAML (`model_params`) -> specs.dit_config()."""
    g = (lambda o, k, d=None: (o.get(k, d) if isinstance(o, dict) else getattr(o, k, d)))
    dit = g(model_params, "DiT")
    wn = g(model_params, "wavenet")
    cfg = dict(version=1, D=g(dit, "hidden_dim"), H=g(dit, "num_heads"), L=g(dit, "depth"), C=g(dit, "in_channels"),
               Dc=g(dit, "content_dim"), style_dim=g(g(model_params, "style_encoder"), "dim"),
               head=g(dit, "final_layer_type"), time_as_token=bool(g(dit, "time_as_token", False)),
               style_as_token=bool(g(dit, "style_as_token", False)), uvit=bool(g(dit, "uvit_skip_connection", False)),
               long_skip=bool(g(dit, "long_skip_connection", False)), style_condition=bool(g(dit, "style_condition", True)),
               codebook=g(dit, "content_codebook_size", 1024), hd=g(dit, "hidden_dim") // g(dit, "num_heads"))
    if cfg["head"] == "wavenet":
        cfg.update(wn_dim=g(wn, "hidden_dim"), wn_layers=g(wn, "num_layers"), wn_kernel=g(wn, "kernel_size"),
                   wn_dilation=g(wn, "dilation_rate"))
    if bool(g(dit, "zero_prompt_speech_token", False)):
        # flow_matching.py:79-80 zeroes mu[..., :prompt_len] (a last-dim slice of the (B, T, C) condition); no shipped preset
        # sets it and the HIP sampler does not reproduce it: refuse instead of silently computing something else
        raise ValueError("DiT.zero_prompt_speech_token=True is not supported by the HIP sampler")
    cfg["name"] = "reference"
    cfg["I"] = specs.ffn_dim(cfg["D"])
    cfg["n_prefix"] = int(cfg["time_as_token"]) + int(cfg["style_as_token"])
    return cfg


def dit_cfg_from_v2_module(estimator):
    """modules.v2.dit_wrapper.DiT instance (configs/v2/vc_wrapper.yaml:15-31) -> specs.dit_config()-style dict, read from
    the module's own attributes and weights."""
    sd = estimator.state_dict()
    D = sd["cond_projection.weight"].shape[0]
    H = int(estimator.num_heads)
    L = sum(1 for k in sd if k.startswith("transformer.layers.") and k.endswith("attention.wqkv.weight"))
    return specs.dit_config("v2", D=D, H=H, L=L, C=int(estimator.in_channels), Dc=int(estimator.content_dim),
                            style_dim=sd["style_in.weight"].shape[1], time_as_token=bool(estimator.time_as_token),
                            style_as_token=bool(estimator.style_as_token), uvit=bool(estimator.uvit_skip_connection))


def patch_cfm(ref_cfm, model_params=None, device=None):
    """Replace `ref_cfm.inference` (modules/flow_matching.py:30; v2 modules/v2/cfm.py:16) by the HIP sampler; same
    signature.  v1: pass the preset's `model_params`; v2 (hydra-built `modules.v2.cfm.CFM`): the configuration is read
    from the estimator module itself."""
    device = device or next(ref_cfm.parameters()).device
    is_v2 = type(ref_cfm.estimator).__module__.endswith("v2.dit_wrapper")
    cfg = dit_cfg_from_v2_module(ref_cfm.estimator) if is_v2 else dit_cfg_from_reference_args(model_params)
    hip = CFM(cfg, ref_cfm.estimator.state_dict(), device)

    if is_v2:
        def inference(self, mu, x_lens, prompt, style, n_timesteps, temperature=1.0, inference_cfg_rate=(0.5, 0.5),
                      random_voice=False):
            return hip.inference(mu, x_lens, prompt, style, None, n_timesteps, temperature, inference_cfg_rate,
                                 random_voice=random_voice)
    else:
        def inference(self, mu, x_lens, prompt, style, f0, n_timesteps, temperature=1.0, inference_cfg_rate=0.5):
            return hip.inference(mu, x_lens, prompt, style, f0, n_timesteps, temperature, inference_cfg_rate)

    ref_cfm.inference = types.MethodType(inference, ref_cfm)
    ref_cfm._seedvc_hip = hip
    return ref_cfm


def wrap_vocoder(module, device=None, precision="fp16p8"):
    """BigVGAN / HiFTGenerator nn.Module (weights loaded) -> callable with the same `vocoder_fn(mel)` contract."""
    device = device or next(module.parameters()).device
    name = type(module).__name__
    if name == "BigVGAN":
        h = dict(module.h)
        return BigVGAN(h, module.state_dict(), device, precision)
    if name == "HiFTGenerator":
        sd = module.state_dict()
        cfg = specs.hift_config(base_channels=sd["conv_pre.bias"].numel(),
                                f0_cond_channels=sd["f0_predictor.classifier.weight"].shape[1],
                                sampling_rate=module.sampling_rate, nb_harmonics=module.nb_harmonics,
                                lrelu_slope=module.lrelu_slope, audio_limit=module.audio_limit)
        return HiFT(cfg, sd, device, precision)
    raise ValueError(f"unsupported vocoder module {name}")


def lr_cfg_from_module(module):
    """InterpolateRegulator nn.Module (v1 or v2) -> specs.lr_config()-style dict, from its own attributes / weights."""
    sd = module.state_dict()
    C = sd["mask_token"].shape[1]
    n_convs = len(module.sampling_ratios)
    tail = f"model.{3 * n_convs}.weight"
    v2 = "vector_quantize" not in type(module).__init__.__code__.co_varnames
    cfg = dict(version=2 if v2 else 1, channels=C, is_discrete=bool(module.is_discrete), n_convs=n_convs,
               codebook_size=sd["embedding.weight"].shape[0], f0_condition=bool(module.f0_condition),
               n_f0_bins=int(getattr(module, "n_f0_bins", 512)),
               in_channels=0 if module.is_discrete else sd["content_in_proj.weight"].shape[1],
               out_channels=sd[tail].shape[0] if tail in sd else C)
    if getattr(module, "n_codebooks", 1) != 1 or hasattr(module, "vq"):
        raise ValueError("multi-codebook / vector-quantised regulators are not on the inference path of the presets")
    return cfg


def patch_length_regulator(module, device=None):
    """Replace `module.forward` (modules/length_regulator.py:90, v2 modules/v2/length_regulator.py:74) by the HIP engine;
    same arguments and return tuple."""
    from .length_regulator import InterpolateRegulator
    device = device or next(module.parameters()).device
    hip = InterpolateRegulator(lr_cfg_from_module(module), module.state_dict(), device)
    if hip.cfg["version"] == 1:
        module.forward = lambda x, ylens=None, n_quantizers=None, f0=None: hip(x, ylens=ylens, n_quantizers=n_quantizers, f0=f0)
    else:
        module.forward = lambda x, ylens=None, f0=None: hip(x, ylens=ylens, f0=f0)
    module._seedvc_hip = hip
    return module


def campplus_cfg_from_module(module):
    """specs.campplus_config of a loaded `CAMPPlus` module, read from its own state dict (block depths, growth rate, ...)."""
    sd = module.state_dict()
    layers = []
    for b in range(1, 5):
        n = sum(1 for k in sd if k.startswith(f"xvector.block{b}.") and k.endswith(".nonlinear1.batchnorm.weight"))
        if n:
            layers.append(n)
    first = sd["xvector.block1.tdnnd1.cam_layer.linear_local.weight"]
    return specs.campplus_config(feat_dim=8 * sd["xvector.tdnn.linear.weight"].shape[1] // sd["head.conv1.weight"].shape[0],
                                 embedding_size=sd["dense.linear.weight"].shape[0], growth_rate=first.shape[0],
                                 bn_size=first.shape[1] // first.shape[0], init_channels=sd["xvector.tdnn.linear.weight"].shape[0],
                                 block_layers=tuple(layers))


def wrap_campplus(module, device=None):
    """`campplus_model` (modules/campplus/DTDNN.py CAMPPlus, weights loaded, inference.py:98-101) -> HIP style encoder with
    the same `campplus_model(feat)` call; the block structure is read from the module's own state dict."""
    from .campplus import CAMPPlus
    device = device or next(module.parameters()).device
    return CAMPPlus(campplus_cfg_from_module(module), module.state_dict(), device)


def make_mel_fn(mel_fn_args, device="cuda:0"):
    """`to_mel = make_mel_fn(mel_fn_args)` replaces `lambda x: mel_spectrogram(x, **mel_fn_args)` (inference.py:315-327).
    The filterbank comes from librosa when the reference environment has it (as modules/audio.py:55 does)."""
    from .audio import MelSpectrogram
    basis = None
    try:
        from librosa.filters import mel as librosa_mel_fn
        basis = torch.from_numpy(librosa_mel_fn(sr=mel_fn_args["sampling_rate"], n_fft=mel_fn_args["n_fft"],
                                                n_mels=mel_fn_args["num_mels"], fmin=mel_fn_args["fmin"],
                                                fmax=None if mel_fn_args["fmax"] in (None, "None") else mel_fn_args["fmax"])).float()
    except Exception:
        basis = None
    return MelSpectrogram(mel_basis=basis, device=device, **mel_fn_args)


def patch_activation1d():
    """Point the reference's fused-activation wrapper (alias_free_activation/cuda/activation1d.py:23-25) at the HIP
    kernel instead of the nvcc JIT extension.  Must run before `BigVGAN(h, use_cuda_kernel=True)` is constructed."""
    import sys
    from . import ops
    mod = types.ModuleType("anti_alias_activation_cuda")
    mod.forward = ops.anti_alias_activation_forward
    load = types.ModuleType("modules.bigvgan.alias_free_activation.cuda.load")
    load.load = lambda: mod
    sys.modules["modules.bigvgan.alias_free_activation.cuda.load"] = load
    return mod
