"""The drop-in shim (seedvc_amd/shim.py) executed end to end on stand-in `nn.Module`s that look to the shim exactly like
the reference's loaded modules (class names / module paths, attributes, `state_dict()` keys from `specs`): every
`patch_*` / `wrap_*` path must give the same result as the direct HIP path, bit for bit.

Reference seams: inference.py:483-506 (cfm.inference / vocoder_fn), seed_vc_wrapper.py:575-603, modules/v2/vc_wrapper.py
(v2 CFM), modules/length_regulator.py:90, modules/bigvgan/alias_free_activation/cuda/activation1d.py:23-25."""
import sys

import pytest
import torch
from torch import nn

import cases
from seedvc_amd import shim, specs, weights

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def module_from_state_dict(sd, cls=nn.Module, **attrs):
    """nn.Module tree whose state_dict() reproduces `sd` key for key (dots = submodules), on the GPU."""
    root = cls.__new__(cls)
    nn.Module.__init__(root)
    for key, val in sd.items():
        parts = key.split(".")
        m = root
        for p in parts[:-1]:
            if p not in m._modules:
                m.add_module(p, nn.Module())
            m = m._modules[p]
        if val.is_floating_point():
            m.register_parameter(parts[-1], nn.Parameter(val.clone(), requires_grad=False))
        else:
            m.register_buffer(parts[-1], val.clone())
    for k, v in attrs.items():
        setattr(root, k, v)
    return root.cuda()


class _RefCFM(nn.Module):
    """stands in for modules.flow_matching.CFM / modules.v2.cfm.CFM: `.estimator` + `.inference`"""

    def inference(self, *a, **k):
        raise AssertionError("the reference sampler must not run once patched")


def _cfm_standin(cfg, sd, v2):
    est_cls = type("DiT", (nn.Module,), {"__module__": "modules.v2.dit_wrapper" if v2 else "modules.diffusion_transformer"})
    attrs = {}
    if v2:
        attrs = dict(num_heads=cfg["H"], in_channels=cfg["C"], content_dim=cfg["Dc"], time_as_token=cfg["time_as_token"],
                     style_as_token=cfg["style_as_token"], uvit_skip_connection=cfg["uvit"])
    ref = _RefCFM()
    ref.estimator = module_from_state_dict(sd, est_cls, **attrs)
    return ref.cuda()


def _model_params(cfg):
    """the preset YAML's `model_params` as the reference passes it around (plain dicts are accepted by the shim)"""
    mp = {"DiT": dict(hidden_dim=cfg["D"], num_heads=cfg["H"], depth=cfg["L"], in_channels=cfg["C"], content_dim=cfg["Dc"],
                      final_layer_type=cfg["head"], time_as_token=cfg["time_as_token"], style_as_token=cfg["style_as_token"],
                      uvit_skip_connection=cfg["uvit"], long_skip_connection=cfg["long_skip"],
                      style_condition=cfg["style_condition"], content_codebook_size=cfg["codebook"],
                      zero_prompt_speech_token=False),
          "style_encoder": dict(dim=cfg["style_dim"])}
    if cfg["head"] == "wavenet":
        mp["wavenet"] = dict(hidden_dim=cfg["wn_dim"], num_layers=cfg["wn_layers"], kernel_size=cfg["wn_kernel"],
                             dilation_rate=cfg["wn_dilation"])
    return mp


@pytest.mark.parametrize("name", ["tiny_r", "small_r", "v2_r"])
def test_patch_cfm_equals_direct_path(name, golden):
    from seedvc_amd.cfm import CFM
    cfg, sd, inp, meta = cases.dit_case(name)
    v2 = cfg["version"] == 2
    ref = _cfm_standin(cfg, sd, v2)
    shim.patch_cfm(ref, None if v2 else _model_params(cfg))
    lens = torch.LongTensor([meta["T"]])
    torch.manual_seed(5)
    z = torch.randn(1, cfg["C"], meta["T"], device="cuda")
    torch.manual_seed(5)          # the patched method draws its noise with torch.randn like the reference (flow_matching.py:50)
    if v2:
        out = ref.inference(inp["mu"].cuda(), lens, inp["prompt"].cuda(), inp["style"].cuda(), meta["n_steps"],
                            inference_cfg_rate=meta["cfg_rate"])
    else:
        out = ref.inference(inp["mu"].cuda(), lens, inp["prompt"].cuda(), inp["style"].cuda(), None, meta["n_steps"],
                            inference_cfg_rate=meta["cfg_rate"])
    direct = CFM(cfg, sd, "cuda:0").inference(inp["mu"].cuda(), lens, inp["prompt"].cuda(), inp["style"].cuda(), None,
                                              meta["n_steps"], inference_cfg_rate=meta["cfg_rate"], z=z)
    assert out.shape == direct.shape and torch.equal(out, direct)
    assert ref.estimator.state_dict().keys() == sd.keys()


def test_patch_cfm_rejects_zero_prompt_speech_token():
    cfg, sd, inp, meta = cases.dit_case("tiny_r")
    mp = _model_params(cfg)
    mp["DiT"]["zero_prompt_speech_token"] = True          # flow_matching.py:79-80: not reproduced -> must fail loudly
    with pytest.raises(ValueError):
        shim.patch_cfm(_cfm_standin(cfg, sd, False), mp)


def test_wrap_vocoder_bigvgan_and_hift(golden):
    from seedvc_amd.vocoder import BigVGAN, HiFT
    h, sd, mel, meta = cases.bigvgan_case("bigvgan_r")
    ref = module_from_state_dict(sd, type("BigVGAN", (nn.Module,), {}), h=dict(h))
    voc = shim.wrap_vocoder(ref)
    y = voc(mel.cuda())
    assert torch.equal(y, BigVGAN(h, sd, "cuda:0")(mel.cuda()))
    assert (y.cpu() - torch.from_numpy(golden["bigvgan_r.wave"])).pow(2).mean().sqrt().item() < 1e-4

    c, sdh, melh, phase0, noise, _ = cases.hift_case("hift_full")
    refh = module_from_state_dict(sdh, type("HiFTGenerator", (nn.Module,), {}), sampling_rate=c["sampling_rate"],
                                  nb_harmonics=c["nb_harmonics"], lrelu_slope=c["lrelu_slope"], audio_limit=c["audio_limit"])
    voch = shim.wrap_vocoder(refh)
    assert voch.cfg["base_channels"] == c["base_channels"] and voch.cfg["f0_cond_channels"] == c["f0_cond_channels"]
    kw = dict(phase0=phase0.cuda(), noise=noise.cuda())
    assert torch.equal(voch(melh.cuda(), **kw), HiFT(c, sdh, "cuda:0")(melh.cuda(), **kw))


@pytest.mark.parametrize("name", ["lr_tiny_r", "lr_base_r", "lr_v2_r"])
def test_patch_length_regulator_equals_direct_path(name, golden):
    from seedvc_amd.length_regulator import InterpolateRegulator
    c, sd, x, ylen, f0, meta = cases.lr_case(name)

    if c["version"] == 1:
        class InterpolateRegulatorV1(nn.Module):
            def __init__(self, vector_quantize=False):      # v1's constructor has this argument, v2's does not
                super().__init__()
        cls = InterpolateRegulatorV1
    else:
        class InterpolateRegulatorV2(nn.Module):
            def __init__(self):
                super().__init__()
        cls = InterpolateRegulatorV2
    ref = module_from_state_dict(sd, cls, sampling_ratios=[1] * c["n_convs"], is_discrete=c["is_discrete"],
                                 f0_condition=c["f0_condition"], n_f0_bins=c["n_f0_bins"], n_codebooks=1)
    got = shim.lr_cfg_from_module(ref)
    for k in ("version", "channels", "is_discrete", "n_convs", "f0_condition", "out_channels"):
        assert got[k] == c[k], (k, got[k], c[k])
    shim.patch_length_regulator(ref)
    ylens = torch.LongTensor([ylen])
    xin = x.cuda()
    f0d = None if f0 is None else f0.cuda()
    if c["version"] == 1:
        out = ref.forward(xin, ylens=ylens, n_quantizers=3, f0=f0d)
        direct = InterpolateRegulator(c, sd, "cuda:0")(xin, ylens=ylens, n_quantizers=3, f0=f0d)
    else:
        out = ref.forward(xin, ylens=ylens, f0=f0d)
        direct = InterpolateRegulator(c, sd, "cuda:0")(xin, ylens=ylens, f0=f0d)
    assert torch.equal(out[0], direct[0])
    assert (out[0].cpu() - torch.from_numpy(golden[name + ".out"])).abs().max().item() < 5e-5


def test_patch_activation1d_routes_the_cuda_seam_to_hip(golden):
    from seedvc_amd import ops
    mod = shim.patch_activation1d()
    load = sys.modules["modules.bigvgan.alias_free_activation.cuda.load"]
    assert load.load() is mod and mod.forward is ops.anti_alias_activation_forward
    x, alpha, beta = cases.act_case("act_mid")
    filt = torch.from_numpy(golden["act_mid.filter"])
    y = load.load().forward(x.cuda(), filt.cuda(), filt.cuda(), alpha.cuda(), beta.cuda())      # activation1d.py:23-25 call form
    assert (y.cpu() - torch.from_numpy(golden["act_mid.snakebeta"])).abs().max().item() < 1e-5


def test_wrap_campplus_equals_direct_path(golden):
    from seedvc_amd.campplus import CAMPPlus
    c, sd, feat = cases.campplus_case("campplus_r")
    ref = module_from_state_dict(sd, type("CAMPPlus", (nn.Module,), {}))
    hip = shim.wrap_campplus(ref)
    for k in ("feat_dim", "embedding_size", "growth_rate", "bn_size", "init_channels"):
        assert hip.cfg[k] == c[k], k
    assert tuple(hip.cfg["block_layers"]) == tuple(c["block_layers"])
    e = hip(feat.cuda())
    assert torch.equal(e, CAMPPlus(c, sd, "cuda:0")(feat.cuda()))
    assert (e.cpu() - torch.from_numpy(golden["campplus_r.emb"])).abs().max().item() < 2e-5


def test_bigvgan_accepts_weight_norm_parametrisation(golden):
    """a BigVGAN whose `remove_weight_norm()` was NOT called hands over `weight_g` / `weight_v`
    (bigvgan.py:413-492 loads them that way; inference.py:108-110 removes them afterwards).  Both forms must give the same
    waveform: w = g * v / ||v|| over dim 0 (torch.nn.utils.weight_norm, dim=0), folded at pack time (model_util.h)."""
    from seedvc_amd.vocoder import BigVGAN
    h, sd, mel, meta = cases.bigvgan_case("bigvgan_r")
    spec_wn = specs.bigvgan_state_spec(h, weight_norm_removed=False)
    sd_wn = {}
    n_split = 0
    for k, shape in spec_wn.items():
        if k.endswith(".weight_g"):
            w = sd[k[:-len("_g")]]
            a = 0.5 + cases.rand("wn." + k, 7, w.shape[0]).reshape(-1, *([1] * (w.dim() - 1)))      # any positive scale of v
            sd_wn[k] = w.flatten(1).norm(dim=1).reshape(shape)
            sd_wn[k[:-2] + "_v"] = w * a
            n_split += 1
        elif not k.endswith(".weight_v"):
            sd_wn[k] = sd[k]
    assert n_split > 10 and set(sd_wn) == set(spec_wn) and not any(k.endswith(".weight") for k in sd_wn if "resblocks" in k and "convs" in k)
    y_removed = BigVGAN(h, sd, "cuda:0")(mel.cuda())
    y_wn = BigVGAN(h, sd_wn, "cuda:0")(mel.cuda())
    rms = (y_wn - y_removed).pow(2).mean().sqrt().item()
    print(f"BigVGAN weight_g/weight_v vs folded weights: waveform RMS {rms:.3e}")
    assert rms < 2e-6                                            # fp32 rounding of g * v / ||v|| only
    assert (y_wn.cpu() - torch.from_numpy(golden["bigvgan_r.wave"])).pow(2).mean().sqrt().item() < 1e-4
    ref = module_from_state_dict(sd_wn, type("BigVGAN", (nn.Module,), {}), h=dict(h))               # through the shim as well
    assert torch.equal(shim.wrap_vocoder(ref)(mel.cuda()), y_wn)
