"""Model shape specs for the hot path (DiT estimator variants + vocoders).

These mirror the hyper-parameters the reference reads from its YAML/JSON configs
(reference: configs/presets/*.yml, configs/v2/vc_wrapper.yaml, configs/hifigan.yml,
modules/bigvgan/config.json) and list, for each model, the state_dict keys and shapes
that the reference modules expose (reference: modules/diffusion_transformer.py:407-483,
modules/wavenet.py:103-136, modules/bigvgan/bigvgan.py:266-358,
modules/hifigan/generator.py:287-377, modules/hifigan/f0_predictor.py:22-49,
modules/v2/dit_wrapper.py:59-112, modules/v2/dit_model.py:20-135).

They exist so that (1) the GPU box can regenerate test weights without the reference,
(2) the packer knows what to ingest.  `tests/golden/make_golden.py` checks every key/shape
against the real reference modules.
"""
from collections import OrderedDict
from copy import deepcopy


def find_multiple(n, k):
    return n if n % k == 0 else n + k - (n % k)


def ffn_dim(d):
    # reference: modules/diffusion_transformer.py:71-74
    return find_multiple(int(2 * (4 * d) / 3), 256)


# --------------------------------------------------------------------------- DiT (v1 / v2)
DIT_PRESETS = {
    # reference: configs/presets/config_dit_mel_seed_uvit_xlsr_tiny.yml:57-79
    "tiny": dict(version=1, D=384, H=6, L=9, C=80, Dc=384, style_dim=192, head="mlp",
                 time_as_token=True, style_as_token=True, uvit=True, long_skip=False,
                 style_condition=True, codebook=1024),
    # reference: configs/presets/config_dit_mel_seed_uvit_whisper_small_wavenet.yml:56-86
    "small": dict(version=1, D=512, H=8, L=13, C=80, Dc=512, style_dim=192, head="wavenet",
                  time_as_token=False, style_as_token=False, uvit=True, long_skip=True,
                  style_condition=True, codebook=1024,
                  wn_dim=512, wn_layers=8, wn_kernel=5, wn_dilation=1),
    # reference: configs/presets/config_dit_mel_seed_uvit_whisper_base_f0_44k.yml:63-93
    "base": dict(version=1, D=768, H=12, L=17, C=128, Dc=768, style_dim=192, head="mlp",
                 time_as_token=False, style_as_token=False, uvit=True, long_skip=False,
                 style_condition=True, codebook=1024),
    # reference: configs/v2/vc_wrapper.yaml:15-31
    "v2": dict(version=2, D=512, H=8, L=13, C=80, Dc=512, style_dim=192, head="mlp",
               time_as_token=True, style_as_token=True, uvit=False, long_skip=False,
               style_condition=True, block_size=8192),
}


def dit_config(name, **overrides):
    """Return a DiT config dict; `overrides` lets tests build reduced-size variants."""
    cfg = deepcopy(DIT_PRESETS[name])
    cfg["name"] = name
    cfg.update(overrides)
    cfg.setdefault("hd", 64)
    assert cfg["D"] == cfg["H"] * cfg["hd"], "hidden = heads * head_dim"
    cfg["I"] = ffn_dim(cfg["D"])
    cfg["n_prefix"] = int(cfg["time_as_token"]) + int(cfg["style_as_token"])
    return cfg


def dit_merge_in_dim(cfg):
    # reference: modules/diffusion_transformer.py:478-480 ; v2: modules/v2/dit_wrapper.py:111
    k = cfg["D"] + 2 * cfg["C"]
    if cfg["version"] == 1 and cfg["style_condition"] and not cfg["style_as_token"]:
        k += cfg["style_dim"]
    return k


def dit_state_spec(cfg):
    """name -> shape of `CFM.estimator.state_dict()` (v1) / v2 `DiT.state_dict()`."""
    D, I, C, L = cfg["D"], cfg["I"], cfg["C"], cfg["L"]
    s = OrderedDict()
    v2 = cfg["version"] == 2
    for i in range(L):
        p = f"transformer.layers.{i}."
        s[p + "attention.wqkv.weight"] = (3 * D, D)
        s[p + "attention.wo.weight"] = (D, D)
        s[p + "feed_forward.w1.weight"] = (I, D)
        s[p + "feed_forward.w3.weight"] = (I, D)
        s[p + "feed_forward.w2.weight"] = (D, I)
        if v2:
            s[p + "ffn_norm.weight"] = (D,)
            s[p + "attention_norm.linear.weight"] = (6 * D, D)
            s[p + "attention_norm.linear.bias"] = (6 * D,)
            s[p + "attention_norm.norm.weight"] = (D,)
        else:
            for n in ("ffn_norm", "attention_norm"):
                s[p + n + ".project_layer.weight"] = (2 * D, D)
                s[p + n + ".project_layer.bias"] = (2 * D,)
                s[p + n + ".norm.weight"] = (D,)
            if cfg["uvit"]:
                s[p + "skip_in_linear.weight"] = (D, 2 * D)
                s[p + "skip_in_linear.bias"] = (D,)
    if v2:
        s["transformer.freqs_cis"] = (cfg["block_size"], cfg["hd"] // 2, 2)
        s["transformer.causal_mask"] = (cfg["block_size"], cfg["block_size"])
        s["transformer.norm.linear.weight"] = (2 * D, D)
        s["transformer.norm.linear.bias"] = (2 * D,)
        s["transformer.norm.norm.weight"] = (D,)
    else:
        s["transformer.norm.project_layer.weight"] = (2 * D, D)
        s["transformer.norm.project_layer.bias"] = (2 * D,)
        s["transformer.norm.norm.weight"] = (D,)
    s["x_embedder.bias"] = (D,)
    s["x_embedder.weight_g"] = (D, 1)
    s["x_embedder.weight_v"] = (D, C)
    if not v2:
        s["cond_embedder.weight"] = (cfg["codebook"], D)
    s["cond_projection.weight"] = (D, cfg["Dc"])
    s["cond_projection.bias"] = (D,)
    if not v2:
        s["t_embedder.freqs"] = (128,)
    s["t_embedder.mlp.0.weight"] = (D, 256)
    s["t_embedder.mlp.0.bias"] = (D,)
    s["t_embedder.mlp.2.weight"] = (D, D)
    s["t_embedder.mlp.2.bias"] = (D,)
    if not v2:
        s["input_pos"] = (16384,)
    if cfg["head"] == "wavenet":
        W, nl, k = cfg["wn_dim"], cfg["wn_layers"], cfg["wn_kernel"]
        s["t_embedder2.freqs"] = (128,)
        s["t_embedder2.mlp.0.weight"] = (W, 256)
        s["t_embedder2.mlp.0.bias"] = (W,)
        s["t_embedder2.mlp.2.weight"] = (W, W)
        s["t_embedder2.mlp.2.bias"] = (W,)
        s["conv1.weight"] = (W, D)
        s["conv1.bias"] = (W,)
        s["conv2.weight"] = (C, W, 1)
        s["conv2.bias"] = (C,)
        s["wavenet.cond_layer.conv.conv.bias"] = (2 * W * nl,)
        s["wavenet.cond_layer.conv.conv.weight_g"] = (2 * W * nl, 1, 1)
        s["wavenet.cond_layer.conv.conv.weight_v"] = (2 * W * nl, W, 1)
        for i in range(nl):
            s[f"wavenet.in_layers.{i}.conv.conv.bias"] = (2 * W,)
            s[f"wavenet.in_layers.{i}.conv.conv.weight_g"] = (2 * W, 1, 1)
            s[f"wavenet.in_layers.{i}.conv.conv.weight_v"] = (2 * W, W, k)
        for i in range(nl):
            o = 2 * W if i < nl - 1 else W
            s[f"wavenet.res_skip_layers.{i}.conv.conv.bias"] = (o,)
            s[f"wavenet.res_skip_layers.{i}.conv.conv.weight_g"] = (o, 1, 1)
            s[f"wavenet.res_skip_layers.{i}.conv.conv.weight_v"] = (o, W, 1)
        s["final_layer.linear.bias"] = (W,)
        s["final_layer.linear.weight_g"] = (W, 1)
        s["final_layer.linear.weight_v"] = (W, W)
        s["final_layer.adaLN_modulation.1.weight"] = (2 * W, W)
        s["final_layer.adaLN_modulation.1.bias"] = (2 * W,)
        s["res_projection.weight"] = (W, D)
        s["res_projection.bias"] = (W,)
    else:
        s["final_mlp.0.weight"] = (D, D)
        s["final_mlp.0.bias"] = (D,)
        s["final_mlp.2.weight"] = (C, D)
        s["final_mlp.2.bias"] = (C,)
    if not v2:
        s["content_mask_embedder.weight"] = (1, D)
        s["skip_linear.weight"] = (D, D + C)
        s["skip_linear.bias"] = (D,)
    s["cond_x_merge_linear.weight"] = (D, dit_merge_in_dim(cfg))
    s["cond_x_merge_linear.bias"] = (D,)
    if v2 or cfg["style_as_token"]:
        s["style_in.weight"] = (D, cfg["style_dim"])
        s["style_in.bias"] = (D,)
    return s


# keys that the forward pass never reads (reference: SURVEY.md 8b) or that are not float weights
DIT_DEAD_KEYS = ("x_embedder.", "cond_embedder.", "content_mask_embedder.", "input_pos",
                 "transformer.causal_mask", "transformer.freqs_cis")


def uvit_layers(cfg):
    """(emit, receive) layer index lists; reference: modules/diffusion_transformer.py:105-107."""
    if not cfg["uvit"] or cfg["version"] == 2:   # v2 accepts the flag but never applies skips
        return [], []
    L = cfg["L"]
    return [i for i in range(L) if i < L // 2], [i for i in range(L) if i > L // 2]


# --------------------------------------------------------------------------- BigVGAN
BIGVGAN_PRESETS = {
    # reference: modules/bigvgan/config.json (nvidia/bigvgan_v2_22khz_80band_256x)
    "22k": dict(num_mels=80, upsample_initial_channel=1536, upsample_rates=[4, 4, 2, 2, 2, 2],
                upsample_kernel_sizes=[8, 8, 4, 4, 4, 4], resblock_kernel_sizes=[3, 7, 11],
                resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                use_tanh_at_final=False, use_bias_at_final=False, snake_logscale=True,
                activation="snakebeta", resblock="1"),
    # 44 kHz 128-band 512x: config is not in the reference tree; public
    # BigVGAN-v2 44k/512x hyper-parameters, parameterised here and confirmed when a local
    # checkpoint directory exists.
    "44k": dict(num_mels=128, upsample_initial_channel=1536, upsample_rates=[8, 4, 2, 2, 2, 2],
                upsample_kernel_sizes=[16, 8, 4, 4, 4, 4], resblock_kernel_sizes=[3, 7, 11],
                resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                use_tanh_at_final=False, use_bias_at_final=False, snake_logscale=True,
                activation="snakebeta", resblock="1"),
}


def bigvgan_config(name, **overrides):
    cfg = deepcopy(BIGVGAN_PRESETS[name])
    cfg["name"] = name
    cfg.update(overrides)
    return cfg


def bigvgan_state_spec(h, weight_norm_removed=True):
    """State dict of BigVGAN after `remove_weight_norm()` (reference: inference.py:108-110)."""
    s = OrderedDict()

    def conv(prefix, shape, bias=True, norm_dim0=None):
        if weight_norm_removed:
            s[prefix + ".weight"] = shape
        else:
            s[prefix + ".weight_g"] = (shape[0], 1, 1)
            s[prefix + ".weight_v"] = shape
        if bias:
            s[prefix + ".bias"] = (norm_dim0 if norm_dim0 is not None else shape[0],)

    c0 = h["upsample_initial_channel"]
    s_pre = OrderedDict()
    # torch orders parameters: bias first for weight-normed modules (weight_g/v registered later)
    if weight_norm_removed:
        s["conv_pre.bias"] = (c0,)
        s["conv_pre.weight"] = (c0, h["num_mels"], 7)
    else:
        s["conv_pre.bias"] = (c0,)
        s["conv_pre.weight_g"] = (c0, 1, 1)
        s["conv_pre.weight_v"] = (c0, h["num_mels"], 7)
    del s_pre
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        cin, cout = c0 // (2 ** i), c0 // (2 ** (i + 1))
        p = f"ups.{i}.0"
        s[p + ".bias"] = (cout,)
        if weight_norm_removed:
            s[p + ".weight"] = (cin, cout, k)
        else:
            s[p + ".weight_g"] = (cin, 1, 1)
            s[p + ".weight_v"] = (cin, cout, k)
    nk = len(h["resblock_kernel_sizes"])
    for i in range(len(h["upsample_rates"])):
        ch = c0 // (2 ** (i + 1))
        for j, (k, dil) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            p = f"resblocks.{i * nk + j}"
            for grp in ("convs1", "convs2"):
                for d in range(len(dil)):
                    q = f"{p}.{grp}.{d}"
                    s[q + ".bias"] = (ch,)
                    if weight_norm_removed:
                        s[q + ".weight"] = (ch, ch, k)
                    else:
                        s[q + ".weight_g"] = (ch, 1, 1)
                        s[q + ".weight_v"] = (ch, ch, k)
            for a in range(2 * len(dil)):
                q = f"{p}.activations.{a}"
                s[q + ".act.alpha"] = (ch,)
                if h["activation"] == "snakebeta":
                    s[q + ".act.beta"] = (ch,)
                s[q + ".upsample.filter"] = (1, 1, 12)
                s[q + ".downsample.lowpass.filter"] = (1, 1, 12)
    s["activation_post.act.alpha"] = (ch,)
    if h["activation"] == "snakebeta":
        s["activation_post.act.beta"] = (ch,)
    s["activation_post.upsample.filter"] = (1, 1, 12)
    s["activation_post.downsample.lowpass.filter"] = (1, 1, 12)
    if h["use_bias_at_final"]:
        s["conv_post.bias"] = (1,)
    if weight_norm_removed:
        s["conv_post.weight"] = (1, ch, 7)
    else:
        s["conv_post.weight_g"] = (1, 1, 1)
        s["conv_post.weight_v"] = (1, ch, 7)
    return s


def bigvgan_total_upsample(h):
    t = 1
    for u in h["upsample_rates"]:
        t *= u
    return t


# --------------------------------------------------------------------------- HiFT
# reference: configs/hifigan.yml
HIFT_PRESET = dict(in_channels=80, base_channels=512, nb_harmonics=8, sampling_rate=22050,
                   nsf_alpha=0.1, nsf_sigma=0.003, nsf_voiced_threshold=10,
                   upsample_rates=[8, 8], upsample_kernel_sizes=[16, 16],
                   istft_n_fft=16, istft_hop=4,
                   resblock_kernel_sizes=[3, 7, 11],
                   resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                   source_resblock_kernel_sizes=[7, 11],
                   source_resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5]],
                   lrelu_slope=0.1, audio_limit=0.99,
                   f0_cond_channels=512)


def hift_config(**overrides):
    cfg = deepcopy(HIFT_PRESET)
    cfg.update(overrides)
    return cfg


def hift_state_spec(c):
    """State dict of HiFTGenerator incl. its f0_predictor; weight-norm is KEPT at inference
    (reference: inference.py:113-122)."""
    s = OrderedDict()
    bc, nfft2 = c["base_channels"], c["istft_n_fft"] + 2
    s["m_source.l_linear.weight"] = (1, c["nb_harmonics"] + 1)
    s["m_source.l_linear.bias"] = (1,)
    s["conv_pre.bias"] = (bc,)
    s["conv_pre.weight_g"] = (bc, 1, 1)
    s["conv_pre.weight_v"] = (bc, c["in_channels"], 7)
    for i, (u, k) in enumerate(zip(c["upsample_rates"], c["upsample_kernel_sizes"])):
        cin, cout = bc // (2 ** i), bc // (2 ** (i + 1))
        s[f"ups.{i}.bias"] = (cout,)
        s[f"ups.{i}.weight_g"] = (cin, 1, 1)
        s[f"ups.{i}.weight_v"] = (cin, cout, k)
    # source downs; reference: generator.py:346-360
    ups = c["upsample_rates"]
    down_rates = [1] + ups[::-1][:-1]
    cum = []
    acc = 1
    for r in down_rates:
        acc *= r
        cum.append(acc)
    for i, u in enumerate(cum[::-1]):
        ch = bc // (2 ** (i + 1))
        if u == 1:
            s[f"source_downs.{i}.weight"] = (ch, nfft2, 1)
        else:
            s[f"source_downs.{i}.weight"] = (ch, nfft2, u * 2)
        s[f"source_downs.{i}.bias"] = (ch,)

    def resblock(p, ch, k, dil):
        for grp in ("convs1", "convs2"):
            for d in range(len(dil)):
                q = f"{p}.{grp}.{d}"
                s[q + ".bias"] = (ch,)
                s[q + ".weight_g"] = (ch, 1, 1)
                s[q + ".weight_v"] = (ch, ch, k)
        for grp in ("activations1", "activations2"):
            for d in range(len(dil)):
                s[f"{p}.{grp}.{d}.alpha"] = (ch,)

    # module registration order in the reference: source_downs / source_resblocks interleaved by
    # attribute (ModuleLists), so all source_downs come first, then source_resblocks.
    tmp = OrderedDict(s)
    s.clear()
    for k_, v_ in tmp.items():
        s[k_] = v_
    for i, (k, dil) in enumerate(zip(c["source_resblock_kernel_sizes"], c["source_resblock_dilation_sizes"])):
        resblock(f"source_resblocks.{i}", bc // (2 ** (i + 1)), k, dil)
    nk = len(c["resblock_kernel_sizes"])
    for i in range(len(ups)):
        ch = bc // (2 ** (i + 1))
        for j, (k, dil) in enumerate(zip(c["resblock_kernel_sizes"], c["resblock_dilation_sizes"])):
            resblock(f"resblocks.{i * nk + j}", ch, k, dil)
    s["conv_post.bias"] = (nfft2,)
    s["conv_post.weight_g"] = (nfft2, 1, 1)
    s["conv_post.weight_v"] = (nfft2, ch, 7)
    fc = c["f0_cond_channels"]
    for li, idx in enumerate((0, 2, 4, 6, 8)):
        cin = c["in_channels"] if li == 0 else fc
        s[f"f0_predictor.condnet.{idx}.bias"] = (fc,)
        s[f"f0_predictor.condnet.{idx}.weight_g"] = (fc, 1, 1)
        s[f"f0_predictor.condnet.{idx}.weight_v"] = (fc, cin, 3)
    s["f0_predictor.classifier.weight"] = (1, fc)
    s["f0_predictor.classifier.bias"] = (1,)
    return s


def hift_total_upsample(c):
    t = c["istft_hop"]
    for u in c["upsample_rates"]:
        t *= u
    return t


# --------------------------------------------------------------------------- v2 AR (NaiveTransformer)
# reference: configs/v2/vc_wrapper.yaml:39-53, modules/v2/ar.py:29-63
AR_PRESET = dict(dim=768, n_head=12, n_local_heads=2, head_dim=64, n_layer=12, intermediate_size=2304,
                 vocab_size=2049, max_seq_len=4096, rope_base=10000.0, norm_eps=1e-5)


def ar_config(**overrides):
    cfg = deepcopy(AR_PRESET)
    cfg.update(overrides)
    assert cfg["dim"] == cfg["n_head"] * cfg["head_dim"]
    return cfg


def ar_state_spec(c):
    """State dict of `NaiveWrapper` before setup_caches (KV caches are runtime state, not weights)."""
    s = OrderedDict()
    D, I, V = c["dim"], c["intermediate_size"], c["vocab_size"]
    kv = c["n_local_heads"] * c["head_dim"]
    s["sep_token_emb"] = (D,)
    s["model.embeddings.weight"] = (V, D)
    for i in range(c["n_layer"]):
        p = f"model.layers.{i}."
        s[p + "attention.wqkv.weight"] = (D + 2 * kv, D)
        s[p + "attention.wo.weight"] = (D, D)
        s[p + "feed_forward.w1.weight"] = (I, D)
        s[p + "feed_forward.w3.weight"] = (I, D)
        s[p + "feed_forward.w2.weight"] = (D, I)
        s[p + "ffn_norm.weight"] = (D,)
        s[p + "attention_norm.weight"] = (D,)
    s["model.norm.weight"] = (D,)
    s["model.output.weight"] = (V, D)
    return s


# --------------------------------------------------------------------------- length regulator
# reference: configs/presets/*.yml `length_regulator:` blocks, configs/v2/vc_wrapper.yaml:32-38,54-60
LR_PRESETS = {
    "tiny": dict(version=1, channels=384, in_channels=1024, is_discrete=False, codebook_size=1024, n_convs=4,
                 f0_condition=False, n_f0_bins=512),
    "small": dict(version=1, channels=512, in_channels=768, is_discrete=False, codebook_size=1024, n_convs=4,
                  f0_condition=False, n_f0_bins=512),
    "base": dict(version=1, channels=768, in_channels=768, is_discrete=False, codebook_size=1024, n_convs=4,
                 f0_condition=True, n_f0_bins=256),
    "v2_cfm": dict(version=2, channels=512, in_channels=0, is_discrete=True, codebook_size=2048, n_convs=4,
                   f0_condition=False, n_f0_bins=512),
    "v2_ar": dict(version=2, channels=768, in_channels=0, is_discrete=True, codebook_size=32, n_convs=0,
                  f0_condition=False, n_f0_bins=512),
}


def lr_config(preset, **overrides):
    cfg = deepcopy(LR_PRESETS[preset])
    cfg.update(overrides)
    cfg.setdefault("out_channels", cfg["channels"])
    return cfg


def lr_has_final_conv(c):
    """v1 always ends in Conv1d(channels, out, 1); v2 uses nn.Identity when out_channels == channels."""
    return c["version"] == 1 or c["out_channels"] != c["channels"]


def lr_state_spec(c):
    """State dict of `InterpolateRegulator` (modules/length_regulator.py:29-88; v2 modules/v2/length_regulator.py:28-72)."""
    s = OrderedDict()
    C = c["channels"]
    for i in range(c["n_convs"]):
        s[f"model.{3 * i}.weight"] = (C, C, 3)
        s[f"model.{3 * i}.bias"] = (C,)
        s[f"model.{3 * i + 1}.weight"] = (C,)
        s[f"model.{3 * i + 1}.bias"] = (C,)
    if lr_has_final_conv(c):
        s[f"model.{3 * c['n_convs']}.weight"] = (c["out_channels"], C, 1)
        s[f"model.{3 * c['n_convs']}.bias"] = (c["out_channels"],)
    s["embedding.weight"] = (c["codebook_size"], C)
    s["mask_token"] = (1, C)
    if c["f0_condition"]:
        s["f0_embedding.weight"] = (c["n_f0_bins"], C)
        s["f0_mask"] = (1, C)
    if not c["is_discrete"]:
        s["content_in_proj.weight"] = (C, c["in_channels"])
        s["content_in_proj.bias"] = (C,)
    return s


# --------------------------------------------------------------------------------------------- CAMPPlus style encoder (8f row 3)
# modules/campplus/DTDNN.py:54-137 as the drivers build it: CAMPPlus(feat_dim=80, embedding_size=192) (inference.py:98)
CAMPPLUS_PRESET = dict(feat_dim=80, embedding_size=192, growth_rate=32, bn_size=4, init_channels=128, m_channels=32,
                       block_layers=(12, 24, 16), block_kernel=(3, 3, 3), block_dilation=(1, 2, 2), seg_len=100)


def campplus_config(**overrides):
    cfg = deepcopy(CAMPPLUS_PRESET)
    cfg.update(overrides)
    assert cfg["feat_dim"] % 8 == 0
    return cfg


def _bn_spec(s, prefix, c, affine=True):
    if affine:
        s[prefix + ".weight"] = (c,)
        s[prefix + ".bias"] = (c,)
    s[prefix + ".running_mean"] = (c,)
    s[prefix + ".running_var"] = (c,)
    s[prefix + ".num_batches_tracked"] = ()


def campplus_state_spec(c):
    """State dict of `CAMPPlus` (modules/campplus/DTDNN.py:54-107, layers.py): FCM head (2-D res blocks), TDNN, three CAM
    dense-TDNN blocks with transit layers, statistics pooling, dense embedding layer."""
    s = OrderedDict()
    m = c["m_channels"]
    s["head.conv1.weight"] = (m, 1, 3, 3)
    _bn_spec(s, "head.bn1", m)
    for layer in ("layer1", "layer2"):
        for b in range(2):
            p = f"head.{layer}.{b}"
            s[p + ".conv1.weight"] = (m, m, 3, 3)
            _bn_spec(s, p + ".bn1", m)
            s[p + ".conv2.weight"] = (m, m, 3, 3)
            _bn_spec(s, p + ".bn2", m)
            if b == 0:            # stride 2 -> projection shortcut (layers.py:279-286)
                s[p + ".shortcut.0.weight"] = (m, m, 1, 1)
                _bn_spec(s, p + ".shortcut.1", m)
    s["head.conv2.weight"] = (m, m, 3, 3)
    _bn_spec(s, "head.bn2", m)
    ch = m * (c["feat_dim"] // 8)
    init, g, bn = c["init_channels"], c["growth_rate"], c["bn_size"] * c["growth_rate"]
    s["xvector.tdnn.linear.weight"] = (init, ch, 5)
    _bn_spec(s, "xvector.tdnn.nonlinear.batchnorm", init)
    ch = init
    for bi, (nl, k) in enumerate(zip(c["block_layers"], c["block_kernel"])):
        for i in range(nl):
            p = f"xvector.block{bi + 1}.tdnnd{i + 1}"
            cin = ch + i * g
            _bn_spec(s, p + ".nonlinear1.batchnorm", cin)
            s[p + ".linear1.weight"] = (bn, cin, 1)
            _bn_spec(s, p + ".nonlinear2.batchnorm", bn)
            s[p + ".cam_layer.linear_local.weight"] = (g, bn, k)
            s[p + ".cam_layer.linear1.weight"] = (bn // 2, bn, 1)
            s[p + ".cam_layer.linear1.bias"] = (bn // 2,)
            s[p + ".cam_layer.linear2.weight"] = (g, bn // 2, 1)
            s[p + ".cam_layer.linear2.bias"] = (g,)
        ch = ch + nl * g
        _bn_spec(s, f"xvector.transit{bi + 1}.nonlinear.batchnorm", ch)
        s[f"xvector.transit{bi + 1}.linear.weight"] = (ch // 2, ch, 1)
        ch //= 2
    _bn_spec(s, "xvector.out_nonlinear.batchnorm", ch)
    s["dense.linear.weight"] = (c["embedding_size"], 2 * ch, 1)
    _bn_spec(s, "dense.nonlinear.batchnorm", c["embedding_size"], affine=False)
    return s
