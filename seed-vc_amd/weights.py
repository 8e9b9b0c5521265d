"""Deterministic, reference-free test-weight generator.

There is no network and no checkpoint in the build/test environments, so every
parity test, `smoke()` and `bench.py` run on weights produced here.  Generation is counter based
(numpy Philox keyed by crc32 of the state-dict key + a user seed), so the GPU box regenerates
bit-identical weights from `specs.*_state_spec()` alone.  Scales are fan-in normalised so that
activations stay O(1) through the whole stack (a more discriminating parity input than N(0,.02)).
"""
import zlib
import numpy as np
import torch


def _rng(key, seed):
    return np.random.Generator(np.random.Philox(key=[zlib.crc32(key.encode()) , seed & 0xFFFFFFFF]))


def _kaiser_sinc_taps():
    """12-tap Kaiser-sinc low-pass, cutoff .25, half-width .3.

    Restated from the published alias-free-torch design used by the reference
    (reference: modules/bigvgan/alias_free_activation/torch/filter.py:30-62)."""
    import math
    ks, cutoff, half_width = 12, 0.25, 0.3
    half = ks // 2
    delta_f = 4 * half_width
    A = 2.285 * (half - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(ks, beta=beta, periodic=False)
    time = torch.arange(-half, half) + 0.5
    f = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    f = f / f.sum()
    return f.view(1, 1, ks).float()


def _gain(key):
    """Residual-branch output convs and the vocoder output convs get a small gain, as in trained
    networks, so activations stay O(1) through 6 upsampling stages and the final clamp rarely bites."""
    if ".convs2." in key:
        return 0.3
    if "conv_post" in key:
        return 0.15
    return 1.0


def make_tensor(key, shape, seed=0):
    """One deterministic fp32 tensor for state-dict entry `key`."""
    shape = tuple(shape)
    g = _rng(key, seed)
    leaf = key.split(".")[-1]

    def normal(std):
        return torch.from_numpy((g.standard_normal(shape) * std).astype(np.float32))

    if leaf == "num_batches_tracked":        # BatchNorm bookkeeping buffer (int64 scalar)
        return torch.tensor(0, dtype=torch.long)
    if leaf == "running_var":                # BatchNorm statistics: positive, O(1)
        return torch.from_numpy((0.6 + 0.8 * g.random(shape)).astype(np.float32))
    if leaf == "running_mean":
        return normal(0.2)
    if key.endswith("input_pos"):
        return torch.arange(shape[0])
    if key.endswith("causal_mask"):
        return torch.tril(torch.ones(shape, dtype=torch.bool))
    if key.endswith("freqs_cis"):
        # v2 registers the RoPE table as a bf16-rounded buffer (reference: v2/dit_model.py:100-102,225-234)
        n_elem = shape[1] * 2
        freqs = 1.0 / (10000 ** (torch.arange(0, n_elem, 2)[: n_elem // 2].float() / n_elem))
        t = torch.arange(shape[0])
        fr = torch.outer(t, freqs)
        cis = torch.polar(torch.ones_like(fr), fr)
        return torch.stack([cis.real, cis.imag], dim=-1).to(torch.bfloat16)
    if key.endswith("t_embedder.freqs") or key.endswith("t_embedder2.freqs"):
        import math
        half = shape[0]
        return torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    if leaf == "filter":
        return _kaiser_sinc_taps()
    if leaf == "weight_g":
        # magnitude of the weight-normed direction: g*v/|v| has per-element variance g^2/fan_in
        gain = _gain(key)
        return torch.from_numpy(((0.8 + 0.4 * g.random(shape)) * gain).astype(np.float32))
    if leaf in ("alpha", "beta"):
        if ".act." in key or key.startswith("activation_post"):      # BigVGAN: log-scale params
            return normal(0.3)
        return torch.from_numpy((0.5 + g.random(shape)).astype(np.float32))   # HiFT Snake: alpha > 0
    if leaf == "bias":
        if key.endswith("f0_predictor.classifier.bias"):
            return torch.full(shape, 150.0)       # puts predicted f0 in the voiced range (> 10 Hz)
        if key.endswith("conv_post.bias") and "hift." in key:
            return normal(0.05) + 1.5            # exp() magnitudes ~4: waveform rms ~0.2, rarely clipped
        return normal(0.05)
    if key.endswith("sep_token_emb"):
        return normal(1.0)
    if "norm.weight" in key or key.endswith("ffn_norm.weight"):
        return 1.0 + normal(0.1)
    if leaf == "weight" and len(shape) == 2 and ("embedder.weight" in key and "t_embedder" not in key):
        return normal(0.5)          # embedding tables (dead at inference)
    # matrices / conv kernels: fan-in normalised
    if leaf in ("weight", "weight_v"):
        if len(shape) == 1:
            return 1.0 + normal(0.1)
        transposed = (".ups." in key or key.startswith("ups.")) and len(shape) == 3
        if transposed:      # ConvTranspose1d weight (Cin, Cout, k): each output sees Cin*k/stride taps
            fan_in = shape[0] * max(shape[2] // 2, 1)
        else:
            fan_in = int(np.prod(shape[1:]))
        if leaf == "weight_v":
            w = normal(1.0)
            # weight_g above is ~1, so scale v such that g*v/|v| ~ N(0, 1/fan_in) per element
            return w
        std = _gain(key) / np.sqrt(fan_in)
        if key.endswith("f0_predictor.classifier.weight"):
            std *= 40.0           # spreads f0 over tens of Hz around the bias
        return normal(std)
    return normal(0.1)


def make_state_dict(spec, seed=0, prefix=""):
    sd = {}
    for k, shp in spec.items():
        sd[k] = make_tensor(prefix + k, shp, seed)
    # weight_g must turn v/|v| into a fan-in-normalised matrix: |row of g*v/|v||^2 = g^2, and the
    # row has fan_in elements, so per-element variance g^2/fan_in already holds with g ~ 1.
    return sd
