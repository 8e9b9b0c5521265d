"""CPU oracle for the seed-vc inference hot path  --  TEST INFRASTRUCTURE ONLY.

A plain-PyTorch fp32 restatement (no reference imports) of the algorithm of
  * the CFM Euler sampler               (reference: modules/flow_matching.py:30-112, modules/v2/cfm.py:16-132)
  * the DiT/UViT estimator              (reference: modules/diffusion_transformer.py:30-48,112-143,173-191,
                                                    222-312,323-364,388-405,486-537)
  * the WaveNet head                    (reference: modules/wavenet.py:138-166, modules/encodec.py:212-228,
                                                    modules/commons.py:131-156)
  * the v2 AdaLN-zero DiT               (reference: modules/v2/dit_wrapper.py:114-152, modules/v2/dit_model.py:20-54,109-143)
  * the BigVGAN v2 vocoder              (reference: modules/bigvgan/bigvgan.py:132-141,360-386,
                                                    modules/bigvgan/activations.py:107-118,
                                                    modules/bigvgan/alias_free_activation/torch/{act,resample,filter}.py)
  * the HiFT vocoder                    (reference: modules/hifigan/generator.py:151-158,196-227,263-279,379-436,
                                                    modules/hifigan/f0_predictor.py:51-55)
  * the chunk / crossfade harness       (reference: inference.py:343-350,470-527)

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this file; it
is the checker, never the product.  Pinning: `tests/golden/make_golden.py` imports the real reference
modules in the build container, loads the SAME generated weights into them and stores input/output
vectors under `tests/golden/*.npz`; `tests/test_oracle_golden.py` holds this file to those vectors
(the reference ships no tests of its own -- SURVEY.md section 4).

Everything takes a flat `sd` (state-dict: name -> tensor) and a config dict from
`seedvc_amd.specs`.  All math is fp32 on CPU; random draws (z, HiFT phases/noise) are inputs.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- small helpers
def wn_weight(sd, prefix):
    """Effective weight of a (possibly) weight-normed layer: g * v / ||v|| over dims != 0
    (torch.nn.utils.weight_norm default dim=0)."""
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"].float()
    g, v = sd[prefix + ".weight_g"].float(), sd[prefix + ".weight_v"].float()
    n = v.reshape(v.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (v.dim() - 1))
    return v * (g / n)


def linear(x, sd, prefix, bias=True):
    b = sd.get(prefix + ".bias") if bias else None
    return F.linear(x, wn_weight(sd, prefix), None if b is None else b.float())


def silu(x):
    return x * torch.sigmoid(x)


def sequence_mask(lengths, max_len):
    # reference: modules/commons.py:152-156
    return torch.arange(max_len)[None, :] < lengths[:, None]


def rmsnorm(x, gamma, eps=1e-5):
    # reference: modules/diffusion_transformer.py:280-285
    return x * torch.rsqrt((x * x).mean(dim=-1, keepdim=True) + eps) * gamma


def rope_table(n_pos, head_dim=64, base=10000.0, bf16_round=False):
    """(n_pos, head_dim/2, 2) cos/sin.  reference: diffusion_transformer.py:288-297 (fp32 table in every
    v1 driver); v2 rounds the table to bf16 (v2/dit_model.py:225-234)."""
    inv = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    ang = torch.outer(torch.arange(n_pos).float(), inv)
    tab = torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1)
    if bf16_round:
        tab = tab.to(torch.bfloat16).float()
    return tab


def apply_rope(x, tab):
    """x: (B, T, H, hd) ; interleaved pairs (x[2i], x[2i+1]).  reference: diffusion_transformer.py:300-312"""
    B, T, H, hd = x.shape
    xr = x.reshape(B, T, H, hd // 2, 2)
    c = tab[:T, :, 0][None, :, None, :]
    s = tab[:T, :, 1][None, :, None, :]
    o0 = xr[..., 0] * c - xr[..., 1] * s
    o1 = xr[..., 1] * c + xr[..., 0] * s
    return torch.stack([o0, o1], dim=-1).reshape(B, T, H, hd)


def attention(x, sd, prefix, n_head, tab, key_mask):
    """Self-attention block.  key_mask: (B, T) bool (True = attend).
    reference: modules/diffusion_transformer.py:222-260 (SDPA with boolean key-padding mask)."""
    B, T, D = x.shape
    hd = D // n_head
    qkv = F.linear(x, sd[prefix + ".wqkv.weight"].float())
    q, k, v = qkv.split([D, D, D], dim=-1)
    q = apply_rope(q.reshape(B, T, n_head, hd), tab).transpose(1, 2)
    k = apply_rope(k.reshape(B, T, n_head, hd), tab).transpose(1, 2)
    v = v.reshape(B, T, n_head, hd).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)
    s = s.masked_fill(~key_mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    y = torch.matmul(p, v).transpose(1, 2).reshape(B, T, D)
    return F.linear(y, sd[prefix + ".wo.weight"].float())


def feed_forward(x, sd, prefix):
    # reference: modules/diffusion_transformer.py:263-271
    a = F.linear(x, sd[prefix + ".w1.weight"].float())
    b = F.linear(x, sd[prefix + ".w3.weight"].float())
    return F.linear(silu(a) * b, sd[prefix + ".w2.weight"].float())


def timestep_embed(t, sd, prefix, freqs=None):
    """reference: modules/diffusion_transformer.py:344-364 (scale 1000, [cos, sin], Linear-SiLU-Linear)."""
    if freqs is None:
        freqs = sd.get(prefix + ".freqs")
    if freqs is None:   # v2 recomputes them on the fly (v2/dit_wrapper.py:41-47)
        freqs = torch.exp(-math.log(10000) * torch.arange(0, 128, dtype=torch.float32) / 128)
    args = 1000.0 * t[:, None].float() * freqs[None].float()
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    h = silu(linear(emb, sd, prefix + ".mlp.0"))
    return linear(h, sd, prefix + ".mlp.2")


# ----------------------------------------------------------------------------- v1 DiT
def adaln_v1(x, c, sd, prefix):
    """w * rmsnorm(x) + b with [w|b] = Linear(c); plain rmsnorm when c is None.
    reference: modules/diffusion_transformer.py:40-48"""
    n = rmsnorm(x, sd[prefix + ".norm.weight"].float())
    if c is None:
        return n
    wb = linear(c, sd, prefix + ".project_layer")
    D = x.shape[-1]
    return wb[..., :D] * n + wb[..., D:]


def sconv1d_reflect(x, w, b, dilation=1):
    """Non-causal, stride-1 `SConv1d`: reflect pad (left = total - total//2, right = total//2).
    reference: modules/encodec.py:212-228 (the padding= kwarg of WN is swallowed, wavenet.py:125)."""
    k = w.shape[-1]
    total = (k - 1) * dilation
    right = total // 2
    left = total - right
    if total > 0:
        T = x.shape[-1]
        extra = 0
        if T <= max(left, right):       # reference: encodec.py:103-111 (small-input guard)
            extra = max(left, right) - T + 1
            x = F.pad(x, (0, extra))
        x = F.pad(x, (left, right), mode="reflect")
        if extra:
            x = x[..., : x.shape[-1] - extra]
    return F.conv1d(x, w, b, dilation=dilation)


def wavenet(x, x_mask, g, sd, prefix, cfg):
    """WN stack.  x (B,W,T), x_mask (B,1,T) float, g (B,W,1).  reference: modules/wavenet.py:138-166"""
    W, nl = cfg["wn_dim"], cfg["wn_layers"]
    out = torch.zeros_like(x)
    gc = F.conv1d(g, wn_weight(sd, prefix + ".cond_layer.conv.conv"), sd[prefix + ".cond_layer.conv.conv.bias"].float())
    for i in range(nl):
        dil = cfg["wn_dilation"] ** i
        xin = sconv1d_reflect(x, wn_weight(sd, f"{prefix}.in_layers.{i}.conv.conv"),
                              sd[f"{prefix}.in_layers.{i}.conv.conv.bias"].float(), dil)
        a = xin + gc[:, 2 * W * i: 2 * W * (i + 1), :]
        acts = torch.tanh(a[:, :W]) * torch.sigmoid(a[:, W:])            # commons.py:131-138
        rs = F.conv1d(acts, wn_weight(sd, f"{prefix}.res_skip_layers.{i}.conv.conv"),
                      sd[f"{prefix}.res_skip_layers.{i}.conv.conv.bias"].float())
        if i < nl - 1:
            x = (x + rs[:, :W]) * x_mask
            out = out + rs[:, W:]
        else:
            out = out + rs
    return out * x_mask


def dit_forward_v1(sd, cfg, x, prompt_x, x_lens, t, style, cond):
    """One estimator evaluation.  x, prompt_x (N,C,T); x_lens (N,) or broadcastable; t (N,) ; style (N,S);
    cond (N,T,Dc).  Returns (N,C,T).  reference: modules/diffusion_transformer.py:486-537"""
    N, C, T = x.shape
    D, H, L = cfg["D"], cfg["H"], cfg["L"]
    t1 = timestep_embed(t, sd, "t_embedder")
    cnd = linear(cond, sd, "cond_projection")
    xt = x.transpose(1, 2)
    x_in = torch.cat([xt, prompt_x.transpose(1, 2), cnd], dim=-1)
    if cfg["style_condition"] and not cfg["style_as_token"]:
        x_in = torch.cat([x_in, style[:, None, :].expand(N, T, -1)], dim=-1)
    h = linear(x_in, sd, "cond_x_merge_linear")
    if cfg["style_as_token"]:
        h = torch.cat([linear(style, sd, "style_in")[:, None, :], h], dim=1)
    if cfg["time_as_token"]:
        h = torch.cat([t1[:, None, :], h], dim=1)
    npre = cfg["n_prefix"]
    Tp = T + npre
    lens = (x_lens + npre).expand(N) if x_lens.numel() != N else x_lens + npre
    key_mask = sequence_mask(lens, Tp)
    c_full = t1[:, None, :]
    c_blk = None if cfg["time_as_token"] else c_full          # diffusion_transformer.py:184
    tab = rope_table(Tp, cfg["hd"])
    emit = [i for i in range(L) if i < L // 2] if cfg["uvit"] else []
    recv = [i for i in range(L) if i > L // 2] if cfg["uvit"] else []
    skips = []
    for i in range(L):
        p = f"transformer.layers.{i}"
        if i in recv:
            h = linear(torch.cat([h, skips.pop()], dim=-1), sd, p + ".skip_in_linear")
        h = h + attention(adaln_v1(h, c_blk, sd, p + ".attention_norm"), sd, p + ".attention", H, tab, key_mask)
        h = h + feed_forward(adaln_v1(h, c_blk, sd, p + ".ffn_norm"), sd, p + ".feed_forward")
        if i in emit:
            skips.append(h)
    h = adaln_v1(h, c_full, sd, "transformer.norm")           # final norm always gets c (:142)
    h = h[:, npre:]
    if cfg["long_skip"]:
        h = linear(torch.cat([h, xt], dim=-1), sd, "skip_linear")
    if cfg["head"] == "wavenet":
        xm = sequence_mask(lens, Tp)[:, None, :].float()       # (N,1,T) -- npre == 0 for wavenet models
        y = linear(h, sd, "conv1").transpose(1, 2)
        t2 = timestep_embed(t, sd, "t_embedder2")
        y = wavenet(y, xm, t2[:, :, None], sd, "wavenet", cfg).transpose(1, 2) + linear(h, sd, "res_projection")
        # FinalLayer: LayerNorm(no affine, eps 1e-6) * (1 + scale) + shift ; chunk order (shift, scale)
        mod = linear(silu(t1), sd, "final_layer.adaLN_modulation.1")
        Wd = cfg["wn_dim"]
        shift, scale = mod[:, :Wd], mod[:, Wd:]
        y = F.layer_norm(y, (Wd,), eps=1e-6) * (1 + scale[:, None, :]) + shift[:, None, :]
        y = linear(y, sd, "final_layer.linear").transpose(1, 2)
        return F.conv1d(y, sd["conv2.weight"].float(), sd["conv2.bias"].float())
    y = linear(silu(linear(h, sd, "final_mlp.0")), sd, "final_mlp.2")
    return y.transpose(1, 2)


# ----------------------------------------------------------------------------- v2 DiT (AdaLN-zero)
def dit_forward_v2(sd, cfg, x, prompt_x, x_lens, t, style, cond):
    """reference: modules/v2/dit_wrapper.py:114-152 + modules/v2/dit_model.py:109-143 (no UViT skips executed)."""
    N, C, T = x.shape
    D, H, L = cfg["D"], cfg["H"], cfg["L"]
    t1 = timestep_embed(t, sd, "t_embedder")
    cnd = linear(cond, sd, "cond_projection")
    x_in = torch.cat([x.transpose(1, 2), prompt_x.transpose(1, 2), cnd], dim=-1)
    h = linear(x_in, sd, "cond_x_merge_linear")
    st = linear(style, sd, "style_in")
    if cfg["style_as_token"]:
        h = torch.cat([st[:, None, :], h], dim=1)
    if cfg["time_as_token"]:
        h = torch.cat([t1[:, None, :], h], dim=1)
    npre = cfg["n_prefix"]
    Tp = T + npre
    lens = (x_lens + npre).expand(N) if x_lens.numel() != N else x_lens + npre
    key_mask = sequence_mask(lens, Tp)
    tab = rope_table(Tp, cfg["hd"], bf16_round=True)
    c = silu(t1[:, None, :])
    for i in range(L):
        p = f"transformer.layers.{i}"
        e = linear(c, sd, p + ".attention_norm.linear")
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = e.chunk(6, dim=-1)
        n = rmsnorm(h, sd[p + ".attention_norm.norm.weight"].float()) * (1 + sc_a) + sh_a
        h = h + g_a * attention(n, sd, p + ".attention", H, tab, key_mask)
        n = rmsnorm(h, sd[p + ".ffn_norm.weight"].float()) * (1 + sc_m) + sh_m
        h = h + g_m * feed_forward(n, sd, p + ".feed_forward")
    e = linear(c, sd, "transformer.norm.linear")
    scale, shift = e.chunk(2, dim=-1)                          # order (scale, shift): v2/dit_model.py:51
    h = rmsnorm(h, sd["transformer.norm.norm.weight"].float()) * (1 + scale) + shift
    h = h[:, npre:]
    y = linear(silu(linear(h, sd, "final_mlp.0")), sd, "final_mlp.2")
    return y.transpose(1, 2)


def dit_forward(sd, cfg, *a):
    return (dit_forward_v2 if cfg["version"] == 2 else dit_forward_v1)(sd, cfg, *a)


# ----------------------------------------------------------------------------- samplers
def t_span_linear(n_steps):
    return torch.linspace(0, 1, n_steps + 1)


def t_span_cosine(n_steps):
    # reference: modules/v2/cfm.py:47-48
    ts = torch.linspace(0, 1, n_steps + 1)
    return ts + (-1) * (torch.cos(torch.pi / 2 * ts) - 1 + ts)


def cfm_sample_v1(sd, cfg, z, x_len, prompt, mu, style, n_steps, cfg_rate, temperature=1.0):
    """Euler sampler for ONE utterance (the reference sampler is B=1 only: flow_matching.py:90,94).
    z (1,C,T) noise, prompt (1,C,P), mu (1,T,Dc), style (1,S).  reference: flow_matching.py:49-112"""
    x = z.clone() * temperature
    t_span = t_span_linear(n_steps)
    P = prompt.shape[-1]
    prompt_x = torch.zeros_like(x)
    prompt_x[..., :P] = prompt
    x[..., :P] = 0
    t = t_span[0]
    lens = torch.tensor([int(x_len)])
    for step in range(1, n_steps + 1):
        dt = t_span[step] - t_span[step - 1]        # recomputed every step (flow_matching.py:83)
        if cfg_rate > 0:
            out = dit_forward(sd, cfg,
                              torch.cat([x, x]), torch.cat([prompt_x, torch.zeros_like(prompt_x)]), lens,
                              torch.stack([t, t]), torch.cat([style, torch.zeros_like(style)]),
                              torch.cat([mu, torch.zeros_like(mu)]))
            v = (1.0 + cfg_rate) * out[0:1] - cfg_rate * out[1:2]
        else:
            v = dit_forward(sd, cfg, x, prompt_x, lens, t[None], style, mu)
        x = x + dt * v
        t = t + dt                                   # (the reference's trailing dt update is dead: :108-109)
        x[:, :, :P] = 0
    return x


def cfm_sample_v2(sd, cfg, z, x_len, prompt, mu, style, n_steps, cfg_rates, temperature=1.0, random_voice=False):
    """v2 sampler: cosine time warp + 1/2/3-way CFG.  reference: modules/v2/cfm.py:44-132"""
    x = z.clone() * temperature
    t_span = t_span_cosine(n_steps)
    P = prompt.shape[-1]
    prompt_x = torch.zeros_like(x)
    prompt_x[..., :P] = prompt
    x[..., :P] = 0
    t = t_span[0]
    dt = t_span[1] - t_span[0]
    lens1 = torch.tensor([int(x_len)])
    a, b = float(cfg_rates[0]), float(cfg_rates[1])
    zp, zs, zm = torch.zeros_like(prompt_x), torch.zeros_like(style), torch.zeros_like(mu)
    for step in range(1, n_steps + 1):
        def est(xs, ps, ss, ms):
            n = len(xs)
            return dit_forward(sd, cfg, torch.cat(xs), torch.cat(ps), lens1.repeat(n), t.repeat(n),
                               torch.cat(ss), torch.cat(ms))
        if random_voice:
            o = est([x, x], [zp, zp], [zs, zs], [mu, zm])
            v = (1.0 + a) * o[0:1] - a * o[1:2]
        elif a == 0 and b == 0:
            v = est([x], [prompt_x], [style], [mu])
        elif a == 0:
            o = est([x, x], [prompt_x, zp], [style, zs], [mu, mu])
            v = (1.0 + b) * o[0:1] - b * o[1:2]
        elif b == 0:
            o = est([x, x], [prompt_x, zp], [style, zs], [mu, zm])
            v = (1.0 + a) * o[0:1] - a * o[1:2]
        else:
            o = est([x, x, x], [prompt_x, zp, zp], [style, zs, zs], [mu, mu, zm])
            v = (1.0 + a + b) * o[0:1] - a * o[2:3] - b * o[1:2]
        x = x + dt * v
        t = t + dt
        if step < n_steps:
            dt = t_span[step + 1] - t
        x[:, :, :P] = 0
    return x


def cfm_sample(sd, cfg, z, x_len, prompt, mu, style, n_steps, cfg_rate, temperature=1.0, random_voice=False):
    if cfg["version"] == 2:
        rates = cfg_rate if isinstance(cfg_rate, (list, tuple)) else [cfg_rate, cfg_rate]
        return cfm_sample_v2(sd, cfg, z, x_len, prompt, mu, style, n_steps, rates, temperature, random_voice)
    return cfm_sample_v1(sd, cfg, z, x_len, prompt, mu, style, n_steps, cfg_rate, temperature)


# ----------------------------------------------------------------------------- anti-aliased activation
def anti_alias_act(x, filt, a, inv_b):
    """Closed form of UpSample1d(x2) -> x + inv_b*sin^2(a*x) -> DownSample1d(x2), replicate padding.
    x (B,C,L); filt (12,); a, inv_b (C,).  reference: alias_free_activation/torch/act.py:25-30,
    resample.py:29-38,55-58, filter.py:94-101."""
    B, C, L = x.shape
    f = filt.reshape(1, 1, 12).float()
    xp = F.pad(x, (5, 5), mode="replicate")
    up = 2.0 * F.conv_transpose1d(xp, f.expand(C, 1, 12), stride=2, groups=C)
    up = up[..., 15:-15]
    s = up + inv_b[None, :, None] * torch.sin(up * a[None, :, None]) ** 2
    sp = F.pad(s, (5, 6), mode="replicate")
    return F.conv1d(sp, f.expand(C, 1, 12), stride=2, groups=C)


def snakebeta_params(sd, prefix, logscale=True, has_beta=True):
    """(a, inv_b): y = x + inv_b * sin^2(a x).  reference: bigvgan/activations.py:107-118"""
    al = sd[prefix + ".alpha"].float()
    be = sd[prefix + ".beta"].float() if has_beta else al
    if logscale:
        al, be = torch.exp(al), torch.exp(be)
    return al, 1.0 / (be + 1e-9)


# ----------------------------------------------------------------------------- BigVGAN
def bigvgan_forward(sd, h, mel):
    """mel (B, num_mels, S) -> (B, 1, S * prod(upsample_rates)).  reference: modules/bigvgan/bigvgan.py:360-386"""
    has_beta = h["activation"] == "snakebeta"
    logs = h["snake_logscale"]
    x = F.conv1d(mel, wn_weight(sd, "conv_pre"), sd["conv_pre.bias"].float(), padding=3)
    nk = len(h["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = F.conv_transpose1d(x, wn_weight(sd, f"ups.{i}.0"), sd[f"ups.{i}.0.bias"].float(),
                               stride=u, padding=(k - u) // 2)
        acc = None
        for j, (rk, dils) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            p = f"resblocks.{i * nk + j}"
            y = x
            for d, dil in enumerate(dils):                        # AMPBlock1: bigvgan.py:132-141
                filt = sd[f"{p}.activations.{2 * d}.upsample.filter"].reshape(12)
                a, ib = snakebeta_params(sd, f"{p}.activations.{2 * d}.act", logs, has_beta)
                xt = anti_alias_act(y, filt, a, ib)
                xt = F.conv1d(xt, wn_weight(sd, f"{p}.convs1.{d}"), sd[f"{p}.convs1.{d}.bias"].float(),
                              dilation=dil, padding=(rk * dil - dil) // 2)
                a, ib = snakebeta_params(sd, f"{p}.activations.{2 * d + 1}.act", logs, has_beta)
                xt = anti_alias_act(xt, filt, a, ib)
                xt = F.conv1d(xt, wn_weight(sd, f"{p}.convs2.{d}"), sd[f"{p}.convs2.{d}.bias"].float(),
                              padding=(rk - 1) // 2)
                y = xt + y
            acc = y if acc is None else acc + y
        x = acc / nk
    filt = sd["activation_post.upsample.filter"].reshape(12)
    a, ib = snakebeta_params(sd, "activation_post.act", logs, has_beta)
    x = anti_alias_act(x, filt, a, ib)
    b = sd.get("conv_post.bias")
    x = F.conv1d(x, wn_weight(sd, "conv_post"), None if b is None else b.float(), padding=3)
    return torch.tanh(x) if h["use_tanh_at_final"] else torch.clamp(x, -1.0, 1.0)


# ----------------------------------------------------------------------------- HiFT
def hift_f0_predictor(sd, mel):
    """mel (B,80,S) -> f0 (B,S) >= 0.  reference: modules/hifigan/f0_predictor.py:51-55"""
    x = mel
    for idx in (0, 2, 4, 6, 8):
        p = f"f0_predictor.condnet.{idx}"
        x = F.elu(F.conv1d(x, wn_weight(sd, p), sd[p + ".bias"].float(), padding=1))
    y = F.linear(x.transpose(1, 2), sd["f0_predictor.classifier.weight"].float(),
                 sd["f0_predictor.classifier.bias"].float())
    return torch.abs(y.squeeze(-1))


def hift_source(sd, c, f0, phase0, noise_sine):
    """f0 (B,S) -> merged harmonic source (B,1,Lw).  phase0 (B,9,1) ~ U(-pi,pi) with [:,0]=0 forced here,
    noise_sine (B,9,Lw) ~ N(0,1).  reference: generator.py:196-227,263-279,379-383.
    The reference accumulates the phase with torch.cumsum on CPU float tensors (double accumulator,
    rounded to fp32 per element); restated with an explicit float64 cumsum."""
    up = hift_total_upsample(c)
    f0u = f0[:, :, None].expand(-1, -1, up).reshape(f0.shape[0], 1, -1)          # nearest upsample
    nh = c["nb_harmonics"] + 1
    mult = torch.arange(1, nh + 1, dtype=torch.float32)[None, :, None]
    fmat = f0u * mult / c["sampling_rate"]                                         # (B,9,Lw) fp32
    cum = torch.cumsum(fmat.double(), dim=-1).float()
    theta = 2 * np.pi * (cum % 1)
    ph = phase0.clone().float()
    ph[:, 0, :] = 0
    sine = c["nsf_alpha"] * torch.sin(theta + ph)
    uv = (f0u > c["nsf_voiced_threshold"]).float()
    namp = uv * c["nsf_sigma"] + (1 - uv) * c["nsf_alpha"] / 3
    sine = sine * uv + namp * noise_sine
    merged = torch.tanh(F.linear(sine.transpose(1, 2), sd["m_source.l_linear.weight"].float(),
                                 sd["m_source.l_linear.bias"].float()))
    return merged.transpose(1, 2)


def hift_total_upsample(c):
    t = c["istft_hop"]
    for u in c["upsample_rates"]:
        t *= u
    return t


def _hann_periodic(n):
    return 0.5 - 0.5 * torch.cos(2 * math.pi * torch.arange(n, dtype=torch.float64) / n)


def stft16(x, n_fft=16, hop=4):
    """(B, Lw) -> real, imag each (B, n_fft/2+1, Lw/hop + 1); centered, reflect padded, periodic Hann.
    Restates torch.stft(center=True) as an explicit DFT.  reference: generator.py:385-391"""
    w = _hann_periodic(n_fft)
    xp = F.pad(x[:, None, :].double(), (n_fft // 2, n_fft // 2), mode="reflect")[:, 0]
    fr = xp.unfold(-1, n_fft, hop) * w                                  # (B, frames, n_fft)
    k = torch.arange(n_fft // 2 + 1, dtype=torch.float64)[:, None]
    n = torch.arange(n_fft, dtype=torch.float64)[None, :]
    ang = 2 * math.pi * k * n / n_fft
    re = torch.matmul(fr, torch.cos(ang).T)
    im = -torch.matmul(fr, torch.sin(ang).T)
    return re.transpose(1, 2).float(), im.transpose(1, 2).float()


def istft16(mag, phase, n_fft=16, hop=4):
    """Inverse of stft16 from magnitude / phase (B, 9, F) -> (B, hop*(F-1)).  reference: generator.py:393-398"""
    mag = torch.clip(mag, max=1e2)
    re = (mag * torch.cos(phase)).double()
    im = (mag * torch.sin(phase)).double()
    B, K, Fr = re.shape
    w = _hann_periodic(n_fft)
    n = torch.arange(n_fft, dtype=torch.float64)[:, None]
    k = torch.arange(K, dtype=torch.float64)[None, :]
    ang = 2 * math.pi * n * k / n_fft
    wk = torch.full((K,), 2.0, dtype=torch.float64)
    wk[0] = 1.0
    wk[-1] = 1.0
    # irfft: imaginary parts of DC / Nyquist bins are ignored
    cosb = torch.cos(ang) * wk[None, :] / n_fft
    sinb = -torch.sin(ang) * wk[None, :] / n_fft
    sinb[:, 0] = 0
    sinb[:, -1] = 0
    fr = torch.matmul(re.transpose(1, 2), cosb.T) + torch.matmul(im.transpose(1, 2), sinb.T)   # (B,F,n_fft)
    fr = fr * w
    Lfull = n_fft + hop * (Fr - 1)
    y = torch.zeros(B, Lfull, dtype=torch.float64)
    env = torch.zeros(Lfull, dtype=torch.float64)
    for f in range(n_fft // hop):          # overlap-add, vectorised over frames with the same phase
        idx = torch.arange(f, Fr, n_fft // hop)
        pos = (idx * hop)[:, None] + torch.arange(n_fft)[None, :]
        y[:, pos.reshape(-1)] += fr[:, idx].reshape(B, -1)
        env[pos.reshape(-1)] += (w * w).repeat(len(idx))
    y = y[:, n_fft // 2: Lfull - n_fft // 2] / env[n_fft // 2: Lfull - n_fft // 2]
    return y.float()


def snake(x, alpha):
    # reference: modules/hifigan/generator.py:79-90 (non-logscale)
    a = alpha[None, :, None]
    return x + (1.0 / (a + 1e-9)) * torch.sin(x * a) ** 2


def hift_resblock(x, sd, p, k, dils):
    # reference: modules/hifigan/generator.py:151-158
    for d, dil in enumerate(dils):
        xt = snake(x, sd[f"{p}.activations1.{d}.alpha"].float())
        xt = F.conv1d(xt, wn_weight(sd, f"{p}.convs1.{d}"), sd[f"{p}.convs1.{d}.bias"].float(),
                      dilation=dil, padding=(k * dil - dil) // 2)
        xt = snake(xt, sd[f"{p}.activations2.{d}.alpha"].float())
        xt = F.conv1d(xt, wn_weight(sd, f"{p}.convs2.{d}"), sd[f"{p}.convs2.{d}.bias"].float(),
                      padding=(k - 1) // 2)
        x = xt + x
    return x


def hift_decode(sd, c, mel, s):
    """mel (B,80,S), s merged source (B,1,Lw) -> waveform (B, Lw).  reference: generator.py:405-436"""
    re, im = stft16(s[:, 0], c["istft_n_fft"], c["istft_hop"])
    s_stft = torch.cat([re, im], dim=1)
    x = F.conv1d(mel, wn_weight(sd, "conv_pre"), sd["conv_pre.bias"].float(), padding=3)
    nk = len(c["resblock_kernel_sizes"])
    nup = len(c["upsample_rates"])
    ups = c["upsample_rates"]
    down_rates = [1] + ups[::-1][:-1]
    cum = list(np.cumprod(down_rates))[::-1]
    for i, (u, k) in enumerate(zip(ups, c["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, c["lrelu_slope"])
        x = F.conv_transpose1d(x, wn_weight(sd, f"ups.{i}"), sd[f"ups.{i}.bias"].float(), stride=u, padding=(k - u) // 2)
        if i == nup - 1:
            x = F.pad(x, (1, 0), mode="reflect")
        r = int(cum[i])
        if r == 1:
            si = F.conv1d(s_stft, sd[f"source_downs.{i}.weight"].float(), sd[f"source_downs.{i}.bias"].float())
        else:
            si = F.conv1d(s_stft, sd[f"source_downs.{i}.weight"].float(), sd[f"source_downs.{i}.bias"].float(),
                          stride=r, padding=r // 2)
        si = hift_resblock(si, sd, f"source_resblocks.{i}", c["source_resblock_kernel_sizes"][i],
                           c["source_resblock_dilation_sizes"][i])
        x = x + si
        acc = None
        for j, (rk, dils) in enumerate(zip(c["resblock_kernel_sizes"], c["resblock_dilation_sizes"])):
            y = hift_resblock(x, sd, f"resblocks.{i * nk + j}", rk, dils)
            acc = y if acc is None else acc + y
        x = acc / nk
    x = F.leaky_relu(x)                                       # default slope 0.01: generator.py:429
    x = F.conv1d(x, wn_weight(sd, "conv_post"), sd["conv_post.bias"].float(), padding=3)
    nb = c["istft_n_fft"] // 2 + 1
    mag = torch.exp(x[:, :nb])
    ph = torch.sin(x[:, nb:])
    y = istft16(mag, ph, c["istft_n_fft"], c["istft_hop"])
    return torch.clamp(y, -c["audio_limit"], c["audio_limit"])


def hift_forward(sd, c, mel, phase0, noise_sine, f0=None):
    """Full HiFT forward with explicit random inputs.  Note the reference also draws a third random tensor
    (`randn_like(uv)`, generator.py:278) whose result is discarded by `_f02source`."""
    if f0 is None:
        f0 = hift_f0_predictor(sd, mel)
    s = hift_source(sd, c, f0, phase0, noise_sine)
    return hift_decode(sd, c, mel, s)


# ----------------------------------------------------------------------------- v2 AR decode step
def ar_new_cache(cfg, dtype=torch.float32):
    shape = (1, cfg["n_local_heads"], cfg["max_seq_len"], cfg["head_dim"])
    return [(torch.zeros(shape, dtype=dtype), torch.zeros(shape, dtype=dtype)) for _ in range(cfg["n_layer"])]


def ar_forward_generate(sd, cfg, x, input_pos, kv_pos, caches):
    """One call of `forward_generate` (prefill with S tokens or a 1-token decode step): x (1,S,dim), input_pos (S,)
    RoPE positions, kv_pos (S,) cache slots.  Returns logits (1,1,vocab) of the LAST token; caches updated in place.
    reference: modules/v2/ar.py:239-267 (forward_generate), :75-93 (KVCache.update), :503-567 (Attention with GQA and
    a causal mask row per kv_pos over the whole cache), :624-651 (bf16-rounded RoPE table)."""
    D, H, Hkv, hd = cfg["dim"], cfg["n_head"], cfg["n_local_heads"], cfg["head_dim"]
    Lmax = cfg["max_seq_len"]
    tab = rope_table(Lmax, hd, cfg["rope_base"], bf16_round=True)[input_pos]          # (S, hd/2, 2)
    mask = torch.tril(torch.ones(Lmax, Lmax, dtype=torch.bool))[kv_pos]               # (S, Lmax)
    S = x.shape[1]
    kvd = Hkv * hd
    h = x
    for i in range(cfg["n_layer"]):
        p = f"model.layers.{i}"
        n = rmsnorm(h, sd[p + ".attention_norm.weight"].float(), cfg["norm_eps"])
        qkv = F.linear(n, sd[p + ".attention.wqkv.weight"].float())
        q, k, v = qkv.split([D, kvd, kvd], dim=-1)

        def rot(t, nh):
            t = t.reshape(1, S, nh, hd // 2, 2)
            c, s_ = tab[None, :, None, :, 0], tab[None, :, None, :, 1]
            return torch.stack([t[..., 0] * c - t[..., 1] * s_, t[..., 1] * c + t[..., 0] * s_], -1).reshape(1, S, nh, hd)

        q = rot(q, H).transpose(1, 2)                       # (1,H,S,hd)
        k = rot(k, Hkv).transpose(1, 2)
        v = v.reshape(1, S, Hkv, hd).transpose(1, 2)
        kc, vc = caches[i]
        kc[:, :, kv_pos] = k.to(kc.dtype)
        vc[:, :, kv_pos] = v.to(vc.dtype)
        kk = kc.float().repeat_interleave(H // Hkv, dim=1)
        vv = vc.float().repeat_interleave(H // Hkv, dim=1)
        sc = torch.matmul(q, kk.transpose(-1, -2)) / math.sqrt(hd)
        sc = sc.masked_fill(~mask[None, None], float("-inf"))
        y = torch.matmul(torch.softmax(sc, dim=-1), vv).transpose(1, 2).reshape(1, S, D)
        h = h + F.linear(y, sd[p + ".attention.wo.weight"].float())
        n = rmsnorm(h, sd[p + ".ffn_norm.weight"].float(), cfg["norm_eps"])
        a = F.linear(n, sd[p + ".feed_forward.w1.weight"].float())
        b = F.linear(n, sd[p + ".feed_forward.w3.weight"].float())
        h = h + F.linear(silu(a) * b, sd[p + ".feed_forward.w2.weight"].float())
    last = rmsnorm(h[:, -1:], sd["model.norm.weight"].float(), cfg["norm_eps"])
    return F.linear(last, sd["model.output.weight"].float())


def ar_logits_to_probs(logits, previous_tokens=None, suppress_tokens=None, temperature=0.7, top_p=0.7,
                       repetition_penalty=1.5):
    """logits (vocab,) -> probs.  reference: modules/v2/ar.py:731-763"""
    logits = logits.clone()
    if previous_tokens is not None:
        pt = previous_tokens.long()
        score = logits[pt]
        score = torch.where(score < 0, score * repetition_penalty, score / repetition_penalty)
        logits[pt] = score
    if suppress_tokens is not None:
        for t in suppress_tokens:
            logits[t] = -float("inf")
    sorted_logits, sorted_idx = torch.sort(logits, descending=True)
    cum = torch.cumsum(torch.softmax(sorted_logits, dim=-1), dim=-1)
    remove_sorted = cum > top_p
    remove_sorted[0] = False
    remove = torch.zeros_like(remove_sorted)
    remove[sorted_idx] = remove_sorted
    logits = logits.masked_fill(remove, -float("inf"))
    logits = logits / max(temperature, 1e-5)
    return torch.softmax(logits, dim=-1)


def ar_sample(probs, exp_noise):
    """argmax(probs / q), q ~ Exp(1) supplied by the caller.  reference: modules/v2/ar.py:723-727"""
    return torch.argmax(probs / exp_noise, dim=-1, keepdim=True).to(torch.int)


def ar_generate(sd, cfg, prompt_text, prompt_target, exp_noise, temperature=0.7, top_p=0.7, repetition_penalty=1.5,
                max_iters=4000):
    """`NaiveWrapper.generate` (modules/v2/ar.py:382-422), B = 1.  exp_noise (n, vocab): row t is the Exp(1) draw used
    for token t.  Quirks kept: EOS suppressed while fewer than 10 tokens exist; the repetition penalty only ever sees
    `previous_tokens[0]`, i.e. the FIRST generated token (`decode_one_token_ar` indexes the 1-D tensor of all previous
    tokens with [0], ar.py:442-444)."""
    V, eos = cfg["vocab_size"], cfg["vocab_size"] - 1
    sep = sd["sep_token_emb"].reshape(1, 1, -1)
    emb = sd["model.embeddings.weight"]
    tgt_emb = emb[prompt_target[0]][None]
    emb_seq = torch.cat([sep, prompt_text, sep, tgt_emb], dim=1)
    input_pos = torch.cat([torch.arange(prompt_text.size(1) + 1), torch.tensor([0]), torch.arange(tgt_emb.size(1)) + 1])
    kv_pos = torch.arange(emb_seq.size(1))
    caches = ar_new_cache(cfg)
    lg = ar_forward_generate(sd, cfg, emb_seq, input_pos, kv_pos, caches)
    pr = ar_logits_to_probs(lg[0, -1], None, [eos], temperature, top_p, repetition_penalty)
    codes = [int(ar_sample(pr, exp_noise[0]))]
    for _ in range(max_iters):
        t = len(codes)
        if int(kv_pos[-1]) + 1 >= cfg["max_seq_len"]:
            break
        input_pos, kv_pos = input_pos[-1:] + 1, kv_pos[-1:] + 1
        lg = ar_forward_generate(sd, cfg, emb[codes[-1]].reshape(1, 1, -1), input_pos, kv_pos, caches)
        pr = ar_logits_to_probs(lg[0, -1], torch.tensor(codes[0]), [eos] if t < 10 else None, temperature, top_p,
                                repetition_penalty)
        nxt = int(ar_sample(pr, exp_noise[t]))
        if nxt == eos:
            break
        codes.append(nxt)
    return torch.tensor(codes, dtype=torch.long)[None, :]


# ----------------------------------------------------------------------------- chunk / crossfade harness
def crossfade(chunk1, chunk2, overlap):
    # reference: inference.py:343-350 (numpy float64 fades applied to float32 chunks, in place on chunk2)
    fade_out = np.cos(np.linspace(0, np.pi / 2, overlap)) ** 2
    fade_in = np.cos(np.linspace(np.pi / 2, 0, overlap)) ** 2
    if len(chunk2) < overlap:
        chunk2[:overlap] = chunk2[:overlap] * fade_in[:len(chunk2)] + (chunk1[-overlap:] * fade_out)[:len(chunk2)]
    else:
        chunk2[:overlap] = chunk2[:overlap] * fade_in + chunk1[-overlap:] * fade_out
    return chunk2


def chunked_convert(sample_fn, vocode_fn, cond, prompt_condition, mel2, style2, hop, max_context_window,
                    overlap_frame_len=16):
    """The driver loop around the hot path.  sample_fn(cat_condition) -> mel (1,C,T) incl. prompt frames,
    vocode_fn(mel (1,C,S)) -> (L,) waveform tensor.  reference: inference.py:470-527"""
    overlap_wave_len = overlap_frame_len * hop
    max_source_window = max_context_window - mel2.size(2)
    processed = 0
    chunks = []
    previous = None
    while processed < cond.size(1):
        chunk_cond = cond[:, processed:processed + max_source_window]
        is_last = processed + max_source_window >= cond.size(1)
        cat_condition = torch.cat([prompt_condition, chunk_cond], dim=1)
        vc_target = sample_fn(cat_condition)[:, :, mel2.size(-1):]
        vc_wave = vocode_fn(vc_target.float()).reshape(1, -1)
        if processed == 0:
            if is_last:
                chunks.append(vc_wave[0].numpy())
                break
            chunks.append(vc_wave[0, :-overlap_wave_len].numpy())
            previous = vc_wave[0, -overlap_wave_len:]
            processed += vc_target.size(2) - overlap_frame_len
        elif is_last:
            chunks.append(crossfade(previous.numpy(), vc_wave[0].numpy(), overlap_wave_len))
            processed += vc_target.size(2) - overlap_frame_len
            break
        else:
            chunks.append(crossfade(previous.numpy(), vc_wave[0, :-overlap_wave_len].numpy(), overlap_wave_len))
            previous = vc_wave[0, -overlap_wave_len:]
            processed += vc_target.size(2) - overlap_frame_len
    return torch.tensor(np.concatenate(chunks))[None, :].float()


# ----------------------------------------------------------------------------------------- length regulator (8f row 1)
def lr_f0_to_coarse(f0, n_bins):
    """f0 (Hz) -> bin index (modules/length_regulator.py:15-26 + the clamp at :129-130).  The reference's last
    line never fires after the line before it zeroed every value >= n_bins, so overflowing f0 maps to bin 0."""
    import math
    mel_min = 1127 * math.log(1 + 50.0 / 700)
    mel_max = 1127 * math.log(1 + 1100.0 / 700)
    mel = 1127 * (1 + f0 / 700).log()
    a = (n_bins - 2) / (mel_max - mel_min)
    b = mel_min * a - 1.0
    mel = torch.where(mel > 0, mel * a - b, mel)
    c = torch.round(mel).long()
    c = c * (c > 0)
    c = c + ((c < 1) * 1)
    c = c * (c < n_bins)
    c = c + ((c >= n_bins) * (n_bins - 1))
    return c.clamp(0, n_bins - 1)


def _nearest_index(n_out, n_in):
    """F.interpolate(mode='nearest'): src = min(floor(dst * float32(n_in / n_out)), n_in - 1)."""
    scale = torch.tensor(n_in, dtype=torch.float32) / torch.tensor(n_out, dtype=torch.float32)
    idx = torch.floor(torch.arange(n_out, dtype=torch.float32) * scale).long()
    return idx.clamp(max=n_in - 1)


def lr_forward(sd, cfg, x, ylen, f0=None):
    """One utterance (B = 1) through InterpolateRegulator.forward (modules/length_regulator.py:90-141; v2
    modules/v2/length_regulator.py:74-105).  x (1, Tin, in) fp32 or (1, Tin) int64 tokens; -> (1, T, out), T = ylen
    (or min(ylen, Tin) rows kept of Tin when the model has no conv stack)."""
    C = cfg["channels"]
    if cfg["is_discrete"]:
        h = sd["embedding.weight"][x[0]]                                    # (Tin, C)
    else:
        h = x[0] @ sd["content_in_proj.weight"].t() + sd["content_in_proj.bias"]
    tin = h.shape[0]
    interpolate = cfg["n_convs"] > 0
    T = ylen if interpolate else tin
    if interpolate:
        h = h[_nearest_index(T, tin)]
    if cfg["f0_condition"]:
        if f0 is None:
            h = h + sd["f0_mask"]
        else:
            q = lr_f0_to_coarse(f0[0], cfg["n_f0_bins"])
            h = h + sd["f0_embedding.weight"][q][_nearest_index(T, q.shape[0])]
    h = h.t()[None]                                                         # (1, C, T)
    for i in range(cfg["n_convs"]):
        h = F.conv1d(h, sd[f"model.{3 * i}.weight"], sd[f"model.{3 * i}.bias"], padding=1)
        mean = h.mean()
        var = ((h - mean) ** 2).mean()
        h = (h - mean) * torch.rsqrt(var + 1e-5) * sd[f"model.{3 * i + 1}.weight"][None, :, None] + sd[f"model.{3 * i + 1}.bias"][None, :, None]
        sp = torch.where(h > 20, h, torch.log1p(torch.exp(torch.clamp(h, max=20.0))))
        h = h * torch.tanh(sp)
    kt = f"model.{3 * cfg['n_convs']}.weight"
    if kt in sd:
        h = F.conv1d(h, sd[kt], sd[f"model.{3 * cfg['n_convs']}.bias"])
    out = h.transpose(1, 2).contiguous()
    if not interpolate:
        keep = min(ylen, tin)
        out = out.clone()
        if cfg["version"] == 1:
            out[:, keep:] = 0                                               # v1 masks by the clamped ylens; v2 does not
    return out


# ----------------------------------------------------------------------------------------- mel front-end (8f row 3)
def mel_spectrogram(y, mel_basis, n_fft, hop_size, win_size):
    """modules/audio.py:45-82 with center=False: reflect pad (n_fft - hop)/2, torch.stft (Hann), sqrt(|.|^2 + 1e-9),
    mel matmul, log(clamp(., 1e-5)).  y (B, L) -> (B, n_mels, frames)."""
    pad = int((n_fft - hop_size) / 2)
    y = F.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.view_as_real(torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=torch.hann_window(win_size),
                                         center=False, pad_mode="reflect", normalized=False, onesided=True, return_complex=True))
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(mel_basis, spec), min=1e-5))


# ----------------------------------------------------------------------------- CAMPPlus style encoder (8f row 3, second half)
def _bn(x, sd, p, affine=True, eps=1e-5):
    """eval-mode BatchNorm over dim 1 (running statistics)."""
    shape = [1, -1] + [1] * (x.dim() - 2)
    y = (x - sd[p + ".running_mean"].reshape(shape)) / torch.sqrt(sd[p + ".running_var"].reshape(shape) + eps)
    if affine:
        y = y * sd[p + ".weight"].reshape(shape) + sd[p + ".bias"].reshape(shape)
    return y


def campplus_forward(sd, c, feat):
    """`CAMPPlus.forward(x)` in eval mode, x (B, T, feat_dim) -> (B, embedding_size).
    reference: modules/campplus/DTDNN.py:132-137 (forward), :13-52 (FCM), layers.py:26-31 (statistics pooling),
    :96-131 (CAMLayer incl. seg_pooling), :134-171 (CAMDenseTDNNLayer), :253-295 (BasicResBlock)."""
    F = torch.nn.functional
    x = feat.permute(0, 2, 1).unsqueeze(1)                                   # (B, 1, F, T)
    out = torch.relu(_bn(F.conv2d(x, sd["head.conv1.weight"], padding=1), sd, "head.bn1"))
    for layer in ("layer1", "layer2"):
        for b in range(2):
            p = f"head.{layer}.{b}"
            stride = (2, 1) if b == 0 else (1, 1)
            y = torch.relu(_bn(F.conv2d(out, sd[p + ".conv1.weight"], stride=stride, padding=1), sd, p + ".bn1"))
            y = _bn(F.conv2d(y, sd[p + ".conv2.weight"], padding=1), sd, p + ".bn2")
            sc = out
            if b == 0:
                sc = _bn(F.conv2d(out, sd[p + ".shortcut.0.weight"], stride=stride), sd, p + ".shortcut.1")
            out = torch.relu(y + sc)
    out = torch.relu(_bn(F.conv2d(out, sd["head.conv2.weight"], stride=(2, 1), padding=1), sd, "head.bn2"))
    x = out.reshape(out.shape[0], out.shape[1] * out.shape[2], out.shape[3])      # (B, 32 * F/8, T), channel = c * F/8 + f
    x = torch.relu(_bn(F.conv1d(x, sd["xvector.tdnn.linear.weight"], stride=2, padding=2), sd, "xvector.tdnn.nonlinear.batchnorm"))
    seg = c["seg_len"]
    for bi, (nl, k, dil) in enumerate(zip(c["block_layers"], c["block_kernel"], c["block_dilation"])):
        for i in range(nl):
            p = f"xvector.block{bi + 1}.tdnnd{i + 1}"
            h = F.conv1d(torch.relu(_bn(x, sd, p + ".nonlinear1.batchnorm")), sd[p + ".linear1.weight"])
            h = torch.relu(_bn(h, sd, p + ".nonlinear2.batchnorm"))
            y = F.conv1d(h, sd[p + ".cam_layer.linear_local.weight"], padding=(k - 1) // 2 * dil, dilation=dil)
            T2 = h.shape[-1]
            sp = F.avg_pool1d(h, kernel_size=seg, stride=seg, ceil_mode=True)
            sp = sp.unsqueeze(-1).expand(*sp.shape, seg).reshape(*sp.shape[:-1], -1)[..., :T2]
            ctx = h.mean(-1, keepdim=True) + sp
            ctx = torch.relu(F.conv1d(ctx, sd[p + ".cam_layer.linear1.weight"], sd[p + ".cam_layer.linear1.bias"]))
            m = torch.sigmoid(F.conv1d(ctx, sd[p + ".cam_layer.linear2.weight"], sd[p + ".cam_layer.linear2.bias"]))
            x = torch.cat([x, y * m], dim=1)
        p = f"xvector.transit{bi + 1}"
        x = F.conv1d(torch.relu(_bn(x, sd, p + ".nonlinear.batchnorm")), sd[p + ".linear.weight"])
    x = torch.relu(_bn(x, sd, "xvector.out_nonlinear.batchnorm"))
    stats = torch.cat([x.mean(dim=-1), x.std(dim=-1, unbiased=True)], dim=-1)
    e = F.conv1d(stats.unsqueeze(-1), sd["dense.linear.weight"]).squeeze(-1)
    return _bn(e, sd, "dense.nonlinear.batchnorm", affine=False)


def kaldi_fbank(wave, num_mel_bins=80, sample_frequency=16000.0, frame_length_ms=25.0, frame_shift_ms=10.0,
                preemphasis=0.97, low_freq=20.0, high_freq=0.0):
    """`torchaudio.compliance.kaldi.fbank(waveform, num_mel_bins=80, dither=0, sample_frequency=16000)` as the drivers call it
    (inference.py:418-428), restated from the published Kaldi algorithm (torchaudio 2.x, compliance/kaldi.py: snip_edges,
    remove_dc_offset, pre-emphasis 0.97 with the first sample against itself, Povey window, power spectrum of the frame
    zero-padded to 512, triangular filters on the Kaldi mel scale 1127 ln(1 + f / 700) from 20 Hz to Nyquist, log with
    the float epsilon floor).  torchaudio is absent from the build image: PARITY UNPINNED for this function (the CAMPPlus
    network itself is pinned by reference outputs).  wave (1, L) or (L,) -> (frames, num_mel_bins)."""
    w = wave.reshape(-1).to(torch.float64)
    win = int(sample_frequency * frame_length_ms * 0.001)
    shift = int(sample_frequency * frame_shift_ms * 0.001)
    nfft = 1
    while nfft < win:
        nfft *= 2
    if w.numel() < win:
        return torch.zeros(0, num_mel_bins)
    n_frames = 1 + (w.numel() - win) // shift
    idx = torch.arange(win)[None, :] + shift * torch.arange(n_frames)[:, None]
    fr = w[idx].to(torch.float32)                                             # torchaudio computes in the input dtype
    fr = fr - fr.mean(dim=1, keepdim=True)
    prev = torch.cat([fr[:, :1], fr[:, :-1]], dim=1)
    fr = fr - preemphasis * prev
    window = torch.hann_window(win, periodic=False, dtype=torch.float32).pow(0.85)
    fr = fr * window[None, :]
    fr = torch.nn.functional.pad(fr, (0, nfft - win))
    spec = torch.fft.rfft(fr).abs().pow(2.0)                                  # (frames, nfft/2 + 1)
    nyq = 0.5 * sample_frequency
    hi = high_freq + nyq if high_freq <= 0.0 else high_freq
    mel = lambda f: 1127.0 * torch.log(1.0 + torch.as_tensor(f, dtype=torch.float32) / 700.0)      # noqa: E731
    mlo, mhi = mel(low_freq), mel(hi)
    delta = (mhi - mlo) / (num_mel_bins + 1)
    b = torch.arange(num_mel_bins, dtype=torch.float32)[:, None]
    left, center, right = mlo + b * delta, mlo + (b + 1.0) * delta, mlo + (b + 2.0) * delta
    fmel = mel((sample_frequency / nfft) * torch.arange(nfft // 2, dtype=torch.float32))[None, :]
    up, down = (fmel - left) / (center - left), (right - fmel) / (right - center)
    fb = torch.clamp(torch.minimum(up, down), min=0.0)                        # (bins, nfft/2)
    fb = torch.nn.functional.pad(fb, (0, 1))
    e = spec @ fb.T
    return torch.log(torch.clamp(e, min=torch.finfo(torch.float32).eps))
