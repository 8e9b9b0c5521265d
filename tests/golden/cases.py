"""Golden-case catalogue shared by `make_golden.py` (which runs the real reference in the build
container) and the test-suite (which replays the same seeded inputs through the oracle / HIP path).

Inputs are never stored: they are regenerated from (case name, seed) with a counter-based RNG, as are
the weights (`seedvc_amd.weights`).  Only the reference's OUTPUTS are committed (tests/golden/*.npz).
"""
import zlib
import numpy as np
import torch

from seedvc_amd import specs, weights


def _gen(tag, seed):
    return np.random.Generator(np.random.Philox(key=[zlib.crc32(tag.encode()), seed & 0xFFFFFFFF]))


def randn(tag, seed, *shape):
    return torch.from_numpy(_gen(tag, seed).standard_normal(shape).astype(np.float32))


def rand(tag, seed, *shape):
    return torch.from_numpy(_gen(tag, seed).random(shape).astype(np.float32))


def logmel(tag, seed, *shape):
    """Log-mel-range prompt: N(-4, 2^2) clamped to [-11.5, 2]."""
    return (randn(tag, seed, *shape) * 2.0 - 4.0).clamp(-11.5, 2.0)


# ---- DiT / CFM cases -----------------------------------------------------------------------------
# name -> (preset, overrides, T, P, n_steps, cfg_rate, seed)
DIT_CASES = {
    # reduced architectures (fast, many variants)
    "tiny_r":  ("tiny",  dict(D=128, H=2, L=5), 40, 16, 4, 0.7, 11),
    "small_r": ("small", dict(D=128, H=2, L=5, Dc=96, wn_dim=128, wn_layers=3), 44, 12, 4, 0.7, 12),
    "base_r":  ("base",  dict(D=192, H=3, L=5, Dc=160, C=128), 36, 10, 3, 0.5, 13),
    "v2_r":    ("v2",    dict(D=128, H=2, L=4, block_size=256), 38, 14, 4, [0.7, 0.7], 14),
    "v2_r_sim": ("v2",   dict(D=128, H=2, L=4, block_size=256), 38, 14, 3, [0.0, 0.7], 15),
    "small_r_nocfg": ("small", dict(D=128, H=2, L=5, Dc=96, wn_dim=128, wn_layers=3), 44, 12, 3, 0.0, 16),
    # the remaining v2 guidance branches (modules/v2/cfm.py:77-89,102-112): random_voice, intelligibility only, no CFG
    "v2_r_rv":    ("v2", dict(D=128, H=2, L=4, block_size=256), 38, 14, 3, [0.7, 0.7], 17, dict(random_voice=True)),
    "v2_r_int":   ("v2", dict(D=128, H=2, L=4, block_size=256), 38, 14, 3, [0.7, 0.0], 18),
    "v2_r_nocfg": ("v2", dict(D=128, H=2, L=4, block_size=256), 38, 14, 3, [0.0, 0.0], 19),
    # full-size architectures, short sequences
    "tiny_full":  ("tiny",  {}, 96, 40, 3, 0.7, 21),
    "small_full": ("small", {}, 96, 40, 3, 0.7, 22),
    "base_full":  ("base",  {}, 64, 24, 2, 0.7, 23),
    "v2_full":    ("v2",    {}, 64, 24, 2, [0.7, 0.7], 24),
}


def dit_case(name):
    preset, ov, T, P, n_steps, rate, seed = DIT_CASES[name][:7]
    extra = DIT_CASES[name][7] if len(DIT_CASES[name]) > 7 else {}
    cfg = specs.dit_config(preset, **ov)
    sd = weights.make_state_dict(specs.dit_state_spec(cfg), seed=seed, prefix=f"dit.{preset}.")
    C, Dc, S = cfg["C"], cfg["Dc"], cfg["style_dim"]
    inp = dict(
        z=randn(name + ".z", seed, 1, C, T),
        mu=randn(name + ".mu", seed, 1, T, Dc),
        prompt=logmel(name + ".prompt", seed, 1, C, P),
        style=randn(name + ".style", seed, 1, S),
        x=randn(name + ".x", seed, 1, C, T),          # estimator-call probe
        t=torch.tensor([0.37], dtype=torch.float32),
    )
    return cfg, sd, inp, dict(T=T, P=P, n_steps=n_steps, cfg_rate=rate, seed=seed, random_voice=bool(extra.get("random_voice", False)))


# ---- vocoder cases -------------------------------------------------------------------------------
BIGVGAN_CASES = {
    # name -> (preset, overrides, S, B, seed)
    "bigvgan_r":  ("22k", dict(upsample_initial_channel=64, num_mels=20), 12, 2, 31),
    "bigvgan_r2": ("22k", dict(upsample_initial_channel=128, num_mels=80, upsample_rates=[4, 2],
                               upsample_kernel_sizes=[8, 4]), 24, 1, 32),
    "bigvgan_full": ("22k", {}, 8, 1, 33),
}


def bigvgan_case(name):
    preset, ov, S, B, seed = BIGVGAN_CASES[name]
    h = specs.bigvgan_config(preset, **ov)
    sd = weights.make_state_dict(specs.bigvgan_state_spec(h), seed=seed, prefix="bigvgan.")
    mel = logmel(name + ".mel", seed, B, h["num_mels"], S)
    return h, sd, mel, dict(S=S, B=B, seed=seed)


HIFT_CASES = {
    # name -> (overrides, S, B, seed)
    "hift_r": (dict(base_channels=64, f0_cond_channels=48), 10, 2, 41),
    "hift_full": ({}, 6, 1, 42),
}


def hift_case(name):
    ov, S, B, seed = HIFT_CASES[name]
    c = specs.hift_config(**ov)
    sd = weights.make_state_dict(specs.hift_state_spec(c), seed=seed, prefix="hift.")
    mel = logmel(name + ".mel", seed, B, c["in_channels"], S)
    Lw = S * specs.hift_total_upsample(c)
    phase0 = (rand(name + ".phase", seed, B, c["nb_harmonics"] + 1, 1) * 2 - 1) * float(np.pi)
    noise = randn(name + ".noise", seed, B, c["nb_harmonics"] + 1, Lw)
    return c, sd, mel, phase0, noise, dict(S=S, B=B, seed=seed)


ACT_CASES = {
    # name -> (B, C, L, seed)
    "act_small": (2, 5, 37, 51),
    "act_edge": (1, 3, 4, 52),
    "act_mid": (1, 24, 300, 53),
}


def act_case(name):
    B, C, L, seed = ACT_CASES[name]
    x = randn(name + ".x", seed, B, C, L) * 2.0
    alpha = randn(name + ".alpha", seed, C) * 0.4
    beta = randn(name + ".beta", seed, C) * 0.4
    return x, alpha, beta


# ---- v2 AR decode step ---------------------------------------------------------------------------
AR_CASES = {
    # name -> (overrides, prefill tokens, decode steps, seed)
    "ar_r": (dict(dim=128, n_head=2, n_local_heads=1, n_layer=3, intermediate_size=256, vocab_size=65, max_seq_len=64), 9, 4, 61),
    "ar_full": ({}, 20, 3, 62),
    # contexts beyond one 512-key batch of the one-token attention kernels (their online-softmax loop over key batches: rescale
    # by exp(m_run - m_new), second and third batch) -- the reference's generate loop runs to 4000 tokens (ar.py:382-422)
    "ar_long600": (dict(dim=128, n_head=2, n_local_heads=1, n_layer=2, intermediate_size=256, vocab_size=65, max_seq_len=1280), 600, 3, 63),
    "ar_long1030": (dict(dim=128, n_head=2, n_local_heads=1, n_layer=2, intermediate_size=256, vocab_size=65, max_seq_len=1280), 1030, 4, 64),
}


def ar_case(name):
    ov, n_prefill, n_decode, seed = AR_CASES[name]
    c = specs.ar_config(**ov)
    sd = weights.make_state_dict(specs.ar_state_spec(c), seed=seed, prefix="ar.")
    x_prefill = randn(name + ".prefill", seed, 1, n_prefill, c["dim"])
    # prefill positions as in NaiveWrapper.generate: [0..n_text, 0, 1..] ; here: text part then target part
    n_text = n_prefill // 2
    input_pos = list(range(n_text + 1)) + list(range(n_prefill - n_text - 1))
    x_steps = randn(name + ".steps", seed, n_decode, 1, 1, c["dim"])
    exp_noise = -torch.log(rand(name + ".expn", seed, n_decode, c["vocab_size"]).clamp_min(1e-9))
    return c, sd, x_prefill, input_pos, x_steps, exp_noise, dict(n_prefill=n_prefill, n_decode=n_decode, seed=seed)


# ------------------------------------------------------------------------------------------ length regulator (8f row 1)
LR_CASES = {
    # name -> (preset, overrides, Tin, ylen, Tf0 (0 = no f0 given), seed)
    "lr_tiny_r": ("tiny", dict(channels=64, in_channels=80), 23, 41, 0, 71),          # in_channels off the k-tile
    "lr_base_r": ("base", dict(channels=96, in_channels=64, n_f0_bins=256), 31, 37, 45, 72),
    "lr_base_r_nof0": ("base", dict(channels=96, in_channels=64, n_f0_bins=256), 31, 26, 0, 73),   # f0_mask branch, shrink
    "lr_v2_r": ("v2_cfm", dict(channels=64, codebook_size=50), 19, 33, 0, 74),
    "lr_v2_ar_r": ("v2_ar", dict(channels=64), 17, 17, 0, 75),                         # embedding only
    "lr_small_full": ("small", {}, 250, 430, 0, 76),
    "lr_base_full": ("base", {}, 250, 430, 500, 77),
}


def lr_case(name):
    preset, ov, tin, ylen, tf0, seed = LR_CASES[name]
    c = specs.lr_config(preset, **ov)
    sd = weights.make_state_dict(specs.lr_state_spec(c), seed=seed, prefix="lr.")
    if c["is_discrete"]:
        x = (rand(name + ".tok", seed, 1, tin) * c["codebook_size"]).long().clamp(max=c["codebook_size"] - 1)
    else:
        x = randn(name + ".x", seed, 1, tin, c["in_channels"])
    f0 = None
    if tf0:
        u = rand(name + ".f0", seed, 1, tf0)
        f0 = 60.0 + 900.0 * u                       # voiced range
        f0 = torch.where(rand(name + ".uv", seed, 1, tf0) < 0.25, torch.zeros_like(f0), f0)      # unvoiced frames
        f0[0, ::17] = 1500.0                        # above f0_max: exercises the reference's overflow-to-bin-0 quirk
    return c, sd, x, ylen, f0, dict(tin=tin, seed=seed)


# AR generate loop (8f row 4): name -> (overrides, Tt condition frames, Tp prompt tokens, seed)
AR_GEN_CASES = {
    "ar_gen_r": (dict(dim=128, n_head=2, n_local_heads=1, n_layer=3, intermediate_size=256, vocab_size=33, max_seq_len=160), 7, 5, 81),
    "ar_gen_r2": (dict(dim=128, n_head=2, n_local_heads=1, n_layer=2, intermediate_size=256, vocab_size=17, max_seq_len=96), 4, 0, 82),
}


def ar_gen_case(name):
    ov, tt, tp, seed = AR_GEN_CASES[name]
    c = specs.ar_config(**ov)
    sd = weights.make_state_dict(specs.ar_state_spec(c), seed=seed, prefix="ar.")
    text = randn(name + ".text", seed, 1, tt, c["dim"])
    target = (rand(name + ".tgt", seed, 1, tp) * (c["vocab_size"] - 1)).long()
    n_noise = c["max_seq_len"]
    exp_noise = -torch.log(rand(name + ".expn", seed, n_noise, c["vocab_size"]).clamp_min(1e-9))
    return c, sd, text, target, exp_noise


# BASELINE configs[4] size: the full ar_base model (configs/v2/vc_wrapper.yaml), 120 condition frames + 200 prompt tokens,
# AR_GEN_FULL_TOKENS generated tokens.  A random-weight model never emits EOS, so the reference loop is cut after that
# many tokens (make_golden.py limits its `tqdm(range(4000))`); the output head is scaled so the sampling distribution is
# peaked like a trained model's (random weights give a flat 2049-way distribution whose top-p boundary moves with
# fp16 rounding), and the Exp(1) draw of every step's winning token is divided by AR_GEN_FULL_BOOST so that the
# exponential race is decided by a clear margin (`winners` = the committed reference tokens; boosting the winner
# cannot change the winner, so the trajectory of the reference run with the plain draws is reproduced).
AR_GEN_FULL = ("ar_gen_full", 120, 200, 83)
AR_GEN_FULL_TOKENS = 160
AR_GEN_FULL_BOOST = 4.0
AR_GEN_FULL_HEAD_SCALE = 8.0


def ar_gen_full_case(winners=None):
    name, tt, tp, seed = AR_GEN_FULL
    c = specs.ar_config()
    sd = weights.make_state_dict(specs.ar_state_spec(c), seed=seed, prefix="ar.")
    sd["model.output.weight"] = sd["model.output.weight"] * AR_GEN_FULL_HEAD_SCALE
    text = randn(name + ".text", seed, 1, tt, c["dim"])
    target = (rand(name + ".tgt", seed, 1, tp) * (c["vocab_size"] - 1)).long()
    exp_noise = -torch.log(rand(name + ".expn", seed, AR_GEN_FULL_TOKENS, c["vocab_size"]).clamp_min(1e-9))
    if winners is not None:
        w = torch.as_tensor(winners).reshape(-1).long()
        exp_noise[torch.arange(w.numel()), w] /= AR_GEN_FULL_BOOST
    return c, sd, text, target, exp_noise


# mel front-end (8f row 3): name -> (n_fft, hop, n_mels, sr, fmin, fmax, L samples, B, seed)
MEL_CASES = {
    "mel_r": (64, 16, 10, 4000, 0, None, 400, 2, 91),
    "mel_22k": (1024, 256, 80, 22050, 0, None, 22050, 1, 92),          # presets: config_dit_mel_seed_uvit_*.yml (1 s clip)
    "mel_44k": (2048, 512, 128, 44100, 0, None, 30000, 2, 93),         # 44.1 kHz preset
}


def mel_case(name):
    from seedvc_amd.audio import slaney_mel_basis
    n_fft, hop, n_mels, sr, fmin, fmax, L, B, seed = MEL_CASES[name]
    t = torch.arange(L, dtype=torch.float32)[None] / sr
    f1 = 110.0 + 600.0 * rand(name + ".f", seed, B, 1)
    y = 0.4 * torch.sin(2 * np.pi * f1 * t) + 0.2 * torch.sin(2 * np.pi * 3.1 * f1 * t) + 0.05 * randn(name + ".n", seed, B, L)
    y[:, L // 2: L // 2 + L // 8] *= 0.001                                   # a near-silent stretch (clamp / log floor)
    basis = slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax)
    return dict(n_fft=n_fft, hop=hop, n_mels=n_mels, sr=sr, fmin=fmin, fmax=fmax), y.clamp(-1, 1), basis


# CAMPPlus style encoder (8f row 3, second half): name -> (overrides, T feature frames, B, seed)
CAMPPLUS_CASES = {
    "campplus_r": (dict(block_layers=(2, 3, 2)), 57, 2, 95),        # fewer dense layers, T2 < seg_len
    "campplus_full": ({}, 500, 1, 96),                              # the drivers' model on a 5 s reference (T2 = 250: 3 segments)
}


def campplus_case(name):
    ov, T, B, seed = CAMPPLUS_CASES[name]
    c = specs.campplus_config(**ov)
    sd = weights.make_state_dict(specs.campplus_state_spec(c), seed=seed, prefix="campplus.")
    feat = randn(name + ".feat", seed, B, T, c["feat_dim"]) * 2.0
    feat = feat - feat.mean(dim=1, keepdim=True)                    # the drivers mean-normalise the fbank (inference.py:429)
    return c, sd, feat


# ---- full-size cases (BASELINE.json sizes: P = S = 430 frames) ------------------------------------------------------
# Outputs of the REFERENCE at full size, stored decimated so the fixture stays small: every FS_MEL_STEP-th generated mel
# frame of the sampler, FS_WAVE_NWIN evenly spaced windows of FS_WAVE_WIN samples of the vocoder waveform.  They let the
# GPU tests hold the HIP path to the reference at BASELINE sizes without a minutes-long CPU oracle run in the test.
FS_P = FS_S = 430
FS_T = FS_P + FS_S
FS_MEL_STEP = 2
FS_WAVE_WIN, FS_WAVE_NWIN = 4096, 8
# name -> (preset, input seed, n_steps, cfg_rate)
FULLSIZE_CFM = {"fs_small": ("small", 300, 25, 0.7), "fs_base": ("base", 500, 50, 0.7), "fs_v2": ("v2", 600, 25, [0.7, 0.7])}
# name -> (preset, weight seed, input seed)
FULLSIZE_VOC = {"fs_bigvgan22k": ("22k", 1234, 301), "fs_bigvgan44k": ("44k", 1234, 501)}


def baseline_inputs(cfg, B, seed, T=FS_T, P=FS_P):
    return dict(mu=randn("bs.mu", seed, B, T, cfg["Dc"]), prompt=logmel("bs.prompt", seed, B, cfg["C"], P),
                style=randn("bs.style", seed, B, cfg["style_dim"]), z=randn("bs.z", seed, B, cfg["C"], T))


def fullsize_cfm_case(name):
    preset, seed, n_steps, cfg_rate = FULLSIZE_CFM[name]
    cfg = specs.dit_config(preset)
    sd = weights.make_state_dict(specs.dit_state_spec(cfg), seed=1234, prefix=f"dit.{preset}.")
    return cfg, sd, baseline_inputs(cfg, 1, seed), dict(n_steps=n_steps, cfg_rate=cfg_rate, T=FS_T, P=FS_P)


def fullsize_voc_case(name):
    preset, wseed, iseed = FULLSIZE_VOC[name]
    h = specs.bigvgan_config(preset)
    sd = weights.make_state_dict(specs.bigvgan_state_spec(h), seed=wseed, prefix="bigvgan.")
    return h, sd, logmel(name + ".mel", iseed, 1, h["num_mels"], FS_S)


def fs_wave_windows(n):
    """Start offsets of the stored waveform windows of an n-sample output."""
    return [int(round(i * (n - FS_WAVE_WIN) / (FS_WAVE_NWIN - 1))) for i in range(FS_WAVE_NWIN)]


# ---- chunk loop (a21): the drivers' while-loop + `_stream_wave_chunks`, driven with a fake sampler / vocoder ----------
# The fakes use only single fp32 multiplies / adds per element (exactly rounded everywhere), so the host that replays
# them reproduces the build container's bits.  name -> (n_src source frames, seed); every case: P = 20 prompt frames,
# max_context_window = 60 (=> 40 source frames per chunk, advance 24), hop 8, 16-frame overlap (128 samples).
CHUNK_C, CHUNK_DC, CHUNK_HOP, CHUNK_P, CHUNK_WINDOW, CHUNK_OVERLAP = 4, 6, 8, 20, 60, 16
CHUNKLOOP_CASES = {"loop1": (40, 111), "loop1s": (17, 112), "loop2": (41, 113), "loop2b": (64, 114), "loop4": (100, 115),
                   "loop5": (113, 116)}
# `_stream_wave_chunks` driven directly, chunk by chunk: name -> ([frames of each chunk], seed).  "short": the last chunk is
# SHORTER than the overlap (10 < 16 frames: the `len(chunk2) < overlap` branch of crossfade, which the drivers' own
# window arithmetic never reaches)
CHUNKSTREAM_CASES = {"short": ([40, 10], 121), "short3": ([40, 33, 3], 122), "even": ([24, 24, 24, 24], 123)}


def fake_sampler(cat_condition, P):
    """Stand-in for `cfm.inference` in the chunk-loop cases: (1, T, Dc) -> (1, C, T), a fixed fp32 mix of the condition."""
    m = cat_condition[0].float().t()                                   # (Dc, T)
    rows = [m[c] * 0.5 + m[c + 1] * float(c + 1) * 0.25 for c in range(CHUNK_C)]
    return torch.stack(rows)[None].contiguous()


def fake_vocoder(mel):
    """Stand-in for `vocoder_fn`: (1, C, S) -> (1, 1, S * hop): sample t*hop + j = mel[0, t] * (j + 1) / 8 + mel[1, t]."""
    j = (torch.arange(CHUNK_HOP, dtype=torch.float32) + 1.0) * 0.125
    w = mel[0, 0][:, None] * j[None, :] + mel[0, 1][:, None]
    return w.reshape(1, 1, -1).contiguous()


def chunkloop_case(name):
    n_src, seed = CHUNKLOOP_CASES[name]
    return dict(cond=randn(name + ".cond", seed, 1, n_src, CHUNK_DC), prompt_condition=randn(name + ".pc", seed, 1, CHUNK_P, CHUNK_DC),
                mel2=logmel(name + ".mel2", seed, 1, CHUNK_C, CHUNK_P), style2=randn(name + ".style", seed, 1, 3))


def chunkstream_case(name):
    frames, seed = CHUNKSTREAM_CASES[name]
    return [randn(f"{name}.w{i}", seed, 1, f * CHUNK_HOP) for i, f in enumerate(frames)], frames


# ---- checkpoint loading (b): a synthetic .pth in the reference's layout through build_model + load_checkpoint ---------
CKPT_T, CKPT_P, CKPT_STEPS, CKPT_SEED = 64, 24, 10, 131


def ckpt_case():
    """tiny preset (configs/presets/config_dit_mel_seed_uvit_xlsr_tiny.yml): generated weights for `cfm.estimator` and the
    length regulator + one short utterance (configs[0]'s plumbing: 10 steps, cfg 0.7)."""
    cfg = specs.dit_config("tiny")
    sd = weights.make_state_dict(specs.dit_state_spec(cfg), seed=CKPT_SEED, prefix="dit.tiny.")
    lc = specs.lr_config("tiny")
    lsd = weights.make_state_dict(specs.lr_state_spec(lc), seed=CKPT_SEED, prefix="lr.")
    name = "ckpt"
    inp = dict(z=randn(name + ".z", CKPT_SEED, 1, cfg["C"], CKPT_T), mu=randn(name + ".mu", CKPT_SEED, 1, CKPT_T, cfg["Dc"]),
               prompt=logmel(name + ".prompt", CKPT_SEED, 1, cfg["C"], CKPT_P), style=randn(name + ".style", CKPT_SEED, 1, cfg["style_dim"]))
    return cfg, sd, lc, lsd, inp
