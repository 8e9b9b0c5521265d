"""Parity at BASELINE.json's own sizes (P = S = 430 frames, T = 860) through the C ABI.

configs[0] (tiny, 10 steps, one 5 s source + reference), one utterance of configs[2] (small+WaveNet + BigVGAN),
configs[3] (base 44.1 kHz, 50 steps + the 6-stage BigVGAN-44k architecture) and configs[4] (v2, 3-way CFG, 25 steps) are
checked end to end against the CPU oracle; configs[1] (B = 64) and the 30 s context window are checked through
size-independent properties: the result does not depend on how the batch is cut into sub-batches (bit for bit on the
same kernel path), every utterance agrees with its own B = 1 run (which runs on the tap-GEMM path instead of the fused
row-panel kernel: fp16-operand rounding apart), prompt frames of the output are zero, reruns are identical, and the
oracle is compared on sampled utterances."""
import pytest
import torch

import cases
import seedvc_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
P = S = 430
T = P + S
HIFT_OWN_F0_RMS = 1e-4      # measured 4.0e-6 at S = 430 (own f0, 5 s of phase integration): the drop-in call meets the north-star bound too


def _inputs(cfg, B, seed):
    return dict(mu=cases.randn("bs.mu", seed, B, T, cfg["Dc"]), prompt=cases.logmel("bs.prompt", seed, B, cfg["C"], P),
                style=cases.randn("bs.style", seed, B, cfg["style_dim"]), z=cases.randn("bs.z", seed, B, cfg["C"], T))


def _cfm(preset, seed=1234):
    from seedvc_amd.cfm import CFM
    cfg = cases.specs.dit_config(preset)
    sd = cases.weights.make_state_dict(cases.specs.dit_state_spec(cfg), seed=seed, prefix=f"dit.{preset}.")
    return CFM(cfg, sd, "cuda:0"), cfg, sd


def test_config0_tiny_10_steps_single_utterance_with_hift():
    """BASELINE configs[0]: seed-uvit-tat-xlsr-tiny, 10 diffusion steps, one 5 s source + 5 s reference, HiFT."""
    from seedvc_amd.vocoder import HiFT
    cfm, cfg, sd = _cfm("tiny")
    i = _inputs(cfg, 1, 100)
    mel = cfm.inference(i["mu"].cuda(), torch.LongTensor([T]), i["prompt"].cuda(), i["style"].cuda(), None, 10,
                        inference_cfg_rate=0.7, z=i["z"].cuda())
    ref = O.cfm_sample(sd, cfg, i["z"], T, i["prompt"], i["mu"], i["style"], 10, 0.7)
    l1 = (mel.cpu() - ref)[:, :, P:].abs().mean().item()
    print(f"config 0 sampler: mel L1 {l1:.3e} over {S} frames (|mel| mean {ref[:, :, P:].abs().mean():.3f})")
    assert l1 < 1e-3                                       # north-star tolerance
    assert mel[:, :, :P].abs().max().item() == 0.0         # prompt region is zeroed after every step (flow_matching.py:110)
    vc = cases.specs.hift_config()
    vsd = cases.weights.make_state_dict(cases.specs.hift_state_spec(vc), seed=1234, prefix="hift.")
    hop = cases.specs.hift_total_upsample(vc)
    nh = vc["nb_harmonics"] + 1
    phase0 = (cases.rand("bs.ph", 100, 1, nh, 1) * 2 - 1) * 3.14159265
    noise = cases.randn("bs.nz", 100, 1, nh, S * hop)
    voc = HiFT(vc, vsd, "cuda:0")
    tgt = ref[:, :, P:].contiguous()                       # vocoder parity on identical mel input
    f0 = O.hift_f0_predictor(vsd, tgt)
    wave, f0_hip = voc(tgt.cuda(), phase0=phase0.cuda(), noise=noise.cuda(), return_f0=True)
    assert ((f0_hip.cpu() - f0).abs() / f0.abs().clamp_min(1.0)).max().item() < 2e-5
    w_hip = voc(tgt.cuda(), f0=f0.cuda(), phase0=phase0.cuda(), noise=noise.cuda()).cpu().reshape(-1)
    w_ref = O.hift_forward(vsd, vc, tgt, phase0, noise, f0=f0).reshape(-1)
    rms = (w_hip - w_ref).pow(2).mean().sqrt().item()
    print(f"config 0 HiFT: {w_ref.numel()} samples, waveform RMS {rms:.3e} (signal rms {w_ref.pow(2).mean().sqrt():.3f})")
    assert w_hip.numel() == S * hop and rms < 1e-4         # north-star tolerance
    # the drop-in call (f0 predicted by the model itself): the sine source integrates f0 over the whole utterance, so the
    # 2e-5 relative f0 difference above turns into a phase drift of the harmonics -- measured and
    # recorded here, bounded by what it really meets
    w_own = wave.cpu().reshape(-1)
    w_ref_own = O.hift_forward(vsd, vc, tgt, phase0, noise).reshape(-1)
    rms_own = (w_own - w_ref_own).pow(2).mean().sqrt().item()
    print(f"config 0 HiFT with its own f0 at S = {S}: waveform RMS {rms_own:.3e} (f0-pinned {rms:.3e})")
    assert rms_own < HIFT_OWN_F0_RMS


def test_config1_batch64_properties():
    """BASELINE configs[1]: tiny, 25 steps, batch of 64 utterances -- size-independent properties + sampled oracle."""
    cfm, cfg, sd = _cfm("tiny")
    B = 64
    i = _inputs(cfg, B, 200)
    dev = {k: v.cuda() for k, v in i.items()}
    lens = torch.LongTensor([T] * B)
    mel = cfm.inference(dev["mu"], lens, dev["prompt"], dev["style"], None, 25, inference_cfg_rate=0.7, z=dev["z"])
    assert mel.shape == (B, cfg["C"], T) and torch.isfinite(mel).all()
    assert mel[:, :, :P].abs().max().item() == 0.0
    again = cfm.inference(dev["mu"], lens, dev["prompt"], dev["style"], None, 25, inference_cfg_rate=0.7, z=dev["z"])
    assert torch.equal(mel, again)                         # deterministic
    # cut differently (two calls of 32 = the fused kernel path as well): bit-identical
    halves = torch.cat([cfm.inference(dev["mu"][s:s + 32], lens[s:s + 32], dev["prompt"][s:s + 32], dev["style"][s:s + 32], None, 25,
                                      inference_cfg_rate=0.7, z=dev["z"][s:s + 32]) for s in (0, 32)])
    assert torch.equal(halves, mel)
    for b in (0, 31, 32, 63):                              # both micro-batches (32 + 32), first and last rows
        one = cfm.inference(dev["mu"][b:b + 1], torch.LongTensor([T]), dev["prompt"][b:b + 1], dev["style"][b:b + 1], None, 25,
                            inference_cfg_rate=0.7, z=dev["z"][b:b + 1])
        d = (one[0] - mel[b])[:, P:].abs().mean().item()   # B = 1 runs on the tap-GEMM path: same math, other rounding
        assert d < 1e-3, (b, d)
    cfm.estimator.set_fused_min_rows(1 << 40)              # on ONE path B independent B = 1 runs are bit-identical
    mel_u = cfm.inference(dev["mu"][:8], lens[:8], dev["prompt"][:8], dev["style"][:8], None, 25, inference_cfg_rate=0.7, z=dev["z"][:8])
    for b in (0, 7):
        one = cfm.inference(dev["mu"][b:b + 1], torch.LongTensor([T]), dev["prompt"][b:b + 1], dev["style"][b:b + 1], None, 25,
                            inference_cfg_rate=0.7, z=dev["z"][b:b + 1])
        assert torch.equal(one[0], mel_u[b]), b
    cfm.estimator.set_fused_min_rows(-1)
    b = 63
    ref = O.cfm_sample(sd, cfg, i["z"][b:b + 1], T, i["prompt"][b:b + 1], i["mu"][b:b + 1], i["style"][b:b + 1], 25, 0.7)
    l1 = (mel[b:b + 1].cpu() - ref)[:, :, P:].abs().mean().item()
    print(f"config 1 utterance {b}: mel L1 {l1:.3e}")
    assert l1 < 1e-3


def test_config2_small_wavenet_bigvgan_one_utterance():
    """One utterance of BASELINE configs[2]: seed-uvit-whisper-small-wavenet, 25 steps, + BigVGAN-22k."""
    from seedvc_amd.vocoder import BigVGAN
    cfm, cfg, sd = _cfm("small")
    i = _inputs(cfg, 1, 300)
    mel = cfm.inference(i["mu"].cuda(), torch.LongTensor([T]), i["prompt"].cuda(), i["style"].cuda(), None, 25,
                        inference_cfg_rate=0.7, z=i["z"].cuda())
    ref = O.cfm_sample(sd, cfg, i["z"], T, i["prompt"], i["mu"], i["style"], 25, 0.7)
    l1 = (mel.cpu() - ref)[:, :, P:].abs().mean().item()
    print(f"config 2 sampler: mel L1 {l1:.3e}")
    assert l1 < 1e-3
    h = cases.specs.bigvgan_config("22k")
    vsd = cases.weights.make_state_dict(cases.specs.bigvgan_state_spec(h), seed=1234, prefix="bigvgan.")
    tgt = ref[:, :, P:].contiguous()
    w_hip = BigVGAN(h, vsd, "cuda:0")(tgt.cuda()).cpu().reshape(-1)
    w_ref = O.bigvgan_forward(vsd, h, tgt).reshape(-1)
    rms = (w_hip - w_ref).pow(2).mean().sqrt().item()
    print(f"config 2 BigVGAN: {w_ref.numel()} samples, waveform RMS {rms:.3e} (signal rms {w_ref.pow(2).mean().sqrt():.3f})")
    assert w_hip.numel() == S * 256 and rms < 1e-4


def test_stress_30s_context_window():
    """The reference's maximum context (inference.py:370: 30 s = 2580 frames, P = 430): T' = 2580 rows of attention."""
    cfm, cfg, sd = _cfm("small")
    Tl = 2580
    mu = cases.randn("st.mu", 400, 1, Tl, cfg["Dc"])
    prompt = cases.logmel("st.p", 400, 1, cfg["C"], P)
    style = cases.randn("st.s", 400, 1, cfg["style_dim"])
    z = cases.randn("st.z", 400, 1, cfg["C"], Tl)
    mel = cfm.inference(mu.cuda(), torch.LongTensor([Tl]), prompt.cuda(), style.cuda(), None, 2, inference_cfg_rate=0.7, z=z.cuda())
    ref = O.cfm_sample(sd, cfg, z, Tl, prompt, mu, style, 2, 0.7)
    l1 = (mel.cpu() - ref)[:, :, P:].abs().mean().item()
    print(f"30 s window, 2 steps: mel L1 {l1:.3e}")
    assert l1 < 1e-3


def test_config3_base_44k_50_steps_one_utterance():
    """BASELINE configs[3]: seed-uvit-whisper-base (D768 L17, 128 mel bands, tap-GEMM path), 50 steps, T = 860, and the
    6-stage BigVGAN "44k" architecture on S = 430 frames (its hyper-parameters are parity-unpinned: that config is absent
    from the reference tree; the kernels are those of the pinned 22 kHz cases)."""
    from seedvc_amd.vocoder import BigVGAN
    cfm, cfg, sd = _cfm("base")
    i = _inputs(cfg, 1, 500)
    mel = cfm.inference(i["mu"].cuda(), torch.LongTensor([T]), i["prompt"].cuda(), i["style"].cuda(), None, 50,
                        inference_cfg_rate=0.7, z=i["z"].cuda())
    ref = O.cfm_sample(sd, cfg, i["z"], T, i["prompt"], i["mu"], i["style"], 50, 0.7)
    l1 = (mel.cpu() - ref)[:, :, P:].abs().mean().item()
    print(f"config 3 sampler (base, 50 steps): mel L1 {l1:.3e} (|mel| mean {ref[:, :, P:].abs().mean():.3f})")
    assert l1 < 1e-3
    assert mel[:, :, :P].abs().max().item() == 0.0
    h = cases.specs.bigvgan_config("44k")
    vsd = cases.weights.make_state_dict(cases.specs.bigvgan_state_spec(h), seed=1234, prefix="bigvgan.")
    hop = cases.specs.bigvgan_total_upsample(h)
    tgt = ref[:, :, P:].contiguous()
    w_hip = BigVGAN(h, vsd, "cuda:0")(tgt.cuda()).cpu().reshape(-1)
    w_ref = O.bigvgan_forward(vsd, h, tgt).reshape(-1)
    rms = (w_hip - w_ref).pow(2).mean().sqrt().item()
    print(f"config 3 BigVGAN-44k: {w_ref.numel()} samples (x{hop}), waveform RMS {rms:.3e} (signal rms {w_ref.pow(2).mean().sqrt():.3f})")
    assert w_hip.numel() == S * hop and rms < 1e-4


def test_config4_v2_three_way_cfg_25_steps():
    """BASELINE configs[4], CFM half: v2 DiT (AdaLN-zero, time + style tokens), cfg [0.7, 0.7] = three estimator streams,
    25 steps of the cosine-warped grid at T = 862 rows, vs the oracle; on both kernel paths."""
    cfm, cfg, sd = _cfm("v2")
    i = _inputs(cfg, 1, 600)
    ref = O.cfm_sample(sd, cfg, i["z"], T, i["prompt"], i["mu"], i["style"], 25, [0.7, 0.7])
    for rows, tag in ((1 << 40, "tap-GEMM path"), (0, "fused row-panel path")):
        cfm.estimator.set_fused_min_rows(rows)
        mel = cfm.inference(i["mu"].cuda(), torch.LongTensor([T]), i["prompt"].cuda(), i["style"].cuda(), None, 25,
                            inference_cfg_rate=[0.7, 0.7], z=i["z"].cuda())
        l1 = (mel.cpu() - ref)[:, :, P:].abs().mean().item()
        print(f"config 4 sampler (v2, 3-way CFG, 25 steps), {tag}: mel L1 {l1:.3e}")
        assert l1 < 1e-3
        assert mel[:, :, :P].abs().max().item() == 0.0
