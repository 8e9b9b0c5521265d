"""Parity at BASELINE.json's own sizes (P = S = 430 frames, T = 860) through the C ABI.

configs[0] (tiny, 10 steps, one 5 s source + reference) is checked end to end against the CPU oracle; one utterance of
configs[2] (small+WaveNet + BigVGAN), configs[3] (base 44.1 kHz, 50 steps + the 6-stage BigVGAN-44k architecture) and
configs[4] (v2, 3-way CFG, 25 steps) against OUTPUTS OF THE REFERENCE ITSELF at those sizes (tests/golden/fullsize.npz,
made by make_golden.py fullsize and stored decimated: the CPU oracle needs 1.5 - 5 minutes per case, too long for a
test); configs[1] (B = 64) and the 30 s context window are checked through
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


def _fs_sampler(golden, name, fused_min_rows=None):
    """HIP sampler on a full-size case vs the reference's stored output frames -> (mel L1, mean |mel|)."""
    from seedvc_amd.cfm import CFM
    cfg, sd, i, meta = cases.fullsize_cfm_case(name)
    cfm = CFM(cfg, sd, "cuda:0")
    if fused_min_rows is not None:
        cfm.estimator.set_fused_min_rows(fused_min_rows)
    mel = cfm.inference(i["mu"].cuda(), torch.LongTensor([T]), i["prompt"].cuda(), i["style"].cuda(), None, meta["n_steps"],
                        inference_cfg_rate=meta["cfg_rate"], z=i["z"].cuda()).cpu()
    ref = torch.from_numpy(golden[name + ".mel"])
    got = mel[0, :, P::cases.FS_MEL_STEP]
    assert got.shape == ref.shape
    assert mel[:, :, :P].abs().max().item() == 0.0         # prompt region is zeroed after every step
    return (got - ref).abs().mean().item(), ref.abs().mean().item()


def _fs_vocoder(golden, name, precision="fp16x3"):
    from seedvc_amd.vocoder import BigVGAN
    h, vsd, mel = cases.fullsize_voc_case(name)
    w = BigVGAN(h, vsd, "cuda:0", precision=precision)(mel.cuda()).cpu().reshape(-1)
    n = int(golden[name + ".n"])
    assert w.numel() == n == S * cases.specs.bigvgan_total_upsample(h)
    ref = torch.from_numpy(golden[name + ".wave"])
    got = torch.stack([w[o:o + cases.FS_WAVE_WIN] for o in cases.fs_wave_windows(n)])
    return (got - ref).pow(2).mean().sqrt().item(), ref.pow(2).mean().sqrt().item(), n


def test_config2_small_wavenet_bigvgan_one_utterance(golden):
    """One utterance of BASELINE configs[2]: seed-uvit-whisper-small-wavenet, 25 steps, + BigVGAN-22k, vs the reference."""
    l1, mag = _fs_sampler(golden, "fs_small")
    print(f"config 2 sampler vs reference: mel L1 {l1:.3e} (|mel| mean {mag:.3f})")
    assert l1 < 1e-3
    rms, sig, n = _fs_vocoder(golden, "fs_bigvgan22k")
    print(f"config 2 BigVGAN vs reference: {n} samples, waveform RMS {rms:.3e} (signal rms {sig:.3f})")
    assert rms < 1e-4


def test_config2_batch64_fused_path_vs_reference(golden):
    """BASELINE configs[2]'s per-GPU workload ON THE KERNEL PATH bench.py TIMES: small+WaveNet, B = 64, T = 860, 25 steps
    -> 55 k rows per launch = `dit_panel_kernel<512,false,1>` (modulated norms, UViT skips, WaveNet window), micro-batches
    of 32.  The reference's full-size case (`fs_small`) sits at utterances 17 and 63 of a seeded batch -- one per
    micro-batch group -- and both must reproduce the reference's stored mel; then BigVGAN over a batch that holds the
    reference's `fs_bigvgan22k` mel at rows 5 and 21 (one per vocoder micro-batch: set to 16 here, default 32)."""
    from seedvc_amd.vocoder import BigVGAN
    cfm, cfg, sd = _cfm("small")
    B = 64
    i = _inputs(cfg, B, 700)
    _, _, fs, meta = cases.fullsize_cfm_case("fs_small")
    assert meta["n_steps"] == 25 and meta["cfg_rate"] == 0.7
    for b in (17, 63):
        for k in i:
            i[k][b] = fs[k][0]
    dev = {k: v.cuda() for k, v in i.items()}
    lens = torch.LongTensor([T] * B)
    assert 2 * 32 * T >= 10240                             # a micro-batch of 32 x 2 CFG streams is a fused-path launch
    mel = cfm.inference(dev["mu"], lens, dev["prompt"], dev["style"], None, 25, inference_cfg_rate=0.7, z=dev["z"])
    assert mel.shape == (B, cfg["C"], T) and torch.isfinite(mel).all()
    assert mel[:, :, :P].abs().max().item() == 0.0         # prompt frames are zeroed after every step
    ref = torch.from_numpy(golden["fs_small.mel"])
    for b in (17, 63):
        got = mel[b, :, P::cases.FS_MEL_STEP].cpu()
        l1 = (got - ref).abs().mean().item()
        print(f"config 2, B = 64 fused path, utterance {b}: mel L1 vs reference {l1:.3e}")
        assert l1 < 1e-3, (b, l1)
    assert torch.equal(mel[17], mel[63])                   # same inputs in both micro-batch groups: same bits
    halves = torch.cat([cfm.inference(dev["mu"][s:s + 32], lens[s:s + 32], dev["prompt"][s:s + 32], dev["style"][s:s + 32], None, 25,
                                      inference_cfg_rate=0.7, z=dev["z"][s:s + 32]) for s in (0, 32)])
    assert torch.equal(halves, mel)
    # the other kernel path on the same utterance (B = 1: tap-GEMMs) stays within the north-star tolerance of this one
    one = cfm.inference(dev["mu"][17:18], torch.LongTensor([T]), dev["prompt"][17:18], dev["style"][17:18], None, 25,
                        inference_cfg_rate=0.7, z=dev["z"][17:18])
    assert (one[0] - mel[17])[:, P:].abs().mean().item() < 1e-3
    # vocoder on a batch (micro-batches of 16)
    h, vsd, m1 = cases.fullsize_voc_case("fs_bigvgan22k")
    Bv = 32
    mels = cases.logmel("bs.vmel", 701, Bv, h["num_mels"], S)
    mels[5] = m1[0]
    mels[21] = m1[0]
    voc = BigVGAN(h, vsd, "cuda:0")
    voc.set_microbatch(16)
    wave = voc(mels.cuda()).cpu().reshape(Bv, -1)
    n = int(golden["fs_bigvgan22k.n"])
    assert wave.shape[1] == n
    refw = torch.from_numpy(golden["fs_bigvgan22k.wave"])
    for b in (5, 21):
        got = torch.stack([wave[b, o:o + cases.FS_WAVE_WIN] for o in cases.fs_wave_windows(n)])
        rms = (got - refw).pow(2).mean().sqrt().item()
        print(f"config 2, BigVGAN batch of {Bv}, utterance {b}: waveform RMS vs reference {rms:.3e}")
        assert rms < 1e-4, (b, rms)
    assert torch.equal(wave[5], wave[21])


@pytest.mark.parametrize("name", ["fs_bigvgan22k", "fs_bigvgan44k"])
def test_bigvgan_fp16p8_mode_at_full_size_vs_reference(golden, name):
    """The "fp16p8" vocoder mode (fp16 hi*hi product + both split-precision correction products in one block-scaled fp8 MFMA
    on the long stride-1 convs) at S = 430 against the reference's own waveform: inside the north-star bound (1e-4) with
    margin, an order of magnitude above fp16x3 (1.7e-6 / 1.4e-6)."""
    rms, sig, n = _fs_vocoder(golden, name, "fp16p8")
    print(f"{name} [fp16p8] vs reference: {n} samples, waveform RMS {rms:.3e} (signal rms {sig:.3f})")
    assert rms < 5e-5


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


def test_config3_base_44k_50_steps_one_utterance(golden):
    """BASELINE configs[3]: seed-uvit-whisper-base (D768 L17, 128 mel bands, tap-GEMM path), 50 steps, T = 860, and the
    6-stage BigVGAN "44k" architecture on S = 430 frames, vs the reference's outputs (the 44k hyper-parameters are
    parity-unpinned: that config is absent from the reference tree; the reference CLASS ran them for the fixture)."""
    l1, mag = _fs_sampler(golden, "fs_base")
    print(f"config 3 sampler (base, 50 steps) vs reference: mel L1 {l1:.3e} (|mel| mean {mag:.3f})")
    assert l1 < 1e-3
    rms, sig, n = _fs_vocoder(golden, "fs_bigvgan44k")
    print(f"config 3 BigVGAN-44k vs reference: {n} samples, waveform RMS {rms:.3e} (signal rms {sig:.3f})")
    assert rms < 1e-4


def test_config4_v2_three_way_cfg_25_steps(golden):
    """BASELINE configs[4], CFM half: v2 DiT (AdaLN-zero, time + style tokens), cfg [0.7, 0.7] = three estimator streams,
    25 steps of the cosine-warped grid at T = 862 rows, vs the reference; on both kernel paths."""
    for rows, tag in ((1 << 40, "tap-GEMM path"), (0, "fused row-panel path")):
        l1, mag = _fs_sampler(golden, "fs_v2", fused_min_rows=rows)
        print(f"config 4 sampler (v2, 3-way CFG, 25 steps) vs reference, {tag}: mel L1 {l1:.3e}")
        assert l1 < 1e-3
