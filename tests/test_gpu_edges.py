"""Edge cases of the hot path through the C ABI: no prompt, minimal lengths, single-frame vocoder input, error paths.
(Empty and ragged batches: test_gpu_dit.py::test_batched_ragged..., test_gpu_lr.py; maximum context: test_gpu_baseline_sizes.py.)"""
import pytest
import torch

import cases
import seedvc_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _cfm(name="tiny_r"):
    from seedvc_amd.cfm import CFM
    cfg, sd, inp, meta = cases.dit_case(name)
    return CFM(cfg, sd, "cuda:0"), cfg, sd


@pytest.mark.parametrize("name", ["tiny_r", "small_r"])
@pytest.mark.parametrize("T,P", [(24, 0), (2, 1), (1, 0), (9, 9)])
def test_sampler_degenerate_lengths(name, T, P):
    """P = 0 (no prompt), a two-frame utterance, a single frame, and a prompt that covers the whole sequence."""
    cfm, cfg, sd = _cfm(name)
    mu = cases.randn("ed.mu", T + P, 1, T, cfg["Dc"])
    prompt = cases.logmel("ed.p", T + P, 1, cfg["C"], P) if P else torch.zeros(1, cfg["C"], 0)
    style = cases.randn("ed.s", T + P, 1, cfg["style_dim"])
    z = cases.randn("ed.z", T + P, 1, cfg["C"], T)
    mel = cfm.inference(mu.cuda(), torch.LongTensor([T]), prompt.cuda(), style.cuda(), None, 3, inference_cfg_rate=0.7, z=z.cuda()).cpu()
    ref = O.cfm_sample(sd, cfg, z, T, prompt, mu, style, 3, 0.7)
    assert mel.shape == ref.shape == (1, cfg["C"], T)
    assert (mel - ref).abs().mean().item() < 1e-3
    assert mel[:, :, :P].abs().max().item() == 0.0 if P else True


def test_sampler_rejects_bad_lengths():
    cfm, cfg, sd = _cfm()
    mu = torch.zeros(1, 8, cfg["Dc"]).cuda()
    prompt = torch.zeros(1, cfg["C"], 4).cuda()
    style = torch.zeros(1, cfg["style_dim"]).cuda()
    with pytest.raises(RuntimeError):
        cfm.inference(mu, torch.LongTensor([9]), prompt, style, None, 2)                  # x_lens > T
    with pytest.raises(RuntimeError):
        cfm.inference(mu, torch.LongTensor([8]), torch.zeros(1, cfg["C"], 12).cuda(), style, None, 2)   # P > T
    with pytest.raises(RuntimeError):
        cfm.inference(mu, torch.LongTensor([8]), prompt, style, None, 0)                  # no steps


@pytest.mark.parametrize("S", [1, 2])
def test_vocoders_on_very_short_mels(S):
    from seedvc_amd.vocoder import BigVGAN, HiFT
    h, vsd, mel, meta = cases.bigvgan_case("bigvgan_r")
    m = mel[:1, :, :S].contiguous()
    y = BigVGAN(h, vsd, "cuda:0")(m.cuda()).cpu()
    ref = O.bigvgan_forward(vsd, h, m)
    assert y.shape == ref.shape and (y - ref).pow(2).mean().sqrt().item() < 1e-4
    c, hsd, hmel, phase0, noise, hmeta = cases.hift_case("hift_r")
    hop = cases.specs.hift_total_upsample(c)
    m = hmel[:1, :, :S].contiguous()
    f0 = O.hift_f0_predictor(hsd, m)
    y = HiFT(c, hsd, "cuda:0")(m.cuda(), f0=f0.cuda(), phase0=phase0[:1].cuda(), noise=noise[:1, :, :S * hop].cuda()).cpu().reshape(-1)
    ref = O.hift_forward(hsd, c, m, phase0[:1], noise[:1, :, :S * hop], f0=f0).reshape(-1)
    assert y.shape == ref.shape and (y - ref).pow(2).mean().sqrt().item() < 1e-4


def test_length_regulator_extreme_lengths():
    from seedvc_amd.length_regulator import InterpolateRegulator
    c, sd, _, _, _, _ = cases.lr_case("lr_base_r")
    m = InterpolateRegulator(c, sd, "cuda:0")
    for tin, ylen, tf0 in ((1, 50, 1), (40, 1, 3), (1, 1, 0)):
        x = cases.randn("ed.lx", tin + ylen, 1, tin, c["in_channels"])
        f0 = torch.zeros(1, tf0) if tf0 else None            # all-unvoiced f0 / the f0_mask branch
        out = m(x.cuda(), ylens=torch.LongTensor([ylen]), f0=None if f0 is None else f0.cuda())[0].cpu()
        ref = O.lr_forward(sd, c, x, ylen, f0)
        assert out.shape == ref.shape and (out - ref).abs().max().item() < 5e-5, (tin, ylen, tf0)


def test_mel_shortest_clip():
    from seedvc_amd.audio import MelSpectrogram
    c, y, basis = cases.mel_case("mel_22k")
    pad = (c["n_fft"] - c["hop"]) // 2
    yy = y[:, :pad + 1 + c["hop"]]                            # reflect padding needs L > pad; -> a single frame or two
    fe = MelSpectrogram(c["n_fft"], c["n_mels"], c["sr"], c["hop"], c["n_fft"], mel_basis=basis)
    m = fe(yy.cuda()).cpu()
    ref = O.mel_spectrogram(yy, basis, c["n_fft"], c["hop"], c["n_fft"])
    assert m.shape == ref.shape and (m - ref).abs().mean().item() < 1e-4
    with pytest.raises(RuntimeError):
        fe(y[:, :pad].cuda())                                 # shorter than the padding: the reference's F.pad raises too


def test_ar_single_token_prefill_and_one_token_generate(golden):
    from seedvc_amd.ar import ARModel
    c, sd, text, target, exp_noise = cases.ar_gen_case("ar_gen_r2")
    m = ARModel(c, sd, "cuda:0")
    toks = m.generate(text.cuda(), target.cuda(), exp_noise=exp_noise.cuda(), max_new=1).cpu()
    ref = torch.from_numpy(golden["ar_gen_r2.codes"])
    assert toks.shape == (1, 1) and int(toks[0, 0]) == int(ref[0, 0])
    x = cases.randn("ed.ar", 3, 1, 1, c["dim"])
    m.setup_caches()
    lg = m.forward_generate(x.cuda(), torch.tensor([0]), torch.tensor([0])).cpu()
    caches = O.ar_new_cache(c)
    ref_lg = O.ar_forward_generate(sd, c, x, torch.tensor([0]), torch.tensor([0]), caches)
    assert (lg.reshape(-1) - ref_lg.reshape(-1)).abs().max().item() < 5e-3


def test_device_chunk_loop_shorter_than_overlap():
    from seedvc_amd.cfm import CFM
    from seedvc_amd.vocoder import BigVGAN
    from seedvc_amd.pipeline import HotPath
    cfg, sd, inp, meta = cases.dit_case("tiny_r")
    h, vsd, mel, vmeta = cases.bigvgan_case("bigvgan_r2")
    hp = HotPath(CFM(cfg, sd, "cuda:0"), BigVGAN(h, vsd, "cuda:0"))
    P = 16
    pc = cases.randn("ed.pc", 3, 1, P, cfg["Dc"]).cuda()
    mel2 = cases.logmel("ed.mel2", 3, 1, cfg["C"], P).cuda()
    style = cases.randn("ed.style", 3, 1, cfg["style_dim"]).cuda()
    noise = lambda T: cases.randn(f"ed.z{T}", 3, 1, cfg["C"], T).cuda()      # noqa: E731
    for S_total in (3, 26):                                  # shorter than the 4-frame overlap; last chunk of 2 frames
        cond = cases.randn(f"ed.cond{S_total}", 3, 1, S_total, cfg["Dc"]).cuda()
        a = hp.convert_long(cond, pc, mel2, style, 2, 0.7, 8, 40, overlap_frame_len=4, noise_fn=noise)
        b = hp.convert_long_device(cond, pc, mel2, style, 2, 0.7, 8, 40, overlap_frame_len=4, noise_fn=noise)
        assert a.shape == b.shape and torch.equal(a.cpu(), b.cpu()), S_total
