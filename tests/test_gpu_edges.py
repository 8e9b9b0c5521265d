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
