"""GPU parity of the caller-side harness (row a21): chunked long-utterance conversion and batched conversion
through the HIP sampler + vocoder against the oracle's restatement of inference.py:470-527."""
import pytest
import torch

import cases
import seedvc_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _models():
    from seedvc_amd.cfm import CFM
    from seedvc_amd.vocoder import BigVGAN
    cfg, sd, inp, meta = cases.dit_case("tiny_r")
    h, vsd, mel, vmeta = cases.bigvgan_case("bigvgan_r2")
    return (CFM(cfg, sd, "cuda:0"), cfg, sd), (BigVGAN(h, vsd, "cuda:0"), h, vsd)


def test_chunked_long_utterance_matches_oracle():
    from seedvc_amd.pipeline import HotPath
    (cfm, cfg, sd), (voc, h, vsd) = _models()
    hop = 8                       # bigvgan_r2: upsample rates [4, 2]
    P, S_total, window = 16, 70, 40
    cond = cases.randn("long.cond", 3, 1, S_total, cfg["Dc"])
    pc = cases.randn("long.pc", 3, 1, P, cfg["Dc"])
    mel2 = cases.logmel("long.mel2", 3, 1, cfg["C"], P)
    style = cases.randn("long.style", 3, 1, cfg["style_dim"])
    noise = lambda T: cases.randn(f"long.z{T}", 3, 1, cfg["C"], T)      # noqa: E731
    hp = HotPath(cfm, voc)
    out = hp.convert_long(cond.cuda(), pc.cuda(), mel2.cuda(), style.cuda(), 3, 0.7, hop, window,
                          overlap_frame_len=4, noise_fn=lambda T: noise(T).cuda())
    ref = O.chunked_convert(
        lambda cc: O.cfm_sample(sd, cfg, noise(cc.size(1)), cc.size(1), mel2, cc, style, 3, 0.7),
        lambda m: O.bigvgan_forward(vsd, h, m).reshape(1, -1), cond, pc, mel2, style, hop, window, overlap_frame_len=4)
    assert out.shape == ref.shape
    rms = (out - ref).pow(2).mean().sqrt().item()
    print(f"chunked conversion: {out.shape[-1]} samples, waveform RMS vs oracle {rms:.3e}")
    assert rms < 5e-3       # mel error (<1e-3 L1) propagated through the vocoder


def test_convert_batch_matches_per_utterance_oracle():
    from seedvc_amd.pipeline import HotPath
    (cfm, cfg, sd), (voc, h, vsd) = _models()
    B, T, P = 3, 40, 16
    mu = cases.randn("cb.mu", 4, B, T, cfg["Dc"])
    prompt = cases.logmel("cb.p", 4, B, cfg["C"], P)
    style = cases.randn("cb.s", 4, B, cfg["style_dim"])
    z = cases.randn("cb.z", 4, B, cfg["C"], T)
    mel, wave = HotPath(cfm, voc).convert_batch(mu.cuda(), prompt.cuda(), style.cuda(), 3, 0.7, z=z.cuda())
    for b in range(B):
        m = O.cfm_sample(sd, cfg, z[b:b + 1], T, prompt[b:b + 1], mu[b:b + 1], style[b:b + 1], 3, 0.7)[:, :, P:]
        assert (mel[b:b + 1].cpu() - m).abs().mean().item() < 1e-3
        w = O.bigvgan_forward(vsd, h, mel[b:b + 1].cpu()).reshape(-1)      # vocoder parity on identical mel input
        assert (wave[b].cpu() - w).pow(2).mean().sqrt().item() < 1e-4


def test_lanes_equal_single_handle():
    """Splitting a batch over independent handle pairs / HIP streams changes nothing: utterances are independent and
    every kernel's per-row reduction order does not depend on which other utterances share the launch."""
    from seedvc_amd.cfm import CFM
    from seedvc_amd.vocoder import BigVGAN
    from seedvc_amd.pipeline import HotPath, Lanes
    (cfm, cfg, sd), (voc, h, vsd) = _models()
    B, T, P = 5, 40, 16
    mu = cases.randn("ln.mu", 5, B, T, cfg["Dc"]).cuda()
    prompt = cases.logmel("ln.p", 5, B, cfg["C"], P).cuda()
    style = cases.randn("ln.s", 5, B, cfg["style_dim"]).cuda()
    z = cases.randn("ln.z", 5, B, cfg["C"], T).cuda()
    mel1, wave1 = HotPath(cfm, voc).convert_batch(mu, prompt, style, 3, 0.7, z=z)
    lanes = Lanes(lambda: (CFM(cfg, sd, "cuda:0"), BigVGAN(h, vsd, "cuda:0")), 2, "cuda:0")
    mel2, wave2 = lanes.convert_batch(mu, prompt, style, 3, 0.7, z=z)
    torch.cuda.synchronize()
    assert mel2.shape == mel1.shape and wave2.shape == wave1.shape
    assert torch.equal(mel1, mel2)
    assert torch.equal(wave1, wave2)


def test_device_chunk_loop_equals_host_chunk_loop():
    """8f row 2: crossfade + concatenation on the device, vocoder overlapped on a second stream -- bit-identical to the
    host loop (same sampler / vocoder outputs, the float64 crossfade arithmetic reproduced exactly)."""
    from seedvc_amd.pipeline import HotPath
    (cfm, cfg, sd), (voc, h, vsd) = _models()
    hop = 8
    P, window = 16, 40
    for S_total in (70, 24, 45):                 # several chunks / single chunk / last chunk shorter than the overlap
        cond = cases.randn(f"dl.cond{S_total}", 3, 1, S_total, cfg["Dc"]).cuda()
        pc = cases.randn("dl.pc", 3, 1, P, cfg["Dc"]).cuda()
        mel2 = cases.logmel("dl.mel2", 3, 1, cfg["C"], P).cuda()
        style = cases.randn("dl.style", 3, 1, cfg["style_dim"]).cuda()
        noise = lambda T: cases.randn(f"dl.z{T}", 3, 1, cfg["C"], T).cuda()      # noqa: E731
        hp = HotPath(cfm, voc)
        a = hp.convert_long(cond, pc, mel2, style, 3, 0.7, hop, window, overlap_frame_len=4, noise_fn=noise)
        b = hp.convert_long_device(cond, pc, mel2, style, 3, 0.7, hop, window, overlap_frame_len=4, noise_fn=noise)
        assert a.shape == b.shape, (S_total, a.shape, b.shape)
        assert torch.equal(a.cpu(), b.cpu()), S_total


class _CaseCFM:
    """cases.fake_sampler behind the HIP sampler's call signature"""
    device = torch.device("cuda:0")

    def inference(self, mu, x_lens, prompt, style, f0, n, inference_cfg_rate=0.7, z=None, **kw):
        return cases.fake_sampler(mu, prompt.size(-1))


@pytest.mark.parametrize("name", ["loop1", "loop1s", "loop2", "loop2b", "loop4", "loop5"])
def test_device_chunk_loop_equals_the_references_own_loop(name, golden):
    """f2 against the reference's loop (tests/golden/chunkloop.npz: the while-loop of SeedVCWrapper.convert_voice +
    _stream_wave_chunks + crossfade, seed_vc_wrapper.py:190-285,560-623, run by make_golden.py with the fake sampler /
    vocoder of cases.py): `convert_long_device` -- `svc_crossfade` on the device, one output buffer, vocoder on a second
    stream -- must give the same samples bit for bit (the fakes are exactly-rounded fp32 multiply / add only)."""
    from seedvc_amd.pipeline import HotPath
    c = {k: v.cuda() for k, v in cases.chunkloop_case(name).items()}
    hp = HotPath(_CaseCFM(), cases_fake_vocoder_cuda)
    out = hp.convert_long_device(c["cond"], c["prompt_condition"], c["mel2"], c["style2"], 10, 0.7, cases.CHUNK_HOP,
                                 cases.CHUNK_WINDOW, overlap_frame_len=cases.CHUNK_OVERLAP)
    want = torch.from_numpy(golden[f"chunkloop.{name}.out"].astype("float32"))
    assert out.shape == (1, want.numel())
    assert torch.equal(out[0].cpu(), want)
    host = hp.convert_long(c["cond"], c["prompt_condition"], c["mel2"], c["style2"], 10, 0.7, cases.CHUNK_HOP,
                           cases.CHUNK_WINDOW, overlap_frame_len=cases.CHUNK_OVERLAP)
    assert torch.equal(host[0], want)


def cases_fake_vocoder_cuda(mel):
    j = (torch.arange(cases.CHUNK_HOP, dtype=torch.float32, device=mel.device) + 1.0) * 0.125
    return (mel[0, 0][:, None] * j[None, :] + mel[0, 1][:, None]).reshape(1, 1, -1).contiguous()
