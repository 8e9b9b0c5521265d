"""GPU parity of the CAMPPlus style encoder (8f row 3, second half) against the reference's outputs, and of the Kaldi fbank
front-end against the CPU restatement (torchaudio is absent from the build image: fbank parity is unpinned)."""
import numpy as np
import pytest
import torch

import cases
import seedvc_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.mark.parametrize("name", list(cases.CAMPPLUS_CASES))
def test_campplus_vs_reference_golden(name, golden):
    from seedvc_amd.campplus import CAMPPlus
    c, sd, feat = cases.campplus_case(name)
    m = CAMPPlus(c, sd, "cuda:0")
    e = m(feat.cuda()).cpu()
    ref = torch.from_numpy(golden[name + ".emb"])
    err = (e - ref).abs().max().item()
    print(f"{name}: embedding max |diff| {err:.3e} (|emb| mean {ref.abs().mean():.3f})")
    assert e.shape == ref.shape and err < 2e-5          # fp32 MFMA throughout (measured 3e-7, ~70 layers deep)
    again = m(feat.cuda()).cpu()
    assert torch.equal(e, again)
    if feat.shape[0] > 1:                                # clips of a batch are independent
        one = m(feat[1:2].cuda()).cpu()
        assert (one - e[1:2]).abs().max().item() < 1e-5


def test_kaldi_fbank_vs_restatement():
    from seedvc_amd.campplus import CAMPPlus
    c, sd, _ = cases.campplus_case("campplus_r")
    m = CAMPPlus(c, sd, "cuda:0")
    sr = 16000
    t = torch.arange(int(2.3 * sr), dtype=torch.float32) / sr
    y = 0.3 * torch.sin(2 * np.pi * 220.0 * t) + 0.2 * torch.sin(2 * np.pi * 1370.0 * t) + 0.02 * cases.randn("fb.n", 1, t.numel())
    got = m.fbank(y.cuda()[None]).cpu()
    ref = O.kaldi_fbank(y[None])
    assert got.shape == ref.shape == (1 + (t.numel() - 400) // 160, 80)
    live = ref > -14.0                                   # bins well above the log floor
    err = (got - ref)[live].abs().max().item()
    print(f"kaldi fbank: {got.shape[0]} frames, max |diff| over live bins {err:.3e}; floor bins equal: {(got - ref)[~live].abs().max().item():.3e}")
    assert err < 2e-3
    assert (got - ref)[~live].abs().max().item() < 0.5
    style = m.style(y.cuda()[None])
    feat = ref - ref.mean(dim=0, keepdim=True)
    want = O.campplus_forward(sd, c, feat[None])
    assert (style.cpu() - want).abs().max().item() < 5e-3
