"""GPU parity of the length regulator through the C ABI: `svc_lr_forward` vs the oracle and the
committed outputs of the reference's `InterpolateRegulator` (v1 continuous +- f0, v2 discrete, v2 embedding-only)."""
import pytest
import torch

import cases
import seedvc_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _model(c, sd):
    from seedvc_amd.length_regulator import InterpolateRegulator
    return InterpolateRegulator(c, sd, "cuda:0")


@pytest.mark.parametrize("name", list(cases.LR_CASES))
def test_lr_matches_reference_outputs(name, golden):
    c, sd, x, ylen, f0, meta = cases.lr_case(name)
    m = _model(c, sd)
    res = m(x.cuda(), ylens=torch.LongTensor([ylen]), n_quantizers=3, f0=None if f0 is None else f0.cuda())
    assert len(res) == (5 if c["version"] == 1 else 2)
    y = res[0].cpu()
    assert res[1].tolist() == [ylen if c["n_convs"] else min(ylen, meta["tin"])]
    ref_o = O.lr_forward(sd, c, x, ylen, f0)
    assert y.shape == ref_o.shape
    err_o = (y - ref_o).abs().max().item()
    g = torch.from_numpy(golden[name + ".out"])
    err_g = ((y[:, ::4] if name.endswith("_full") else y) - g).abs().max().item()
    print(f"{name}: max|hip - oracle| {err_o:.2e}  max|hip - reference| {err_g:.2e}")
    assert err_o < 5e-5 and err_g < 5e-5          # fp32 MFMA contractions, fp32 elementwise; |out| ~ 0.5


def test_lr_batch_is_independent_utterances():
    """Ragged batch: own input length, output length and f0 length per utterance; one empty utterance."""
    c, sd, _, _, _, _ = cases.lr_case("lr_base_r")
    m = _model(c, sd)
    tins, ylens, tf0s = [31, 12, 20, 9], [37, 40, 11, 0], [45, 30, 45, 8]
    B, Tin, Tf = len(tins), max(tins), max(tf0s)
    x = cases.randn("lrb.x", 5, B, Tin, c["in_channels"])
    f0 = 80.0 + 700.0 * cases.rand("lrb.f0", 5, B, Tf)
    out, olens, *_ = m(x.cuda(), ylens=torch.LongTensor(ylens), f0=f0.cuda(), in_lens=tins, f0_lens=tf0s)
    out = out.cpu()
    assert out.shape == (B, max(ylens), c["out_channels"]) and olens.tolist() == ylens
    for b in range(B):
        if ylens[b] == 0:
            assert out[b].abs().max().item() == 0.0
            continue
        ref = O.lr_forward(sd, c, x[b:b + 1, :tins[b]], ylens[b], f0[b:b + 1, :tf0s[b]])
        assert (out[b, :ylens[b]] - ref[0]).abs().max().item() < 5e-5
        assert out[b, ylens[b]:].abs().max().item() == 0.0 if ylens[b] < max(ylens) else True


def test_lr_feeds_the_sampler():
    """content -> length regulator -> `mu` -> CFM sampler, all on the device, against the oracle chain."""
    from seedvc_amd.cfm import CFM
    cfg, dsd, inp, meta = cases.dit_case("tiny_r")
    lc = cases.specs.lr_config("tiny", channels=cfg["Dc"], in_channels=64)
    lsd = cases.weights.make_state_dict(cases.specs.lr_state_spec(lc), seed=91, prefix="lr.")
    T = inp["mu"].shape[1]
    P = inp["prompt"].shape[-1]
    content = cases.randn("lrs.c", 9, 1, 15, 64)
    mu = _model(lc, lsd)(content.cuda(), ylens=torch.LongTensor([T]))[0]
    mel = CFM(cfg, dsd, "cuda:0").inference(mu, torch.LongTensor([T]), inp["prompt"].cuda(), inp["style"].cuda(), None, 3,
                                            inference_cfg_rate=0.7, z=inp["z"].cuda()).cpu()
    mu_o = O.lr_forward(lsd, lc, content, T)
    ref = O.cfm_sample(dsd, cfg, inp["z"], T, inp["prompt"], mu_o, inp["style"], 3, 0.7)
    assert (mel - ref)[:, :, P:].abs().mean().item() < 1e-3
