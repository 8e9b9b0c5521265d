"""The fused row-panel kernel (csrc/fused.hip: wo + residual, ffn-norm, SwiGLU MLP, UViT skip linear, next layer's
attention-norm + QKV + RoPE in one launch) against the reference outputs and against the tap-GEMM path.

The kernel is selected by launch size (>= 10240 token rows); here it is forced on (`set_fused_min_rows(0)`) so the
committed reference goldens of the full-width architectures exercise it at fixture size."""
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _run(cfm, inp, meta, **kw):
    return cfm.inference(inp["mu"].cuda(), torch.LongTensor([meta["T"]]), inp["prompt"].cuda(), inp["style"].cuda(), None,
                         meta["n_steps"], inference_cfg_rate=meta["cfg_rate"], z=inp["z"].cuda(), **kw).cpu()


@pytest.mark.parametrize("name", ["tiny_full", "small_full", "v2_full"])
def test_fused_sampler_vs_reference_golden(name, golden):
    """D = 384 (tiny: prefix tokens, UViT skips, no modulation), D = 512 (small: modulated norms, skips, WaveNet head),
    v2 (AdaLN-zero gates, 3-way CFG) -- sampler output vs the REFERENCE's, north-star tolerance."""
    from seedvc_amd.cfm import CFM
    cfg, sd, inp, meta = cases.dit_case(name)
    cfm = CFM(cfg, sd, "cuda:0")
    assert cfm.estimator.fused_available
    cfm.estimator.set_fused_min_rows(0)
    out = _run(cfm, inp, meta)
    ref = torch.from_numpy(golden[name + ".sample"])
    l1 = (out - ref).abs().mean().item()
    cfm.estimator.set_fused_min_rows(1 << 40)
    out_u = _run(cfm, inp, meta)
    d = (out - out_u).abs().mean().item()
    print(f"{name}: fused vs reference L1 {l1:.3e}; fused vs tap-GEMM path L1 {d:.3e} (|mel| mean {ref.abs().mean():.3f})")
    assert l1 < 1e-3
    assert d < 1e-3          # two fp16-operand paths: each is ~1e-4..4e-4 from the fp32 reference


@pytest.mark.parametrize("name", ["tiny_full", "small_full", "v2_full"])
def test_fused_estimator_vs_reference_golden(name, golden):
    from seedvc_amd.cfm import CFM
    cfg, sd, inp, meta = cases.dit_case(name)
    cfm = CFM(cfg, sd, "cuda:0")
    cfm.estimator.set_fused_min_rows(0)
    T, P = meta["T"], meta["P"]
    prompt_x = torch.zeros(1, cfg["C"], T)
    prompt_x[..., :P] = inp["prompt"]
    est = cfm.estimator(inp["x"].cuda(), prompt_x.cuda(), torch.LongTensor([T]), inp["t"], inp["style"].cuda(), inp["mu"].cuda()).cpu()
    ref = torch.from_numpy(golden[name + ".est"])
    err = (est - ref).abs().mean().item() / ref.abs().mean().item()
    print(f"{name}: fused estimator rel L1 {err:.3e}")
    assert err < 5e-3


def test_fused_ragged_batch_and_window():
    """Ragged batch through the fused path: every utterance equals its own run on the same path bit for bit (rows are
    independent, panels straddle sequences), short-prompt window of the last layer included."""
    from seedvc_amd.cfm import CFM
    cfg, sd, inp, meta = cases.dit_case("tiny_full")
    cfm = CFM(cfg, sd, "cuda:0")
    cfm.estimator.set_fused_min_rows(0)
    B, T, P = 5, 200, 90
    mu = cases.randn("fz.mu", 7, B, T, cfg["Dc"]).cuda()
    prompt = cases.logmel("fz.p", 7, B, cfg["C"], P).cuda()
    style = cases.randn("fz.s", 7, B, cfg["style_dim"]).cuda()
    z = cases.randn("fz.z", 7, B, cfg["C"], T).cuda()
    lens = [200, 173, 200, 96, 131]
    plens = [90, 40, 64, 33, 90]
    out = cfm.inference(mu, torch.LongTensor(lens), prompt, style, None, 3, inference_cfg_rate=0.7, z=z, prompt_lens=plens)
    for b in range(B):
        one = cfm.inference(mu[b:b + 1, :lens[b]], torch.LongTensor([lens[b]]), prompt[b:b + 1, :, :plens[b]], style[b:b + 1], None, 3,
                            inference_cfg_rate=0.7, z=z[b:b + 1, :, :lens[b]])
        assert torch.isfinite(one).all()
        d = (one[0] - out[b, :, :lens[b]]).abs().max().item()
        assert d < 2e-4, (b, d)          # windows differ between the batch and the single run (shortest prompt): rounding only
    cfm.estimator.set_fused_min_rows(1 << 40)
    out_u = cfm.inference(mu, torch.LongTensor(lens), prompt, style, None, 3, inference_cfg_rate=0.7, z=z, prompt_lens=plens)
    for b in range(B):
        d = (out[b, :, :lens[b]] - out_u[b, :, :lens[b]]).abs().mean().item()
        assert d < 2e-4, (b, d)


@pytest.mark.parametrize("name", ["tiny_full", "small_full"])
def test_attention32_form_in_the_model(name, golden, monkeypatch):
    """SVC_ATTN32=1: the 32x32x16 attention kernel with its own V^T column order, which the QKV epilogues of BOTH paths (tap-GEMM
    and fused row-panel kernel) must then write (vt_pos mode 2) -- sampler vs the reference on both paths."""
    from seedvc_amd.cfm import CFM
    monkeypatch.setenv("SVC_ATTN32", "1")
    cfg, sd, inp, meta = cases.dit_case(name)
    cfm = CFM(cfg, sd, "cuda:0")
    ref = torch.from_numpy(golden[name + ".sample"])
    for rows in (0, 1 << 40):
        cfm.estimator.set_fused_min_rows(rows)
        l1 = (_run(cfm, inp, meta) - ref).abs().mean().item()
        print(f"{name}, attn32, fused_min_rows {rows}: L1 vs reference {l1:.3e}")
        assert l1 < 1e-3
