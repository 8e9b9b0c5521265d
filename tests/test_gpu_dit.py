"""GPU parity of the DiT estimator and the CFM sampler (HIP path through the C ABI) against the CPU
oracle and against the committed reference outputs.  Tolerance: north_star's mel L1 < 1e-3."""
import pytest
import torch

import cases
import seedvc_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

MEL_L1 = 1e-3

_cache = {}


def _model(name):
    from seedvc_amd.cfm import CFM
    if name not in _cache:
        cfg, sd, inp, meta = cases.dit_case(name)
        _cache[name] = (CFM(cfg, sd, "cuda:0"), cfg, sd, inp, meta)
    return _cache[name]


@pytest.mark.parametrize("name", list(cases.DIT_CASES))
def test_estimator_vs_reference_golden(name, golden):
    cfm, cfg, sd, inp, meta = _model(name)
    T, P = meta["T"], meta["P"]
    prompt_x = torch.zeros(1, cfg["C"], T)
    prompt_x[..., :P] = inp["prompt"]
    y = cfm.estimator(inp["x"].cuda(), prompt_x.cuda(), torch.LongTensor([T]), inp["t"], inp["style"].cuda(),
                      inp["mu"].cuda()).cpu()
    ref = torch.from_numpy(golden[name + ".est"])
    l1 = (y - ref).abs().mean().item()
    print(f"{name}: estimator L1 {l1:.3e} max {(y - ref).abs().max().item():.3e}")
    assert l1 < MEL_L1, f"{name}: estimator L1 {l1:.3e}"
    assert (y - ref).abs().max().item() < 3e-2


@pytest.mark.parametrize("name", list(cases.DIT_CASES))
def test_sampler_vs_reference_golden(name, golden):
    cfm, cfg, sd, inp, meta = _model(name)
    y = cfm.inference(inp["mu"].cuda(), torch.LongTensor([meta["T"]]), inp["prompt"].cuda(), inp["style"].cuda(), None,
                      meta["n_steps"], inference_cfg_rate=meta["cfg_rate"], z=inp["z"].cuda(),
                      random_voice=meta["random_voice"]).cpu()
    ref = torch.from_numpy(golden[name + ".sample"])
    l1 = (y - ref).abs().mean().item()
    print(f"{name}: sampler mel L1 {l1:.3e}")
    assert l1 < MEL_L1, f"{name}: sampler mel L1 {l1:.3e}"
    assert float(y[..., :meta["P"]].abs().max()) == 0.0


def test_batched_ragged_equals_independent_runs():
    """B > 1 is defined as B independent B=1 reference runs (each with its own length / prompt length)."""
    cfm, cfg, sd, inp, meta = _model("small_r")
    T, P = meta["T"], meta["P"]
    lens, plens = [T, T - 9, T - 17], [P, P - 4, 3]
    B = len(lens)
    mu = torch.cat([cases.randn(f"rag.mu{b}", 1, 1, T, cfg["Dc"]) for b in range(B)])
    z = torch.cat([cases.randn(f"rag.z{b}", 1, 1, cfg["C"], T) for b in range(B)])
    prompt = torch.cat([cases.logmel(f"rag.p{b}", 1, 1, cfg["C"], P) for b in range(B)])
    style = torch.cat([cases.randn(f"rag.s{b}", 1, 1, cfg["style_dim"]) for b in range(B)])
    y = cfm.inference(mu.cuda(), torch.LongTensor(lens), prompt.cuda(), style.cuda(), None, 3, inference_cfg_rate=0.7,
                      z=z.cuda(), prompt_lens=plens).cpu()
    for b in range(B):
        Tb, Pb = lens[b], plens[b]
        ref = O.cfm_sample(sd, cfg, z[b:b + 1, :, :Tb], Tb, prompt[b:b + 1, :, :Pb], mu[b:b + 1, :Tb], style[b:b + 1], 3, 0.7)
        l1 = (y[b:b + 1, :, :Tb] - ref).abs().mean().item()
        assert l1 < MEL_L1, f"utterance {b}: {l1:.3e}"


def test_microbatch_invariance():
    cfm, cfg, sd, inp, meta = _model("tiny_r")
    T, P = meta["T"], meta["P"]
    B = 5
    mu = cases.randn("mb.mu", 2, B, T, cfg["Dc"])
    z = cases.randn("mb.z", 2, B, cfg["C"], T)
    prompt = cases.logmel("mb.p", 2, B, cfg["C"], P)
    style = cases.randn("mb.s", 2, B, cfg["style_dim"])
    outs = []
    for mb in (16, 2):
        cfm.estimator.set_microbatch(mb)
        outs.append(cfm.inference(mu.cuda(), torch.LongTensor([T] * B), prompt.cuda(), style.cuda(), None, 2,
                                  inference_cfg_rate=0.7, z=z.cuda()).cpu())
    cfm.estimator.set_microbatch(0)
    assert torch.equal(outs[0], outs[1])


def test_temperature_and_errors():
    cfm, cfg, sd, inp, meta = _model("tiny_r")
    with pytest.raises(RuntimeError):
        cfm.inference(inp["mu"].cuda(), torch.LongTensor([meta["T"] + 5]), inp["prompt"].cuda(), inp["style"].cuda(), None, 2,
                      z=inp["z"].cuda())
    y = cfm.inference(inp["mu"].cuda(), torch.LongTensor([meta["T"]]), inp["prompt"].cuda(), inp["style"].cuda(), None, 2,
                      temperature=0.5, inference_cfg_rate=0.7, z=inp["z"].cuda()).cpu()
    ref = O.cfm_sample(sd, cfg, inp["z"], meta["T"], inp["prompt"], inp["mu"], inp["style"], 2, 0.7, temperature=0.5)
    assert (y - ref).abs().mean().item() < MEL_L1


@pytest.mark.parametrize("name", ["tiny_full", "small_r", "v2_r"])
def test_euler_loop_graph_replay_is_bit_identical(name, golden):
    """The sampler's Euler loop is captured into a hipGraph on the second call with the same shape and replayed from the
    third: eager (1st), capture + first replay (2nd) and replay (3rd) must agree bit for bit, also after a call with
    another shape in between, and with graphs switched off."""
    cfm, cfg, sd, inp, meta = _model(name)
    eager = cfm.inference(inp["mu"].cuda(), torch.LongTensor([meta["T"]]), inp["prompt"].cuda(), inp["style"].cuda(), None,
                          meta["n_steps"], inference_cfg_rate=meta["cfg_rate"], z=inp["z"].cuda()).cpu()
    cfm.estimator.set_graphs(True)

    def run(T=None):
        T = T or meta["T"]
        return cfm.inference(inp["mu"][:, :T].cuda(), torch.LongTensor([T]), inp["prompt"].cuda(), inp["style"].cuda(), None,
                             meta["n_steps"], inference_cfg_rate=meta["cfg_rate"], z=inp["z"][:, :, :T].cuda()).cpu()

    a, b, c = run(), run(), run()
    assert torch.equal(a, b) and torch.equal(a, c)
    other = run(meta["T"] - 8)                       # another key; the first graph stays cached
    assert other.shape[-1] == meta["T"] - 8
    d = run()
    assert torch.equal(a, d)
    cfm.estimator.set_graphs(False)
    e = run()
    assert torch.equal(a, e) and torch.equal(a, eager)
    # several micro-batch groups inside one call share a key: the second group captures, the third replays
    cfm.estimator.set_graphs(True)
    cfm.estimator.set_microbatch(1)
    B = 3
    rep = lambda t: t.repeat(B, *([1] * (t.dim() - 1)))      # noqa: E731
    many = cfm.inference(rep(inp["mu"]).cuda(), torch.LongTensor([meta["T"]] * B), rep(inp["prompt"]).cuda(), rep(inp["style"]).cuda(),
                         None, meta["n_steps"], inference_cfg_rate=meta["cfg_rate"], z=rep(inp["z"]).cuda()).cpu()
    for b in range(B):
        assert torch.equal(many[b:b + 1], a)
    ref = torch.from_numpy(golden[name + ".sample"])
    assert (a - ref).abs().mean().item() < 1e-3


def test_sampler_on_checkpoint_loaded_by_the_reference(golden):
    """BASELINE configs[0]'s plumbing: `ckpt.tiny.sample` is `model.cfm.inference` (10 steps, cfg 0.7) of a model the
    REFERENCE built with `build_model` and filled with `load_checkpoint` from a synthetic .pth (make_golden.gen_ckpt, which
    also asserts that the loaded `cfm.estimator.state_dict()` equals these generated tensors key for key): the HIP sampler
    packed from the same state dict must reproduce it within the north-star tolerance."""
    from seedvc_amd.cfm import CFM
    cfg, sd, lc, lsd, inp = cases.ckpt_case()
    cfm = CFM(cfg, sd, "cuda:0")
    out = cfm.inference(inp["mu"].cuda(), torch.LongTensor([cases.CKPT_T]), inp["prompt"].cuda(), inp["style"].cuda(), None,
                        cases.CKPT_STEPS, inference_cfg_rate=0.7, z=inp["z"].cuda()).cpu()
    ref = torch.from_numpy(golden["ckpt.tiny.sample"])
    l1 = (out - ref)[:, :, cases.CKPT_P:].abs().mean().item()
    print(f"sampler on the reference-loaded checkpoint: mel L1 {l1:.3e}")
    assert l1 < 1e-3 and out[:, :, :cases.CKPT_P].abs().max().item() == 0.0
