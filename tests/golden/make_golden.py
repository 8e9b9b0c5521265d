"""Generates tests/golden/*.npz by running the REAL reference modules (imported from /root/reference in
the build container; three absent third-party packages are stubbed in tests/golden/_stubs) on the
seeded inputs and generated weights of `cases.py`.  The reference never travels: only its outputs are
committed.  Also asserts that `seedvc_amd.specs.*_state_spec` matches the reference modules key-for-key.

Run:  python tests/golden/make_golden.py            (needs /root/reference; not run on the GPU box)
"""
import ast
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SEEDVC_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(HERE, "_stubs"))
sys.path.insert(1, REF)

import numpy as np
import torch

import _pkgload
_pkgload.load_package()
from seedvc_amd import specs
import cases

torch.set_grad_enabled(False)
torch.set_num_threads(8)


def check_spec(spec, module, what):
    ref = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    mine = {k: tuple(v) for k, v in spec.items()}
    assert set(ref) == set(mine), (what, sorted(set(ref) ^ set(mine))[:20])
    for k in ref:
        assert ref[k] == mine[k], (what, k, ref[k], mine[k])


def load_sd(module, sd):
    cur = module.state_dict()
    cast = {k: v.to(cur[k].dtype) for k, v in sd.items()}
    module.load_state_dict(cast, strict=True)
    module.eval()


# ------------------------------------------------------------------------------------------ DiT
def build_ref_cfm(cfg):
    from munch import Munch
    if cfg["version"] == 2:
        from modules.v2.dit_wrapper import DiT
        from modules.v2.cfm import CFM
        est = DiT(time_as_token=cfg["time_as_token"], style_as_token=cfg["style_as_token"],
                  uvit_skip_connection=cfg["uvit"], block_size=cfg["block_size"], depth=cfg["L"],
                  num_heads=cfg["H"], hidden_dim=cfg["D"], in_channels=cfg["C"], content_dim=cfg["Dc"],
                  style_encoder_dim=cfg["style_dim"], class_dropout_prob=0.1, dropout_rate=0.0,
                  attn_dropout_rate=0.0)
        return CFM(est)
    from modules.flow_matching import CFM
    dit = Munch(hidden_dim=cfg["D"], num_heads=cfg["H"], depth=cfg["L"], class_dropout_prob=0.1,
                block_size=8192, in_channels=cfg["C"], style_condition=cfg["style_condition"],
                final_layer_type=cfg["head"], target="mel", content_dim=cfg["Dc"],
                content_codebook_size=cfg["codebook"], content_type="discrete", f0_condition=False,
                n_f0_bins=512, content_codebooks=1, is_causal=False, long_skip_connection=cfg["long_skip"],
                zero_prompt_speech_token=False, time_as_token=cfg["time_as_token"],
                style_as_token=cfg["style_as_token"], uvit_skip_connection=cfg["uvit"],
                add_resblock_in_transformer=False)
    args = Munch(dit_type="DiT", reg_loss_type="l1", DiT=dit, style_encoder=Munch(dim=cfg["style_dim"]))
    if cfg["head"] == "wavenet":
        args.wavenet = Munch(hidden_dim=cfg["wn_dim"], num_layers=cfg["wn_layers"], kernel_size=cfg["wn_kernel"],
                             dilation_rate=cfg["wn_dilation"], p_dropout=0.2, style_condition=cfg["style_condition"])
    cfm = CFM(args)
    cfm.estimator.setup_caches(max_batch_size=1, max_seq_length=8192)
    return cfm


def gen_dit(out):
    for name in cases.DIT_CASES:
        cfg, sd, inp, meta = cases.dit_case(name)
        cfm = build_ref_cfm(cfg)
        check_spec(specs.dit_state_spec(cfg), cfm.estimator, name)
        load_sd(cfm.estimator, sd)
        T, P = meta["T"], meta["P"]
        lens = torch.LongTensor([T])
        # (1) one estimator evaluation on the conditional inputs
        prompt_x = torch.zeros(1, cfg["C"], T)
        prompt_x[..., :P] = inp["prompt"]
        if cfg["version"] == 2:
            est = cfm.estimator(inp["x"], prompt_x, lens, inp["t"], inp["style"], inp["mu"])
        else:
            est = cfm.estimator(inp["x"], prompt_x, lens, inp["t"], inp["style"], inp["mu"], False)
        # (2) full sampler with explicit noise: patch torch.randn to return our z
        real_randn = torch.randn
        torch.randn = lambda *a, **k: inp["z"].clone()
        try:
            if cfg["version"] == 2:
                smp = cfm.inference(inp["mu"], lens, inp["prompt"], inp["style"], meta["n_steps"],
                                    inference_cfg_rate=meta["cfg_rate"], random_voice=meta["random_voice"])
            else:
                smp = cfm.inference(inp["mu"], lens, inp["prompt"], inp["style"], None, meta["n_steps"],
                                    inference_cfg_rate=meta["cfg_rate"])
        finally:
            torch.randn = real_randn
        out[name + ".est"] = est.numpy()
        out[name + ".sample"] = smp.numpy()
        print(f"{name}: est |mean| {est.abs().mean():.4f}  sample |mean| {smp.abs().mean():.4f}", flush=True)


# ------------------------------------------------------------------------------------------ BigVGAN
def gen_bigvgan(out):
    from modules.bigvgan import bigvgan
    from modules.bigvgan.env import AttrDict
    for name in cases.BIGVGAN_CASES:
        h, sd, mel, meta = cases.bigvgan_case(name)
        model = bigvgan.BigVGAN(AttrDict(dict(h)), use_cuda_kernel=False)
        check_spec(specs.bigvgan_state_spec(h, weight_norm_removed=False), model, name + "(wn)")
        model.remove_weight_norm()
        check_spec(specs.bigvgan_state_spec(h), model, name)
        load_sd(model, sd)
        y = model(mel)
        out[name + ".wave"] = y.numpy()
        print(f"{name}: wave {tuple(y.shape)} rms {y.pow(2).mean().sqrt():.4f} clip {(y.abs() >= 1).float().mean():.3f}", flush=True)


def gen_act(out):
    from modules.bigvgan.alias_free_activation.torch.act import Activation1d
    from modules.bigvgan.activations import SnakeBeta, Snake
    for name in cases.ACT_CASES:
        x, alpha, beta = cases.act_case(name)
        act = SnakeBeta(x.shape[1], alpha_logscale=True)
        act.alpha.data.copy_(alpha)
        act.beta.data.copy_(beta)
        m = Activation1d(activation=act)
        out[name + ".snakebeta"] = m(x).numpy()
        act2 = Snake(x.shape[1], alpha_logscale=True)
        act2.alpha.data.copy_(alpha)
        out[name + ".snake"] = Activation1d(activation=act2)(x).numpy()
        out[name + ".filter"] = m.upsample.filter.reshape(-1).numpy()


# ------------------------------------------------------------------------------------------ HiFT
def gen_hift(out):
    from modules.hifigan.generator import HiFTGenerator
    from modules.hifigan.f0_predictor import ConvRNNF0Predictor
    import modules.hifigan.generator as G
    for name in cases.HIFT_CASES:
        c, sd, mel, phase0, noise, meta = cases.hift_case(name)
        f0p = ConvRNNF0Predictor(num_class=1, in_channels=c["in_channels"], cond_channels=c["f0_cond_channels"])
        model = HiFTGenerator(in_channels=c["in_channels"], base_channels=c["base_channels"],
                              nb_harmonics=c["nb_harmonics"], sampling_rate=c["sampling_rate"],
                              nsf_alpha=c["nsf_alpha"], nsf_sigma=c["nsf_sigma"],
                              nsf_voiced_threshold=c["nsf_voiced_threshold"],
                              upsample_rates=c["upsample_rates"], upsample_kernel_sizes=c["upsample_kernel_sizes"],
                              istft_params={"n_fft": c["istft_n_fft"], "hop_len": c["istft_hop"]},
                              resblock_kernel_sizes=c["resblock_kernel_sizes"],
                              resblock_dilation_sizes=c["resblock_dilation_sizes"],
                              source_resblock_kernel_sizes=c["source_resblock_kernel_sizes"],
                              source_resblock_dilation_sizes=c["source_resblock_dilation_sizes"],
                              lrelu_slope=c["lrelu_slope"], audio_limit=c["audio_limit"], f0_predictor=f0p)
        check_spec(specs.hift_state_spec(c), model, name)
        load_sd(model, sd)

        # route the reference's random draws to our explicit tensors
        class FixedUniform:
            def __init__(self, low, high):
                pass

            def sample(self, sample_shape):
                assert tuple(sample_shape) == tuple(phase0.shape), (sample_shape, phase0.shape)
                return phase0.clone()

        calls = []

        def fixed_randn_like(t):
            calls.append(tuple(t.shape))
            if len(calls) == 1:
                assert tuple(t.shape) == tuple(noise.shape)
                return noise.clone()
            return torch.zeros_like(t)      # randn_like(uv): result is discarded by _f02source

        real_u, real_r = G.Uniform, torch.randn_like
        G.Uniform, torch.randn_like = FixedUniform, fixed_randn_like
        try:
            f0 = model.f0_predictor(mel)
            y = model(mel)
            calls.clear()
            f0_fixed = torch.full_like(f0, 0.0) + cases.rand(name + ".f0", meta["seed"], *f0.shape) * 300.0
            y2 = model(mel, f0=f0_fixed)
        finally:
            G.Uniform, torch.randn_like = real_u, real_r
        out[name + ".f0"] = f0.numpy()
        out[name + ".wave"] = y.numpy()
        out[name + ".f0_fixed"] = f0_fixed.numpy()
        out[name + ".wave_f0fixed"] = y2.numpy()
        print(f"{name}: f0 mean {f0.mean():.2f} voiced {(f0 > 10).float().mean():.2f} wave rms {y.pow(2).mean().sqrt():.4f}", flush=True)


# ------------------------------------------------------------------------------------------ v2 AR
def gen_ar(out):
    from modules.v2.ar import NaiveWrapper, NaiveTransformer, NaiveModelArgs, logits_to_probs, multinomial_sample_one_no_sync
    for name in cases.AR_CASES:
        c, sd, x_prefill, input_pos, x_steps, exp_noise, meta = cases.ar_case(name)
        args = NaiveModelArgs(dropout=0.0, rope_base=c["rope_base"], dim=c["dim"], head_dim=c["head_dim"],
                              n_local_heads=c["n_local_heads"], intermediate_size=c["intermediate_size"],
                              n_head=c["n_head"], n_layer=c["n_layer"], vocab_size=c["vocab_size"], max_seq_len=c["max_seq_len"])
        wrap = NaiveWrapper(NaiveTransformer(args))
        check_spec(specs.ar_state_spec(c), wrap, name)
        load_sd(wrap, sd)
        wrap.setup_caches(1, c["max_seq_len"], dtype=torch.float32, device=torch.device("cpu"))
        n_prefill = meta["n_prefill"]
        ip = torch.tensor(input_pos)
        kv = torch.arange(n_prefill)
        logits = [wrap.model.forward_generate(x_prefill, ip, kv).logits.clone()]
        prev = []
        probs_all, idx_all = [], []
        for s in range(meta["n_decode"]):
            ip = ip[-1:] + 1
            kv = kv[-1:] + 1
            lg = wrap.model.forward_generate(x_steps[s], ip, kv).logits.clone()
            logits.append(lg)
            pt = torch.tensor(prev, dtype=torch.long) if prev else None
            pr = logits_to_probs(lg[0, -1].clone(), previous_tokens=pt, suppress_tokens=[c["vocab_size"] - 1],
                                 temperature=0.7, top_p=0.7, repetition_penalty=1.5)
            probs_all.append(pr)
            # the reference divides by q = Exp(1) noise drawn in place; replay it with our explicit noise
            idx = torch.argmax(pr / exp_noise[s], dim=-1, keepdim=True).to(torch.int)
            torch.manual_seed(1000 + s)
            ref_idx = multinomial_sample_one_no_sync(pr)
            torch.manual_seed(1000 + s)
            q = torch.empty_like(pr).exponential_(1)
            assert int(torch.argmax(pr / q)) == int(ref_idx), "exponential-race replay mismatch"
            idx_all.append(idx)
            prev.append(int(idx))
        out[name + ".logits"] = torch.cat(logits, dim=0).numpy()
        out[name + ".probs"] = torch.stack(probs_all).numpy()
        out[name + ".idx"] = torch.cat(idx_all).numpy()
        print(f"{name}: logits |mean| {torch.cat(logits).abs().mean():.4f} sampled {prev}", flush=True)


def gen_argen(out):
    """NaiveWrapper.generate with its Exp(1) draws replaced by the case's noise rows (the reference draws them inside
    multinomial_sample_one_no_sync; that function is swapped for one that reads our rows in call order)."""
    import modules.v2.ar as ar_mod
    for name in cases.AR_GEN_CASES:
        c, sd, text, target, exp_noise = cases.ar_gen_case(name)
        args = ar_mod.NaiveModelArgs(dropout=0.0, rope_base=c["rope_base"], dim=c["dim"], head_dim=c["head_dim"],
                                     n_local_heads=c["n_local_heads"], intermediate_size=c["intermediate_size"],
                                     n_head=c["n_head"], n_layer=c["n_layer"], vocab_size=c["vocab_size"],
                                     max_seq_len=c["max_seq_len"])
        wrap = ar_mod.NaiveWrapper(ar_mod.NaiveTransformer(args))
        load_sd(wrap, sd)
        wrap.setup_caches(1, c["max_seq_len"], dtype=torch.float32, device=torch.device("cpu"))
        counter = [0]
        orig = ar_mod.multinomial_sample_one_no_sync

        def replay(probs_sort):
            q = exp_noise[counter[0]]
            counter[0] += 1
            return torch.argmax(probs_sort / q, dim=-1, keepdim=True).to(dtype=torch.int)

        ar_mod.multinomial_sample_one_no_sync = replay
        try:
            codes = wrap.generate(text, target.clone(), top_p=0.7, temperature=0.7, repetition_penalty=1.5)
        except IndexError:
            codes = None
        finally:
            ar_mod.multinomial_sample_one_no_sync = orig
        assert codes is not None, f"{name}: generation ran past max_seq_len without EOS; pick another seed"
        out[name + ".codes"] = codes.numpy().astype(np.int64)
        print(f"{name}: {codes.shape[-1]} tokens {codes.flatten().tolist()[:24]}", flush=True)


def gen_argenfull(out):
    """BASELINE configs[4] size (full ar_base, 120 condition frames + 200 prompt tokens): NaiveWrapper.generate cut after
    AR_GEN_FULL_TOKENS tokens (its `tqdm(range(4000))` is limited; a random-weight model never emits EOS).  Run twice:
    with the plain Exp(1) rows, then with every winner's draw divided by AR_GEN_FULL_BOOST (cases.ar_gen_full_case) --
    the tokens must not change; the committed tokens are what the GPU test must reproduce with the boosted rows."""
    import itertools
    import modules.v2.ar as ar_mod

    def run(exp_noise, c, sd, text, target):
        args = ar_mod.NaiveModelArgs(dropout=0.0, rope_base=c["rope_base"], dim=c["dim"], head_dim=c["head_dim"],
                                     n_local_heads=c["n_local_heads"], intermediate_size=c["intermediate_size"],
                                     n_head=c["n_head"], n_layer=c["n_layer"], vocab_size=c["vocab_size"],
                                     max_seq_len=c["max_seq_len"])
        wrap = ar_mod.NaiveWrapper(ar_mod.NaiveTransformer(args))
        load_sd(wrap, sd)
        wrap.setup_caches(1, c["max_seq_len"], dtype=torch.float32, device=torch.device("cpu"))
        counter = [0]
        orig, orig_tqdm = ar_mod.multinomial_sample_one_no_sync, ar_mod.tqdm

        def replay(probs_sort):
            q = exp_noise[counter[0]]
            counter[0] += 1
            return torch.argmax(probs_sort / q, dim=-1, keepdim=True).to(dtype=torch.int)

        ar_mod.multinomial_sample_one_no_sync = replay
        ar_mod.tqdm = lambda it: itertools.islice(it, cases.AR_GEN_FULL_TOKENS - 1)
        try:
            return wrap.generate(text, target.clone(), top_p=0.7, temperature=0.7, repetition_penalty=1.5)
        finally:
            ar_mod.multinomial_sample_one_no_sync, ar_mod.tqdm = orig, orig_tqdm

    c, sd, text, target, exp_noise = cases.ar_gen_full_case()
    codes = run(exp_noise, c, sd, text, target)
    assert codes.shape[-1] == cases.AR_GEN_FULL_TOKENS, codes.shape
    c, sd, text, target, boosted = cases.ar_gen_full_case(winners=codes)
    codes2 = run(boosted, c, sd, text, target)
    assert torch.equal(codes, codes2), "boosting the winners changed the trajectory"
    out["ar_gen_full.codes"] = codes.numpy().astype(np.int64)
    print(f"ar_gen_full: {codes.shape[-1]} tokens, {len(set(codes.flatten().tolist()))} distinct, first {codes.flatten().tolist()[:16]}", flush=True)


def gen_fullsize(out):
    """BASELINE.json sizes (P = S = 430): the reference sampler for small / base (50 steps) / v2 (3-way CFG) and the
    reference BigVGAN in its 22 kHz and "44k" architectures, outputs stored decimated (cases.FS_*).  Minutes of CPU."""
    from modules.bigvgan import bigvgan
    from modules.bigvgan.env import AttrDict
    for name in cases.FULLSIZE_CFM:
        cfg, sd, inp, meta = cases.fullsize_cfm_case(name)
        cfm = build_ref_cfm(cfg)
        load_sd(cfm.estimator, sd)
        lens = torch.LongTensor([meta["T"]])
        real_randn = torch.randn
        torch.randn = lambda *a, **k: inp["z"].clone()
        try:
            if cfg["version"] == 2:
                smp = cfm.inference(inp["mu"], lens, inp["prompt"], inp["style"], meta["n_steps"], inference_cfg_rate=meta["cfg_rate"])
            else:
                smp = cfm.inference(inp["mu"], lens, inp["prompt"], inp["style"], None, meta["n_steps"],
                                    inference_cfg_rate=meta["cfg_rate"])
        finally:
            torch.randn = real_randn
        assert smp[:, :, :meta["P"]].abs().max().item() == 0.0
        out[name + ".mel"] = smp[0, :, meta["P"]::cases.FS_MEL_STEP].contiguous().numpy()
        print(f"{name}: generated mel |mean| {smp[:, :, meta['P']:].abs().mean():.4f}", flush=True)
    for name in cases.FULLSIZE_VOC:
        h, sd, mel = cases.fullsize_voc_case(name)
        model = bigvgan.BigVGAN(AttrDict(dict(h)), use_cuda_kernel=False)
        model.remove_weight_norm()
        load_sd(model, sd)
        y = model(mel).reshape(-1)
        out[name + ".wave"] = torch.stack([y[o:o + cases.FS_WAVE_WIN] for o in cases.fs_wave_windows(y.numel())]).numpy()
        out[name + ".n"] = np.int64(y.numel())
        print(f"{name}: {y.numel()} samples rms {y.pow(2).mean().sqrt():.4f}", flush=True)


def gen_campplus(out):
    from modules.campplus.DTDNN import CAMPPlus
    for name in cases.CAMPPLUS_CASES:
        c, sd, feat = cases.campplus_case(name)
        m = CAMPPlus(feat_dim=c["feat_dim"], embedding_size=c["embedding_size"])
        if tuple(c["block_layers"]) != (12, 24, 16):
            # reduced variant: rebuild the xvector stack with fewer dense layers exactly as CAMPPlus.__init__ does
            from collections import OrderedDict
            from modules.campplus.layers import TDNNLayer, CAMDenseTDNNBlock, TransitLayer, DenseLayer, get_nonlinear
            ch = m.head.out_channels
            xv = torch.nn.Sequential(OrderedDict([("tdnn", TDNNLayer(ch, c["init_channels"], 5, stride=2, dilation=1, padding=-1))]))
            ch = c["init_channels"]
            for i, (nl, k, d) in enumerate(zip(c["block_layers"], c["block_kernel"], c["block_dilation"])):
                xv.add_module("block%d" % (i + 1), CAMDenseTDNNBlock(num_layers=nl, in_channels=ch, out_channels=c["growth_rate"],
                                                                      bn_channels=c["bn_size"] * c["growth_rate"], kernel_size=k,
                                                                      dilation=d, memory_efficient=True))
                ch = ch + nl * c["growth_rate"]
                xv.add_module("transit%d" % (i + 1), TransitLayer(ch, ch // 2, bias=False))
                ch //= 2
            xv.add_module("out_nonlinear", get_nonlinear("batchnorm-relu", ch))
            m.xvector = xv
            m.dense = DenseLayer(ch * 2, c["embedding_size"], config_str="batchnorm_")
        check_spec(specs.campplus_state_spec(c), m, name)
        torch.nn.Module.load_state_dict(m, {k: v.to(m.state_dict()[k].dtype) for k, v in sd.items()}, strict=True)
        m.eval()
        e = m(feat)
        out[name + ".emb"] = e.numpy()
        print(f"{name}: emb {tuple(e.shape)} |mean| {e.abs().mean():.4f} max {e.abs().max():.3f}", flush=True)


def gen_mel(out):
    """modules.audio.mel_spectrogram with its module-level caches pre-filled: librosa (the only source of the mel
    filterbank) is absent from this image, so the reference function runs its own padding / STFT / log path on the
    filterbank restated in seedvc_amd.audio.slaney_mel_basis (parity of that matrix itself stays unpinned)."""
    import modules.audio as A
    for name in cases.MEL_CASES:
        c, y, basis = cases.mel_case(name)
        A.mel_basis[f"{c['sr']}_{c['fmax']}_{y.device}"] = basis
        A.hann_window[f"{c['sr']}_{y.device}"] = torch.hann_window(c["n_fft"])
        m = A.mel_spectrogram(y, c["n_fft"], c["n_mels"], c["sr"], c["hop"], c["n_fft"], c["fmin"], c["fmax"], center=False)
        out[name + ".mel"] = m.numpy()
        print(f"{name}: mel {tuple(m.shape)} range [{m.min():.2f}, {m.max():.2f}]", flush=True)


# ------------------------------------------------------------------------------------------ length regulator
def gen_lr(out):
    for name in cases.LR_CASES:
        c, sd, x, ylen, f0, meta = cases.lr_case(name)
        kw = dict(channels=c["channels"], sampling_ratios=[1] * c["n_convs"], is_discrete=c["is_discrete"],
                  in_channels=c["in_channels"] or None, codebook_size=c["codebook_size"], out_channels=c["out_channels"],
                  f0_condition=c["f0_condition"], n_f0_bins=c["n_f0_bins"])
        if c["version"] == 1:
            from modules.length_regulator import InterpolateRegulator
            m = InterpolateRegulator(vector_quantize=False, n_codebooks=1, quantizer_dropout=0.0, **kw)
        else:
            from modules.v2.length_regulator import InterpolateRegulator
            m = InterpolateRegulator(**kw)
        check_spec(specs.lr_state_spec(c), m, name)
        load_sd(m, sd)
        ylens = torch.LongTensor([ylen])
        if c["version"] == 1:
            y = m(x, ylens=ylens, n_quantizers=3, f0=f0)[0]
        else:
            y = m(x, ylens=ylens, f0=f0)[0]
        out[name + ".out"] = (y[:, ::4] if name.endswith("_full") else y).numpy()      # full-size: every 4th frame
        print(f"{name}: out {tuple(y.shape)} |mean| {y.abs().mean():.4f}", flush=True)


# ------------------------------------------------------------------------------------------ harness
def gen_crossfade(out):
    """`crossfade` lives in inference.py, whose module-level imports (librosa, torchaudio) are absent here;
    the function itself only needs numpy, so it is extracted by name and executed on its own."""
    src = open(os.path.join(REF, "inference.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "crossfade"][0]
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "inference.py", "exec"), ns)
    rng = np.random.default_rng(7)
    for tag, n1, n2, ov in (("a", 64, 80, 16), ("b", 32, 10, 16)):
        c1 = rng.standard_normal(n1).astype(np.float32)
        c2 = rng.standard_normal(n2).astype(np.float32)
        out[f"crossfade.{tag}.c1"] = c1
        out[f"crossfade.{tag}.c2"] = c2.copy()
        out[f"crossfade.{tag}.out"] = ns["crossfade"](c1.copy(), c2.copy(), ov)


def _wrapper_methods(*names):
    """Methods of `SeedVCWrapper` (seed_vc_wrapper.py) by name, as AST nodes: the module's imports (librosa, torchaudio,
    pydub, transformers models) are absent or heavy, the methods themselves need numpy / torch only."""
    src = open(os.path.join(REF, "seed_vc_wrapper.py")).read()
    cls = [n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "SeedVCWrapper"][0]
    found = {n.name: n for n in cls.body if isinstance(n, ast.FunctionDef)}
    return [found[n] for n in names]


def reference_chunk_machinery():
    """-> (fake-self class holding the reference's `crossfade` + `_stream_wave_chunks`, `loop(self, ...)`): `loop` is the
    tail of `SeedVCWrapper.convert_voice` from `max_source_window = ...` to its end (seed_vc_wrapper.py:560-623: the
    while-loop over chunks with its window arithmetic and the calls of cfm.inference / bigvgan_fn /
    `_stream_wave_chunks`), re-wrapped as a function of the names it reads.  Nothing of it is stored: it is extracted and
    compiled at run time, like `gen_crossfade` does for inference.py."""
    crossfade_fn, stream_fn, convert = _wrapper_methods("crossfade", "_stream_wave_chunks", "convert_voice")
    start = [i for i, n in enumerate(convert.body) if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Name)
             and n.targets[0].id == "max_source_window"][0]
    params = ["self", "cond", "prompt_condition", "mel2", "style2", "max_context_window", "inference_module", "diffusion_steps",
              "inference_cfg_rate", "bigvgan_fn", "overlap_wave_len", "stream_output", "sr"]
    loop = ast.FunctionDef(name="loop", args=ast.arguments(posonlyargs=[], args=[ast.arg(arg=p) for p in params], kwonlyargs=[],
                                                          kw_defaults=[], defaults=[]),
                           body=convert.body[start:], decorator_list=[])
    cls = ast.ClassDef(name="FakeSelf", bases=[], keywords=[], body=[crossfade_fn, stream_fn], decorator_list=[])
    mod = ast.fix_missing_locations(ast.Module(body=[cls, loop], type_ignores=[]))
    ns = {"np": np, "torch": torch}
    exec(compile(mod, "seed_vc_wrapper.py", "exec"), ns)
    return ns["FakeSelf"], ns["loop"]


def gen_chunkloop(out):
    """a21 pinned by the reference's own loop: (1) the chunk while-loop of `convert_voice` + `_stream_wave_chunks` +
    `crossfade` over 1-, 2-, 4- and 5-chunk inputs with the fake sampler / vocoder of cases.py; (2) `_stream_wave_chunks`
    driven chunk by chunk, incl. a last chunk shorter than the overlap."""
    from munch import Munch
    FakeSelf, loop = reference_chunk_machinery()
    me = FakeSelf()
    me.overlap_frame_len, me.device, me.fp16, me.bitrate = cases.CHUNK_OVERLAP, torch.device("cpu"), False, "320k"
    ovw = cases.CHUNK_OVERLAP * cases.CHUNK_HOP
    for name in cases.CHUNKLOOP_CASES:
        c = cases.chunkloop_case(name)
        calls = []

        def inference(cat_condition, x_lens, mel2, style2, f0, n, inference_cfg_rate=0.7):
            assert int(x_lens[0]) == cat_condition.size(1) and f0 is None
            calls.append(cat_condition.size(1))
            return cases.fake_sampler(cat_condition, mel2.size(-1))

        mod = Munch(cfm=Munch(inference=inference))
        g = loop(me, c["cond"], c["prompt_condition"], c["mel2"], c["style2"], cases.CHUNK_WINDOW, mod, 10, 0.7,
                 cases.fake_vocoder, ovw, False, 22050)
        try:
            next(g)
            raise AssertionError("non-streaming run must not yield")
        except StopIteration as e:
            wave = e.value
        out[f"chunkloop.{name}.out"] = np.asarray(wave)
        out[f"chunkloop.{name}.calls"] = np.asarray(calls, dtype=np.int64)
        print(f"chunkloop {name}: {len(calls)} chunks {calls} -> {len(wave)} samples", flush=True)
    for name in cases.CHUNKSTREAM_CASES:
        waves, frames = cases.chunkstream_case(name)
        chunks, prev, processed = [], None, 0
        for i, (w, f) in enumerate(zip(waves, frames)):
            last = i == len(waves) - 1
            processed, prev, brk, mp3, full = me._stream_wave_chunks(w, processed, torch.zeros(1, 1, f), ovw, chunks, prev, last, False, 22050)
            assert brk == last and mp3 is None
        out[f"chunkstream.{name}.out"] = np.asarray(full)
        out[f"chunkstream.{name}.processed"] = np.int64(processed)
        print(f"chunkstream {name}: frames {frames} -> {len(full)} samples, processed {processed}", flush=True)


def gen_ckpt(out):
    """BASELINE configs[0]'s plumbing with the reference's own loader: a synthetic .pth in the reference's
    {"net": {"cfm", "length_regulator"}} layout (keys carry DDP's `module.` prefix; `estimator.input_pos` deliberately has
    another shape, as a checkpoint trained with another block_size would; one key the model does not have) ->
    `build_model(recursive_munch(yaml))` + `load_checkpoint` (modules/commons.py:387-479) -> `model.cfm.inference`.
    Stores the sampler output of the LOADED model; asserts that what the loader left in the modules is exactly what the
    HIP path's `svc_dit_create` / `svc_lr_create` would be handed through `module.state_dict()`."""
    import tempfile
    import yaml
    from modules.commons import build_model, load_checkpoint, recursive_munch
    from seedvc_amd import shim
    cfg, sd, lc, lsd, inp = cases.ckpt_case()
    config = yaml.safe_load(open(os.path.join(REF, "configs", "presets", "config_dit_mel_seed_uvit_xlsr_tiny.yml")))
    model_params = recursive_munch(config["model_params"])
    model_params.dit_type = "DiT"                                                   # inference.py:73
    model = build_model(model_params, stage="DiT")
    net = {"cfm": {"module.estimator." + k: v.clone() for k, v in sd.items()},
           "length_regulator": {"module." + k: v.clone() for k, v in lsd.items()}}
    net["cfm"]["module.estimator.input_pos"] = torch.arange(4096)                   # shape mismatch: must be dropped
    net["cfm"]["module.estimator.not_in_the_model"] = torch.zeros(3)                # unknown key: must be dropped
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "synthetic.pth")
        torch.save({"net": net, "epoch": 3, "iters": 7}, path)
        model, _, _, _ = load_checkpoint(model, None, path, load_only_params=True, ignore_modules=[], is_distributed=False)
    model.cfm.estimator.setup_caches(max_batch_size=1, max_seq_length=8192)         # inference.py:90
    got = model.cfm.estimator.state_dict()
    assert set(got) == set(sd), sorted(set(got) ^ set(sd))[:10]
    for k, v in sd.items():
        want = torch.arange(16384) if k == "input_pos" else v          # diffusion_transformer.py:443
        assert torch.equal(got[k].cpu(), want.to(got[k].dtype)), k
    lgot = model.length_regulator.state_dict()
    assert set(lgot) == set(lsd)
    for k, v in lsd.items():
        assert torch.equal(lgot[k], v.to(lgot[k].dtype)), k
    assert not model.cfm.training and not model.length_regulator.training
    mine = shim.dit_cfg_from_reference_args(model_params)                          # the object patch_cfm receives
    for k, v in cfg.items():
        if k != "name":
            assert mine.get(k) == v, (k, mine.get(k), v)
    assert shim.lr_cfg_from_module(model.length_regulator) == {k: lc[k] for k in shim.lr_cfg_from_module(model.length_regulator)}
    lens = torch.LongTensor([cases.CKPT_T])
    real_randn = torch.randn
    torch.randn = lambda *a, **k: inp["z"].clone()
    try:
        smp = model.cfm.inference(inp["mu"], lens, inp["prompt"], inp["style"], None, cases.CKPT_STEPS, inference_cfg_rate=0.7)
    finally:
        torch.randn = real_randn
    out["ckpt.tiny.sample"] = smp.numpy()
    print(f"ckpt: loaded through load_checkpoint; sample |mean| {smp.abs().mean():.4f}", flush=True)



def main():
    which = sys.argv[1:] or ["dit", "bigvgan", "act", "hift", "crossfade", "ar", "lr", "argen", "argenfull", "mel", "campplus", "chunkloop", "ckpt"]
    for w in which:
        out = {}
        globals()["gen_" + w](out)
        path = os.path.join(HERE, f"{w}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
