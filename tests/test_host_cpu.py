"""CPU-only checks: the C-ABI library loads and exports every declared symbol, host-side logic (sharding,
gather over gloo with world_size 2, crossfade harness, specs / weight generator determinism)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from seedvc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "seedvc_hip.h")).read()
    declared = set(re.findall(r"\b(svc_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/seedvc_hip.h but not exported"
    assert set(_lib.EXPORTS) <= declared
    assert lib.svc_abi_version() == 1


def test_product_path_has_no_cpu_fallback():
    # importing the host mirror never imports the oracle, and the oracle dir is not on the package path
    for f in os.listdir(os.path.join(ROOT, "seed-vc_amd")):
        if f.endswith(".py"):
            src = open(os.path.join(ROOT, "seed-vc_amd", f)).read()
            assert "seedvc_oracle" not in src and "oracle" not in src.replace("# oracle", ""), f


def test_weight_generator_is_deterministic():
    from seedvc_amd import specs, weights
    cfg = specs.dit_config("tiny", D=128, H=2, L=3)
    a = weights.make_state_dict(specs.dit_state_spec(cfg), seed=5, prefix="dit.tiny.")
    b = weights.make_state_dict(specs.dit_state_spec(cfg), seed=5, prefix="dit.tiny.")
    c = weights.make_state_dict(specs.dit_state_spec(cfg), seed=6, prefix="dit.tiny.")
    k = "transformer.layers.0.attention.wqkv.weight"
    assert torch.equal(a[k], b[k]) and not torch.equal(a[k], c[k])
    # spot checksum pins the generator across machines (numpy Philox)
    assert abs(float(a[k].double().sum()) - float(b[k].double().sum())) == 0.0


def test_specs_shapes():
    from seedvc_amd import specs
    assert specs.dit_config("tiny")["I"] == 1024 and specs.dit_config("small")["I"] == 1536 and specs.dit_config("base")["I"] == 2048
    assert specs.uvit_layers(specs.dit_config("small")) == ([0, 1, 2, 3, 4, 5], [7, 8, 9, 10, 11, 12])
    assert specs.uvit_layers(specs.dit_config("tiny")) == ([0, 1, 2, 3], [5, 6, 7, 8])
    assert specs.uvit_layers(specs.dit_config("v2")) == ([], [])
    assert specs.bigvgan_total_upsample(specs.bigvgan_config("22k")) == 256
    assert specs.hift_total_upsample(specs.hift_config()) == 256


def test_shard_range_covers_everything():
    from seedvc_amd.pipeline import shard_range
    for n in (0, 1, 7, 64, 513):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1


def test_crossfade_matches_reference_golden(golden):
    from seedvc_amd.pipeline import crossfade
    for tag in ("a", "b"):
        out = crossfade(golden[f"crossfade.{tag}.c1"].copy(), golden[f"crossfade.{tag}.c2"].copy(), 16)
        np.testing.assert_array_equal(out, golden[f"crossfade.{tag}.out"])


class _FakeCFM:
    """Stands in for the HIP sampler so the chunk loop can be exercised on CPU (host logic only)."""

    def __init__(self, C):
        self.C = C
        self.calls = []

    def inference(self, mu, x_lens, prompt, style, f0, n, inference_cfg_rate=0.7, z=None, **kw):
        self.calls.append(mu.size(1))
        t = torch.arange(mu.size(1), dtype=torch.float32)
        return (mu[..., :1].transpose(1, 2) + 0 * t).expand(1, self.C, -1).clone()


def test_chunk_loop_equals_oracle_harness():
    """convert_long reproduces the reference's chunk boundaries / crossfade (inference.py:470-527) -- checked
    against the oracle's restatement with the same fake sampler / vocoder."""
    import seedvc_oracle as O
    from seedvc_amd.pipeline import HotPath
    C, hop, P = 4, 8, 30
    cond = torch.randn(1, 250, 6)
    pc = torch.randn(1, P, 6)
    mel2 = torch.randn(1, C, P)
    style = torch.randn(1, 3)
    voc = lambda m: torch.repeat_interleave(m[:, 0, :], hop, dim=1)     # noqa: E731
    fake = _FakeCFM(C)
    hp = HotPath(fake, voc)
    out = hp.convert_long(cond, pc, mel2, style, 2, 0.7, hop, max_context_window=100)
    ref = O.chunked_convert(lambda cc: _FakeCFM(C).inference(cc, None, None, None, None, 2), voc, cond, pc, mel2, style, hop, 100)
    assert torch.equal(out, ref)
    assert fake.calls[0] == 100 and len(fake.calls) == 5


class _CaseCFM:
    """The chunk-loop cases' fake sampler behind the HIP sampler's call signature (host logic only)."""

    def __init__(self):
        self.calls = []

    def inference(self, mu, x_lens, prompt, style, f0, n, inference_cfg_rate=0.7, z=None, **kw):
        import cases
        assert int(x_lens[0]) == mu.size(1) and f0 is None
        self.calls.append(mu.size(1))
        return cases.fake_sampler(mu, prompt.size(-1))


@pytest.mark.parametrize("name", ["loop1", "loop1s", "loop2", "loop2b", "loop4", "loop5"])
def test_chunk_loop_equals_the_references_own_loop(name, golden):
    """a21 pinned by the reference itself: tests/golden/chunkloop.npz holds what the while-loop of
    `SeedVCWrapper.convert_voice` + `_stream_wave_chunks` + `crossfade` (seed_vc_wrapper.py:190-285,560-623, extracted and run by
    make_golden.py) produce with the fake sampler / vocoder of cases.py; `convert_long` and the oracle's restatement
    must give the same samples bit for bit and call the sampler with the same window lengths."""
    import cases
    import seedvc_oracle as O
    from seedvc_amd.pipeline import HotPath, chunk_plan
    c = cases.chunkloop_case(name)
    fake = _CaseCFM()
    out = HotPath(fake, cases.fake_vocoder).convert_long(c["cond"], c["prompt_condition"], c["mel2"], c["style2"], 10, 0.7,
                                                         cases.CHUNK_HOP, cases.CHUNK_WINDOW, overlap_frame_len=cases.CHUNK_OVERLAP)
    want = golden[f"chunkloop.{name}.out"]
    assert out.dtype == torch.float32 and out.shape == (1, len(want))
    np.testing.assert_array_equal(out[0].numpy(), want.astype(np.float32))
    assert fake.calls == golden[f"chunkloop.{name}.calls"].tolist()
    plan = chunk_plan(c["cond"].size(1), cases.CHUNK_WINDOW - cases.CHUNK_P, cases.CHUNK_OVERLAP)
    assert [cases.CHUNK_P + n for _, n, _ in plan] == fake.calls and plan[-1][2]
    ref = O.chunked_convert(lambda cc: cases.fake_sampler(cc, cases.CHUNK_P), lambda m: cases.fake_vocoder(m).reshape(1, -1), c["cond"],
                            c["prompt_condition"], c["mel2"], c["style2"], cases.CHUNK_HOP, cases.CHUNK_WINDOW,
                            overlap_frame_len=cases.CHUNK_OVERLAP)
    np.testing.assert_array_equal(ref[0].numpy(), want.astype(np.float32))


@pytest.mark.parametrize("name", ["short", "short3", "even"])
def test_stream_wave_chunks_equals_reference_method(name, golden):
    """`pipeline.stream_wave_chunks` against `SeedVCWrapper._stream_wave_chunks` driven chunk by chunk (incl. a last chunk
    SHORTER than the overlap, which the drivers' own window arithmetic never produces)."""
    import cases
    from seedvc_amd.pipeline import stream_wave_chunks
    waves, frames = cases.chunkstream_case(name)
    ovw = cases.CHUNK_OVERLAP * cases.CHUNK_HOP
    chunks, prev, processed = [], None, 0
    for i, (w, f) in enumerate(zip(waves, frames)):
        last = i == len(waves) - 1
        processed, prev, brk = stream_wave_chunks(w, processed, f, ovw, cases.CHUNK_OVERLAP, chunks, prev, last)
        assert brk == last
    np.testing.assert_array_equal(np.concatenate(chunks), golden[f"chunkstream.{name}.out"])
    assert processed == int(golden[f"chunkstream.{name}.processed"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/modules"), reason="reference tree not present")
def test_reference_checkpoint_loader_leaves_what_the_shim_packs(golden):
    """(b) `checkpoint loading stays unchanged`: a synthetic .pth in the reference's {"net": {"cfm", "length_regulator"}} layout
    (DDP `module.` prefixes, one shape-mismatched and one unknown key) goes through the REFERENCE's `build_model` +
    `load_checkpoint` (modules/commons.py:387-479); make_golden.gen_ckpt asserts that the loaded modules' state dicts are
    exactly the generated tensors, that `shim.dit_cfg_from_reference_args(<the Munch>)` / `lr_cfg_from_module` equal
    `specs`, and re-runs `model.cfm.inference`: its output must equal the committed fixture bit for bit."""
    script = f"""
import sys
sys.path.insert(0, {os.path.join(ROOT, 'tests', 'golden')!r})
import make_golden as G
out = {{}}
G.gen_ckpt(out)
import numpy as np
want = np.load({os.path.join(ROOT, 'tests', 'golden', 'ckpt.npz')!r})["ckpt.tiny.sample"]
assert np.array_equal(out["ckpt.tiny.sample"], want), float(abs(out["ckpt.tiny.sample"] - want).max())
print("CKPT_OK")
"""
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CKPT_OK" in r.stdout, r.stderr[-3000:]


def test_gather_audio_gloo_world2(tmp_path):
    """N > 1 path: two CPU ranks (gloo) shard 5 utterances of ragged length and gather them on rank 0."""
    script = tmp_path / "w.py"
    script.write_text(f"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {ROOT!r})
import _pkgload; _pkgload.load_package()
from seedvc_amd.pipeline import shard_range, gather_audio
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = 5
lens = [100, 37, 64, 1, 80]
s, e = shard_range(n, rank, world)
loc = torch.zeros(e - s, max(lens[s:e]))
for i in range(s, e):
    loc[i - s, :lens[i]] = torch.arange(lens[i]) + 1000 * i
out = gather_audio(loc, lens[s:e], n)
if rank == 0:
    assert len(out) == n
    for i in range(n):
        assert torch.equal(out[i], torch.arange(lens[i]) + 1000.0 * i), i
    print("GATHER_OK")
else:
    assert out is None
dist.destroy_process_group()
""")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "GATHER_OK" in r.stdout


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs/presets"), reason="reference tree not present")
def test_shim_config_mapping_matches_presets():
    """shim.dit_cfg_from_reference_args on the reference's own preset YAMLs (read as data) == specs presets."""
    import yaml
    from seedvc_amd import shim, specs
    for preset, fname in (("tiny", "config_dit_mel_seed_uvit_xlsr_tiny.yml"),
                          ("small", "config_dit_mel_seed_uvit_whisper_small_wavenet.yml"),
                          ("base", "config_dit_mel_seed_uvit_whisper_base_f0_44k.yml")):
        mp = yaml.safe_load(open(os.path.join("/root/reference/configs/presets", fname)))["model_params"]
        got = shim.dit_cfg_from_reference_args(mp)
        want = specs.dit_config(preset)
        for k in ("D", "H", "L", "C", "Dc", "style_dim", "head", "time_as_token", "style_as_token", "uvit", "long_skip",
                  "style_condition", "I", "n_prefix", "hd"):
            assert got[k] == want[k], (preset, k, got[k], want[k])
        if want["head"] == "wavenet":
            for k in ("wn_dim", "wn_layers", "wn_kernel", "wn_dilation"):
                assert got[k] == want[k]


def test_chunk_plan_matches_reference_loop_arithmetic():
    """pipeline.chunk_plan vs a literal walk of the driver loop (inference.py:473-527)."""
    from seedvc_amd.pipeline import chunk_plan
    for n_src, msw, ov in ((70, 24, 4), (24, 24, 4), (25, 24, 4), (6029, 2150, 16), (45, 24, 4), (1, 24, 4)):
        ref, processed = [], 0
        while processed < n_src:
            chunk = list(range(processed, min(processed + msw, n_src)))
            is_last = processed + msw >= n_src
            ref.append((processed, len(chunk), is_last))
            if is_last:
                break
            processed += len(chunk) - ov
        assert chunk_plan(n_src, msw, ov) == ref
        assert ref[-1][2] and ref[-1][0] + ref[-1][1] == n_src


def test_slaney_mel_basis_properties():
    """Restated librosa filterbank (parity unpinned: librosa is absent): shape, support, Slaney area normalisation."""
    import numpy as np
    from seedvc_amd.audio import slaney_mel_basis
    for sr, n_fft, n_mels in ((22050, 1024, 80), (44100, 2048, 128)):
        w = slaney_mel_basis(sr, n_fft, n_mels, 0, None).numpy().astype(np.float64)
        assert w.shape == (n_mels, n_fft // 2 + 1) and (w >= 0).all()
        peaks = w.argmax(axis=1)
        assert (np.diff(peaks) > 0).all()                       # centre frequencies increase
        assert ((w > 0).sum(axis=1) >= 1).all()                 # no empty filter at these resolutions
        df = sr / n_fft
        area = w.sum(axis=1) * df                               # triangles normalised to unit area (norm='slaney')
        assert abs(np.median(area) - 1.0) < 0.02 and np.abs(area - 1.0).max() < 0.3      # narrow low filters are coarsely sampled


def test_lr_and_mel_symbols_have_python_mirrors():
    from seedvc_amd import specs
    for preset in specs.LR_PRESETS:
        c = specs.lr_config(preset)
        spec = specs.lr_state_spec(c)
        assert ("content_in_proj.weight" in spec) == (not c["is_discrete"])
        assert (f"model.{3 * c['n_convs']}.weight" in spec) == specs.lr_has_final_conv(c)


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` (no WORLD_SIZE in the environment) must start two ranks itself and print ONE JSON line
    with n_gpus = 2.  --dry-run keeps the ranks on the CPU (gloo rendezvous, barrier, max-over-ranks, rank-0 print): the
    launcher / rendezvous / single-line contract is what is tested, the GPU work is covered by the -m gpu tests."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["max_over_ranks"] == 2.0
    # the N > 1 diagnostics object a real SCALE run carries (bench.rank_stats): rank count as torch.distributed reports it
    # after init, per-rank step time (min / mean / max) and the audio gather timed on its own
    rk = rec["ranks"]
    assert rk["nccl_ranks"] == 2 and rk["backend"] == "gloo"
    assert rk["step_ms_per_rank"] == [1.0, 2.0] and (rk["step_ms_min"], rk["step_ms_mean"], rk["step_ms_max"]) == (1.0, 1.5, 2.0)
    assert rk["gather_ms_per_rank"] == [0.1, 0.2] and rk["gather_ms_max"] == 0.2


def test_bench_launcher_command_is_the_drivers_form():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse(["--gpus", "4", "--master-port", "29555"])
    cmd = bench.rank_command(a, ["--gpus", "4", "--steps", "3"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    # a failing child is reported through the exit code
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--model", "nonexistent"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0


@pytest.mark.skipif(not os.path.isdir("/root/reference/modules/v2"), reason="reference tree not present")
def test_shim_reads_configs_from_the_real_reference_modules():
    """shim.dit_cfg_from_v2_module / lr_cfg_from_module / campplus_cfg_from_module on the REAL reference modules (imported from /root/reference
    with the absent third-party packages stubbed as in tests/golden/make_golden.py) == specs presets."""
    script = f"""
import sys
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests', 'golden', '_stubs')!r}); sys.path.insert(1, '/root/reference')
import torch
import _pkgload; _pkgload.load_package()
from seedvc_amd import shim, specs
from modules.v2.dit_wrapper import DiT
cfg = specs.dit_config('v2')
est = DiT(time_as_token=cfg['time_as_token'], style_as_token=cfg['style_as_token'], uvit_skip_connection=cfg['uvit'],
          block_size=cfg['block_size'], depth=cfg['L'], num_heads=cfg['H'], hidden_dim=cfg['D'], in_channels=cfg['C'],
          content_dim=cfg['Dc'], style_encoder_dim=cfg['style_dim'], class_dropout_prob=0.1, dropout_rate=0.0, attn_dropout_rate=0.0)
got = shim.dit_cfg_from_v2_module(est)
for k in ('version', 'D', 'H', 'L', 'C', 'Dc', 'style_dim', 'time_as_token', 'style_as_token', 'uvit', 'I', 'n_prefix', 'head'):
    assert got[k] == cfg[k], (k, got[k], cfg[k])
for preset, c in specs.LR_PRESETS.items():
    c = specs.lr_config(preset)
    kw = dict(channels=c['channels'], sampling_ratios=[1] * c['n_convs'], is_discrete=c['is_discrete'],
              in_channels=c['in_channels'] or None, codebook_size=c['codebook_size'], out_channels=c['out_channels'],
              f0_condition=c['f0_condition'], n_f0_bins=c['n_f0_bins'])
    if c['version'] == 1:
        from modules.length_regulator import InterpolateRegulator as LR
        m = LR(vector_quantize=False, n_codebooks=1, quantizer_dropout=0.0, **kw)
    else:
        from modules.v2.length_regulator import InterpolateRegulator as LR2
        m = LR2(**kw)
    got = shim.lr_cfg_from_module(m)
    for k in ('version', 'channels', 'is_discrete', 'n_convs', 'codebook_size', 'f0_condition', 'n_f0_bins', 'in_channels', 'out_channels'):
        assert got[k] == c[k], (preset, k, got[k], c[k])
from modules.campplus.DTDNN import CAMPPlus
cp = specs.campplus_config()
got = shim.campplus_cfg_from_module(CAMPPlus(feat_dim=80, embedding_size=192))       # as built at inference.py:98
assert got == cp, (got, cp)
print('SHIM_CFG_OK')
"""
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHIM_CFG_OK" in r.stdout, r.stderr[-3000:]
