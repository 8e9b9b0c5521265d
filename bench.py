"""bench.py -- mel-frames/sec of the hot path (CFM Euler sampler, 25 steps, CFG 0.7 + vocoder) on MI355X.

Workload at N=1 = BASELINE.json configs[1]: seed-uvit-tat-xlsr-tiny (D384 L9, time/style tokens, UViT skips),
25 diffusion steps, fp16 MFMA operands with fp32 accumulate/state, batch = 64 synthetic 22.05 kHz utterances
(P = S = 430 frames -> T = 860) + HiFT vocoder.  One "step" = one pass over one such batch per rank.
N > 1: one process per GPU (torch.distributed / RCCL), every rank converts its own 64-utterance shard and the
output audio is gathered on rank 0 (weak scaling); value = all frames of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.  Extra objects: `roofline` (dominant kernel = fp16 tap-GEMM, timed live with HIP
events on the launch stream in a separate single-lane pass after the timed region) and `cpu_baseline` (the CPU oracle on the
host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="tiny", choices=["tiny", "small", "base"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--diffusion-steps", type=int, default=0, help="default 25 (tiny/small), 50 (base)")
    ap.add_argument("--frames", type=int, default=430, help="prompt frames = source frames")
    ap.add_argument("--microbatch", type=int, default=0)
    ap.add_argument("--vocoder-precision", default="fp16x3", choices=["fp32", "fp16", "fp16x3"])
    ap.add_argument("--lanes", type=int, default=2, help="independent handle pairs / HIP streams per GPU")
    ap.add_argument("--dist-backend", default="nccl", help="rehearsal only: gloo runs the N > 1 code path without RCCL")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0 (1-GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


PEAK_F16_TFLOPS = 2500.0        # dense fp16/bf16 MFMA (MI355X_MICROARCH.md)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if a.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.dist_backend)
    torch.set_grad_enabled(False)

    from _pkgload import load_package
    load_package()
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from seedvc_amd import specs, weights, _lib
    from seedvc_amd.cfm import CFM
    from seedvc_amd.vocoder import HiFT, BigVGAN
    from seedvc_amd.pipeline import gather_audio
    import cases

    cfg = specs.dit_config(a.model)
    sd = weights.make_state_dict(specs.dit_state_spec(cfg), seed=1234, prefix=f"dit.{a.model}.")
    if a.model == "tiny":
        vc = specs.hift_config()
        vsd = weights.make_state_dict(specs.hift_state_spec(vc), seed=1234, prefix="hift.")
        hop = specs.hift_total_upsample(vc)
    else:
        vc = specs.bigvgan_config("44k" if a.model == "base" else "22k")
        vsd = weights.make_state_dict(specs.bigvgan_state_spec(vc), seed=1234, prefix="bigvgan.")
        hop = specs.bigvgan_total_upsample(vc)
    sr = 44100.0 if a.model == "base" else 22050.0
    if not a.diffusion_steps:
        a.diffusion_steps = 50 if a.model == "base" else 25
    model_name = {"tiny": "seed-uvit-tat-xlsr-tiny", "small": "seed-uvit-whisper-small-wavenet",
                  "base": "seed-uvit-whisper-base (44.1 kHz SVC)"}[a.model]
    voc_name = "HiFT" if a.model == "tiny" else ("BigVGAN-44k" if a.model == "base" else "BigVGAN-22k")

    def make_pair():
        cfm_ = CFM(cfg, sd, dev)
        if a.microbatch:
            cfm_.estimator.set_microbatch(a.microbatch)
        voc_ = (HiFT if a.model == "tiny" else BigVGAN)(vc, vsd, dev, precision=a.vocoder_precision)
        return cfm_, voc_

    from seedvc_amd.pipeline import Lanes
    lanes = Lanes(make_pair, max(1, min(a.lanes, a.batch)), dev)

    B, P, S = a.batch, a.frames, a.frames
    T = P + S
    seed = 1234 + rank
    mu = cases.randn("bench.mu", seed, B, T, cfg["Dc"]).to(dev)
    prompt = cases.logmel("bench.prompt", seed, B, cfg["C"], P).to(dev)
    style = cases.randn("bench.style", seed, B, cfg["style_dim"]).to(dev)
    z = cases.randn("bench.z", seed, B, cfg["C"], T).to(dev)
    lens = torch.LongTensor([T] * B)
    vkw = {}
    if a.model == "tiny":
        nh = vc["nb_harmonics"] + 1
        g = torch.Generator(device=dev).manual_seed(seed)
        vkw = dict(phase0=(torch.rand(B, nh, 1, device=dev, generator=g) * 2 - 1) * 3.14159265,
                   noise=torch.randn(B, nh, S * hop, device=dev, generator=g))

    def step():
        mel, wave = lanes.convert_batch(mu, prompt, style, a.diffusion_steps, 0.7, z=z, vocoder_kwargs=vkw)
        if world > 1:
            gather_audio(wave, [wave.size(1)] * B, B * world)
        return wave

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    frames = world * B * S * a.steps
    value = frames / dt

    out = {
        "metric": "mel_frames_per_sec", "value": round(value, 1), "unit": "mel-frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "rtf": round((dt / a.steps) / (B * S * hop / sr), 6),
        "realtime_factor_per_gpu": round(value / world / (sr / hop), 1),
        "config": {"workload": f"{model_name} CFM {a.diffusion_steps} steps cfg 0.7 + {voc_name}, "
                               f"batch {B} x (P={P},S={S}) per GPU",
                   "batch_per_gpu": B, "global_batch": B * world, "prompt_frames": P, "source_frames": S,
                   "diffusion_steps": a.diffusion_steps, "cfg_rate": 0.7, "vocoder_precision": a.vocoder_precision,
                   "lanes_per_gpu": len(lanes.lanes),
                   "parallelism": f"utterance-sharded x{world}, audio gather on rank 0"},
    }

    if rank == 0 and not a.no_roofline:
        import ctypes as C
        L = _lib.lib()
        # per-kernel durations are taken in a serial pass (one lane, one stream): with several lanes the kernels of
        # different streams overlap and an event pair around one launch would also time its neighbours.
        L.svc_prof_enable(1)
        lanes.lanes[0][0].convert_batch(mu, prompt, style, a.diffusion_steps, 0.7, z=z, vocoder_kwargs=vkw)
        torch.cuda.synchronize()
        buf = (C.c_double * 12)()
        L.svc_prof_collect(buf, 3)
        L.svc_prof_enable(0)
        n, ms, fl, by = buf[0], buf[1], buf[2], buf[3]
        names = ["kgemm_f16", "kgemm_f32", "attention"]
        detail = {}
        for i, nm in enumerate(names):
            if buf[i * 4] > 0:
                detail[nm] = {"launches": int(buf[i * 4]), "total_ms": round(buf[i * 4 + 1], 3),
                              "tflops": round(buf[i * 4 + 2] / (buf[i * 4 + 1] * 1e-3) / 1e12, 2),
                              "alg_GBps": round(buf[i * 4 + 3] / (buf[i * 4 + 1] * 1e-3) / 1e9, 1)}
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # HBM traffic per launch of the same kernel class comes from rocprofv3 PMC passes of THIS command
        # (FETCH_SIZE / WRITE_SIZE in separate passes, tools/pmc_traffic.py applies the gfx950 corrections);
        # it cannot be collected from inside the process, so the last committed measurement is reported.
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pmc.get("workload") == f"{a.model}-b{a.batch}":
                traffic = round(pmc["kgemm_f16"]["hbm_bytes_per_launch"])
        except Exception:
            traffic = None
        out["roofline"] = {"bound": "mfma", "kernel": "kgemm_kernel<f16> (tap-GEMM, all DiT linears)",
                           "achieved": round(ach, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_F16_TFLOPS, 4), "traffic": traffic,
                           "launches": int(n), "avg_launch_ms": round(ms / max(n, 1), 4),
                           "alg_flop_per_launch": round(fl / max(n, 1)), "per_class": detail,
                           "alg_gflop_per_frame": round(sum(buf[i * 4 + 2] for i in range(3)) / (B * S) / 1e9, 3),
                           "end_to_end_tflops": round(value / world * sum(buf[i * 4 + 2] for i in range(3))
                                                      / (B * S) / 1e12, 2)}

    if rank == 0 and world == 1 and not a.no_cpu_baseline:      # reported at N = 1 only
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import seedvc_oracle as O
        nthreads = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(nthreads)
        nb = 1
        t1 = time.perf_counter()
        for b in range(nb):
            m = O.cfm_sample(sd, cfg, z[b:b + 1].cpu(), T, prompt[b:b + 1].cpu(), mu[b:b + 1].cpu(), style[b:b + 1].cpu(),
                             a.diffusion_steps, 0.7)[:, :, P:]
            if a.model == "tiny":
                O.hift_forward(voc_sd(None, vc, weights, specs), vc, m, vkw["phase0"][b:b + 1].cpu(), vkw["noise"][b:b + 1].cpu())
            else:
                O.bigvgan_forward(voc_sd(None, vc, weights, specs), vc, m)
        cpu_dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(nb * S / cpu_dt, 2), "unit": "mel-frames/s", "cores": nthreads, "kind": "port",
                               "sample": f"{nb} utterance(s) of the same workload (P=S={S}, {a.diffusion_steps} steps + vocoder), "
                                         f"torch fp32 oracle, {cpu_dt:.1f} s"}

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def voc_sd(voc, vc, weights, specs):  # noqa: ARG001
    if "in_channels" in vc:
        return weights.make_state_dict(specs.hift_state_spec(vc), seed=1234, prefix="hift.")
    return weights.make_state_dict(specs.bigvgan_state_spec(vc), seed=1234, prefix="bigvgan.")


if __name__ == "__main__":
    main()
