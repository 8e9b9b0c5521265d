"""bench.py -- mel-frames/sec of the hot path (CFM Euler sampler, 25 steps, CFG 0.7 + vocoder) on MI355X.

Workload at N=1 = BASELINE.json configs[1]: seed-uvit-tat-xlsr-tiny (D384 L9, time/style tokens, UViT skips),
25 diffusion steps, fp16 MFMA operands with fp32 accumulate/state, batch = 64 synthetic 22.05 kHz utterances
(P = S = 430 frames -> T = 860) + HiFT vocoder.  One "step" = one pass over one such batch per rank.
N > 1: one process per GPU (torch.distributed / RCCL), every rank converts its own 64-utterance shard and the
output audio is gathered on rank 0 (weak scaling); value = all frames of all ranks / max-over-ranks time.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process; the parent never touches the
GPU and relays the child's output and exit code).  Started under torch.distributed.run it is one of the ranks.

Prints ONE JSON line on rank 0.  Extra objects: `roofline` (dominant kernel class = fp16 tap-GEMM, timed live with HIP
events on the launch stream in a separate single-lane pass after the timed region), `cpu_baseline` (the CPU oracle on
the host cores: warm-up + median of 3, bounded sample) and `secondary` (N = 1 only: the north-star's own target
workload, seed-uvit-whisper-small + BigVGAN, as a B = 64 throughput line and a B = 1 latency line, measured in the
same process).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_TFLOPS = 2500.0        # dense fp16/bf16 MFMA (MI355X_MICROARCH.md)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="tiny", choices=["tiny", "small", "base", "v2"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--diffusion-steps", type=int, default=0, help="default 25 (tiny/small), 50 (base)")
    ap.add_argument("--frames", type=int, default=430, help="prompt frames = source frames")
    ap.add_argument("--microbatch", type=int, default=0)
    ap.add_argument("--fused-min-rows", type=int, default=-1, help="tuning: row threshold of the fused DiT row-panel path (-1 = library default)")
    ap.add_argument("--vocoder-precision", default="fp16p8", choices=["fp32", "fp16", "fp16x3", "fp16p8"])
    ap.add_argument("--lanes", type=int, default=2, help="independent handle pairs / HIP streams per GPU")
    ap.add_argument("--dist-backend", default="nccl", help="rehearsal only: gloo runs the N > 1 code path without RCCL")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0 (1-GPU box)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-started ranks (0 = pick a free one)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher self-test: ranks rendezvous over gloo on the CPU and rank 0 prints a JSON line without "
                         "touching the GPU (exercised by tests/test_host_cpu.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ N > 1 launcher
def rank_command(a, argv):
    """Command line that starts the a.gpus ranks of this script on one node (the driver's own form)."""
    port = a.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(a, argv):
    """Parent of the self-started ranks: never imports torch.cuda, never re-execs; relays output and exit code."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(rank_command(a, argv), env=env)


def dry_run_rank(a, rank, world):
    """Launcher self-test: same rendezvous / barrier / max-over-ranks / single JSON line as the real run, on the CPU."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    ranks = None
    if world > 1:
        mine = torch.tensor([1.0 + rank, 0.1 * (1 + rank)], dtype=torch.float64)
        allv = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        allv = torch.stack(allv)
        ranks = rank_stats(allv[:, 0].tolist(), allv[:, 1].tolist(), "gloo", world)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        line = {"metric": "mel_frames_per_sec", "value": 0.0, "unit": "mel-frames/s", "n_gpus": world,
                "steps": a.steps, "warmup": a.warmup, "dry_run": True, "max_over_ranks": float(t.item())}
        if ranks is not None:
            line["ranks"] = ranks
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def rank_stats(per_rank_ms, gather_ms, backend, world):
    """The N > 1 diagnostics object of the JSON line (also produced by --dry-run, so its shape is tested on the CPU)."""
    import torch.distributed as dist
    return {"nccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "requested_backend": backend,
            "step_ms_per_rank": [round(v, 3) for v in per_rank_ms],
            "step_ms_min": round(min(per_rank_ms), 3), "step_ms_mean": round(sum(per_rank_ms) / len(per_rank_ms), 3),
            "step_ms_max": round(max(per_rank_ms), 3),
            "gather_ms_per_rank": [round(v, 3) for v in gather_ms], "gather_ms_max": round(max(gather_ms), 3),
            "note": "step_ms = each rank's own time per step up to its local device sync (sampler + vocoder + its part of "
                    "the audio gather), before the closing barrier; gather_ms = the length all-gather + padded audio "
                    "gather alone, timed separately after the timed region (barrier, 3 gathers, device sync)"}


def rank_diagnostics(w, dt_local, steps, dev, world, B, backend):
    """Per-rank step time and the audio gather alone, so that a poor scaling figure can be attributed (load imbalance
    between ranks vs the collective) from the single JSON line."""
    import torch
    import torch.distributed as dist
    from seedvc_amd.pipeline import gather_audio
    _, wave = w.step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        gather_audio(wave, [wave.size(1)] * B, B * world)
    torch.cuda.synchronize()
    g_ms = (time.perf_counter() - t1) / 3 * 1e3
    mine = torch.tensor([dt_local / steps * 1e3, g_ms], device=dev, dtype=torch.float64)
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    allv = torch.stack(allv).cpu()
    return rank_stats(allv[:, 0].tolist(), allv[:, 1].tolist(), backend, world)


# ------------------------------------------------------------------------------------------------ workloads
MODEL_NAMES = {"tiny": "seed-uvit-tat-xlsr-tiny", "small": "seed-uvit-whisper-small-wavenet",
               "base": "seed-uvit-whisper-base (44.1 kHz SVC)", "v2": "v2 hubert-bsqvae-small CFM (3-way CFG)"}


class Workload:
    """One (model, vocoder, batch) configuration with its synthetic inputs resident on the device."""

    def __init__(self, model, batch, frames, diffusion_steps, lanes, dev, seed, microbatch=0, vocoder_precision="fp16p8",
                 fused_min_rows=-1):
        import torch
        from seedvc_amd import specs, weights
        from seedvc_amd.cfm import CFM
        from seedvc_amd.vocoder import HiFT, BigVGAN
        from seedvc_amd.pipeline import Lanes
        import cases
        self.model, self.B, self.P, self.S = model, batch, frames, frames
        self.T = self.P + self.S
        self.cfg = cfg = specs.dit_config(model)
        self.sd = sd = weights.make_state_dict(specs.dit_state_spec(cfg), seed=1234, prefix=f"dit.{model}.")
        if model == "tiny":
            self.vc = vc = specs.hift_config()
            self.vsd = vsd = weights.make_state_dict(specs.hift_state_spec(vc), seed=1234, prefix="hift.")
            self.hop = specs.hift_total_upsample(vc)
        else:
            self.vc = vc = specs.bigvgan_config("44k" if model == "base" else "22k")
            self.vsd = vsd = weights.make_state_dict(specs.bigvgan_state_spec(vc), seed=1234, prefix="bigvgan.")
            self.hop = specs.bigvgan_total_upsample(vc)
        self.sr = 44100.0 if model == "base" else 22050.0
        self.n_steps = diffusion_steps or (50 if model == "base" else 25)
        self.voc_name = "HiFT" if model == "tiny" else ("BigVGAN-44k" if model == "base" else "BigVGAN-22k")

        def make_pair():
            cfm_ = CFM(cfg, sd, dev)
            if microbatch:
                cfm_.estimator.set_microbatch(microbatch)
            if fused_min_rows >= 0:
                cfm_.estimator.set_fused_min_rows(fused_min_rows)
            voc_ = (HiFT if model == "tiny" else BigVGAN)(vc, vsd, dev, precision=vocoder_precision)
            return cfm_, voc_

        self.lanes = Lanes(make_pair, max(1, min(lanes, batch)), dev)
        B, P, S, T = self.B, self.P, self.S, self.T
        self.mu = cases.randn("bench.mu", seed, B, T, cfg["Dc"]).to(dev)
        self.prompt = cases.logmel("bench.prompt", seed, B, cfg["C"], P).to(dev)
        self.style = cases.randn("bench.style", seed, B, cfg["style_dim"]).to(dev)
        self.z = cases.randn("bench.z", seed, B, cfg["C"], T).to(dev)
        self.vkw = {}
        if model == "tiny":
            nh = vc["nb_harmonics"] + 1
            g = torch.Generator(device=dev).manual_seed(seed)
            self.vkw = dict(phase0=(torch.rand(B, nh, 1, device=dev, generator=g) * 2 - 1) * 3.14159265,
                            noise=torch.randn(B, nh, S * self.hop, device=dev, generator=g))

    def step(self):
        return self.lanes.convert_batch(self.mu, self.prompt, self.style, self.n_steps, 0.7, z=self.z, vocoder_kwargs=self.vkw)

    def step_single_lane(self):
        return self.lanes.lanes[0][0].convert_batch(self.mu, self.prompt, self.style, self.n_steps, 0.7, z=self.z,
                                                    vocoder_kwargs=self.vkw)

    def describe(self):
        return (f"{MODEL_NAMES[self.model]} CFM {self.n_steps} steps cfg 0.7 + {self.voc_name}, "
                f"batch {self.B} x (P={self.P},S={self.S}) per GPU")

    def timed(self, steps, warmup):
        import torch
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        return time.perf_counter() - t0


def roofline(w, frames_per_sec, pmc_key):
    """`roofline` object of one workload: a serial pass (one lane, one stream) with the library's per-launch HIP-event
    timer on (`svc_prof_*`): with several lanes the kernels of different streams overlap and an event pair around one
    launch would also time its neighbours.  Dominant class = all fp16 MFMA GEMM launches of the DiT and the vocoder
    (tap-GEMM, resident-tile conv and the fused row-panel DiT kernels)."""
    import ctypes as C
    import torch
    from seedvc_amd import _lib
    L = _lib.lib()
    L.svc_prof_enable(1)
    w.step_single_lane()
    torch.cuda.synchronize()
    ncls = 4
    buf = (C.c_double * (4 * ncls))()
    L.svc_prof_collect(buf, ncls)
    L.svc_prof_enable(0)
    n, ms, fl, by = buf[0], buf[1], buf[2], buf[3]
    names = ["kgemm_f16", "kgemm_f32", "attention", "fused_dit"]
    detail = {}
    for i, nm in enumerate(names):
        if buf[i * 4] > 0:
            detail[nm] = {"launches": int(buf[i * 4]), "total_ms": round(buf[i * 4 + 1], 3),
                          "tflops": round(buf[i * 4 + 2] / (buf[i * 4 + 1] * 1e-3) / 1e12, 2),
                          "alg_GBps": round(buf[i * 4 + 3] / (buf[i * 4 + 1] * 1e-3) / 1e9, 1)}
    n += buf[12]; ms += buf[13]; fl += buf[14]; by += buf[15]
    ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    # HBM traffic per launch of the same kernel class comes from rocprofv3 PMC passes of THIS command
    # (FETCH_SIZE / WRITE_SIZE in separate passes, tools/pmc_traffic.py applies the gfx950 corrections);
    # counters cannot be collected from inside the process, so the last committed measurement is reported and
    # labelled as such (traffic_source).
    traffic, traffic_source = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        pmc = pmc.get("workloads", {}).get(pmc_key) or (pmc if pmc.get("workload") == pmc_key else None)
        if pmc:
            traffic = round(pmc["kgemm_f16"]["hbm_bytes_per_launch"])
            traffic_source = {"file": "profiles/pmc_traffic.json", "commit": pmc.get("commit"),
                              "collected_by": "tools/profile_round.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
                              "note": "replayed from the committed profile, not measured in this run"}
    except Exception:
        traffic = None
    ntot = max(n, 1)
    alg = sum(buf[i * 4 + 2] for i in range(ncls)) / (w.B * w.S)
    return {"bound": "mfma", "kernel": "fp16 MFMA GEMM class (kgemm_kernel<f16> tap-GEMM, kconv, fused DiT row-panel kernels)",
            "achieved": round(ach, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_F16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "launches": int(n), "avg_launch_ms": round(ms / ntot, 4),
            "alg_flop_per_launch": round(fl / ntot), "alg_bytes_per_launch": round(by / ntot),
            "per_class": detail, "alg_gflop_per_frame": round(alg / 1e9, 3),
            "end_to_end_tflops": round(frames_per_sec * alg / 1e12, 2)}


def secondary_lines(a, dev):
    """The north-star's own target (>= 20x real time per stream on whisper-small @ 25 steps) in the driver's record."""
    import torch
    out = []
    for batch, lanes, steps, warmup in ((64, 2, 8, 2), (1, 1, 16, 3)):
        w = Workload("small", batch, a.frames, 0, lanes, dev, 1234, vocoder_precision=a.vocoder_precision)
        dt = w.timed(steps, warmup)
        per = dt / steps
        fps = batch * w.S / per
        out.append({"workload": w.describe(), "metric": "mel_frames_per_sec", "value": round(fps, 1),
                    "ms_per_step": round(per * 1e3, 3), "ms_per_utterance": round(per * 1e3 / batch, 3),
                    "realtime_factor": round(fps / (w.sr / w.hop), 1),
                    "realtime_factor_per_stream": round((w.S * w.hop / w.sr) / per, 1) if batch == 1 else None,
                    "steps": steps, "warmup": warmup, "lanes": lanes})
        if not a.no_roofline:
            out[-1]["roofline"] = roofline(w, fps, f"small-b{batch}")
        del w
        torch.cuda.empty_cache()
    return out


def host_cores():
    """Host cores this process may use: the affinity mask, capped by the cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return n


def cpu_baseline(w, nthreads):
    """CPU oracle (oracle/seedvc_oracle.py, a port of the reference's fp32 path) on one utterance of the same workload:
    a short warm-up run, then the median of three timed runs on all host cores."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import seedvc_oracle as O
    torch.set_num_threads(nthreads)
    z, prompt, mu, style = (t[:1].cpu() for t in (w.z, w.prompt, w.mu, w.style))
    vkw = {k: v[:1].cpu() for k, v in w.vkw.items()}

    def run(n_steps):
        m = O.cfm_sample(w.sd, w.cfg, z, w.T, prompt, mu, style, n_steps, [0.7, 0.7] if w.cfg["version"] == 2 else 0.7)[:, :, w.P:]
        if w.model == "tiny":
            O.hift_forward(w.vsd, w.vc, m, vkw["phase0"], vkw["noise"])
        else:
            O.bigvgan_forward(w.vsd, w.vc, m)

    run(2)                                             # warm-up: allocator, thread pool, first-touch
    times = []
    for _ in range(3):
        t1 = time.perf_counter()
        run(w.n_steps)
        times.append(time.perf_counter() - t1)
    med = sorted(times)[1]
    return {"value": round(w.S / med, 2), "unit": "mel-frames/s", "cores": nthreads, "kind": "port",
            "sample": f"1 utterance of the same workload (P=S={w.S}, {w.n_steps} steps + vocoder), torch fp32 oracle, "
                      f"2-step warm-up + median of 3 runs ({', '.join(f'{t:.1f}' for t in times)} s)"}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a, sys.argv[1:]))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if a.dry_run:
        return dry_run_rank(a, rank, world)
    import torch
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
    from seedvc_amd import _lib
    from seedvc_amd.pipeline import gather_audio

    w = Workload(a.model, a.batch, a.frames, a.diffusion_steps, a.lanes, dev, 1234 + rank, a.microbatch, a.vocoder_precision,
                 a.fused_min_rows)
    B, S = w.B, w.S

    def step():
        mel, wave = w.step()
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
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0                 # this rank's own work (its gathers included), before the barrier
    sync_all()
    dt = time.perf_counter() - t0
    ranks_info = None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ranks_info = rank_diagnostics(w, dt_local, a.steps, dev, world, B, a.dist_backend)
    frames = world * B * S * a.steps
    value = frames / dt

    out = {
        "metric": "mel_frames_per_sec", "value": round(value, 1), "unit": "mel-frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "rtf": round((dt / a.steps) / (B * S * w.hop / w.sr), 6),
        "realtime_factor_per_gpu": round(value / world / (w.sr / w.hop), 1),
        "config": {"workload": w.describe(),
                   "batch_per_gpu": B, "global_batch": B * world, "prompt_frames": w.P, "source_frames": S,
                   "diffusion_steps": w.n_steps, "cfg_rate": 0.7, "vocoder_precision": a.vocoder_precision,
                   "lanes_per_gpu": len(w.lanes.lanes),
                   "parallelism": f"utterance-sharded x{world}, audio gather on rank 0"},
    }

    if ranks_info is not None:
        out["ranks"] = ranks_info

    if rank == 0 and not a.no_roofline:
        out["roofline"] = roofline(w, value / world, f"{a.model}-b{a.batch}")

    if rank == 0 and world == 1 and not a.no_secondary and a.model == "tiny":
        out["secondary"] = secondary_lines(a, dev)

    if rank == 0 and world == 1 and not a.no_cpu_baseline:      # reported at N = 1 only
        out["cpu_baseline"] = cpu_baseline(w, host_cores())

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
