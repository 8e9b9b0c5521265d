"""Long-utterance conversion: host chunk loop vs device chunk loop (crossfade on the device,
vocoder of chunk k overlapped with the sampler of chunk k+1).  small+WaveNet + BigVGAN-22k, 25 steps, P = 430,
`--seconds` of source audio."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
_pkgload.load_package()
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import cases
from seedvc_amd import specs, weights
from seedvc_amd.cfm import CFM
from seedvc_amd.vocoder import BigVGAN
from seedvc_amd.pipeline import HotPath

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=70.0)
a = ap.parse_args()
torch.set_grad_enabled(False)
dev = "cuda:0"
cfg = specs.dit_config("small")
cfm = CFM(cfg, weights.make_state_dict(specs.dit_state_spec(cfg), seed=1234, prefix="dit.small."), dev)
vc = specs.bigvgan_config("22k")
voc = BigVGAN(vc, weights.make_state_dict(specs.bigvgan_state_spec(vc), seed=1234, prefix="bigvgan."), dev)
hop, P = 256, 430
S = int(a.seconds * 22050 / hop)
cond = cases.randn("lb.cond", 1, 1, S, cfg["Dc"]).to(dev)
pc = cases.randn("lb.pc", 1, 1, P, cfg["Dc"]).to(dev)
mel2 = cases.logmel("lb.mel2", 1, 1, cfg["C"], P).to(dev)
style = cases.randn("lb.style", 1, 1, cfg["style_dim"]).to(dev)
noise = lambda T: torch.randn(1, cfg["C"], T, device=dev, generator=torch.Generator(device=dev).manual_seed(T))
hp = HotPath(cfm, voc)
window = 22050 // hop * 30
for name, fn in (("host loop", hp.convert_long), ("device loop", hp.convert_long_device)):
    for _ in range(2):
        w = fn(cond, pc, mel2, style, 25, 0.7, hop, window, noise_fn=noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        w = fn(cond, pc, mel2, style, 25, 0.7, hop, window, noise_fn=noise)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name:12s}: {a.seconds:.0f} s of audio ({S} frames) in {dt * 1e3:7.1f} ms -> {a.seconds / dt:6.1f}x real time, "
          f"{w.shape[-1]} samples", flush=True)
