"""v2 AR decode-step timing (config 5 of BASELINE.json: 320-token prefill, 256 one-token steps): tokens/s and the
achieved fraction of the HBM roofline (weights streamed once per token + valid KV prefix)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import _pkgload
_pkgload.load_package()
import torch
from seedvc_amd import specs, weights
from seedvc_amd.ar import ARModel

torch.set_grad_enabled(False)
c = specs.ar_config()
sd = weights.make_state_dict(specs.ar_state_spec(c), seed=7, prefix="ar.")
ar = ARModel(c, sd, "cuda:0")
ar.setup_caches()
n_prefill, n_steps = 320, 256
x = torch.randn(1, n_prefill, c["dim"], device="cuda")
t0 = time.perf_counter()
ar.forward_generate(x, torch.arange(n_prefill), torch.arange(n_prefill))
torch.cuda.synchronize()
t_prefill = time.perf_counter() - t0
xs = torch.randn(n_steps, c["dim"], device="cuda")
res = {}
for mode in ("eager", "graph"):
    ar.setup_caches()
    ar.forward_generate(x, torch.arange(n_prefill), torch.arange(n_prefill))
    pos = n_prefill
    if mode == "graph":
        ar.decode_step(xs[0], pos, pos)        # capture + first step
        pos += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(1, n_steps):
        if mode == "graph":
            ar.decode_step(xs[s])
        else:
            ar.forward_generate(xs[s].reshape(1, 1, -1), torch.tensor([pos]), torch.tensor([pos]))
            pos += 1
    torch.cuda.synchronize()
    res[mode] = (time.perf_counter() - t0) / (n_steps - 1)
n_params = sum(v.numel() for k, v in sd.items() if "layers" in k or k.endswith("output.weight"))
bytes_tok = n_params * 2 + 2 * c["n_layer"] * c["n_local_heads"] * (n_prefill + n_steps // 2) * 64 * 4
out = {"prefill_ms": round(t_prefill * 1e3, 2), "eager_us_per_token": round(res["eager"] * 1e6, 1),
       "graph_us_per_token": round(res["graph"] * 1e6, 1), "tokens_per_s_graph": round(1.0 / res["graph"], 1),
       "alg_bytes_per_token": bytes_tok, "achieved_GBps": round(bytes_tok / res["graph"] / 1e9, 1),
       "hbm_peak_GBps": 8000, "frac": round(bytes_tok / res["graph"] / 8e12, 4)}
# whole generate loop on the device (8f row 4): prefill + n_steps tokens incl. embedding lookup and sampling
text = torch.randn(1, 120, c["dim"], device="cuda")
target = torch.randint(0, c["vocab_size"] - 1, (1, 200), device="cuda")
noise = torch.empty(n_steps, c["vocab_size"], device="cuda").exponential_(1)
noise[:, c["vocab_size"] - 1] = 1e30        # the EOS token never wins the exponential race: every run generates n_steps tokens
for _ in range(2):
    toks = ar.generate(text, target, exp_noise=noise, max_new=n_steps, check_every=16)
torch.cuda.synchronize()
t0 = time.perf_counter()
toks = ar.generate(text, target, exp_noise=noise, max_new=n_steps, check_every=16)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out["generate_tokens"] = int(toks.shape[-1])
out["generate_ms"] = round(dt * 1e3, 2)
out["generate_tokens_per_s"] = round(toks.shape[-1] / dt, 1)
print(json.dumps(out))
