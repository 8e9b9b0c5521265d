"""Bandwidth record of the a16 seam kernel (`svc_anti_alias_act_fwd` = aa_act_rows_kernel, the drop-in for the reference's
only native code, alias_free_activation/cuda/anti_alias_activation_cuda.cu:43-246) against the HBM roofline.

Algorithmic bytes = 2 * B * C * L * sizeof(T) (every sample read once and written once); shapes are the
BigVGAN-22k stage shapes of a 16-utterance micro-batch at S = 430 frames: (16, 768, 1720), (16, 96, 27520),
(16, 24, 110080), fp32 and fp16.  Timed with HIP events on the launch stream (torch.cuda.Event on the current stream,
which is the stream the op is enqueued on).  Prints one JSON object; run under `rocprofv3 --kernel-trace --stats` to put
the kernel line beside it (tools/profile_round.sh does)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkgload
_pkgload.load_package()
import torch
from seedvc_amd import _lib, ops, weights

HBM_PEAK = 8000.0       # GB/s (MI355X_MICROARCH.md)
torch.set_grad_enabled(False)
filt = weights.make_tensor("x.filter", (1, 1, 12)).reshape(-1).cuda()
rows = []
for dtype, name in ((torch.float32, "f32"), (torch.float16, "f16")):
    for B, C, L in ((16, 768, 1720), (16, 96, 27520), (16, 24, 110080)):
        x = (torch.randn(B, C, L, device="cuda") * 2).to(dtype)
        alpha = torch.randn(C, device="cuda") * 0.4
        beta = torch.randn(C, device="cuda") * 0.4
        for _ in range(3):
            ops.anti_alias_activation_forward(x, filt, filt, alpha, beta)
        # timed loop: the C entry point itself on pre-marshalled pointers (the Python wrapper's host work per call is longer
        # than the kernel at the short shapes and would be timed as gaps between launches)
        y = torch.empty_like(x)
        args = (_lib.ptr(x), _lib.ptr(y), _lib.ptr(filt), _lib.ptr(filt), _lib.ptr(alpha), _lib.ptr(beta), B, C, L,
                {"f32": 0, "f16": 1}[name], _lib.stream_ptr())
        fn = _lib.lib().svc_anti_alias_act_fwd
        iters = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn(*args)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        nbytes = 2 * x.numel() * x.element_size()
        gbps = nbytes / (us * 1e-6) / 1e9
        rows.append({"dtype": name, "B": B, "C": C, "L": L, "alg_bytes": nbytes, "us_per_launch": round(us, 2),
                     "achieved_GBps": round(gbps, 1), "frac_of_hbm_peak": round(gbps / HBM_PEAK, 4)})
print(json.dumps({"kernel": "aa_act_rows_kernel (svc_anti_alias_act_fwd)", "bound": "hbm", "peak_GBps": HBM_PEAK,
                  "alg_bytes_formula": "2*B*C*L*sizeof(T)", "shapes": rows}))
