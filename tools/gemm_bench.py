"""Kernel-tuning harness: times single tap-GEMM shapes of the DiT on the GPU (not part of the product path).
args = debug words: bit0 skip loads, bit1 skip epilogue, 0x10 / 0x20 / 0x80 tile variants (kgemm.hip launch_wide)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkgload
_pkgload.load_package()
import torch
from seedvc_amd import _lib

L = _lib.lib()
torch.cuda.init()
shapes = [("wo    tiny", 27648, 384, 384, 0, 0), ("w13   tiny", 27648, 2048, 384, 0, 1), ("qkv   tiny", 27648, 1152, 384, 0, 3),
          ("w2    tiny", 27648, 384, 1024, 0, 0), ("w13  small", 27648, 3072, 512, 0, 1), ("w2   small", 27648, 512, 1536, 0, 0),
          ("big 8k^3/8", 8192, 8192, 1024, 0, 0), ("f32 conv", 27520, 96, 1056, 1, 0)]
dbg = [int(x) for x in (sys.argv[1:] or ["0"])]
for name, M, N, K, dt, epi in shapes:
    row = []
    for d in dbg:
        ms = C.c_float()
        _lib.check(L.svc_op_gemm_bench(M, N, K, dt, epi, 20, d, C.byref(ms), None))
        row.append(f"dbg{d}: {ms.value * 1e3:8.1f} us {2.0 * M * N * K / ms.value / 1e9:8.1f} TF")
    print(f"{name:12s} M={M} N={N} K={K}  " + " | ".join(row), flush=True)

# calibration only: what the vendor library (hipBLASLt through torch.mm) reaches on the same shapes and clocks
if os.environ.get("GEMM_BENCH_VENDOR"):
    for name, M, N, K, dt, epi in shapes:
        a = torch.randn(M, K, device="cuda", dtype=torch.float16 if dt == 0 else torch.float32)
        w = torch.randn(N, K, device="cuda", dtype=a.dtype)
        for _ in range(3):
            torch.mm(a, w.t())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            torch.mm(a, w.t())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:12s} vendor mm (plain store): {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:8.1f} TF", flush=True)
