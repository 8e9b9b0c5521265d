"""Tuning harness (not the product path): tap-GEMM tile forms on the large-M DiT shapes, pseudo-random operands.
args = launch_wide form words (kgemm.hip): 0 default rule, 0x10/0x40 128x128 rings, 0xb0 256x128 four waves, 0x90 256x256
AGPR form; | 1 skips the ring refills, | 2 skips the epilogue (ablations)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkgload
_pkgload.load_package()
import torch
from seedvc_amd import _lib
L = _lib.lib()
torch.cuda.init()
shapes = [("w13 base", 55296, 4096, 768, 1), ("qkv base", 55296, 2304, 768, 3), ("w2 base", 55296, 768, 2048, 0),
          ("wo base", 55296, 768, 768, 0), ("skip base", 55296, 768, 1536, 0), ("w13 small", 55296, 3072, 512, 1),
          ("w2 small", 55296, 512, 1536, 0), ("wn small", 55296, 1024, 2560, 2)]
dbg = [int(x, 0) for x in (sys.argv[1:] or ["0"])]
for name, M, N, K, epi in shapes:
    row = []
    for d in dbg:
        ms = C.c_float()
        _lib.check(L.svc_op_gemm_bench(M, N, K, 0, epi, 20, d, C.byref(ms), None))
        row.append(f"{d:#x}: {ms.value * 1e3:7.1f} us {2.0 * M * N * K / ms.value / 1e9:7.1f} TF")
    print(f"{name:10s} M={M} N={N} K={K}  " + " | ".join(row), flush=True)
