"""Kernel-tuning harness for the small-M (single-utterance) tap-GEMM variants: times the B = 1 shapes of the small /
tiny / base models for each `launch_wide` small-M variant (debug bits 8..11, kgemm.hip).  Not part of the product path."""
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
shapes = [("wo   small", 1720, 512, 512, 0), ("w13  small", 1720, 3072, 512, 1), ("qkv  small", 1720, 1536, 512, 3),
          ("w2   small", 1720, 512, 1536, 0), ("skip small", 1720, 512, 1024, 0), ("wn in small", 900, 1024, 2560, 2),
          ("wn rs small", 900, 1024, 512, 0), ("wo    tiny", 1736, 384, 384, 0), ("w13   tiny", 1736, 2048, 384, 1),
          ("qkv   tiny", 1736, 1152, 384, 3), ("w2    tiny", 1736, 384, 1024, 0), ("wo    base", 1720, 768, 768, 0),
          ("w13   base", 1720, 4096, 768, 1), ("w2    base", 1720, 768, 2048, 0)]
variants = [int(x) for x in (sys.argv[1:] or range(0, 13))]
print("variant:      " + " ".join(f"{v:>6d}" for v in variants))
for name, M, N, K, epi in shapes:
    row = []
    for v in variants:
        ms = C.c_float()
        _lib.check(L.svc_op_gemm_bench(M, N, K, 0, epi, 50, ((v if v else 15) << 8) | (0x1000 if v else 0), C.byref(ms), None))
        row.append(f"{ms.value * 1e3:6.1f}")
    print(f"{name:12s}  " + " ".join(row) + f"   us  (M={M} N={N} K={K} epi={epi})", flush=True)
