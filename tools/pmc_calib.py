"""Known-traffic calibration for the FETCH_SIZE / WRITE_SIZE counters: one elementwise pass over 1 GiB (reads 1 GiB,
writes 1 GiB, far beyond the 256 MiB Infinity Cache) run under the same rocprofv3 --pmc command as the bench passes.
tools/pmc_traffic.py reads the biggest dispatch of this run and reports known bytes / counter."""
import torch

n = 1 << 28                                    # 2^28 floats = 1 GiB
x = torch.empty(n, device="cuda").uniform_()
torch.cuda.synchronize()
y = x * 1.0001                                 # vectorized_elementwise_kernel: 16-byte loads and stores, each byte once
torch.cuda.synchronize()
print("calib bytes", n * 4)
