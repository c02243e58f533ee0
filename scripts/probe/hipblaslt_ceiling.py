"""What the vendor GEMM reaches on the f16 pipe at this project's matrix-bound shapes (a practical ceiling for the split-f16 loop, whose
three products per fp32 product make a launch of K an f16 GEMM of 3 K): torch.matmul (hipBLASLt / rocBLAS) in fp16 with fp32 accumulate.
usage: python scripts/probe/hipblaslt_ceiling.py"""
import torch
shapes = [(50176, 2304, 256), (50176, 1024, 256), (12544, 2048, 512), (12544, 4608, 512), (200704, 1152, 128), (802816, 576, 64),
          (8192, 8192, 8192)]
for (M, K, N) in shapes:
    for mult in (1, 3):
        a = torch.randn(M, K * mult, device="cuda", dtype=torch.float16)
        b = torch.randn(K * mult, N, device="cuda", dtype=torch.float16)
        for _ in range(3): torch.matmul(a, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): torch.matmul(a, b)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        tf = 2.0 * M * K * mult * N / us / 1e6
        print(f"M={M:7d} K={K * mult:6d} N={N:5d}  {us:8.1f} us  {tf:7.1f} TFLOP/s f16  ({tf / 2500:.2f} of 2.5 PFLOP/s)" + (f"   = {tf / 3:6.1f} TFLOP/s fp32-equivalent at three products" if mult == 3 else ""), flush=True)
        del a, b
