"""Which kernels the vendor library picks for the forward's GEMM shapes (reference point only; run under rocprofv3 --kernel-trace --stats): the Tensile kernel names
spell out macro tile, depth, LDS buffering and stream-K / split-K -- what a hand-written kernel is up against on each shape."""
import torch
SHAPES = [(2112, 3584, 3584), (2112, 4608, 3584), (8192, 1280, 1280), (8192, 3840, 1280), (8192, 6912, 1280), (8192, 1280, 3456), (2112, 37888, 3584), (2112, 3584, 18944),
          (32768, 2304, 576), (32768, 576, 2304), (32768, 1728, 576), (32768, 576, 576), (8192, 8192, 8192)]
for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
print("ok")
