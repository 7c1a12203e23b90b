"""A/B of the fused stage-1 MLP (csrc/hiera_mlp.hip) inside the SAM2-L image encoder: ms per 8-frame chunk with rga3.model.sam2._MLP_FUSE on / off, interleaved, same box.
python tools/ab_hiera_mlp.py [iters]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
import rga3.model.sam2 as S2
from rga3.hip import ops
it = int(sys.argv[1]) if len(sys.argv) > 1 else 5
torch.manual_seed(1)
m = S2.SAM2().to(torch.bfloat16).cuda().eval()
with torch.no_grad():
    for n, p_ in m.named_parameters():
        if p_.dim() >= 2: p_.normal_(0, 0.02)
    x = torch.randn(8, 3, 1024, 1024, device="cuda").to(torch.bfloat16)
    def run(flag):
        S2._MLP_FUSE = flag
        for _ in range(2): m.sam2_model.forward_image(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(it): m.sam2_model.forward_image(x)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / it * 1e3
    def run_dims(dims):
        S2._MLP_FUSE, S2._MLP_FUSE_DIMS = bool(dims), tuple(dims)
        for _ in range(2): m.sam2_model.forward_image(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(it): m.sam2_model.forward_image(x)
        torch.cuda.synchronize()
        S2._MLP_FUSE, S2._MLP_FUSE_DIMS = True, (144, 288)
        return (time.perf_counter() - t0) / it * 1e3
    for rep in range(3):
        print(f"encoder per 8 frames: stage 1 + 2 fused {run_dims((144, 288)):.2f} ms   stage 1 only {run_dims((144,)):.2f} ms   unfused {run_dims(()):.2f} ms", flush=True)
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 10 * 1e3
    # the kernel alone on one 8-frame activation of each stage
    for bi, C, rows in ((0, 144, 8 * 65536), (3, 288, 8 * 16384)):
        b = m.sam2_model.image_encoder.trunk.blocks[bi]
        xx = torch.randn(rows, C, device="cuda").to(torch.bfloat16)
        l1 = b.mlp.layers[1]
        wf, colc, biasf = b._folded("fc1")
        tf = t(lambda: ops.hiera_mlp(xx, wf, colc, biasf, l1.weight, l1.bias, 1e-6))
        tu = t(lambda: l1(ops.gemm_ln(xx, ops.layernorm_stats(xx, 1e-6), wf, colc, biasf, act="gelu"), residual=xx))
        fl = 2 * 2 * rows * C * 4 * C
        print(f"kernel alone, C = {C}, 8 frames ({rows} rows): fused {tf:.3f} ms = {fl / tf / 1e9:.0f} TFLOP/s   unfused (stats + fc1 + fc2) {tu:.3f} ms = {fl / tu / 1e9:.0f} TFLOP/s")
