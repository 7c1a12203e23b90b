"""SAM2-L image encoder (Hiera-L + FPN) alone: ms per 8-frame chunk.  python tools/sam2_encoder_probe.py [iters]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.model.sam2 import SAM2
it = int(sys.argv[1]) if len(sys.argv) > 1 else 5
torch.manual_seed(1)
m = SAM2().to(torch.bfloat16).cuda().eval()
with torch.no_grad():
    for n, p_ in m.named_parameters():
        if p_.dim() >= 2: p_.normal_(0, 0.02)
    NF = int(os.environ.get("NF", "8"))
    x = torch.randn(NF, 3, 1024, 1024, device="cuda").to(torch.bfloat16)
    for _ in range(2): m.sam2_model.forward_image(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): m.sam2_model.forward_image(x)
    torch.cuda.synchronize()
    print(f"{(time.perf_counter()-t0)/it*1e3:.1f} ms per {NF} frames")
    # A/B in one process (boards differ by +-2 %): module switches named in AB="_LN_SUMS,_MLP_FUSE" are turned off one at a time, interleaved with the default
    import rga3.model.sam2 as S2
    def timed():
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(it): m.sam2_model.forward_image(x)
        torch.cuda.synchronize(); return (time.perf_counter() - t1) / it * 1e3
    for name in [n for n in os.environ.get("AB", "").split(",") if n]:
        res = {"on": [], "off": []}
        for _ in range(3):
            res["on"].append(timed())
            setattr(S2, name, False); m.sam2_model.forward_image(x); res["off"].append(timed()); setattr(S2, name, True); m.sam2_model.forward_image(x)
        print(f"A/B {name}: on {min(res['on']):.2f} ms  off {min(res['off']):.2f} ms   (min of 3 x {it}, {NF} frames)")
from rga3.hip import tuner
if os.environ.get("RGA3_TUNE_SAVE"):      # decisions of this run, so that a profiled run (RGA3_TUNE_LOAD) holds no trial launches
    tuner.save(os.environ["RGA3_TUNE_SAVE"])
for k, v in tuner.timings().items():
    Mb, N, K = k[0], k[1], k[2]
    best = tuner.table().get(k)
    print(k[:4], "best", best, {t: round(2.0 * Mb * 256 * N * K / ms / 1e9) for t, ms in v.items()})
