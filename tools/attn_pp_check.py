import os, sys, torch
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
from rga3.hip import ops
torch.manual_seed(0)
def case(lens_q, lens_k, Hq, Hkv, D, causal):
    Tq, Tk = sum(lens_q), sum(lens_k)
    q = torch.randn(Tq, Hq, D, device="cuda").to(torch.bfloat16)
    k = torch.randn(Tk, Hkv, D, device="cuda").to(torch.bfloat16)
    v = torch.randn(Tk, Hkv, D, device="cuda").to(torch.bfloat16)
    cq = torch.tensor([0] + list(torch.tensor(lens_q).cumsum(0)), dtype=torch.int32, device="cuda")
    ck = torch.tensor([0] + list(torch.tensor(lens_k).cumsum(0)), dtype=torch.int32, device="cuda")
    o0, l0 = ops.attn_varlen(q, k, v, cq, ck, max(lens_q), D ** -0.5, causal, return_lse=True, impl=0)
    for rep in range(3):
        o1, l1 = ops.attn_varlen(q, k, v, cq, ck, max(lens_q), D ** -0.5, causal, return_lse=True, impl=4)
        ok = torch.equal(o0, o1) and torch.equal(l0, l1)
        if not ok:
            d = (o0.float() - o1.float()).abs()
            print("MISMATCH", lens_q, lens_k, Hq, Hkv, D, causal, "max", float(d.max()), "frac", float((d > 0).float().mean()), "nan", int(torch.isnan(o1.float()).sum()))
            return False
    print("ok", lens_q, lens_k, Hq, Hkv, D, causal)
    return True
allok = True
for D in (128, 64):
    for causal in (False, True):
        allok &= case([2112], [2112], 28, 4, D, causal)
        allok &= case([200, 64, 1000, 129], [200, 64, 1000, 129], 4, 2, D, causal)
        allok &= case([65], [65], 2, 2, D, causal)
        allok &= case([128, 513], [300, 1100], 6, 3, D, causal)     # Lk != Lq (causal shift)
        allok &= case([4160], [4160], 8, 4, D, causal)
allok &= case([300], [300], 4, 4, 120, True)   # D < DP
print("ALL OK" if allok else "FAILED")
