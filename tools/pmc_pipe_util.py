#!/usr/bin/env python3
"""Matrix-pipe / issue utilisation of the kernels that SHIP, from rocprofv3 --pmc passes.

  run        python3 tools/pmc_pipe_util.py run          launches every target kernel a few times at its bench shape (the program a --pmc pass profiles)
  summarise  python3 tools/pmc_pipe_util.py sum <dir> <out.json>    reads every *counter_collection.csv under <dir> (one sub-directory per pass)

Targets (kernel name needle -> the call that launches it at the shape of the headline step / configs[1] forward / configs[3] stream):
  gemm_nt_sk_kernel<2, false, 4, false / true>   LLM gate-up with the SwiGLU epilogue, 2112 x 37888 x 3584 (tile 22 / tile 27: ragged last tile row on workgroups of its own)
  gemm_nt_w4_kernel<0, false>   the four-wave kernel at the LM head's shape, 2112 x 152064 x 3584 (tile 28)
  gemm_nt_sk_kernel<0, false, 3>   the three-phase 192-row loop at the decoder's q|k|v shape, 2112 x 4608 x 3584 (tile 31)
  hiera_mlp_kernel<HmCfg<144 / 288   fused Hiera MLP of stage 1 / stage 2 on 8 frames
  gemm_nt_sk_kernel<0,       LLM down projection + residual, 2112 x 3584 x 18944 (tile 22)
  gemm_nt_kernel<128, 192    Hiera-L stage-3 qkv, 32768 x 1728 x 576 (8 frames; tile 5)
  gemm_nt_kernel<128, 256    LLM o-proj + residual, 2112 x 3584 x 3584 (tile 3)
  gemm_nt_pp_kernel<0, false, 2, true>   Hiera stage-3 fc2 + residual + the LayerNorm partial sums of the rows written (round 6), 65536 x 576 x 2304 on 256 x 192 tiles (tile 23)
  gemm_nt_pp_kernel<1, false, 1, false>   Hiera stage-3 fc1 + GELU, LayerNorm folded (statistics from the producer's partial sums), 65536 x 2304 x 576 (tile 20)
  attn_fwd_kernel<128        causal decoder attention, S = 2112, 28 / 4 heads x 128
  attn_bwd_dkv / attn_bwd_dq its backward
  attn_win_kernel<96, 8      Hiera stage-3 windows: 128 windows x 256 tokens, 8 heads x 72 (round 6: 32 query rows per wave, one workgroup per window and head)
  memattn_cross_kernel       SAM2 memory cross-attention, 4096 queries x 28 736 keys

Derivations (MI355X_MICROARCH.md, rocprofv3 PMC slots + cycle constants):
  SQ_VALU_MFMA_BUSY_CYCLES  = matrix-pipe cycles summed over the chip's 1024 SIMDs (16 per 16x16x32 bf16 MFMA, 32 per 32x32x16)
  GRBM_GUI_ACTIVE           = busy cycles summed over the 8 XCDs  ->  kernel cycles = GRBM_GUI_ACTIVE / 8
  mfma_busy                 = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024)
  SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY count quad-cycles per wave: their RATIOS say where a resident wave spends its time
  (parked at s_waitcnt / barrier, issue-stalled, issuing); SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = share of LDS-array cycles lost to conflicts.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TARGETS = ["gemm_nt_sk_kernel<2, false, 4, true>", "gemm_nt_sk_kernel<2, false, 4, false>", "gemm_nt_sk_kernel<0, false, 4, false>", "gemm_nt_sk_kernel<0, false, 3, false>", "gemm_nt_w4_kernel<0, false>", "gemm_nt_kernel<128, 192", "gemm_nt_kernel<128, 256", "gemm_nt_pp_kernel<0, false, 2, true>", "gemm_nt_pp_kernel<1, false, 1, false>", "attn_fwd_kernel<128", "attn_causal32_kernel",
           "attn_bwd_dkv_kernel", "attn_bwd_dq_kernel", "attn_win_kernel<96, 8", "memattn_cross_kernel", "hiera_mlp_kernel<rga3::HmCfg<144", "hiera_mlp_kernel<rga3::HmCfg<288"]


def run():
    import torch
    sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
    from rga3.hip import ops
    dev = "cuda"
    torch.manual_seed(0)
    bf = torch.bfloat16
    R = 4

    def rn(*s, scale=1.0):
        return (torch.randn(*s, device=dev) * scale).to(bf)

    # ---- GEMMs (weights rotated so every launch streams them from HBM, as inside the model)
    x = rn(2112, 3584)
    wgu = [rn(37888, 3584, scale=0.02) for _ in range(2)]
    for i in range(R):
        ops.gemm(x, wgu[i % 2], act="swiglu", tile=22)
    for i in range(R):   # the ragged last tile row on workgroups of its own (round 5: what the forward runs)
        ops.gemm(x, wgu[i % 2], act="swiglu", tile=27)
    del wgu
    wlm = rn(152064, 3584, scale=0.02)   # the four-wave kernel at the LM head's shape (round 5; tile 28)
    for i in range(2):
        ops.gemm(x, wlm, tile=28)
    del wlm
    h = rn(2112, 18944)
    wd = [rn(3584, 18944, scale=0.02) for _ in range(2)]
    for i in range(R):
        ops.gemm(h, wd[i % 2], residual=x, tile=22)
    del wd, h
    # the three-phase 192-row loop at the decoder's q|k|v shape (round 5; tile 31: 11 x 18 tiles, one round)
    wqkv = [rn(4608, 3584, scale=0.02) for _ in range(4)]
    for i in range(2 * R):
        ops.gemm(x, wqkv[i % 4], bias=rn(4608), tile=31)
    del wqkv
    wo = [rn(3584, 3584, scale=0.02) for _ in range(8)]
    for i in range(2 * R):
        ops.gemm(x, wo[i % 8], residual=x, tile=3)
    xs = rn(32768, 576)
    wq = [rn(1728, 576, scale=0.04) for _ in range(4)]
    for i in range(R):
        ops.gemm(xs, wq[i % 4], bias=rn(1728), tile=5)
    # ---- causal decoder attention forward + backward
    S, Hq, Hkv, D = 2112, 28, 4, 128
    qkv = rn(S, Hq + 2 * Hkv, D)
    q, k, v = qkv[:, :Hq], qkv[:, Hq:Hq + Hkv], qkv[:, Hq + Hkv:]
    cu = torch.tensor([0, S], dtype=torch.int32, device=dev)
    for i in range(R):
        o, lse = ops.attn_varlen(q, k, v, cu, cu, S, D ** -0.5, causal=True, return_lse=True)
    do = rn(S, Hq, D)
    for i in range(R):
        ops.attn_varlen_bwd(q, k, v, o, do, lse, cu, cu, S, S, D ** -0.5, True)
    # ---- Hiera stage-3 windows (8 frames: 128 windows of 256 tokens, 8 heads x 72)
    T, Hh, Dh = 32768, 8, 72
    qkvh = rn(T, 3 * Hh, Dh)
    cuw = torch.arange(0, T + 1, 256, dtype=torch.int32, device=dev)
    for i in range(R):
        ops.attn_varlen(qkvh[:, :Hh], qkvh[:, Hh:2 * Hh], qkvh[:, 2 * Hh:], cuw, cuw, 256, Dh ** -0.5, max_k=256)
    # ---- Hiera stage-3 fc2 + residual on the 256 x 192 ping-pong tiles (tile 23) and fc1 + GELU (LayerNorm folded) on the 256 x 256 ping-pong tiles (16 frames)
    xh, xr = rn(65536, 2304), rn(65536, 576)
    w2h = [rn(576, 2304, scale=0.04) for _ in range(2)]
    for i in range(R):
        xo, parts = ops.gemm_lnsum(xh, w2h[i % 2], rn(576), residual=xr, tile=23)
    wf, colc, bfold = ops.fold_layernorm(rn(2304, 576, scale=0.04), rn(2304), rn(576) + 1, rn(576, scale=0.1))
    for i in range(R):
        ops.gemm_ln(xo, ops.LnSums(parts, 1e-6), wf, colc, bfold, act="gelu", tile=20)
    del xh, w2h
    # ---- fused Hiera MLP, stage 1 (8 frames: 524 288 rows x 144) and stage 2 (131 072 rows x 288)
    for C, rows in ((144, 8 * 65536), (288, 8 * 16384)):
        xm = rn(rows, C)
        wf2, colc2, bf2 = ops.fold_layernorm(rn(4 * C, C, scale=0.05), rn(4 * C), rn(C) + 1, rn(C, scale=0.1))
        w2m, b2m = rn(C, 4 * C, scale=0.05), rn(C)
        for i in range(R):
            ops.hiera_mlp(xm, wf2, colc2, bf2, w2m, b2m, 1e-6)
        del xm
    # ---- SAM2 memory cross-attention at the full bank
    mq, mk, mm = rn(4096, 256), rn(28736, 256), rn(28736, 64)
    for i in range(R):
        ops.memattn_cross(mq, mk, mm, 256 ** -0.5, partials=True)
    torch.cuda.synchronize()
    print("ok")


def summarise(d, out):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                for t in TARGETS:
                    if t in name:
                        acc[t][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = {}
    for t, cs in acc.items():
        m = {k: sum(v) / len(v) for k, v in cs.items()}
        e = {"launches": {k: len(v) for k, v in cs.items()}, "counters_mean_per_launch": {k: round(v, 1) for k, v in m.items()}}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0
            e["kernel_cycles"] = round(cyc, 1)
            e["mfma_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), 4)
        if m.get("SQ_WAVE_CYCLES"):
            w = m["SQ_WAVE_CYCLES"]
            for k, lab in (("SQ_WAIT_ANY", "wave_parked"), ("SQ_WAIT_INST_ANY", "wave_issue_stalled"), ("SQ_ACTIVE_INST_ANY", "wave_issuing"),
                           ("SQ_ACTIVE_INST_VALU", "wave_valu"), ("SQ_ACTIVE_INST_LDS", "wave_lds_issue"), ("SQ_WAIT_INST_LDS", "wave_lds_stalled")):
                if k in m:
                    e[lab] = round(m[k] / w, 4)
        if m.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_share"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4)
            if m.get("GRBM_GUI_ACTIVE"):
                e["lds_array_busy"] = round(m["SQ_LDS_IDX_ACTIVE"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 256.0), 4)
        res[t] = e
    res["_derivation"] = ("mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); wave_* = share of SQ_WAVE_CYCLES; lds_array_busy = SQ_LDS_IDX_ACTIVE / "
                          "(kernel cycles x 256 CUs); separate rocprofv3 --pmc passes per counter set, program directly after '--'")
    sys.path.insert(0, os.path.join(ROOT, "rga3-release_amd"))
    from rga3.utils.fingerprint import tree_fingerprint
    res["_tree"] = tree_fingerprint()     # bench.py quotes this file only while the running tree has the same fingerprint
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk in ("mfma_busy", "wave_parked", "wave_issue_stalled", "wave_issuing", "lds_array_busy", "lds_conflict_share")}
                      for k, v in res.items() if isinstance(v, dict)}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        summarise(sys.argv[2], sys.argv[3])
