"""ORACLE (test infrastructure only — never imported by the product path).

fp32 PyTorch-CPU restatements of the single ops the HIP kernels implement, following the arithmetic of the
third-party code the reference calls into:
  * RMSNorm        — HF modeling_qwen2_5_vl.py:65-79
  * rotary         — HF modeling_qwen2_5_vl.py:153-171 (rotate_half, apply_rotary_pos_emb_vision)
  * attention      — HF modeling_qwen2_5_vl.py:188-208 (eager_attention_forward: fp32 softmax, GQA repeat)
  * SwiGLU MLP     — HF modeling_qwen2_5_vl.py:84-96
  * CE             — HF ForCausalLMLoss (fp32 logits, ignore_index=-100)
"""
import torch
import torch.nn.functional as F


def linear_ref(a, w, bias=None, residual=None, act="none"):
    a, w = a.float(), w.float()
    y = a @ w.t()
    if bias is not None:
        y = y + bias.float()
    if act == "gelu":
        y = F.gelu(y)
    elif act == "relu":
        y = F.relu(y)
    elif act == "swiglu":
        # rows of w interleaved in blocks of 16: [16 gate | 16 up] -> out block of 16
        n = y.shape[1]
        yb = y.view(y.shape[0], n // 32, 2, 16)
        y = (F.silu(yb[:, :, 0]) * yb[:, :, 1]).reshape(y.shape[0], n // 2)
    if residual is not None:
        y = y + residual.float()
    return y


def rmsnorm_ref(x, w, eps):
    x = x.float()
    var = x.pow(2).mean(-1, keepdim=True)
    return w.float() * (x * torch.rsqrt(var + eps))


def layernorm_ref(x, w, b, eps):
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), None if b is None else b.float(), eps)


def rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def rope_ref(x, cos, sin):
    """x [T, H, D]; cos/sin [T, D]"""
    x = x.float()
    return x * cos[:, None, :].float() + rotate_half(x) * sin[:, None, :].float()


def attn_varlen_ref(q, k, v, cu_q, cu_k, scale, causal):
    """q [Tq,Hq,D], k/v [Tk,Hkv,D]; returns (out [Tq,Hq,D] fp32, lse [Hq,Tq])."""
    q, k, v = q.float(), k.float(), v.float()
    Hq, Hkv = q.shape[1], k.shape[1]
    rep = Hq // Hkv
    out = torch.zeros_like(q)
    lse = torch.zeros(Hq, q.shape[0])
    for s in range(len(cu_q) - 1):
        q0, q1, k0, k1 = int(cu_q[s]), int(cu_q[s + 1]), int(cu_k[s]), int(cu_k[s + 1])
        if q1 == q0:
            continue
        qq = q[q0:q1].transpose(0, 1)                      # [Hq, Lq, D]
        kk = k[k0:k1].transpose(0, 1).repeat_interleave(rep, 0)
        vv = v[k0:k1].transpose(0, 1).repeat_interleave(rep, 0)
        sc = qq @ kk.transpose(1, 2) * scale
        if causal:
            Lq, Lk = q1 - q0, k1 - k0
            i = torch.arange(Lq)[:, None]
            j = torch.arange(Lk)[None, :]
            sc = sc.masked_fill(j > i + (Lk - Lq), float("-inf"))
        lse[:, q0:q1] = torch.logsumexp(sc, -1)
        pr = torch.softmax(sc, -1)
        out[q0:q1] = (pr @ vv).transpose(0, 1)
    return out, lse


def ce_rows_ref(logits, labels):
    logits = logits.float()
    loss = F.cross_entropy(logits, labels, ignore_index=-100, reduction="none")
    return loss


def dropout_mask_ref(n: int, p: float, seed: int):
    """Keep mask (bool [n]) and scale of rga3_dropout_bf16: element e of 8-element group g keeps iff the 16-bit half (e & 1) of
    hash(seed, 4g + e // 2) is >= round(p * 65536); scale = 1 / (1 - thr / 65536) (fp32).  This RNG is the build's own (torch's Philox
    stream on the reference side is not reproducible across devices either); the oracle restates it for exact mask parity."""
    import numpy as np

    assert n % 8 == 0
    seed &= 0x7FFFFFFFFFFFFFFF
    thr = np.uint32(int(np.float32(p) * np.float32(65536.0) + np.float32(0.5)))
    ctr = np.arange(n // 2, dtype=np.uint64)             # one hash per element pair
    M = np.uint64(0xFFFFFFFF)
    x = ((ctr & M) ^ np.uint64(seed & 0xFFFFFFFF)).astype(np.uint32)
    y = ((ctr >> np.uint64(32)) ^ np.uint64(seed >> 32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        x = x * np.uint32(0xcc9e2d51); x = (x << np.uint32(15)) | (x >> np.uint32(17)); x = x * np.uint32(0x1b873593)
        y = y ^ x; y = (y << np.uint32(13)) | (y >> np.uint32(19)); y = y * np.uint32(5) + np.uint32(0xe6546b64)
        y ^= y >> np.uint32(16); y = y * np.uint32(0x85ebca6b); y ^= y >> np.uint32(13); y = y * np.uint32(0xc2b2ae35); y ^= y >> np.uint32(16)
    keep = np.empty(n, dtype=bool)
    keep[0::2] = (y & np.uint32(0xffff)) >= thr
    keep[1::2] = (y >> np.uint32(16)) >= thr
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(thr) / np.float32(65536.0))
    return keep, float(scale)


def quant_fp8_rows_ref(x):
    """(q float8_e4m3fn [rows, K], scale f32 [rows]) of rga3_quant_fp8_rows: scale = amax / 448 (1 for a zero row), q = e4m3(x / scale) (fp32 divide, RNE)."""
    xf = x.float()
    amax = xf.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    return (xf / scale[:, None]).to(torch.float8_e4m3fn), scale


def gemm_fp8_ref(qa, sa, qw, sw, bias=None, residual=None):
    """fp32 value of rga3_gemm_fp8 before the bf16 rounding(s): products of e4m3 values are exact in fp32, only the summation order differs."""
    y = (qa.float() @ qw.float().t()) * sa[:, None] * sw[None, :]
    if bias is not None:
        y = y + bias.float()
    y = y.to(torch.bfloat16).float()
    if residual is not None:
        y = (y + residual.float()).to(torch.bfloat16).float()
    return y
