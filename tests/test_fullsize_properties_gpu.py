"""GPU, BASELINE.json full sizes: size-independent properties of the hot-path kernels (the oracle is too slow at 7B dims).
  * GEMM linearity and consistency across tilings at the 7B shapes;
  * attention: V = 1 gives output 1 (softmax rows sum to 1); causality (perturbing future tokens leaves the past unchanged, bitwise);
  * windowed ViT attention is invariant to the window order; RoPE preserves pair norms; gather/scatter round trip."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rnd(shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(dev)


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


@pytest.mark.parametrize("M,N,K", [(8192, 3840, 1280), (2112, 4608, 3584), (2112, 3584, 18944)])
def test_gemm_linearity_and_tilings(dev, M, N, K):
    from rga3.hip import ops

    a1, a2, w = rnd((M, K), dev, seed=1), rnd((M, K), dev, seed=2), rnd((N, K), dev, 0.02, seed=3)
    y1, y2 = ops.gemm(a1, w, out_dtype=torch.float32, tile=10), ops.gemm(a2, w, out_dtype=torch.float32, tile=10)
    asum = (a1.float() + a2.float())
    exact = asum.to(torch.bfloat16).float().equal(asum)  # sum may round; compare against the rounded-sum GEMM plus slack
    y12 = ops.gemm(asum.to(torch.bfloat16), w, out_dtype=torch.float32, tile=10)
    assert rel(y12, y1 + y2) < (1e-5 if exact else 6e-3)
    ref = ops.gemm(a1, w, tile=10)
    for tile in (11, 12, 3, 4, 20, 21, 22, 26, 27, 28, 31, 32):
        assert rel(ops.gemm(a1, w, tile=tile), ref) < 2e-3, tile   # same products, different summation split points only


def test_attention_rowsum_causality_window_invariance(dev):
    from rga3.hip import ops

    S, Hq, Hk, D = 2112, 28, 4, 128
    q, k = rnd((S, Hq, D), dev, seed=1), rnd((S, Hk, D), dev, seed=2)
    cu = torch.tensor([0, S], dtype=torch.int32, device=dev)
    ones = torch.ones((S, Hk, D), dtype=torch.bfloat16, device=dev)
    o = ops.attn_varlen(q, k, ones, cu, cu, S, D ** -0.5, causal=True)
    assert (o.float() - 1).abs().max().item() < 1e-2
    v = rnd((S, Hk, D), dev, seed=3)
    o1 = ops.attn_varlen(q, k, v, cu, cu, S, D ** -0.5, causal=True)
    k2, v2 = k.clone(), v.clone()
    k2[1500:] = rnd((S - 1500, Hk, D), dev, seed=4)
    v2[1500:] = rnd((S - 1500, Hk, D), dev, seed=5)
    o2 = ops.attn_varlen(q, k2, v2, cu, cu, S, D ** -0.5, causal=True)
    assert torch.equal(o1[:1500], o2[:1500])          # the past never sees the future: bit-exact
    # ViT windows: 128 windows x 64 tokens, 16 heads x 80; permuting whole windows permutes the output
    W, L, H, d = 128, 64, 16, 80
    qkv = rnd((W * L, 3, H, d), dev, seed=6)
    cw = (torch.arange(W + 1, dtype=torch.int32) * L).to(dev)
    a = ops.attn_varlen(qkv[:, 0], qkv[:, 1], qkv[:, 2], cw, cw, L, d ** -0.5)
    perm = torch.randperm(W, generator=torch.Generator().manual_seed(0)).to(dev)
    idx = (perm[:, None] * L + torch.arange(L, device=dev)[None]).reshape(-1)
    qp = qkv[idx].contiguous()
    b = ops.attn_varlen(qp[:, 0], qp[:, 1], qp[:, 2], cw, cw, L, d ** -0.5)
    assert torch.equal(b, a[idx])


def test_rope_norms_and_gather_roundtrip(dev):
    from rga3.hip import ops

    T, H, D = 2112, 32, 128
    x = rnd((T, H, D), dev, seed=1)
    ang = torch.rand(T, D // 2, generator=torch.Generator().manual_seed(1)) * 100.0
    emb = torch.cat([ang, ang], -1)
    cos, sin = emb.cos().to(dev), emb.sin().to(dev)
    y = x.clone()
    ops.rope_(y, cos, sin, 0, H)
    n0 = x.float()[..., : D // 2] ** 2 + x.float()[..., D // 2:] ** 2
    n1 = y.float()[..., : D // 2] ** 2 + y.float()[..., D // 2:] ** 2
    assert ((n1 - n0).abs() / (n0 + 1e-3)).max().item() < 3e-2
    ops.rope_(y, cos, (-sin).contiguous(), 0, H)      # inverse rotation returns to the start (up to bf16 rounding twice)
    assert rel(y, x) < 8e-3
    tab = rnd((8192, 1280), dev, seed=2)
    perm = torch.randperm(2048, generator=torch.Generator().manual_seed(2)).to(dev)
    g = ops.gather_rows(tab, perm, rows_per_idx=4)
    back = torch.empty_like(tab)
    ops.scatter_rows_(back, perm, g, rows_per_idx=4)
    assert torch.equal(back, tab)
