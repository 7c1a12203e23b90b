"""GPU parity of the SAM2-side HIP kernels (through the C ABI) against torch fp32 references of the ops the reference calls
(F.conv2d / F.max_pool2d / F.interpolate / conv_transpose2d / complex RoPE / BCE+dice sums)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import sam2 as S

pytestmark = pytest.mark.gpu


def rnd(shape, dev, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(dev)


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def test_gemm_ktail_and_colscale(dev):
    from rga3.hip import ops

    for K in (8, 24, 144, 152, 200):
        a, w = rnd((70, K), dev, seed=K), rnd((40, K), dev, 0.1, seed=K + 1)
        assert rel(ops.gemm(a, w), a.float().cpu() @ w.float().cpu().t()) < 6e-3, K
    a, w, b = rnd((65, 64), dev, seed=1), rnd((48, 64), dev, 0.1, seed=2), rnd((48,), dev, 0.1, seed=3)
    g, r = rnd((48,), dev, 1.0, seed=4), rnd((65, 48), dev, seed=5)
    ref = r.float().cpu() + g.float().cpu() * (a.float().cpu() @ w.float().cpu().t() + b.float().cpu())
    assert rel(ops.gemm(a, w, b, residual=r, colscale=g), ref) < 8e-3


def test_patch_embed_im2col(dev):
    from rga3.hip import ops

    img = rnd((2, 3, 64, 64), dev, seed=1)
    w, b = rnd((16, 3, 7, 7), dev, 0.1, seed=2), rnd((16,), dev, 0.1, seed=3)
    cols, (Ho, Wo) = ops.im2col(img, 7, 4, 3)
    wl = ops.pad_cols(w.reshape(16, -1).contiguous(), cols.shape[1])
    out = ops.gemm(cols, wl, b).view(2, Ho, Wo, 16)
    ref = F.conv2d(img.float().cpu(), w.float().cpu(), b.float().cpu(), stride=4, padding=3).permute(0, 2, 3, 1)
    assert rel(out, ref) < 6e-3


def test_maxpool_win_and_upsample_add(dev):
    from rga3.hip import ops
    from rga3.model.sam2 import relayout

    Fn, H, W, C, w = 2, 16, 16, 24, 8
    x = rnd((Fn * H * W, 3 * C), dev, seed=1)
    xw = relayout(x, Fn, H, W, 0, w)
    y = ops.maxpool2x2_win(xw[:, C:2 * C], Fn * (H // w) * (W // w), w)          # strided slice read
    y = relayout(y, Fn, H // 2, W // 2, w // 2, 0)
    ref = F.max_pool2d(x[:, C:2 * C].float().cpu().view(Fn, H, W, C).permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).reshape(-1, C)
    assert torch.equal(y.float().cpu(), ref)
    a, b = rnd((Fn * H * W, C), dev, seed=2), rnd((Fn * H * W // 4, C), dev, seed=3)
    up = F.interpolate(b.float().cpu().view(Fn, H // 2, W // 2, C).permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest").permute(0, 2, 3, 1).reshape(-1, C)
    assert rel(ops.upsample2x_add(a, b, Fn, H, W), a.float().cpu() + up) < 4e-3
    pe = rnd((H * W, C), dev, seed=4)
    assert rel(ops.add_bcast(a, pe, 0.1), a.float().cpu() + 0.1 * pe.float().cpu().repeat(Fn, 1)) < 4e-3


@pytest.mark.parametrize("size", [(128, 128), (37, 53), (9, 200)])
def test_bilinear(dev, size):
    from rga3.hip import ops

    x = torch.randn(5, 32, 32)
    ref = F.interpolate(x[None], size=size, mode="bilinear", align_corners=False)[0]
    assert (ops.bilinear(x.to(dev), size).cpu() - ref).abs().max().item() < 1e-5
    idx = torch.tensor([3, 0, 3], dtype=torch.int32)
    out = ops.bilinear(x.to(dev), size, idx.to(dev)).cpu()
    assert (out - ref[idx.long()]).abs().max().item() < 1e-5
    xb = x.to(torch.bfloat16)
    assert (ops.bilinear(xb.to(dev), size).cpu() - F.interpolate(xb.float()[None], size=size, mode="bilinear", align_corners=False)[0]).abs().max().item() < 1e-5


def test_conv3x3s2_dwconv_pixel_shuffle(dev):
    from rga3.hip import ops

    Fn, H, W = 2, 16, 16
    for cin in (1, 4, 16, 64):
        x = rnd((Fn * H * W, cin), dev, seed=cin)
        w, b = rnd((cin * 4, cin, 3, 3), dev, 0.2, seed=cin + 1), rnd((cin * 4,), dev, 0.1, seed=cin + 2)
        ref = F.conv2d(x.float().cpu().view(Fn, H, W, cin).permute(0, 3, 1, 2), w.float().cpu(), b.float().cpu(), stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, cin * 4)
        assert rel(ops.conv3x3s2(x, w, b, Fn, H, W), ref) < 6e-3, cin
    m = torch.randn(Fn, H, W) * 4
    w, b = rnd((4, 1, 3, 3), dev, 0.2, seed=9), rnd((4,), dev, 0.1, seed=10)
    mm = (torch.sigmoid(m) * 20 - 10).to(torch.bfloat16).float()
    ref = F.conv2d(mm[:, None], w.float().cpu(), b.float().cpu(), stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, 4)
    assert rel(ops.conv3x3s2(m.to(dev), w, b, Fn, H, W, 20.0, -10.0), ref) < 6e-3
    C = 32
    x, w, b = rnd((Fn * H * W, C), dev, seed=11), rnd((C, 1, 7, 7), dev, 0.1, seed=12), rnd((C,), dev, 0.1, seed=13)
    ref = F.conv2d(x.float().cpu().view(Fn, H, W, C).permute(0, 3, 1, 2), w.float().cpu(), b.float().cpu(), padding=3, groups=C).permute(0, 2, 3, 1).reshape(-1, C)
    assert rel(ops.dwconv7x7(x, w, b, Fn, H, W), ref) < 6e-3
    cin, co = 32, 16
    x, wt, b = rnd((Fn * H * W, cin), dev, seed=14), rnd((cin, co, 2, 2), dev, 0.1, seed=15), rnd((co,), dev, 0.1, seed=16)
    add = rnd((Fn * 4 * H * W, co), dev, seed=17)
    g = ops.gemm(x, wt.permute(2, 3, 1, 0).reshape(-1, cin).contiguous())
    out = ops.pixel_shuffle2x(g, b, add, Fn, H, W)
    ref = F.conv_transpose2d(x.float().cpu().view(Fn, H, W, cin).permute(0, 3, 1, 2), wt.float().cpu(), b.float().cpu(), stride=2).permute(0, 2, 3, 1).reshape(-1, co) + add.float().cpu()
    assert rel(out, ref) < 8e-3
    assert rel(ops.pixel_shuffle2x(g, b, add, Fn, H, W, act="gelu"), F.gelu(ref)) < 1e-2


def test_rope_axial_and_layernorm_gelu(dev):
    from rga3.hip import ops

    nq, nk, C = 16, 40, 64
    cos, sin = S.compute_axial_cis(C, 4, 4)
    q, k = rnd((nq, C), dev, seed=1), rnd((nk, C), dev, seed=2)
    rq, rk = S.apply_rotary_enc(q.float().cpu()[None, None], k.float().cpu()[None, None][:, :, :32], cos, sin, repeat_freqs_k=True)
    ops.rope_axial_(q, cos.contiguous().to(dev), sin.contiguous().to(dev), nq)
    k_ref = torch.cat([rk[0, 0], k.float().cpu()[32:]], 0)
    ops.rope_axial_(k, cos.contiguous().to(dev), sin.contiguous().to(dev), 32)
    assert rel(q, rq[0, 0]) < 4e-3 and rel(k, k_ref) < 4e-3
    x, w, b = rnd((33, 64), dev, 2.0, seed=3), rnd((64,), dev, seed=4), rnd((64,), dev, seed=5)
    ref = F.gelu(F.layer_norm(x.float().cpu(), (64,), w.float().cpu(), b.float().cpu(), 1e-6))
    assert rel(ops.layernorm(x, w, b, 1e-6, act="gelu"), ref) < 8e-3


def test_bce_dice_sums(dev):
    from rga3.hip import ops

    x, t = torch.randn(3, 40, 50) * 3, (torch.randn(3, 40, 50) > 0.2).float()
    out = ops.bce_dice_sums(x.to(dev), t.to(dev)).cpu()
    bce = F.binary_cross_entropy_with_logits(x, t, reduction="none").flatten(1).sum(1)
    p = torch.sigmoid(x)
    ref = torch.stack([bce, (p * t).flatten(1).sum(1), p.flatten(1).sum(1), t.flatten(1).sum(1)], 1)
    assert ((out - ref).abs() / ref.abs().clamp_min(1)).max().item() < 1e-4
    # large planes (many blocks per mask): the block sums are added in block order, so two runs agree bit for bit
    x, t = (torch.randn(5, 480, 854) * 3).to(dev), (torch.randn(5, 480, 854) > 0.2).float().to(dev)
    a, b = ops.bce_dice_sums(x, t), ops.bce_dice_sums(x, t)
    assert torch.equal(a, b)
    refb = F.binary_cross_entropy_with_logits(x.cpu(), t.cpu(), reduction="none").flatten(1).sum(1)
    assert ((a[:, 0].cpu() - refb).abs() / refb.abs()).max().item() < 1e-4


def test_layernorm_narrow_rows(dev):
    """LayerNorm2d(+GELU) on 4 / 12 / 6 channels (memory encoder's first mask-downsampler stage, reference model/sam2.py:611-643)."""
    from rga3.hip import ops

    for dim in (4, 12, 6):
        x = rnd((1000, dim), dev, seed=dim)
        w, b = rnd((dim,), dev, 0.5, seed=dim + 1), rnd((dim,), dev, 0.2, seed=dim + 2)
        xf = x.float().cpu()
        ref = F.layer_norm(xf, (dim,), w.float().cpu(), b.float().cpu(), 1e-6)
        assert rel(ops.layernorm(x, w, b, 1e-6), ref) < 6e-3, dim
        refg = F.gelu(ref.to(torch.bfloat16).float())
        assert rel(ops.layernorm(x, w, b, 1e-6, act="gelu"), refg) < 8e-3, dim


@pytest.mark.parametrize("rows,dim", [(5000, 32), (4097, 64), (1000, 16), (3000, 128), (700, 256), (70000, 256), (3, 256), (900, 512), (500, 384), (300, 1024)])
def test_layernorm_backward(dev, rows, dim):
    """dx, dw, db of LayerNorm vs fp32 autograd (lane-group kernel with per-workgroup partial rows for 16..512 channels, wave-per-row + atomics for other widths)."""
    from rga3.hip import ops

    x, dy = rnd((rows, dim), dev, seed=1), rnd((rows, dim), dev, seed=2)
    w = (rnd((dim,), dev, 0.3, seed=3).float() + 1).to(torch.bfloat16)
    dx, dw, db = ops.layernorm_bwd(x, w, dy, 1e-6)
    xr = x.float().cpu().requires_grad_(True)
    wr = w.float().cpu().requires_grad_(True)
    br = torch.zeros(dim, requires_grad=True)
    F.layer_norm(xr, (dim,), wr, br, 1e-6).backward(dy.float().cpu())
    assert rel(dx, xr.grad) < 8e-3 and rel(dw, wr.grad) < 5e-3 and rel(db, br.grad) < 5e-3, (rows, dim)
    if dim in (16, 32, 64, 128, 256, 512):   # the partial-row path is bit-reproducible
        dx2, dw2, db2 = ops.layernorm_bwd(x, w, dy, 1e-6)
        assert torch.equal(dw, dw2) and torch.equal(db, db2) and torch.equal(dx, dx2)


@pytest.mark.parametrize("rows,cols,ld", [(65536, 256, 256), (100000, 32, 32), (262144, 64, 64), (7, 256, 256), (1000, 2048, 2048), (513, 24, 40), (3000, 4096, 4096),
                                          (900, 4, 4), (700, 36, 36)])
def test_colsum(dev, rows, cols, ld):
    """bias gradients: two-stage deterministic column sums (cols % 8 == 0) and the atomic fall-back for other widths, vs an fp64 sum of the same bf16 values."""
    from rga3.hip import ops

    buf = rnd((rows, ld), dev, seed=rows + cols)
    x = buf[:, :cols]
    out = ops.colsum(x)
    ref = x.double().sum(0).cpu()
    assert ((out.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item() < 2e-5, (rows, cols)
    if cols % 8 == 0:
        assert torch.equal(out, ops.colsum(x))


@pytest.mark.parametrize("n,hi,wi,ho,wo", [(3, 64, 64, 256, 256), (2, 37, 53, 120, 168), (2, 120, 168, 37, 53), (1, 16, 16, 16, 16), (2, 5, 7, 50, 9), (1, 256, 256, 1024, 1024)])
def test_bilinear_backward_gather(dev, n, hi, wi, ho, wo):
    """gradient of the bilinear resize (align_corners=False) by gathering, up- and down-sampling, vs torch's autograd of F.interpolate; run-to-run identical."""
    from rga3.hip import ops

    g = torch.Generator().manual_seed(hi * wo)
    dout = torch.randn(n, ho, wo, generator=g)
    xin = torch.zeros(n, 1, hi, wi, requires_grad=True)
    F.interpolate(xin, size=(ho, wo), mode="bilinear", align_corners=False).backward(dout[:, None])
    got = ops.bilinear_bwd(dout.to(dev), (n, hi, wi))
    assert rel(got, xin.grad[:, 0]) < 1e-5
    assert torch.equal(got, ops.bilinear_bwd(dout.to(dev), (n, hi, wi)))
    # with a plane selection the atomic form is used: same numbers within f32 summation order
    idx = torch.arange(n, dtype=torch.int32, device=dev)
    assert rel(ops.bilinear_bwd(dout.to(dev), (n, hi, wi), idx), xin.grad[:, 0]) < 1e-5


@pytest.mark.parametrize("B,P,C", [(16, 65536, 32), (3, 1000, 32), (2, 4096, 16), (1, 77, 8)])
def test_mask_product_forward_backward(dev, B, P, C):
    """masks[b] = hyper[b] @ up[b]^T for all frames in one launch (reference model/sam2.py:2142-2149) and its two gradients vs fp32 einsum autograd."""
    from rga3.hip import ops

    hyper, up = rnd((B, 4, C), dev, seed=1), rnd((B * P, C), dev, 0.5, seed=2)
    dm = torch.randn(B, 4, P, generator=torch.Generator().manual_seed(3))
    hr = hyper.float().cpu().requires_grad_(True)
    ur = up.float().cpu().view(B, P, C).requires_grad_(True)
    ref = torch.einsum("bmc,bpc->bmp", hr, ur)
    ref.backward(dm)
    masks = ops.mask_product(hyper, up, P)
    assert masks.dtype == torch.float32 and rel(masks, ref.detach()) < 1e-5
    dh, du = ops.mask_product_bwd(dm.to(dev), hyper, up, P)
    assert rel(du, ur.grad.view(B * P, C)) < 4e-3 and rel(dh, hr.grad) < 4e-3
    dh2, du2 = ops.mask_product_bwd(dm.to(dev), hyper, up, P)
    assert torch.equal(dh, dh2) and torch.equal(du, du2)


@pytest.mark.parametrize("Nq,Nk,nsplit", [(4096, 28736, 0), (300, 1000, 0), (256, 70, 0), (513, 4160, 1), (64, 4096 + 4, 7), (4096, 4096 + 4, 0)])
def test_memattn_cross_low_rank_values(dev, Nq, Nk, nsplit):
    """csrc/memattn.hip: softmax(scale q k^T) m with the values kept in the 64-wide memory space, against fp32 attention on the same bf16 inputs -- the full
    SAM2-L bank (4096 x 28 736), ragged query / key counts (partial 256-row blocks, a last tile of 40 / 6 / 4 keys), one slice and an uneven slice count, and keys
    whose scores grow along the bank (the running maximum moves in every tile, so the rescaling path of the online softmax is exercised, not skipped)."""
    from rga3.hip import ops

    g = torch.Generator().manual_seed(Nq * 7 + Nk)
    q = (torch.randn(Nq, 256, generator=g) * 1.0).to(torch.bfloat16)
    k = (torch.randn(Nk, 256, generator=g) * (0.5 + 1.5 * torch.linspace(0, 1, Nk)[:, None])).to(torch.bfloat16)
    m = torch.randn(Nk, 64, generator=g).to(torch.bfloat16)
    out = ops.memattn_cross(q.to(dev), k.to(dev), m.to(dev), 256 ** -0.5, nsplit=nsplit)
    assert tuple(out.shape) == (Nq, 64) and out.dtype == torch.bfloat16
    torch.set_num_threads(max(1, min(64, (__import__("os").cpu_count() or 2) // 2)))
    ref = torch.softmax(q.float() @ k.float().t() * 256 ** -0.5, dim=-1) @ m.float()
    assert rel(out, ref) < 1e-2, rel(out, ref)
    # strided operands (a 256-wide column group of a fused [*, 1024] projection) give the same bits
    kw = torch.zeros(Nk, 1024, dtype=torch.bfloat16)
    kw[:, 512:768] = k
    out2 = ops.memattn_cross(q.to(dev), kw.to(dev)[:, 512:768], m.to(dev), 256 ** -0.5, nsplit=nsplit)
    assert torch.equal(out, out2)
    assert torch.equal(out, ops.memattn_cross(q.to(dev), k.to(dev), m.to(dev), 256 ** -0.5, nsplit=nsplit))     # run-to-run identical (fixed summation order)


def test_fused_decoder_heads_and_selection(dev):
    """csrc/dechead.hip: (a) three-layer MLPs on single token rows, several MLPs x several frames in one launch (mixed output widths, sigmoid on one, outputs written
    into strided views) against fp32 F.linear / relu on the same bf16 weights; (b) argmax over IoU 1..3 (first maximum on ties), plane index, obj_ptr_proj of the chosen
    token and the hard object gate against the same arithmetic in torch -- indices bit-exact."""
    from rga3.hip import ops

    B, nq, C = 5, 9, 256
    g = torch.Generator().manual_seed(5)
    hs = torch.randn(B, nq, C, generator=g).to(torch.bfloat16)
    def mk(i, h, o):
        return [(torch.randn(a, b, generator=g) * 0.08).to(torch.bfloat16) if k == 0 else (torch.randn(a, generator=g) * 0.1).to(torch.bfloat16)
                for a, b in ((h, i), (h, h), (o, h)) for k in (0, 1)]
    sets = [mk(C, 256, 32), mk(C, 256, 32), mk(C, 64, 4), mk(C, 256, 1)]
    toks = [2, 3, 1, 0]
    hsd = hs.to(dev)
    hyper = torch.empty((B, 2, 32), dtype=torch.bfloat16, device=dev)
    specs = []
    for i, (w, t) in enumerate(zip(sets, toks)):
        specs.append((hsd.view(-1)[t * C:], nq * C, tuple(x.to(dev) for x in w), i == 2, hyper[:, i] if i < 2 else None))
    outs = ops.mlp3_rows(specs, B)
    for i, (w, t) in enumerate(zip(sets, toks)):
        x = hs[:, t].float()
        for li in range(3):
            x = F.linear(x, w[2 * li].float(), w[2 * li + 1].float())
            x = x.to(torch.bfloat16).float()
            if li < 2:
                x = F.relu(x)
        if i == 2:
            x = torch.sigmoid(x)
        assert rel(outs[i], x) < 6e-3, (i, rel(outs[i], x))
    assert torch.equal(outs[0], hyper[:, 0]) and torch.equal(outs[1], hyper[:, 1])
    # (b) selection
    iou = torch.tensor([[0.9, 0.2, 0.7, 0.7], [0.1, 0.5, 0.5, 0.4], [0.3, 0.1, 0.2, 0.6], [0.0, 0.8, 0.1, 0.3], [0.2, 0.25, 0.5, 0.125]]).to(torch.bfloat16)
    obj = torch.tensor([[1.5], [-0.5], [0.0], [2.0], [0.25]]).to(torch.bfloat16)
    proj = mk(C, C, C)
    no_obj = (torch.randn(C, generator=g)).to(torch.bfloat16)
    mask_toks = hsd[:, 2:6]
    best, sel, sel64, ptr = ops.sam_select_objptr(iou.to(dev), obj.to(dev), mask_toks, tuple(x.to(dev) for x in proj), no_obj.to(dev))
    rbest = torch.argmax(iou[:, 1:].float(), dim=-1)
    assert torch.equal(best.cpu(), rbest) and rbest.tolist() == [1, 0, 2, 0, 1]
    assert torch.equal(sel.cpu().long(), torch.arange(B) * 4 + 1 + rbest) and torch.equal(sel64.cpu(), sel.cpu().long())
    x = hs[torch.arange(B), 3 + rbest].float()
    for li in range(3):
        x = F.linear(x, proj[2 * li].float(), proj[2 * li + 1].float()).to(torch.bfloat16).float()
        if li < 2:
            x = F.relu(x)
    ref = torch.where(obj.float() > 0, x, no_obj.float()[None])
    assert rel(ptr, ref) < 6e-3
    assert torch.equal(ptr[1].cpu(), no_obj) and torch.equal(ptr[2].cpu(), no_obj)


@pytest.mark.parametrize("C,M", [(144, 65536), (144, 1000), (144, 256 * 3 + 17), (288, 16384), (288, 1000), (288, 128 * 3 + 17), (288, 131072)])
def test_hiera_stage1_mlp_fused(dev, C, M):
    """csrc/hiera_mlp.hip: x + W2 gelu(LayerNorm(x) W1^T + b1) + b2 (C -> 4 C -> C; C = 144: Hiera-L stage 1, C = 288: stage 2, reference model/sam2.py:1035-1117,
    :2305-2329) in one launch against fp32 torch (LayerNorm eps 1e-6, exact-erf GELU) on the same bf16 operands, and against the unfused pair it replaces
    (rga3_layernorm_stats + rga3_gemm_ln_bf16 + rga3_gemm_bf16) -- a full frame's tokens (65 536 / 16 384), eight frames of stage 2, and ragged row counts (a partial
    last workgroup, a partial last 32-token wave).  Run to run bit-identical."""
    from rga3.hip import ops

    g = torch.Generator().manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * 1.5 + 0.3 * torch.randn(M, 1, generator=g)).to(torch.bfloat16)
    w1 = (torch.randn(4 * C, C, generator=g) * 0.08 * (144 / C) ** 0.5).to(torch.bfloat16)
    b1 = (torch.randn(4 * C, generator=g) * 0.1).to(torch.bfloat16)
    w2 = (torch.randn(C, 4 * C, generator=g) * 0.05 * (144 / C) ** 0.5).to(torch.bfloat16)
    b2 = (torch.randn(C, generator=g) * 0.1).to(torch.bfloat16)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(torch.bfloat16)
    beta = (0.1 * torch.randn(C, generator=g)).to(torch.bfloat16)
    xd = x.to(dev)
    wf, colc, biasf = ops.fold_layernorm(w1.to(dev), b1.to(dev), gamma.to(dev), beta.to(dev))
    y = ops.hiera_mlp(xd, wf, colc, biasf, w2.to(dev), b2.to(dev), 1e-6)
    xf = x.float()
    ref = xf + F.linear(F.gelu(F.linear(F.layer_norm(xf, (C,), gamma.float(), beta.float(), 1e-6), w1.float(), b1.float())), w2.float(), b2.float())
    assert rel(y, ref) < 1e-2, rel(y, ref)
    hmid = ops.gemm_ln(xd, ops.layernorm_stats(xd, 1e-6), wf, colc, biasf, act="gelu")
    y2 = ops.gemm(hmid, w2.to(dev), b2.to(dev), residual=xd)
    assert rel(y, y2) < 4e-3, rel(y, y2)
    assert torch.equal(y, ops.hiera_mlp(xd, wf, colc, biasf, w2.to(dev), b2.to(dev), 1e-6))
    assert bool(torch.isfinite(y.float()).all())


def test_in_launch_reductions_equal_two_launch_forms(dev):
    """The optional `counters` forms of rga3_gemm_tn_bf16 (K-slice sum by the last workgroup of a tile) and rga3_colsum (second stage by the last workgroup of a column
    block) add in the same order as their second launches: bit-identical results, run after run.  (They are NOT the default: measured slower)"""
    from rga3.hip import ops

    for (K, M, N) in ((65536, 128, 256), (2112, 128, 3584), (4160, 512, 128)):
        a, b = rnd((K, M), dev, 0.5, K), rnd((K, N), dev, 0.5, K + 1)
        ref = ops.gemm_tn(a, b)
        for _ in range(3):
            assert torch.equal(ops.gemm_tn(a, b, fused_sum=True), ref), (K, M, N)
        assert torch.equal(ops.gemm_tn(a, b, out_dtype=torch.float32, fused_sum=True), ops.gemm_tn(a, b, out_dtype=torch.float32))
    for (rows, cols) in ((65536, 256), (144, 256), (1048576, 32), (4096, 2048)):
        x = rnd((rows, cols), dev, 1.0, rows)
        ref = ops.colsum(x)
        for _ in range(3):
            assert torch.equal(ops.colsum(x, fused_finish=True), ref), (rows, cols)


@pytest.mark.parametrize("M,nq", [(4096, 4096), (1000, 256), (16 * 3 + 5, 64)])
def test_memory_layer_row_chain(dev, M, nq):
    """csrc/memlayer.hip: the three one-launch chains of a memory-attention layer against the launches they replace (same library: GEMM with residual epilogue,
    LayerNorm, axial RoPE, the cross-attention merge) and against fp32 torch on the same bf16 operands -- a whole 64 x 64 frame, ragged row counts (partial last
    16-row workgroup), table rows wrapping (token % nq)."""
    from rga3.hip import ops

    g = torch.Generator().manual_seed(M + nq)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(dev)
    x = (torch.randn(M, 256, generator=g) * 1.2 + 0.4 * torch.randn(M, 1, generator=g)).to(torch.bfloat16).to(dev)
    gam, bet = (1 + 0.2 * torch.randn(256, generator=g)).to(torch.bfloat16).to(dev), r(256, sc=0.1)
    ang = torch.rand(nq, 128, generator=g) * 6.28
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)

    def rope_ref(y, cols):       # fp32 complex rotation of consecutive pairs, table row = token % nq, pair = (column % 256) / 2
        y = y.clone()
        t = torch.arange(M) % nq
        for c0 in range(0, cols, 256):
            blk = y[:, c0:c0 + 256].reshape(M, 128, 2)
            c, s = cos.cpu()[t], sin.cpu()[t]
            y[:, c0:c0 + 256] = torch.stack([blk[..., 0] * c - blk[..., 1] * s, blk[..., 0] * s + blk[..., 1] * c], -1).reshape(M, 256)
        return y

    # (1) norm -> qkv (256 -> 768) -> RoPE on q | k
    wqkv, bqkv = r(768, 256, sc=0.06), r(768, sc=0.1)
    _, _, y = ops.memlayer_rows(x, (gam, bet), 1e-5, w2=wqkv, b2=bqkv, rope=(cos, sin), rope_cols=512)
    y_un = ops.gemm(ops.layernorm(x, gam, bet, 1e-5), wqkv, bqkv)
    ops.rope_axial_(y_un[:, :512], cos.repeat(1, 2).contiguous(), sin.repeat(1, 2).contiguous(), M)
    ref = rope_ref(F.linear(F.layer_norm(x.float().cpu(), (256,), gam.float().cpu(), bet.float().cpu(), 1e-5), wqkv.float().cpu(), bqkv.float().cpu()), 512)
    assert rel(y, ref) < 1e-2, rel(y, ref)
    assert rel(y, y_un) < 3e-3, rel(y, y_un)

    # (2) out-projection + residual -> norm -> q projection -> RoPE
    a, wo, bo, wq, bq = r(M, 256), r(256, 256, sc=0.06), r(256, sc=0.1), r(256, 256, sc=0.06), r(256, sc=0.1)
    x2, t2, q2 = ops.memlayer_rows(x, (gam, bet), 1e-5, a=a, w1=wo, b1=bo, want_t=True, w2=wq, b2=bq, rope=(cos, sin), rope_cols=256)
    x_un = ops.gemm(a, wo, bo, residual=x)
    t_un = ops.layernorm(x_un, gam, bet, 1e-5)
    q_un = ops.gemm(t_un, wq, bq)
    ops.rope_axial_(q_un, cos, sin, M)
    assert rel(x2, x_un) < 2e-3, rel(x2, x_un)                     # same rounding points; the library GEMM may cut K differently (stream-K)
    assert rel(t2, t_un) < 3e-3 and rel(q2, q_un) < 3e-3, (rel(t2, t_un), rel(q2, q_un))
    xr = F.linear(a.float().cpu(), wo.float().cpu(), bo.float().cpu()) + x.float().cpu()
    qr = rope_ref(F.linear(F.layer_norm(xr, (256,), gam.float().cpu(), bet.float().cpu(), 1e-5), wq.float().cpu(), bq.float().cpu()), 256)
    assert rel(x2, xr) < 1e-2 and rel(q2, qr) < 1.5e-2, (rel(x2, xr), rel(q2, qr))

    # (3) merge of the cross-attention slices -> (Wo Wv) + residual -> norm
    Nk = 1500
    qq, kk, mm = r(M, 256), (torch.randn(Nk, 256, generator=g) * (0.5 + 1.5 * torch.linspace(0, 1, Nk)[:, None])).to(torch.bfloat16).to(dev), r(Nk, 64)
    wov, bov = r(256, 64, sc=0.1), r(256, sc=0.1)
    for nsplit in (0, 1, 5):
        pm = ops.memattn_cross(qq, kk, mm, 256 ** -0.5, nsplit=nsplit)
        x_un = ops.gemm(pm, wov, bov, residual=x)
        t_un = ops.layernorm(x_un, gam, bet, 1e-5)
        parts = ops.memattn_cross(qq, kk, mm, 256 ** -0.5, nsplit=nsplit, partials=True)
        x3, t3, _ = ops.memlayer_rows(x, (gam, bet), 1e-5, partials=parts, w1=wov, b1=bov, want_t=True)
        assert rel(x3, x_un) < 2e-3, (nsplit, rel(x3, x_un))
        assert rel(t3, t_un) < 3e-3, rel(t3, t_un)
    # plain 64-wide rows as the operand of product 1, nothing but the normalised rows written
    _, t4, _ = ops.memlayer_rows(x, (gam, bet), 1e-5, a=pm, w1=wov, b1=bov, want_x=False)
    assert torch.equal(t4, t3)
    assert torch.equal(q2, ops.memlayer_rows(x, (gam, bet), 1e-5, a=a, w1=wo, b1=bo, w2=wq, b2=bq, rope=(cos, sin), rope_cols=256)[2])


@pytest.mark.parametrize("Fn,S", [(1, 1024), (2, 64), (1, 8)])
def test_mask_downsampler_narrow_stages_fused(dev, Fn, S):
    """csrc/sam2ops.hip conv3x3s2_ln_gelu: Conv2d(k 3, s 2, p 1) + LayerNorm2d + GELU of the two narrow mask-down-sampler stages in one launch each -- the same numbers as
    the three launches they replace (conv3x3s2, then layernorm with the fused GELU), and fp32 torch on the same bf16 operands; a whole 1024 x 1024 mask, two frames, and a
    8 x 8 map where many taps fall on the zero padding."""
    from rga3.hip import ops

    g = torch.Generator().manual_seed(S)
    mask = (torch.randn(Fn, S, S, generator=g) * 4).to(dev)
    w1, b1 = rnd((4, 1, 3, 3), dev, 0.3, seed=1), rnd((4,), dev, 0.1, seed=2)
    g1, h1 = (1 + 0.2 * torch.randn(4, generator=g)).to(torch.bfloat16).to(dev), rnd((4,), dev, 0.1, seed=3)
    w2, b2 = rnd((16, 4, 3, 3), dev, 0.2, seed=4), rnd((16,), dev, 0.1, seed=5)
    g2, h2 = (1 + 0.2 * torch.randn(16, generator=g)).to(torch.bfloat16).to(dev), rnd((16,), dev, 0.1, seed=6)
    y1 = ops.conv3x3s2_ln_gelu(mask, w1, b1, g1, h1, 1e-6, Fn, S, S, 20.0, -10.0)
    u1 = ops.layernorm(ops.conv3x3s2(mask, w1, b1, Fn, S, S, 20.0, -10.0), g1, h1, 1e-6, act="gelu")

    def same(a, b):     # the same arithmetic at the same rounding points; the compilers' contraction choices differ in a few elements per million, by one bf16 ulp
        d = (a.float() - b.float()).abs()
        return float((d > 0).float().mean()) < 2e-5 and bool((d <= b.float().abs() * 2 ** -7 + 1e-30).all())

    assert same(y1, u1)
    y2 = ops.conv3x3s2_ln_gelu(y1, w2, b2, g2, h2, 1e-6, Fn, S // 2, S // 2)
    u2 = ops.layernorm(ops.conv3x3s2(y1, w2, b2, Fn, S // 2, S // 2), g2, h2, 1e-6, act="gelu")
    assert same(y2, u2)
    mm = (torch.sigmoid(mask.cpu()) * 20 - 10).to(torch.bfloat16).float()
    r1 = F.conv2d(mm[:, None], w1.float().cpu(), b1.float().cpu(), stride=2, padding=1).permute(0, 2, 3, 1)
    r1 = F.gelu(F.layer_norm(r1, (4,), g1.float().cpu(), h1.float().cpu(), 1e-6))
    assert rel(y1, r1.reshape(-1, 4)) < 2e-2
    r2 = F.conv2d(y1.float().cpu().view(Fn, S // 2, S // 2, 4).permute(0, 3, 1, 2), w2.float().cpu(), b2.float().cpu(), stride=2, padding=1).permute(0, 2, 3, 1)
    r2 = F.gelu(F.layer_norm(r2, (16,), g2.float().cpu(), h2.float().cpu(), 1e-6))
    assert rel(y2, r2.reshape(-1, 16)) < 2e-2


def test_copy_many(dev):
    """csrc/sam2ops.hip copy_many: up to 24 device-to-device copies per launch (more are chunked), sizes from 16 B to a few MB; odd-sized / strided pairs fall back to
    Tensor.copy_."""
    from rga3.hip import ops

    g = torch.Generator().manual_seed(3)
    sizes = [8, 256, 4096 * 64, 1000 * 8, 16 * 4096 * 32, 24] + [256] * 25
    srcs = [torch.randn(n, generator=g).to(torch.bfloat16).to(dev) for n in sizes]
    dsts = [torch.zeros_like(s_) for s_ in srcs]
    odd_s, odd_d = torch.randn(7, generator=g).to(torch.bfloat16).to(dev), torch.zeros(7, dtype=torch.bfloat16, device=dev)        # 14 bytes
    big = torch.randn(64, 64, generator=g).to(dev)
    view_d = torch.zeros(64, 64, device=dev)
    ops.copy_many(list(zip(dsts, srcs)) + [(odd_d, odd_s), (view_d[:, :32], big[:, :32])])
    for d_, s_ in zip(dsts, srcs):
        assert torch.equal(d_, s_)
    assert torch.equal(odd_d, odd_s) and torch.equal(view_d[:, :32], big[:, :32]) and float(view_d[:, 32:].abs().max()) == 0.0


@pytest.mark.parametrize("Fn,H,W,C", [(1, 64, 64, 256), (2, 20, 12, 72), (1, 7, 9, 8)])
def test_dwconv7x7_tiled(dev, Fn, H, W, C):
    """csrc/sam2ops.hip dwconv7_kernel (8 x 8 pixel tile x 64 channels per workgroup, halo and taps in LDS): the memory encoder's map, maps that are not multiples of
    the tile with a partial channel block, a map smaller than the filter -- against F.conv2d(groups = C) on the same bf16 operands."""
    from rga3.hip import ops

    x, w, b = rnd((Fn * H * W, C), dev, seed=H), rnd((C, 1, 7, 7), dev, 0.1, seed=W), rnd((C,), dev, 0.1, seed=C)
    ref = F.conv2d(x.float().cpu().view(Fn, H, W, C).permute(0, 3, 1, 2), w.float().cpu(), b.float().cpu(), padding=3, groups=C).permute(0, 2, 3, 1).reshape(-1, C)
    out = ops.dwconv7x7(x, w, b, Fn, H, W)
    assert rel(out, ref) < 6e-3
    assert torch.equal(out, ops.dwconv7x7(x, w, b, Fn, H, W))
    off = torch.zeros(8 + C * 49, dtype=torch.bfloat16, device=dev)      # a filter that does not start on a 16-byte boundary
    off[3:3 + C * 49] = w.reshape(-1)
    assert torch.equal(out, ops.dwconv7x7(x, off[3:3 + C * 49].view(C, 1, 7, 7), b, Fn, H, W))


@pytest.mark.parametrize("B,hw,nk", [(1, 4096, 9), (3, 24, 9), (2, 40, 16), (1, 16, 1)])
def test_decoder_image_side_block_boundary(dev, B, hw, nk):
    """csrc/decimg.hip: image-to-token attention + norm4 + the next token-to-image k / v projections in one launch, against the eight launches it replaces (add_bcast,
    q_proj, the attention kernel, out_proj + residual, LayerNorm, add_bcast, k_proj, v_proj) and against fp32 torch on the same bf16 operands -- a whole 64 x 64 frame,
    several small frames whose 16-row blocks straddle a frame boundary, the maximum token count, a single token."""
    from rga3.hip import ops

    g = torch.Generator().manual_seed(B * 100 + hw + nk)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(dev)
    M = B * hw
    keys, pe = (torch.randn(M, 256, generator=g) * 1.2 + 0.3 * torch.randn(M, 1, generator=g)).to(torch.bfloat16).to(dev), r(hw, 256, sc=0.5)
    kt, vt = r(B * nk, 128), r(B * nk, 128)
    wq, bq, wo, bo = r(128, 256, sc=0.06), r(128, sc=0.1), r(256, 128, sc=0.09), r(256, sc=0.1)
    gam, bet = (1 + 0.2 * torch.randn(256, generator=g)).to(torch.bfloat16).to(dev), r(256, sc=0.1)
    wk, bk, wv, bv = r(128, 256, sc=0.06), r(128, sc=0.1), r(128, 256, sc=0.06), r(128, sc=0.1)
    out, k2, v2 = ops.decimg_rows(keys, pe, kt, vt, nk, (wq, bq), (wo, bo), (gam, bet), 1e-5, (wk, bk), (wv, bv), scale=0.25)
    # the launches replaced
    kin = ops.add_bcast(keys, pe)
    qp = ops.gemm(kin, wq, bq)
    cuq = torch.arange(0, M + 1, hw, dtype=torch.int32, device=dev)
    cuk = torch.arange(0, B * nk + 1, nk, dtype=torch.int32, device=dev)
    o = ops.attn_varlen(qp.view(-1, 8, 16), kt.view(-1, 8, 16), vt.view(-1, 8, 16), cuq, cuk, hw, 0.25, max_k=nk)
    x_un = ops.layernorm(ops.gemm(o.reshape(-1, 128), wo, bo, residual=keys), gam, bet, 1e-5)
    k_un, v_un = ops.gemm(ops.add_bcast(x_un, pe), wk, bk), ops.gemm(x_un, wv, bv)
    assert rel(out, x_un) < 4e-3 and rel(k2, k_un) < 6e-3 and rel(v2, v_un) < 6e-3, (rel(out, x_un), rel(k2, k_un), rel(v2, v_un))
    # fp32
    f = lambda t: t.float().cpu()
    pe_rows = f(pe).repeat(B, 1)
    qf_ = (F.linear(f(keys) + pe_rows, f(wq), f(bq))).view(B, hw, 8, 16).permute(0, 2, 1, 3)
    kf_, vf_ = f(kt).view(B, nk, 8, 16).permute(0, 2, 1, 3), f(vt).view(B, nk, 8, 16).permute(0, 2, 1, 3)
    of_ = torch.softmax(qf_ @ kf_.transpose(-1, -2) * 0.25, dim=-1) @ vf_
    xr = F.layer_norm(F.linear(of_.permute(0, 2, 1, 3).reshape(M, 128), f(wo), f(bo)) + f(keys), (256,), f(gam), f(bet), 1e-5)
    assert rel(out, xr) < 1.5e-2, rel(out, xr)
    assert rel(k2, F.linear(xr + pe_rows, f(wk), f(bk))) < 2e-2 and rel(v2, F.linear(xr, f(wv), f(bv))) < 2e-2
    if hw % 16 == 0:     # the transposed form of v2 ([frames * 128, hw], what attn_fewq reads): the same numbers
        _, k2t, v2t = ops.decimg_rows(keys, pe, kt, vt, nk, (wq, bq), (wo, bo), (gam, bet), 1e-5, (wk, bk), (wv, bv), scale=0.25, v_transposed=True)
        assert torch.equal(k2t.permute(1, 0, 2).reshape(M, 128), k2) and torch.equal(v2t.view(B, 128, hw).permute(0, 2, 1).reshape(M, 128), v2)      # k2 head-major [8, M, 16]
    only, n1, n2 = ops.decimg_rows(keys, pe, kt, vt, nk, (wq, bq), (wo, bo), (gam, bet), 1e-5, scale=0.25)
    assert n1 is None and n2 is None and torch.equal(only, out)
    assert torch.equal(out, ops.decimg_rows(keys, pe, kt, vt, nk, (wq, bq), (wo, bo), (gam, bet), 1e-5, (wk, bk), (wv, bv), scale=0.25)[0])


def test_two_way_transformer_fused_image_side_matches_separate_launches(dev):
    """The mask decoder's two-way transformer at SAM2-L dims (4096 image tokens, 9 tokens): the path with the one-launch image side against the per-launch path."""
    from rga3.model import sam2 as S2

    torch.manual_seed(5)
    tr = S2.TwoWayTransformer(2, 256, 8, 2048).to(torch.bfloat16).to(dev).eval()
    with torch.no_grad():
        for p_ in tr.parameters():
            if p_.dim() >= 2:
                p_.normal_(0, 0.05)
    g = torch.Generator().manual_seed(6)
    keys = torch.randn(4096, 256, generator=g).to(torch.bfloat16).to(dev)
    pe = (torch.randn(4096, 256, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    toks = torch.randn(9, 256, generator=g).to(torch.bfloat16).to(dev)
    with torch.no_grad():
        assert tr._fusable(keys, pe, 9)
        q1, k1 = tr(keys, pe, toks, 1, 9, 4096)
        S2._DECIMG = False
        try:
            q0, k0 = tr(keys, pe, toks, 1, 9, 4096)
        finally:
            S2._DECIMG = True
    assert rel(q1, q0) < 1e-2 and rel(k1, k0) < 1e-2, (rel(q1, q0), rel(k1, k0))


@pytest.mark.parametrize("frames,nq,nk,H", [(1, 9, 4096, 8), (3, 16, 1000, 8), (2, 1, 20, 2), (1, 5, 64, 1)])
def test_attention_few_queries(dev, frames, nq, nk, H):
    """csrc/decimg.hip attn_fewq: up to 16 queries over up to 4096 keys per frame, heads of 16, values given transposed, one workgroup per (frame, head) -- against fp32
    attention on the same bf16 operands and against the general kernel; ragged key counts (a last 16-key tile of 8 / 4), one query, the value bias added after the
    softmax."""
    from rga3.hip import ops

    g = torch.Generator().manual_seed(frames * 1000 + nk)
    q = (torch.randn(frames * nq, H * 16, generator=g) * 1.5).to(torch.bfloat16).to(dev)
    k = (torch.randn(frames * nk, H * 16, generator=g) * (0.5 + 1.5 * torch.linspace(0, 1, frames * nk)[:, None])).to(torch.bfloat16).to(dev)
    v = torch.randn(frames * nk, H * 16, generator=g).to(torch.bfloat16).to(dev)
    vb = (torch.randn(H * 16, generator=g) * 0.3).to(torch.bfloat16).to(dev)
    vt = v.view(frames, nk, H * 16).permute(0, 2, 1).reshape(frames * H * 16, nk).contiguous()
    out = ops.attn_fewq(q, k, vt, nq, nk, H, 0.25, vb)
    qf_, kf_, vf_ = (t.float().cpu().view(frames, -1, H, 16).permute(0, 2, 1, 3) for t in (q, k, v))
    ref = (torch.softmax(qf_ @ kf_.transpose(-1, -2) * 0.25, dim=-1) @ vf_).permute(0, 2, 1, 3).reshape(frames * nq, H * 16) + vb.float().cpu()
    assert rel(out, ref) < 1e-2, rel(out, ref)
    cuq = torch.arange(0, frames * nq + 1, nq, dtype=torch.int32, device=dev)
    cuk = torch.arange(0, frames * nk + 1, nk, dtype=torch.int32, device=dev)
    gen = ops.attn_varlen(q.view(-1, H, 16), k.view(-1, H, 16), v.view(-1, H, 16), cuq, cuk, nq, 0.25, max_k=nk).reshape(frames * nq, H * 16)
    assert rel(ops.attn_fewq(q, k, vt, nq, nk, H, 0.25), gen) < 8e-3
    assert torch.equal(out, ops.attn_fewq(q, k, vt, nq, nk, H, 0.25, vb))
    khm = k.view(frames * nk, H, 16).permute(1, 0, 2).contiguous()          # head-major keys: the same bits
    assert torch.equal(out, ops.attn_fewq(q, khm, vt, nq, nk, H, 0.25, vb, k_head_major=True))
