"""Tensor-level wrappers over the C ABI (include/rga3_hip.h).

PyTorch is used for device memory and streams only; every function here enqueues hand-written HIP kernels
on ``torch.cuda.current_stream()`` and raises if the extension is missing or rejects its arguments.
"""
from __future__ import annotations

import torch

from . import lib as _lib
from . import tuner as _tuner

BF16, F32 = 0, 1
ACT = {"none": 0, "gelu": 1, "swiglu": 2, "relu": 3}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.Rga3Error("rga3 HIP ops need device tensors (no CPU fallback on the product path)")


def pad_cols(x: torch.Tensor, ld: int) -> torch.Tensor:
    """[rows, cols] bf16 -> [rows, ld] with zero-filled tail columns."""
    _need_cuda(x)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    out = torch.empty((rows, ld), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().rga3_pad_cols(x.data_ptr(), out.data_ptr(), rows, cols, x.stride(0), ld, _stream()), "pad_cols")
    return out


_gemm_ws = {}
_ws_scope = [None]


class workspace_scope:
    """``with ops.workspace_scope(token): ...`` -- every scratch buffer this module hands to a kernel (stream-K slabs / flags, the memory cross-attention's partial
    sums, the TN ticket words) is keyed by (device, token) instead of (device, current stream) inside the block.  A captured hipGraph bakes the POINTERS it saw, and a
    graph is replayed on whatever stream is current -- so the scratch of a capture must belong to the graph's owner, not to a stream handle: torch.cuda.Stream() hands
    out streams of a pool of 32 round-robin, and a later eager launch on a stream that happens to BE some capture stream would share that graph's scratch while the
    graph replays elsewhere (round 6: an intermittent mismatch of the concurrent object slots at the configs[3] shape).  The owner replays its graphs one after the
    other; warm-up and capture both run inside the scope so that every buffer exists before the capture."""

    def __init__(self, token):
        self.token = ("scope", token)

    def __enter__(self):
        self.old, _ws_scope[0] = _ws_scope[0], self.token
        return self

    def __exit__(self, *exc):
        _ws_scope[0] = self.old


def _ws_key(device):
    dev = device.index if device.index is not None else torch.cuda.current_device()
    return (dev, _ws_scope[0] if _ws_scope[0] is not None else torch.cuda.current_stream(device).cuda_stream)


def gemm_workspace(device) -> torch.Tensor:
    """The caller-owned workspace of the stream-K / split-K GEMM tilings for (device, current stream) -- or (device, workspace_scope token): allocated from PyTorch's
    caching allocator on first use, zeroed once (the flag words; the kernels leave them zero), then kept for the life of the process."""
    key = _ws_key(device)
    t = _gemm_ws.get(key)
    if t is None:
        with torch.cuda.device(device):
            n = int(_lib.load().rga3_gemm_workspace_bytes())
        if n <= 0:
            raise _lib.Rga3Error("rga3_gemm_workspace_bytes failed: " + _lib.last_error())
        t = _gemm_ws[key] = torch.zeros(n, dtype=torch.uint8, device=device)
    return t


def gemm_stream_k_timeouts(device=None) -> int:
    """Sum of the bounded-spin give-ups recorded in this process's GEMM workspaces (diagnostic; expected 0)."""
    tot = 0
    for (dev, _), t in _gemm_ws.items():
        if device is None or dev == (device.index if device.index is not None else torch.cuda.current_device()):
            with torch.cuda.device(dev):
                tot += int(_lib.load().rga3_gemm_stream_k_timeouts(t.data_ptr()))
    return tot


_LNSUM_TILES = (20, 3, 5, 12, 13, 23)      # the tilings rga3_gemm_lnsum_bf16 has kernels for


def gemm(a: torch.Tensor, w: torch.Tensor, bias=None, residual=None, act: str = "none", out_dtype=torch.bfloat16,
         out=None, tile: int = -1, colscale=None, rms_in=None, rms_out=None) -> torch.Tensor:
    """out = residual + colscale * act(a @ w.T + bias).  a [M,K], w [N,K] (nn.Linear layout), bf16.

    RMSNorm folded around the product (rga3_gemm_rms_bf16; M > 16, bf16 output, no column scale):
      rms_in = (row_sumsq [M] int64, norm_width, eps): ``a`` holds the UN-normalised rows and ``w`` the weight with the norm weight folded in
               (w * gamma[None, :]); the accumulators are scaled by 1 / sqrt(row_sumsq / 2^20 / norm_width + eps) before bias / activation;
      rms_out = [M] int64 (zeroed by the caller): the sums of squares of the bf16 rows written are ADDED to it as 2^20 fixed-point integers
                (integer atomics: any order, same bits) -- the rms_in of the product that consumes the normalised rows."""
    _need_cuda(a, w, bias, residual, out, colscale)
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    assert a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1], (a.shape, w.shape)
    assert a.stride(1) == 1 and w.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    if K % 8 != 0 or a.stride(0) % 8 != 0 or w.stride(0) % 8 != 0:
        Kp = (K + 7) // 8 * 8
        if a.shape[1] != Kp or a.stride(0) % 8 != 0:
            a = pad_cols(a, Kp)
        if w.shape[1] != Kp or w.stride(0) % 8 != 0:
            w = pad_cols(w, Kp)
        K = Kp
    n_out = N // 2 if act == "swiglu" else N
    if out is None:
        out = torch.empty((M, n_out), dtype=out_dtype, device=a.device)
    assert out.shape == (M, n_out) and out.stride(1) == 1 and out.dtype == out_dtype
    if bias is not None:
        assert bias.dtype == torch.bfloat16 and bias.numel() == N and bias.is_contiguous()
    ldr = 0
    if residual is not None:
        assert residual.dtype == torch.bfloat16 and residual.shape == (M, n_out) and residual.stride(1) == 1
        ldr = residual.stride(0)
    if colscale is not None:
        assert colscale.dtype == torch.bfloat16 and colscale.numel() == n_out and colscale.is_contiguous()
    fn = _lib.load().rga3_gemm_bf16
    odt = BF16 if out_dtype == torch.bfloat16 else F32
    ws = gemm_workspace(a.device)

    rms = rms_in is not None or rms_out is not None
    if rms:
        assert M > 16 and out_dtype == torch.bfloat16 and colscale is None, "gemm: the RMSNorm-folded form takes M > 16, bf16 output, no column scale"
        rs_t, rs_width, rs_eps = rms_in if rms_in is not None else (None, 0, 0.0)
        for t_ in (rs_t, rms_out):
            assert t_ is None or (t_.is_cuda and t_.dtype == torch.int64 and t_.numel() == M and t_.is_contiguous())
        fn_rms = _lib.load().rga3_gemm_rms_bf16

    def run(t, final=True):
        ro = rms_out if (rms and final) else None      # tuner trials must not ADD into the caller's row sums: only the final launch carries rms_out
        if rms and (rs_t is not None or ro is not None):
            _lib.check(fn_rms(a.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(), M, N, K, a.stride(0), w.stride(0), out.stride(0), ldr,
                              ACT[act], t, ws.data_ptr(), ws.numel(), _ptr(rs_t), int(rs_width), float(rs_eps), _ptr(ro), _stream()), "gemm_rms_bf16")
        else:
            _lib.check(fn(a.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(residual), _ptr(colscale), out.data_ptr(), M, N, K, a.stride(0), w.stride(0),
                          out.stride(0), ldr, ACT[act], odt, t, ws.data_ptr(), ws.numel(), _stream()), "gemm_bf16")

    if rms:
        if tile == -1:
            tile = _tuner.pick(_tuner.key_of(M, N, K, act, odt, bias is not None, residual is not None), lambda t: run(t, final=False))
            if tile in (14, 25, 40, 41):     # decided for the plain product of the same shape: those tilings have their own epilogues
                tile = -1
        run(tile)
        return out
    if tile == -1 and M <= 4 and colscale is None:
        tile = 40   # decode step: a weight stream, not a tiled product (gemv_kernel)
    if tile == -1 and M * N * K >= (1 << 24) and not (residual is not None and residual.data_ptr() == out.data_ptr()):
        # few output tiles over a very long K (weight gradients dW = dY^T X): also try the split-K form
        plain = act == "none" and residual is None and colscale is None
        extra = (25,) if (M * N <= (1 << 20) and K >= 4096 and plain) else ()
        if plain and bias is None and N % 8 == 0 and M * N <= (1 << 19) and K >= 512:   # skinny products (LoRA x A^T, dY B): 64 x 64 tiles with a K split
            extra = extra + (14,)
        if M <= 16 and colscale is None and out_dtype == torch.bfloat16 and act != "swiglu":   # token rows (smaller products take tile 41 without asking)
            extra = extra + (41,)
        if act == "none" and N % 192 == 0 and N % 256 != 0 and M >= 1024:   # widths of 3 / 9 x 192 (Hiera stage 3): the ping-pong loop on 256 x 192 tiles
            extra = extra + (23,)
        tile = _tuner.pick(_tuner.key_of(M, N, K, act, odt, bias is not None, residual is not None), run, extra)
    run(tile)
    return out


def gemm_swiglu_pre(a, w, bias=None, tile: int = -1):
    """(silu(gate) * up [M, N / 2], pre-activations [M, N]) of a @ w.T + bias on the gate | up pack (16-row blocks interleaved) in ONE launch
    (rga3_gemm_swiglu_pre_bf16): what gemm(a, w, bias) followed by swiglu_fwd produces, without writing and re-reading the pre-activations in between."""
    _need_cuda(a, w, bias)
    assert a.dtype == w.dtype == torch.bfloat16 and a.dim() == w.dim() == 2 and a.shape[1] == w.shape[1] and a.stride(1) == w.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    assert N % 32 == 0 and M > 4
    out = torch.empty((M, N // 2), dtype=torch.bfloat16, device=a.device)
    pre = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    ws = gemm_workspace(a.device)
    fn = _lib.load().rga3_gemm_swiglu_pre_bf16

    def run(t):
        _lib.check(fn(a.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(), pre.data_ptr(), M, N, K, a.stride(0), w.stride(0), out.stride(0), pre.stride(0), t,
                      ws.data_ptr(), ws.numel(), _stream()), "gemm_swiglu_pre_bf16")

    if tile == -1 and M * N * K >= (1 << 24):
        tile = _tuner.pick(_tuner.key_of(M, N, K, "swiglu+pre", BF16, bias is not None, False), run)
        if tile in (14, 25, 40, 41):    # forced (tuner.force) tilings with an epilogue of their own: no pre-activation store there
            tile = -1
    run(tile)
    return out, pre


def gemm_cat(a, w, bias=None, a2=None, w2=None, wn=None, tile: int = -1):
    """Concatenated operands (rga3_gemm_cat_bf16): out = [a | a2] @ [w | w2].T + bias  and, with wn [N2, K], out_n = a @ wn.T from the same launch.
    a [M, K], w [N, K], a2 [M, K2], w2 [N, K2] bf16 (row strides free; K, K2 multiples of 64).  Returns out or (out, out_n)."""
    _need_cuda(a, w, bias, a2, w2, wn)
    assert a.dtype == w.dtype == torch.bfloat16 and a.dim() == w.dim() == 2 and a.shape[1] == w.shape[1] and a.stride(1) == w.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    assert (a2 is None) == (w2 is None) and (a2 is not None or wn is not None)
    K2 = 0
    if a2 is not None:
        assert a2.dtype == w2.dtype == torch.bfloat16 and a2.shape[0] == M and w2.shape[0] == N and a2.shape[1] == w2.shape[1] and a2.stride(1) == w2.stride(1) == 1
        K2 = a2.shape[1]
    N2 = 0
    out_n = None
    if wn is not None:
        assert wn.dtype == torch.bfloat16 and wn.dim() == 2 and wn.shape[1] == K and wn.stride(1) == 1
        N2 = wn.shape[0]
        out_n = torch.empty((M, N2), dtype=torch.bfloat16, device=a.device)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    fn = _lib.load().rga3_gemm_cat_bf16

    def run(t):
        _lib.check(fn(a.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(), M, N, K, a.stride(0), w.stride(0), out.stride(0), _ptr(a2), _ptr(w2), K2,
                      a2.stride(0) if a2 is not None else 0, w2.stride(0) if w2 is not None else 0, _ptr(wn), _ptr(out_n), N2, wn.stride(0) if wn is not None else 0,
                      out_n.stride(0) if out_n is not None else 0, t, _stream()), "gemm_cat_bf16")

    if tile == -1:
        cands = tuple(t for t in ((12, 3, 6, 13) if wn is not None else (12, 3, 4, 5, 6, 13)) if wn is None or N % (256 if t in (3, 6) else 64 if t == 13 else 128) == 0)
        tile = _tuner.pick(_tuner.key_of(M, N + N2, K + K2, "cat", BF16, bias is not None, False), run, candidates=cands)
        if tile not in cands:    # a tiling forced for the plain GEMMs (tuner.force) that this entry point does not have
            tile = -1
    run(tile)
    return out if out_n is None else (out, out_n)


def gemm_rows16_many(sets):
    """Up to 4 token-row products (M <= 16) in one launch (csrc/gemm_bf16.hip gemm_rows16_many_kernel).  sets: list of (a, a2 or None, weight, bias or None[, residual
    [, act]]): out_i = act((a + a2) @ weight.T + bias) (+ residual), a + a2 rounded to bf16 first.  -> list of [M, N] bf16."""
    import ctypes
    n = len(sets)
    assert 1 <= n <= 4
    ptrs, dims, outs = (ctypes.c_void_p * (6 * n))(), (ctypes.c_int64 * (9 * n))(), []
    for i, st in enumerate(sets):
        a, a2, w, b = st[:4]
        res = st[4] if len(st) > 4 else None
        act = st[5] if len(st) > 5 else "none"
        _need_cuda(a, a2, w, b, res)
        assert a.dtype == w.dtype == torch.bfloat16 and a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1] and a.stride(1) == 1 and w.stride(1) == 1
        M, K = a.shape
        N = w.shape[0]
        assert M <= 16 and K % 8 == 0 and (a2 is None or (a2.shape == a.shape and a2.stride(1) == 1 and a2.dtype == torch.bfloat16))
        assert b is None or (b.numel() == N and b.is_contiguous() and b.dtype == torch.bfloat16)
        assert res is None or (tuple(res.shape) == (M, N) and res.stride(1) == 1 and res.stride(0) >= N and res.dtype == torch.bfloat16)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
        outs.append(out)
        for j, t in enumerate((a, a2, w, b, res, out)):
            ptrs[6 * i + j] = _ptr(t) or None
        for j, v in enumerate((M, N, K, ACT[act], a.stride(0), a2.stride(0) if a2 is not None else 0, w.stride(0), N, res.stride(0) if res is not None else 0)):
            dims[9 * i + j] = int(v)
    _lib.check(_lib.load().rga3_gemm_rows16_many(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(dims, ctypes.c_void_p), n, _stream()), "gemm_rows16_many")
    return outs


def layernorm_stats(x, eps: float):
    """[rows, 2] f32 = (mean, 1/sqrt(var + eps)) of the rows of x [rows, dim] bf16 (row stride free): the input of gemm_ln."""
    _need_cuda(x)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    st = torch.empty((x.shape[0], 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().rga3_layernorm_stats(x.data_ptr(), st.data_ptr(), x.shape[0], x.shape[1], x.stride(0), float(eps), _stream()), "layernorm_stats")
    return st


def fold_layernorm(weight, bias, gamma, beta):
    """(Wf, colc, biasf) for gemm_ln: LayerNorm(x; gamma, beta) @ weight.T + bias  ==  rinv (x @ Wf.T - mean colc) + biasf.
    Wf = bf16(weight * gamma) [N, K]; colc = row sums of THAT bf16 matrix in f32 (what the product really multiplies); biasf = bf16(beta @ weight.T + bias)."""
    w = weight.detach().float()
    wf = (w * gamma.detach().float()[None, :]).to(torch.bfloat16).contiguous()
    colc = wf.float().sum(1).contiguous()
    # (elementwise product + row sum rather than `w @ beta`: no vendor-BLAS launch on the product path, even at pack time)
    bf = (w * beta.detach().float()[None, :]).sum(1) if beta is not None else torch.zeros(w.shape[0], dtype=torch.float32, device=w.device)
    if bias is not None:
        bf = bf + bias.detach().float()
    return wf, colc, bf.to(torch.bfloat16).contiguous()


_hm288_packs = {}   # (pointers + versions of the folded operands) -> (packed chunk images, the operands themselves: kept alive so that an address cannot be recycled)


def _hiera_mlp288_pack(wf, colc, biasf, w2):
    """The 36 chunk images rga3_hiera_mlp288 streams by LDS-DMA (csrc/hiera_mlp.hip), built once per frozen block."""
    key = (wf.data_ptr(), wf._version, colc.data_ptr(), colc._version, biasf.data_ptr(), biasf._version, w2.data_ptr(), w2._version)
    hit = _hm288_packs.get(key)
    if hit is None:
        L = _lib.load()
        pack = torch.empty(int(L.rga3_hiera_mlp288_pack_bytes()), dtype=torch.uint8, device=wf.device)
        _lib.check(L.rga3_hiera_mlp288_pack(wf.data_ptr(), colc.data_ptr(), biasf.data_ptr(), w2.data_ptr(), pack.data_ptr(), _stream()), "hiera_mlp288_pack")
        if len(_hm288_packs) > 64:
            _hm288_packs.clear()
        hit = _hm288_packs[key] = (pack, (wf, colc, biasf, w2))
    return hit[0]


def hiera_mlp(x, wf, colc, biasf, w2, b2, eps: float):
    """x + W2 gelu(LayerNorm(x) W1^T + b1) + b2 for the C -> 4 C -> C MLP of a frozen Hiera block in one launch (csrc/hiera_mlp.hip), C = 144 (stage 1) or 288 (stage 2:
    the weights travel as packed chunk images, built on first use and kept per weight version); (wf, colc, biasf) = fold_layernorm(W1, b1, gamma, beta)."""
    _need_cuda(x, wf, colc, biasf, w2, b2)
    assert x.dtype == wf.dtype == w2.dtype == b2.dtype == biasf.dtype == torch.bfloat16 and colc.dtype == torch.float32
    C = x.shape[1]
    assert x.dim() == 2 and C in (144, 288) and x.is_contiguous() and tuple(wf.shape) == (4 * C, C) and tuple(w2.shape) == (C, 4 * C) and wf.is_contiguous() and w2.is_contiguous()
    assert colc.numel() == 4 * C and biasf.numel() == 4 * C and b2.numel() == C and colc.is_contiguous() and biasf.is_contiguous() and b2.is_contiguous()
    y = torch.empty_like(x)
    if C == 288:
        pack = _hiera_mlp288_pack(wf, colc, biasf, w2)
        _lib.check(_lib.load().rga3_hiera_mlp288(x.data_ptr(), pack.data_ptr(), b2.data_ptr(), y.data_ptr(), x.shape[0], float(eps), _stream()), "hiera_mlp288")
    else:
        _lib.check(_lib.load().rga3_hiera_mlp144(x.data_ptr(), wf.data_ptr(), colc.data_ptr(), biasf.data_ptr(), w2.data_ptr(), b2.data_ptr(), y.data_ptr(), x.shape[0],
                                                 float(eps), _stream()), "hiera_mlp144")
    return y


hiera_mlp144 = hiera_mlp      # the round-3 name (stage 1 only)


_LN_TILES = (20, 3, 5, 12, 13, 6, 7)     # (6 / 7: the three-stage 128 x 256 / 128 x 192 forms -- two K-tiles in flight; offered to the tuner since round 6)


class LnSums:
    """Row statistics as the PRODUCER's partial sums: t [M, slices, 2] f32 = (sum x, sum x^2) per row and tile column of the product that wrote the rows
    (gemm_lnsum), eps of the LayerNorm that will consume them.  gemm_ln takes it in place of layernorm_stats' (mean, 1 / std)."""
    __slots__ = ("t", "eps")

    def __init__(self, t, eps):
        self.t, self.eps = t, float(eps)


def gemm_lnsum(a, w, bias=None, residual=None, tile: int = -1):
    """(out, parts): out = residual + a @ w.T + bias as gemm(), parts [M, slices, 2] f32 = per row and tile column the (sum, sum of squares) of the bf16 values
    written (rga3_gemm_lnsum_bf16) -- LnSums(parts, eps) is the statistics input of the gemm_ln that consumes `out`.  M > 16, bf16, no activation."""
    _need_cuda(a, w, bias, residual)
    assert a.dtype == w.dtype == torch.bfloat16 and a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1] and a.stride(1) == 1 and w.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    assert M > 16 and K % 8 == 0 and a.stride(0) % 8 == 0 and w.stride(0) % 8 == 0
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    ldr = 0
    if residual is not None:
        assert residual.dtype == torch.bfloat16 and residual.shape == (M, N) and residual.stride(1) == 1
        ldr = residual.stride(0)
    L = _lib.load()
    if tile == -1 and M * N * K >= (1 << 24):
        ws = gemm_workspace(a.device)

        def trial(t):      # the plain kernel of the same tiling (same main loop; the tuner only ranks tilings)
            _lib.check(L.rga3_gemm_bf16(a.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(residual), None, out.data_ptr(), M, N, K, a.stride(0), w.stride(0), out.stride(0), ldr,
                                        ACT["none"], BF16, t, ws.data_ptr(), ws.numel(), _stream()), "gemm_bf16")

        cands = _LNSUM_TILES if (N % 192 == 0 and N % 256 != 0 and M >= 1024) else tuple(t for t in _LNSUM_TILES if t != 23)
        tile = _tuner.pick(_tuner.key_of(M, N, K, "lnsum", BF16, bias is not None, residual is not None), trial, candidates=cands)
    if tile not in _LNSUM_TILES:
        tile = -1
    ns = int(L.rga3_gemm_lnsum_slices(N, tile))
    if ns < 1:
        raise _lib.Rga3Error("rga3_gemm_lnsum_slices failed")
    parts = torch.empty((M, ns, 2), dtype=torch.float32, device=a.device)
    _lib.check(L.rga3_gemm_lnsum_bf16(a.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(), M, N, K, a.stride(0), w.stride(0), out.stride(0), ldr,
                                      tile, parts.data_ptr(), _stream()), "gemm_lnsum_bf16")
    return out, parts


def gemm_ln(a, stats, wf, colc, biasf, act: str = "none", out=None, tile: int = -1):
    """act(LayerNorm(a) @ W.T + b) with the LayerNorm folded into the product: a [M, K] UN-normalised bf16 rows, stats = layernorm_stats(a) or the LnSums the
    producer of `a` left (rga3_gemm_lnq_bf16), (wf, colc, biasf) = fold_layernorm(...).  The normalised activations are never materialised."""
    sums = stats if isinstance(stats, LnSums) else None
    if sums is not None:
        stats = sums.t
    _need_cuda(a, stats, wf, colc, biasf, out)
    assert a.dtype == wf.dtype == torch.bfloat16 and a.dim() == 2 and wf.dim() == 2 and a.shape[1] == wf.shape[1] and a.stride(1) == 1 and wf.is_contiguous()
    assert stats.dtype == colc.dtype == torch.float32 and stats.is_contiguous() and colc.numel() == wf.shape[0]
    assert (stats.dim() == 3 and stats.shape[0] == a.shape[0] and stats.shape[2] == 2) if sums is not None else stats.shape == (a.shape[0], 2)
    assert act in ("none", "gelu", "relu") and a.shape[1] % 8 == 0 and wf.shape[0] % 4 == 0
    M, K = a.shape
    N = wf.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.bfloat16
    fn = _lib.load().rga3_gemm_ln_bf16
    fnq = _lib.load().rga3_gemm_lnq_bf16

    def run(t):
        if sums is not None:
            _lib.check(fnq(a.data_ptr(), wf.data_ptr(), _ptr(biasf), colc.data_ptr(), stats.data_ptr(), stats.shape[1], K, sums.eps, out.data_ptr(), M, N, K, a.stride(0),
                           wf.stride(0), out.stride(0), ACT[act], t, _stream()), "gemm_lnq_bf16")
        else:
            _lib.check(fn(a.data_ptr(), wf.data_ptr(), _ptr(biasf), colc.data_ptr(), stats.data_ptr(), out.data_ptr(), M, N, K, a.stride(0), wf.stride(0), out.stride(0),
                          ACT[act], t, _stream()), "gemm_ln_bf16")

    if tile == -1 and M * N * K >= (1 << 24):
        tile = _tuner.pick(_tuner.key_of(M, N, K, "ln+" + act, BF16, biasf is not None, False), run, candidates=_LN_TILES)
        if tile not in _LN_TILES:    # a tiling forced for the plain GEMMs (tuner.force) that this entry point does not have
            tile = -1
    run(tile)
    return out


def gemm_tn(a: torch.Tensor, b: torch.Tensor, out_dtype=torch.bfloat16, out=None, fused_sum: bool = False) -> torch.Tensor:
    """out [M, N] = a^T @ b for a [K, M], b [K, N] bf16 (row stride free): dW = dY^T X without transposing either operand."""
    _need_cuda(a, b, out)
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0]
    assert a.stride(1) == 1 and b.stride(1) == 1
    K, M = a.shape
    N = b.shape[1]
    if M % 8 or N % 8 or a.stride(0) % 8 or b.stride(0) % 8:   # ragged widths: the NT kernel pads its reduction dim, which is what these are there
        return gemm(transpose(a), transpose(b), out_dtype=out_dtype, out=out)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == out_dtype
    ws = cnt = None
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles < 128 and K >= 1024:      # few output tiles over many tokens: give the kernel room to split K (<= 64 f32 slabs)
        ws = torch.empty((min(64, 256 // tiles, max(2, K // 256)) * M * N,), dtype=torch.float32, device=a.device)
        if fused_sum:
            cnt = _tn_counters(a.device)[:128]
        # (cnt stays None by default: the in-launch slab sum -- last workgroup per tile, rga3_gemm_tn_bf16's `counters` -- was measured 3x SLOWER than the second launch it saves:
        #  one workgroup per tile adds Z x 64 KiB behind two agent-scope fences while the chip idles; DESIGN.md 4, rejected list)
    _lib.check(_lib.load().rga3_gemm_tn_bf16(a.data_ptr(), b.data_ptr(), None, out.data_ptr(), M, N, K, a.stride(0), b.stride(0), out.stride(0),
                                             BF16 if out_dtype == torch.bfloat16 else F32, _ptr(ws), ws.numel() * 4 if ws is not None else 0, _ptr(cnt), _stream()),
               "gemm_tn_bf16")
    return out


def gemm_tn_many(pairs, out_dtype=torch.bfloat16):
    """[a_i^T @ b_i for (a_i [K_i, M_i], b_i [K_i, N_i]) in pairs] (<= 4 products, bf16 operands) in ONE launch + one slab-sum launch (rga3_gemm_tn_many) --
    bit-identical to gemm_tn on each pair; products the grouped entry point does not take (ragged widths) fall back to gemm_tn."""
    import ctypes
    n = len(pairs)
    assert 1 <= n <= 4
    ok = all(a.dtype == b.dtype == torch.bfloat16 and a.dim() == b.dim() == 2 and a.shape[0] == b.shape[0] and a.stride(1) == b.stride(1) == 1 and a.shape[1] % 8 == 0 and
             b.shape[1] % 8 == 0 and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 for a, b in pairs)
    if n == 1 or not ok:
        return [gemm_tn(a, b, out_dtype=out_dtype) for a, b in pairs]
    ptrs, dims, outs, need = (ctypes.c_void_p * (3 * n))(), (ctypes.c_int64 * (7 * n))(), [], 0
    for i, (a, b) in enumerate(pairs):
        _need_cuda(a, b)
        K, M = a.shape
        N = b.shape[1]
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
        outs.append(out)
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        nk = (K + 31) // 32
        z = max(1, min(64, 256 // tiles, nk // 8)) if (tiles < 128 and nk >= 32) else 1
        need += z * M * N
        for j, t in enumerate((a, b, out)):
            ptrs[3 * i + j] = t.data_ptr()
        for j, v in enumerate((M, N, K, a.stride(0), b.stride(0), out.stride(0), BF16 if out_dtype == torch.bfloat16 else F32)):
            dims[7 * i + j] = int(v)
    ws = torch.empty((need,), dtype=torch.float32, device=pairs[0][0].device)
    _lib.check(_lib.load().rga3_gemm_tn_many(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(dims, ctypes.c_void_p), n, ws.data_ptr(), ws.numel() * 4, _stream()),
               "gemm_tn_many")
    return outs


_tn_cnt = {}


def _tn_counters(device):
    """128 ticket words per (device, stream or workspace_scope) for the fused slab sum of gemm_tn: zeroed once, the kernel leaves them zero."""
    key = _ws_key(device)
    t = _tn_cnt.get(key)
    if t is None:
        t = _tn_cnt[key] = torch.zeros(256, dtype=torch.int32, device=device)     # [0, 128): gemm_tn tiles; [128, 256): colsum column blocks
    return t


_attn_impl_or = [0]


def set_causal32(on: bool):
    """A/B switch (bench.py variants, tests): False keeps the long causal rows at D = 128 on the general kernel instead of attn_causal32_kernel."""
    _attn_impl_or[0] = 0 if on else 4


def attn_varlen(q, k, v, cu_q, cu_k, max_q: int, scale: float, causal: bool = False, out=None, return_lse=False,
                impl: int = 0, block=None, max_k: int = 0):
    """softmax(q k^T * scale) v over packed variable-length segments.

    q [Tq, Hq, D], k/v [Tk, Hkv, D] (arbitrary token/head strides, unit stride on D); cu_* int32 [nseg+1].
    """
    _need_cuda(q, k, v, cu_q, cu_k)
    assert q.dtype == k.dtype == v.dtype == torch.bfloat16
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    assert cu_q.dtype == torch.int32 and cu_k.dtype == torch.int32
    Tq, Hq, D = q.shape
    Hkv = k.shape[1]
    nseg = cu_q.numel() - 1
    if out is None:
        out = torch.empty((Tq, Hq, D), dtype=torch.bfloat16, device=q.device)
    lse = torch.empty((Hq, Tq), dtype=torch.float32, device=q.device) if return_lse else None
    # few query blocks x heads over a long key range (SAM2 memory attention: one head, 4096 queries, <= 28 736 keys): hand the kernel a
    # workspace so it can split the keys over up to 8 workgroups per query block
    bq, bk = (int(block[0]), int(block[1])) if block else (0, 0)   # block-diagonal visibility inside a segment: (query block, key block), powers of two
    if not max_k and cu_k is cu_q:      # self-attention over the same packing: the longest key segment is the longest query segment (no read-back)
        max_k = int(max_q)
    split_ws = None
    if not causal and ((int(max_q) + 63) // 64) * Hq * nseg < 128 and k.shape[0] >= 1024 and D % 4 == 0:
        if not max_k:    # longest key segment: known to most callers (max_k); reading it back from cu_k is a device -> host sync
            max_k = k.shape[0] if nseg == 1 else int((cu_k[1:] - cu_k[:-1]).max())
        if max_k >= 1024:
            split_ws = torch.empty((8 * Tq * Hq * (D + 1),), dtype=torch.float32, device=q.device)
    rc = _lib.load().rga3_attn_varlen_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _ptr(lse),
                                          cu_q.data_ptr(), cu_k.data_ptr(), nseg, int(max_q), Tq, Hq, Hkv, D,
                                          q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                                          out.stride(0), out.stride(1), float(scale), int(bool(causal)), impl | _attn_impl_or[0], _ptr(split_ws),
                                          split_ws.numel() if split_ws is not None else 0, int(max_k), bq, bk, _stream())
    _lib.check(rc, "attn_varlen_fwd")
    return (out, lse) if return_lse else out


def attn_rope_win_ok(max_q: int, D: int) -> bool:
    """Windowed attention (<= 64 queries per segment): q and k can be rotated while the attention kernel loads them (rga3_attn_varlen_fwd_rope)."""
    return int(max_q) <= 64 and D <= 128 and D % 16 == 0


def attn_varlen_rope(q, k, v, cu_q, cu_k, max_q: int, scale: float, cos, sin, causal: bool = False, rope_k: bool = False, out=None, return_lse=False):
    """attn_varlen with RoPE applied while q is loaded (q un-rotated); rope_k=False: k is already rotated; rope_k=True: k (un-rotated, same token
    packing as q: self-attention) is rotated while it is staged.  cos / sin [T, D] f32."""
    _need_cuda(q, k, v, cu_q, cu_k, cos, sin)
    assert q.dtype == k.dtype == v.dtype == torch.bfloat16 and q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    assert cu_q.dtype == torch.int32 and cu_k.dtype == torch.int32
    Tq, Hq, D = q.shape
    assert cos.dtype == sin.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous() and tuple(cos.shape) == (Tq, D) == tuple(sin.shape)
    assert not rope_k or k.shape[0] == Tq, "rope_k needs keys packed like the queries (self-attention)"
    if out is None:
        out = torch.empty((Tq, Hq, D), dtype=torch.bfloat16, device=q.device)
    lse = torch.empty((Hq, Tq), dtype=torch.float32, device=q.device) if return_lse else None
    rc = _lib.load().rga3_attn_varlen_fwd_rope(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _ptr(lse), cu_q.data_ptr(), cu_k.data_ptr(),
                                               cu_q.numel() - 1, int(max_q), Tq, Hq, k.shape[1], D, q.stride(0), q.stride(1), k.stride(0), k.stride(1),
                                               v.stride(0), v.stride(1), out.stride(0), out.stride(1), float(scale), int(bool(causal)), cos.data_ptr(),
                                               sin.data_ptr(), cos.data_ptr() if rope_k else None, sin.data_ptr() if rope_k else None, _stream())
    _lib.check(rc, "attn_varlen_fwd_rope")
    return (out, lse) if return_lse else out


def rmsnorm(x, weight, eps: float, add=None, return_residual=False):
    """HF RMSNorm on [rows, dim] bf16; with add: normalises (x + add) and optionally returns that sum."""
    _need_cuda(x, weight, add)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    rows, dim = x.shape
    y = torch.empty((rows, dim), dtype=torch.bfloat16, device=x.device)
    res = None
    if add is not None:
        assert add.shape == x.shape and add.stride() == x.stride()
        if return_residual:
            res = torch.empty_strided(x.shape, x.stride(), dtype=torch.bfloat16, device=x.device)
    if x.stride(0) != dim:
        x = x.contiguous()
        add = add.contiguous() if add is not None else None
    rc = _lib.load().rga3_rmsnorm_fwd(x.data_ptr(), _ptr(add), weight.data_ptr(), y.data_ptr(), _ptr(res), rows, dim,
                                      x.stride(0), float(eps), _stream())
    _lib.check(rc, "rmsnorm_fwd")
    return (y, res) if return_residual else y


def layernorm(x, weight, bias, eps: float, act: str = "none", out=None):
    """LayerNorm over the last dim of [rows, dim] (row stride free); act="gelu" fuses the exact GELU."""
    _need_cuda(x, weight, bias)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    rows, dim = x.shape
    y = out if out is not None else torch.empty((rows, dim), dtype=torch.bfloat16, device=x.device)
    rc = _lib.load().rga3_layernorm_fwd(x.data_ptr(), weight.data_ptr(), _ptr(bias), y.data_ptr(), rows, dim, x.stride(0),
                                        y.stride(0), float(eps), 1 if act == "gelu" else 0, _stream())
    _lib.check(rc, "layernorm_fwd")
    return y


def rope_(x, cos, sin, h0: int, nh: int):
    """In-place rotate heads [h0, h0+nh) of x [T, H, D]; cos/sin [T, D] fp32."""
    _need_cuda(x, cos, sin)
    assert x.dtype == torch.bfloat16 and x.dim() == 3 and x.stride(2) == 1
    assert cos.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape == (x.shape[0], x.shape[2])
    rc = _lib.load().rga3_rope_inplace(x.data_ptr(), cos.data_ptr(), sin.data_ptr(), x.shape[0], h0, nh, x.shape[2],
                                       x.stride(0), x.stride(1), _stream())
    _lib.check(rc, "rope_inplace")
    return x


def gather_rows(table, idx, rows_per_idx: int = 1):
    """out[i*r + j] = table[idx[i]*r + j]"""
    _need_cuda(table, idx)
    assert table.dtype == torch.bfloat16 and table.dim() == 2 and table.stride(1) == 1 and idx.dtype == torch.int64
    n = idx.numel()
    out = torch.empty((n * rows_per_idx, table.shape[1]), dtype=torch.bfloat16, device=table.device)
    rc = _lib.load().rga3_gather_rows(table.data_ptr(), idx.data_ptr(), out.data_ptr(), n, rows_per_idx, table.shape[1],
                                      table.stride(0), out.stride(0), _stream())
    _lib.check(rc, "gather_rows")
    return out


def scatter_rows_(out, idx, src, rows_per_idx: int = 1):
    """out[idx[i]*r + j] = src[i*r + j] (in place)."""
    _need_cuda(out, idx, src)
    assert out.dtype == src.dtype == torch.bfloat16 and idx.dtype == torch.int64
    assert out.dim() == 2 and src.dim() == 2 and out.stride(1) == 1 and src.stride(1) == 1 and out.shape[1] == src.shape[1]
    n = idx.numel()
    if n == 0:
        return out
    rc = _lib.load().rga3_scatter_rows(src.data_ptr(), idx.data_ptr(), out.data_ptr(), n, rows_per_idx, src.shape[1],
                                       src.stride(0), out.stride(0), _stream())
    _lib.check(rc, "scatter_rows")
    return out


def silu_mul(a, b):
    _need_cuda(a, b)
    assert a.dtype == b.dtype == torch.bfloat16 and a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    out = torch.empty_like(a)
    _lib.check(_lib.load().rga3_silu_mul(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "silu_mul")
    return out


def add(a, b):
    _need_cuda(a, b)
    assert a.dtype == b.dtype == torch.bfloat16 and a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    out = torch.empty_like(a)
    _lib.check(_lib.load().rga3_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "add")
    return out


def cross_entropy_rows(logits, labels, want_grad=False, grad_scale: float = 1.0):
    """Per-row CE with ignore_index=-100 (rows with negative labels give 0)."""
    _need_cuda(logits, labels)
    assert logits.dim() == 2 and logits.stride(1) == 1 and labels.dtype == torch.int64 and labels.numel() == logits.shape[0]
    rows, V = logits.shape
    loss = torch.empty((rows,), dtype=torch.float32, device=logits.device)
    dl = torch.empty((rows, V), dtype=torch.bfloat16, device=logits.device) if want_grad else None
    dt = BF16 if logits.dtype == torch.bfloat16 else F32
    ld = logits.stride(0)
    if dl is not None and ld != V:
        logits = logits.contiguous()
        ld = V
    rc = _lib.load().rga3_cross_entropy_rows(logits.data_ptr(), dt, labels.data_ptr(), loss.data_ptr(), _ptr(dl), rows, V,
                                             ld, float(grad_scale), _stream())
    _lib.check(rc, "cross_entropy_rows")
    return (loss, dl) if want_grad else loss


# ------------------------------------------------------------------------------------------------ SAM2-side kernels
def im2col(img, ks: int, stride: int, pad: int):
    """NCHW bf16 images -> [F*Ho*Wo, ld] patch rows, columns (c, kh, kw), ld = C*ks*ks rounded up to 8."""
    _need_cuda(img)
    assert img.dtype == torch.bfloat16 and img.dim() == 4 and img.is_contiguous()
    F, C, H, W = img.shape
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    ld = (C * ks * ks + 7) // 8 * 8
    out = torch.empty((F * Ho * Wo, ld), dtype=torch.bfloat16, device=img.device)
    _lib.check(_lib.load().rga3_im2col(img.data_ptr(), out.data_ptr(), F, C, H, W, ks, stride, pad, ld, _stream()), "im2col")
    return out, (Ho, Wo)


def maxpool2x2_win(x, nwin: int, w: int):
    """x [nwin*w*w, C] window-major tokens (row stride free) -> [nwin*(w/2)^2, C]."""
    _need_cuda(x)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1 and x.shape[0] == nwin * w * w
    C = x.shape[1]
    y = torch.empty((nwin * (w // 2) ** 2, C), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().rga3_maxpool2x2_win(x.data_ptr(), y.data_ptr(), nwin, w, C, x.stride(0), C, _stream()), "maxpool2x2_win")
    return y


def upsample2x_add(a, b, F: int, H: int, W: int):
    """a [F*H*W, C] + nearest-x2(b [F*(H/2)*(W/2), C]) (raster token order)."""
    _need_cuda(a, b)
    assert a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype == torch.bfloat16
    out = torch.empty_like(a)
    _lib.check(_lib.load().rga3_upsample2x_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), F, H, W, a.shape[1], _stream()), "upsample2x_add")
    return out


def add_bcast(a, b, alpha: float = 1.0):
    """a [rows, C] + alpha * b[rows_b, C] with b broadcast (row r uses b[r % rows_b])."""
    _need_cuda(a, b)
    assert a.dtype == b.dtype == torch.bfloat16 and a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[1]
    assert a.stride(1) == 1 and b.stride(1) == 1 and a.shape[0] % b.shape[0] == 0
    out = torch.empty((a.shape[0], a.shape[1]), dtype=torch.bfloat16, device=a.device)
    rc = _lib.load().rga3_add_bcast(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape[0], b.shape[0], a.shape[1], a.stride(0), b.stride(0),
                                    out.stride(0), float(alpha), _stream())
    _lib.check(rc, "add_bcast")
    return out


def bilinear(x, size, plane_idx=None):
    """F.interpolate(bilinear, align_corners=False) on planes x [N, Hi, Wi] (f32 / bf16) -> f32 [N_out, Ho, Wo]."""
    _need_cuda(x, plane_idx)
    assert x.dim() == 3 and x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16)
    n_out = x.shape[0] if plane_idx is None else plane_idx.numel()
    if plane_idx is not None:
        assert plane_idx.dtype == torch.int32
    out = torch.empty((n_out, size[0], size[1]), dtype=torch.float32, device=x.device)
    rc = _lib.load().rga3_bilinear(x.data_ptr(), BF16 if x.dtype == torch.bfloat16 else F32, out.data_ptr(), _ptr(plane_idx), n_out, x.shape[1],
                                   x.shape[2], size[0], size[1], _stream())
    _lib.check(rc, "bilinear")
    return out


_conv_pack = {}


def decimg_rows(keys, pe, kt, vt, nk: int, q_proj, out_proj, norm, eps: float, k_next=None, v_next=None, scale: float = 0.25, v_transposed: bool = False):
    """The image side of a mask-decoder block boundary in one launch (csrc/decimg.hip): keys [M, 256], pe [hw, 256] (broadcast over frames), kt / vt [frames * nk, 128]
    token-side keys / values (projected), q_proj / out_proj / k_next / v_next = (weight, bias) pairs, norm = (weight, bias).  -> (keys', k2, v2) (k2 = v2 = None
    without k_next / v_next); v_transposed: v2 comes as [frames * 128, hw] and k2 head-major [8, M, 16] (the layouts attn_fewq reads)."""
    _need_cuda(keys, pe, kt, vt, *q_proj, *out_proj, *norm)
    assert keys.dtype == torch.bfloat16 and keys.dim() == 2 and keys.shape[1] == 256 and keys.stride(1) == 1 and pe.shape[1] == 256 and pe.stride(1) == 1
    M, hw = keys.shape[0], pe.shape[0]
    assert M % hw == 0 and hw >= 16 and 1 <= nk <= 16 and kt.is_contiguous() and vt.is_contiguous() and tuple(kt.shape) == tuple(vt.shape) == (M // hw * nk, 128)
    (wq, bq), (wo, bo), (gw, gb) = q_proj, out_proj, norm
    assert tuple(wq.shape) == (128, 256) and tuple(wo.shape) == (256, 128) and wq.is_contiguous() and wo.is_contiguous()
    out = torch.empty((M, 256), dtype=torch.bfloat16, device=keys.device)
    k2 = v2 = None
    wk2 = bk2 = wv2 = bv2 = None
    if k_next is not None:
        (wk2, bk2), (wv2, bv2) = k_next, v_next
        assert tuple(wk2.shape) == tuple(wv2.shape) == (128, 256) and wk2.is_contiguous() and wv2.is_contiguous()
        k2 = torch.empty((8, M, 16) if v_transposed else (M, 128), dtype=torch.bfloat16, device=keys.device)
        assert not v_transposed or hw % 16 == 0
        v2 = torch.empty((M // hw * 128, hw) if v_transposed else (M, 128), dtype=torch.bfloat16, device=keys.device)
    _lib.check(_lib.load().rga3_decimg_rows(keys.data_ptr(), keys.stride(0), pe.data_ptr(), pe.stride(0), hw, kt.data_ptr(), vt.data_ptr(), int(nk), wq.data_ptr(), _ptr(bq),
                                            wo.data_ptr(), _ptr(bo), gw.data_ptr(), _ptr(gb), float(eps), _ptr(wk2), _ptr(bk2), _ptr(wv2), _ptr(bv2), out.data_ptr(), 256,
                                            _ptr(k2), _ptr(v2), 128, 1 if (v_transposed and k_next is not None) else 0, float(scale), M, _stream()), "decimg_rows")
    return out, k2, v2


def attn_fewq(q, k, vt, nq: int, nk: int, H: int, scale: float, vbias=None, k_head_major: bool = False):
    """softmax(scale q k^T) v (+ vbias) for nq <= 16 queries over nk <= 4096 keys per frame, heads of 16 dims, one launch without a merge (csrc/decimg.hip):
    q [frames * nq, H * 16], k [frames * nk, H * 16] (row stride free) or, k_head_major, [H, frames * nk, 16] (what decimg_rows(v_transposed=True) writes),
    vt [frames * H * 16, nk] the values transposed.  -> [frames * nq, H * 16] bf16."""
    _need_cuda(q, k, vt, vbias)
    assert q.dtype == k.dtype == vt.dtype == torch.bfloat16 and q.stride(1) == 1 and vt.is_contiguous() and q.shape[1] == 16 * H
    frames = q.shape[0] // nq
    assert q.shape[0] == frames * nq and tuple(vt.shape) == (frames * H * 16, nk) and nq <= 16 and nk <= 4096 and nk % 4 == 0
    if k_head_major:
        assert k.is_contiguous() and k.numel() == H * frames * nk * 16
        kst, khst = 16, frames * nk * 16
    else:
        assert k.dim() == 2 and k.stride(1) == 1 and tuple(k.shape) == (frames * nk, 16 * H)
        kst, khst = k.stride(0), 16
    out = torch.empty((frames * nq, 16 * H), dtype=torch.bfloat16, device=q.device)
    _lib.check(_lib.load().rga3_attn_fewq(q.data_ptr(), q.stride(0), k.data_ptr(), kst, khst, vt.data_ptr(), _ptr(vbias), out.data_ptr(), out.stride(0), frames, int(nq),
                                          int(nk), int(H), float(scale), _stream()), "attn_fewq")
    return out


def copy_many(pairs):
    """[(dst, src), ...] dense tensors of equal byte size -> dst[i] <- src[i], 24 copies per launch (csrc/sam2ops.hip copy_many); pairs that are not 16-byte
    shaped / aligned go through Tensor.copy_."""
    import ctypes
    todo = []
    for d, s in pairs:
        nb = d.numel() * d.element_size()
        if (d.is_cuda and s.is_cuda and d.is_contiguous() and s.is_contiguous() and d.dtype == s.dtype and d.numel() == s.numel() and nb > 0 and nb % 16 == 0
                and d.data_ptr() % 16 == 0 and s.data_ptr() % 16 == 0):
            todo.append((d.data_ptr(), s.data_ptr(), nb))
        else:
            d.copy_(s)
    for i0 in range(0, len(todo), 24):
        part = todo[i0:i0 + 24]
        n = len(part)
        dp, sp, nb = (ctypes.c_void_p * n)(*[p_[0] for p_ in part]), (ctypes.c_void_p * n)(*[p_[1] for p_ in part]), (ctypes.c_int64 * n)(*[p_[2] for p_ in part])
        _lib.check(_lib.load().rga3_copy_many(ctypes.cast(dp, ctypes.c_void_p), ctypes.cast(sp, ctypes.c_void_p), ctypes.cast(nb, ctypes.c_void_p), n, _stream()),
                   "copy_many")


def conv3x3s2_ln_gelu_ok(x, weight) -> bool:
    """The fused narrow stages: 1 -> 4 channels from an f32 plane, 4 -> 16 from bf16."""
    co, ci = weight.shape[0], weight.shape[1]
    return (x.dtype == torch.float32 and (ci, co) == (1, 4)) or (x.dtype == torch.bfloat16 and (ci, co) == (4, 16))


def conv3x3s2_ln_gelu(x, weight, bias, ln_w, ln_b, eps: float, F: int, H: int, W: int, sig_scale: float = 0.0, sig_bias: float = 0.0):
    """conv3x3s2 + LayerNorm over the output channels + exact GELU in one launch (csrc/sam2ops.hip); the arithmetic of conv3x3s2 followed by layernorm(act="gelu")."""
    _need_cuda(x, weight, bias, ln_w, ln_b)
    Cout, Cin = weight.shape[0], weight.shape[1]
    assert conv3x3s2_ln_gelu_ok(x, weight) and weight.dtype == torch.bfloat16 and weight.is_contiguous() and x.is_contiguous() and x.numel() == F * H * W * Cin
    y = torch.empty((F * (H // 2) * (W // 2), Cout), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().rga3_conv3x3s2_ln_gelu(x.data_ptr(), F32 if x.dtype == torch.float32 else BF16, weight.data_ptr(), _ptr(bias), ln_w.data_ptr(), _ptr(ln_b),
                                                  float(eps), y.data_ptr(), F, H, W, Cin, Cout, float(sig_scale), float(sig_bias), _stream()), "conv3x3s2_ln_gelu")
    return y


def conv3x3s2(x, weight, bias, F: int, H: int, W: int, sig_scale: float = 0.0, sig_bias: float = 0.0):
    """token-major x [F*H*W, Cin] (bf16, or f32 single plane with the sigmoid affine) -> [F*(H/2)*(W/2), Cout]."""
    _need_cuda(x, weight, bias)
    Cout, Cin = weight.shape[0], weight.shape[1]
    assert weight.dtype == torch.bfloat16 and weight.is_contiguous() and x.is_contiguous() and x.numel() == F * H * W * Cin
    if x.dtype == torch.bfloat16 and Cin % 8 == 0 and Cin >= 16:
        # wide stages: im2col (16-byte copies) + the NT GEMM on the weight repacked [Cout, (kh, kw, ci)] (cached per weight version)
        key = (weight.data_ptr(), weight._version)
        wp = _conv_pack.get(key)
        if wp is None:
            if len(_conv_pack) > 64:
                _conv_pack.clear()
            wp = _conv_pack[key] = weight.detach().permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
        cols = torch.empty((F * (H // 2) * (W // 2), 9 * Cin), dtype=torch.bfloat16, device=x.device)
        _lib.check(_lib.load().rga3_im2col3x3s2(x.data_ptr(), cols.data_ptr(), F, H, W, Cin, _stream()), "im2col3x3s2")
        return gemm(cols, wp, bias)
    y = torch.empty((F * (H // 2) * (W // 2), Cout), dtype=torch.bfloat16, device=x.device)
    rc = _lib.load().rga3_conv3x3s2(x.data_ptr(), F32 if x.dtype == torch.float32 else BF16, weight.data_ptr(), _ptr(bias), y.data_ptr(), F, H, W,
                                    Cin, Cout, float(sig_scale), float(sig_bias), _stream())
    _lib.check(rc, "conv3x3s2")
    return y


def dwconv7x7(x, weight, bias, F: int, H: int, W: int):
    _need_cuda(x, weight, bias)
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and weight.is_contiguous()
    y = torch.empty_like(x)
    _lib.check(_lib.load().rga3_dwconv7x7(x.data_ptr(), weight.data_ptr(), _ptr(bias), y.data_ptr(), F, H, W, x.shape[1], _stream()), "dwconv7x7")
    return y


def rope_axial_(x, cos, sin, n_rope: int):
    """In-place axial complex RoPE on x [T, C] (row stride free); rows t < n_rope use table row t % nq; cos/sin [nq, C/2] f32."""
    _need_cuda(x, cos, sin)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1 and cos.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous()
    assert cos.shape[1] * 2 == x.shape[1]
    _lib.check(_lib.load().rga3_rope_axial_inplace(x.data_ptr(), cos.data_ptr(), sin.data_ptr(), int(n_rope), cos.shape[0], x.shape[1], x.stride(0),
                                                  _stream()), "rope_axial")
    return x


def mlp3_rows(specs, B: int):
    """Several 3-layer MLPs (Linear+ReLU, Linear+ReLU, Linear [+ sigmoid]) on one row per frame in ONE launch (csrc/dechead.hip).
    specs: list of (x, x_frame_stride, (w0, b0, w1, b1, w2, b2), sigmoid[, y]) with x a bf16 tensor whose data_ptr is frame 0's row and y an optional [B, out] view
    to write into (any frame stride).  -> list of [B, out] bf16."""
    import ctypes
    n = len(specs)
    assert 1 <= n <= 8
    ptrs, dims, outs, keep = (ctypes.c_void_p * (8 * n))(), (ctypes.c_int64 * (6 * n))(), [], []
    for i, sp in enumerate(specs):
        x, xs, (w0, b0, w1, b1, w2, b2), sig = sp[:4]
        _need_cuda(x, w0, b0, w1, b1, w2, b2)
        for t in (w0, w1, w2):
            assert t.dtype == torch.bfloat16 and t.is_contiguous()
        assert x.dtype == torch.bfloat16 and tuple(w1.shape) == (w0.shape[0], w0.shape[0]) and w2.shape[1] == w0.shape[0], (w0.shape, w1.shape, w2.shape)
        y = sp[4] if len(sp) > 4 and sp[4] is not None else torch.empty((B, w2.shape[0]), dtype=torch.bfloat16, device=x.device)
        assert y.dtype == torch.bfloat16 and tuple(y.shape) == (B, w2.shape[0]) and y.stride(1) == 1
        for j, t in enumerate((x, w0, b0, w1, b1, w2, b2, y)):
            ptrs[8 * i + j] = _ptr(t)
        for j, v in enumerate((int(xs), y.stride(0), w0.shape[1], w0.shape[0], w2.shape[0], 1 if sig else 0)):
            dims[6 * i + j] = v
        outs.append(y)
    _lib.check(_lib.load().rga3_mlp3_rows(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(dims, ctypes.c_void_p), n, int(B), _stream()), "mlp3_rows")
    return outs


def sam_select_objptr(iou, obj, toks, proj, no_obj_ptr):
    """-> (best [B] int64, sel [B] int32, sel64 [B] int64, obj_ptr [B, C] bf16): argmax over IoU 1..3, plane index b*4+1+best, obj_ptr_proj of the chosen multimask token, gated by obj > 0.
    iou [B, 4], obj [B, 1] bf16 contiguous; toks [B, 4, C] bf16 (any frame stride); proj = (w0, b0, w1, b1, w2, b2)."""
    _need_cuda(iou, obj, toks, no_obj_ptr, *proj)
    B, _, C = toks.shape
    assert iou.dtype == obj.dtype == toks.dtype == torch.bfloat16 and iou.is_contiguous() and obj.is_contiguous() and toks.stride(2) == 1 and toks.stride(1) == C
    best = torch.empty((2, B), dtype=torch.int64, device=toks.device)
    sel = torch.empty(B, dtype=torch.int32, device=toks.device)
    ptr = torch.empty((B, C), dtype=torch.bfloat16, device=toks.device)
    w0, b0, w1, b1, w2, b2 = proj
    _lib.check(_lib.load().rga3_sam_select_objptr(iou.data_ptr(), obj.data_ptr(), toks.data_ptr(), toks.stride(0), C, w0.data_ptr(), _ptr(b0), w1.data_ptr(), _ptr(b1),
                                                  w2.data_ptr(), _ptr(b2), no_obj_ptr.data_ptr(), best.data_ptr(), sel.data_ptr(), ptr.data_ptr(), B, _stream()),
               "sam_select_objptr")
    return best[0], sel, best[1], ptr


_memattn_ws = {}


def memattn_cross(q, k, m, scale: float, nsplit: int = 0, partials: bool = False):
    """SAM2 memory cross-attention with the values kept in memory space (csrc/memattn.hip): softmax(scale q k^T) m -> [Nq, 64] bf16.
    q [Nq, 256], k [Nk, 256] bf16 (projected and rotated; row strides free), m [Nk, 64] bf16 the un-projected memory rows.  The caller applies the value
    projection to the result.  nsplit = 0 picks the number of key slices that fills the chip (one 256-row query block x slice per CU)."""
    _need_cuda(q, k, m)
    assert q.dtype == k.dtype == m.dtype == torch.bfloat16 and q.dim() == k.dim() == m.dim() == 2
    assert q.shape[1] == 256 and k.shape[1] == 256 and m.shape[1] == 64 and k.shape[0] == m.shape[0], (q.shape, k.shape, m.shape)
    assert q.stride(1) == 1 and k.stride(1) == 1 and m.stride(1) == 1
    Nq, Nk = q.shape[0], k.shape[0]
    if nsplit <= 0:
        nqb = (Nq + 255) // 256
        nsplit = max(1, min(32, 256 // nqb, (Nk + 63) // 64))
    L = _lib.load()
    n = int(L.rga3_memattn_cross_ws_floats(Nq, nsplit))
    if n < 0:
        raise _lib.Rga3Error("memattn_cross: bad workspace query")
    key = _ws_key(q.device)
    ws = _memattn_ws.get(key)
    if ws is None or ws.numel() < n:
        ws = _memattn_ws[key] = torch.empty(n, dtype=torch.float32, device=q.device)
    if partials:     # leave the per-slice partial results in the workspace: (sums [nsplit, Nq, 64], (max, sum) [nsplit, Nq, 2], nsplit) for memlayer_rows
        _lib.check(L.rga3_memattn_cross(q.data_ptr(), k.data_ptr(), m.data_ptr(), None, Nq, Nk, q.stride(0), k.stride(0), m.stride(0), 0,
                                        float(scale), int(nsplit), ws.data_ptr(), _stream()), "memattn_cross")
        return ws[:nsplit * Nq * 64], ws[nsplit * Nq * 64:nsplit * Nq * 66], nsplit
    out = torch.empty((Nq, 64), dtype=torch.bfloat16, device=q.device)
    _lib.check(L.rga3_memattn_cross(q.data_ptr(), k.data_ptr(), m.data_ptr(), out.data_ptr(), Nq, Nk, q.stride(0), k.stride(0), m.stride(0), out.stride(0),
                                    float(scale), int(nsplit), ws.data_ptr(), _stream()), "memattn_cross")
    return out


def memlayer_rows(res, ln, eps: float, a=None, partials=None, w1=None, b1=None, want_x=True, want_t=False, w2=None, b2=None, rope=None, rope_cols: int = 0):
    """The row-wise chain between the attention kernels of a SAM2 memory-attention layer in one launch (csrc/memlayer.hip), model width 256:
         x' = bf16(bf16(a w1^T + b1) + res)   (a [M, 64 | 256] rows, or partials = memattn_cross(..., partials=True); without w1: x' = res)
         t  = LayerNorm(x'; ln = (weight, bias), eps)
         y  = bf16(t w2^T + b2), columns < rope_cols rotated by rope = (cos, sin) [nq, 128] f32 (row = token % nq)
    -> (x' or None, t or None, y or None)."""
    wln, bln = ln
    _need_cuda(res, wln, bln, a, w1, b1, w2, b2)
    assert res.dtype == torch.bfloat16 and res.dim() == 2 and res.shape[1] == 256 and res.stride(1) == 1
    M, dev = res.shape[0], res.device
    x = t = y = None
    K1, po, pml, ns = 0, None, None, 0
    if w1 is not None:
        assert w1.dtype == torch.bfloat16 and w1.is_contiguous() and w1.shape[0] == 256
        K1 = w1.shape[1]
        if partials is not None:
            po, pml, ns = partials
            assert a is None and K1 == 64 and po.numel() == ns * M * 64 and pml.numel() == ns * M * 2
        else:
            assert a is not None and a.dtype == torch.bfloat16 and tuple(a.shape) == (M, K1) and a.stride(1) == 1
        if want_x:
            x = torch.empty((M, 256), dtype=torch.bfloat16, device=dev)
    else:
        assert a is None and partials is None
    if want_t or w2 is None:
        t = torch.empty((M, 256), dtype=torch.bfloat16, device=dev)
    N2, cos, sin, nq = 0, None, None, 0
    if w2 is not None:
        assert w2.dtype == torch.bfloat16 and w2.is_contiguous() and w2.shape[1] == 256 and w2.shape[0] in (256, 768)
        N2 = w2.shape[0]
        y = torch.empty((M, N2), dtype=torch.bfloat16, device=dev)
        if rope_cols:
            cos, sin = rope
            assert cos.dtype == sin.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape[1] == 128 and cos.shape == sin.shape
            nq = cos.shape[0]
    _lib.check(_lib.load().rga3_memlayer_rows(_ptr(a), a.stride(0) if a is not None else 0, K1, _ptr(po), _ptr(pml), ns, _ptr(w1), _ptr(b1), res.data_ptr(), res.stride(0),
                                              _ptr(x), 256, wln.data_ptr(), _ptr(bln), float(eps), _ptr(t), 256, _ptr(w2), _ptr(b2), N2, _ptr(y), N2 if N2 else 0,
                                              _ptr(cos), _ptr(sin), int(rope_cols), nq, M, _stream()), "memlayer_rows")
    return x, t, y


def pixel_shuffle2x(g, bias, add, F: int, H: int, W: int, act: str = "none"):
    """g [F*H*W, 4*Co] -> [F*2H*2W, Co] (+bias, +add, then optional GELU)."""
    _need_cuda(g, bias, add)
    Co = g.shape[1] // 4
    assert g.dtype == torch.bfloat16 and g.is_contiguous()
    out = torch.empty((F * 4 * H * W, Co), dtype=torch.bfloat16, device=g.device)
    _lib.check(_lib.load().rga3_pixel_shuffle2x(g.data_ptr(), _ptr(bias), _ptr(add), out.data_ptr(), F, H, W, Co, 1 if act == "gelu" else 0, _stream()),
               "pixel_shuffle2x")
    return out


def bce_dice_sums(logits, targets):
    """logits/targets f32 [n, H, W] -> f32 [n, 4] = {sum bce, sum sig*t, sum sig, sum t}."""
    _need_cuda(logits, targets)
    assert logits.dtype == targets.dtype == torch.float32 and logits.is_contiguous() and targets.is_contiguous() and logits.shape == targets.shape
    n = logits.shape[0]
    out = torch.empty((n, 4), dtype=torch.float32, device=logits.device)
    if n == 0:
        return out
    L = _lib.load()
    hw = logits[0].numel()
    nws = int(L.rga3_bce_dice_sums_ws_floats(n, hw))
    ws = torch.empty(nws, dtype=torch.float32, device=logits.device)
    _lib.check(L.rga3_bce_dice_sums_det(logits.data_ptr(), targets.data_ptr(), out.data_ptr(), ws.data_ptr(), nws, n, hw, _stream()), "bce_dice_sums_det")   # reproducible sums
    return out


# ------------------------------------------------------------------------------------------------ training-step kernels
def attn_varlen_bwd(q, k, v, o, dout, lse, cu_q, cu_k, max_q: int, max_k: int, scale: float, causal: bool, dq=None, dk=None, dv=None):
    """Gradients of attn_varlen w.r.t. q, k, v (bf16, same [T, H, D] shapes; outputs may be strided views)."""
    import ctypes
    _need_cuda(q, k, v, o, dout, lse)
    Tq, Hq, D = q.shape
    Hkv = k.shape[1]
    dq = torch.empty_like(q) if dq is None else dq
    dk = torch.empty((k.shape[0], Hkv, D), dtype=torch.bfloat16, device=q.device) if dk is None else dk
    dv = torch.empty((k.shape[0], Hkv, D), dtype=torch.bfloat16, device=q.device) if dv is None else dv
    for t in (q, k, v, o, dout, dq, dk, dv):
        assert t.dtype == torch.bfloat16 and t.stride(2) == 1
    delta = torch.empty((Hq, Tq), dtype=torch.float32, device=q.device)
    Tk = k.shape[0]
    dkv_ws = torch.empty((2 * Hq * Tk * D,), dtype=torch.float32, device=q.device) if (Hq > Hkv and D % 4 == 0) else None
    st = (ctypes.c_int64 * 16)(q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1), o.stride(0), o.stride(1),
                               dout.stride(0), dout.stride(1), dq.stride(0), dq.stride(1), dk.stride(0), dk.stride(1), dv.stride(0), dv.stride(1))
    rc = _lib.load().rga3_attn_varlen_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), dout.data_ptr(), lse.data_ptr(), dq.data_ptr(),
                                          dk.data_ptr(), dv.data_ptr(), delta.data_ptr(), cu_q.data_ptr(), cu_k.data_ptr(), cu_q.numel() - 1, int(max_q),
                                          int(max_k), Tq, Hq, Hkv, D, ctypes.cast(st, ctypes.c_void_p), float(scale), int(bool(causal)), _ptr(dkv_ws), Tk, _stream())
    _lib.check(rc, "attn_varlen_bwd")
    return dq, dk, dv


def rmsnorm_bwd(x, weight, dy, eps: float, add=None):
    _need_cuda(x, weight, dy, add)
    assert x.is_contiguous() and dy.is_contiguous() and x.dtype == dy.dtype == torch.bfloat16 and (add is None or add.is_contiguous())
    dx = torch.empty_like(x)
    _lib.check(_lib.load().rga3_rmsnorm_bwd(x.data_ptr(), weight.data_ptr(), dy.data_ptr(), _ptr(add), dx.data_ptr(), x.shape[0], x.shape[1], float(eps),
                                            _stream()), "rmsnorm_bwd")
    return dx


def quant_fp8_rows(x):
    """bf16 [rows, K] (row stride free, K % 8 == 0) -> (uint8 [rows, K] holding OCP e4m3, f32 scales [rows]): q = e4m3(x / scale), scale = amax / 448."""
    _need_cuda(x)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] % 8 == 0
    rows, K = x.shape
    q = torch.empty((rows, K), dtype=torch.uint8, device=x.device)
    sc = torch.empty((rows,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().rga3_quant_fp8_rows(x.data_ptr(), q.data_ptr(), sc.data_ptr(), rows, K, x.stride(0), q.stride(0), _stream()), "quant_fp8_rows")
    return q, sc


def gemm_fp8(aq, sa, wq, sw, bias=None, residual=None, out=None):
    """(aq [M,K] e4m3 as uint8, sa [M]) x (wq [N,K], sw [N]) -> bf16 [M,N] = (aq . wq^T) * sa * sw (+ bias) (+ residual); K % 128 == 0."""
    _need_cuda(aq, wq, sa, sw, bias, residual)
    assert aq.dtype == torch.uint8 and wq.dtype == torch.uint8 and aq.stride(1) == 1 and wq.stride(1) == 1 and aq.shape[1] == wq.shape[1]
    M, K = aq.shape
    N = wq.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=aq.device)
    assert out.dtype == torch.bfloat16 and out.stride(1) == 1 and sa.dtype == torch.float32 and sw.dtype == torch.float32
    _lib.check(_lib.load().rga3_gemm_fp8(aq.data_ptr(), wq.data_ptr(), sa.data_ptr(), sw.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(), M, N, K,
                                         aq.stride(0), wq.stride(0), out.stride(0), residual.stride(0) if residual is not None else 0, _stream()), "gemm_fp8")
    return out


def swiglu_fwd_quant(gu):
    """(e4m3 [T, I], scales [T]) of silu(gate) * up, as quant_fp8_rows(swiglu_fwd(gu)) gives them, without the bf16 tensor in between."""
    _need_cuda(gu)
    assert gu.dtype == torch.bfloat16 and gu.is_contiguous() and gu.shape[1] % 32 == 0
    T, I = gu.shape[0], gu.shape[1] // 2
    q = torch.empty((T, I), dtype=torch.uint8, device=gu.device)
    sc = torch.empty((T,), dtype=torch.float32, device=gu.device)
    _lib.check(_lib.load().rga3_swiglu_fwd_quant_fp8(gu.data_ptr(), q.data_ptr(), sc.data_ptr(), T, I, _stream()), "swiglu_fwd_quant_fp8")
    return q, sc


def swiglu_bwd_quant(gu, da):
    """(e4m3 [T, 2I], scales [T]) of the SwiGLU backward, as quant_fp8_rows(swiglu_bwd(gu, da)) gives them."""
    _need_cuda(gu, da)
    assert gu.dtype == da.dtype == torch.bfloat16 and gu.is_contiguous() and da.is_contiguous() and gu.shape[1] == 2 * da.shape[1]
    T, I = da.shape
    q = torch.empty((T, 2 * I), dtype=torch.uint8, device=gu.device)
    sc = torch.empty((T,), dtype=torch.float32, device=gu.device)
    _lib.check(_lib.load().rga3_swiglu_bwd_quant_fp8(gu.data_ptr(), da.data_ptr(), q.data_ptr(), sc.data_ptr(), T, I, _stream()), "swiglu_bwd_quant_fp8")
    return q, sc


def dropout(x, p: float, seed: int, out=None, accumulate: bool = False):
    """out = (accumulate ? out : 0) + dropout(x) with the counter-hash mask of (seed, element index); x bf16 contiguous, numel % 8 == 0."""
    _need_cuda(x, out)
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.numel() % 8 == 0
    if out is None:
        assert not accumulate
        out = torch.empty_like(x)
    assert out.is_contiguous() and out.shape == x.shape and out.dtype == torch.bfloat16
    _lib.check(_lib.load().rga3_dropout_bf16(x.data_ptr(), out.data_ptr(), x.numel(), float(p), int(seed) & 0x7FFFFFFFFFFFFFFF, int(bool(accumulate)), _stream()),
               "dropout")
    return out


def dropout_pair(xa, pa: float, seed_a: int, xb, pb: float, seed_b: int, accumulate_into=None):
    """The two LoRA branches of a layer in one launch (rga3_dropout_pair_bf16).  accumulate_into None: (dropout_a(xa), dropout_b(xb)) -- xa may be xb (read once);
    else accumulate_into += dropout_a(xa) + dropout_b(xb) with one rounding, returns it.  Masks as dropout()'s for the same seeds."""
    _need_cuda(xa, xb, accumulate_into)
    assert xa.dtype == xb.dtype == torch.bfloat16 and xa.is_contiguous() and xb.is_contiguous() and xa.shape == xb.shape and xa.numel() % 8 == 0
    L = _lib.load()
    if accumulate_into is None:
        ya, yb = torch.empty_like(xa), torch.empty_like(xb)
        _lib.check(L.rga3_dropout_pair_bf16(xa.data_ptr(), xb.data_ptr(), ya.data_ptr(), yb.data_ptr(), xa.numel(), float(pa), int(seed_a), float(pb), int(seed_b), 0,
                                            _stream()), "dropout_pair_bf16")
        return ya, yb
    out = accumulate_into
    assert out.dtype == torch.bfloat16 and out.is_contiguous() and out.shape == xa.shape
    _lib.check(L.rga3_dropout_pair_bf16(xa.data_ptr(), xb.data_ptr(), out.data_ptr(), None, xa.numel(), float(pa), int(seed_a), float(pb), int(seed_b), 1, _stream()),
               "dropout_pair_bf16")
    return out


def swiglu_fwd(gu):
    """silu(gate) * up from interleaved pre-activations gu [T, 2I] -> [T, I]."""
    _need_cuda(gu)
    assert gu.dtype == torch.bfloat16 and gu.is_contiguous() and gu.shape[1] % 32 == 0
    a = torch.empty((gu.shape[0], gu.shape[1] // 2), dtype=torch.bfloat16, device=gu.device)
    _lib.check(_lib.load().rga3_swiglu_fwd(gu.data_ptr(), a.data_ptr(), gu.shape[0], gu.shape[1] // 2, _stream()), "swiglu_fwd")
    return a


def swiglu_bwd(gu, da):
    _need_cuda(gu, da)
    assert gu.is_contiguous() and da.is_contiguous() and gu.shape[1] == 2 * da.shape[1]
    dgu = torch.empty_like(gu)
    _lib.check(_lib.load().rga3_swiglu_bwd(gu.data_ptr(), da.data_ptr(), dgu.data_ptr(), gu.shape[0], da.shape[1], _stream()), "swiglu_bwd")
    return dgu


def transpose(x):
    """[R, C] bf16 (row stride free) -> contiguous [C, R]."""
    _need_cuda(x)
    assert x.dim() == 2 and x.stride(1) == 1 and x.element_size() == 2
    R, C = x.shape
    out = torch.empty((C, R), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().rga3_transpose16(x.data_ptr(), out.data_ptr(), R, C, x.stride(0), R, _stream()), "transpose16")
    return out


def transpose_many(mats):
    """Contiguous 2-D 16-bit matrices -> their contiguous transposes, 64 per launch (csrc/trainops.hip)."""
    import ctypes
    outs = []
    for a in range(0, len(mats), 64):
        grp = mats[a:a + 64]
        n = len(grp)
        ptrs, dims = (ctypes.c_void_p * (2 * n))(), (ctypes.c_int64 * (2 * n))()
        for i, x in enumerate(grp):
            _need_cuda(x)
            assert x.dim() == 2 and x.is_contiguous() and x.element_size() == 2
            o = torch.empty((x.shape[1], x.shape[0]), dtype=x.dtype, device=x.device)
            ptrs[2 * i], ptrs[2 * i + 1] = x.data_ptr(), o.data_ptr()
            dims[2 * i], dims[2 * i + 1] = x.shape[0], x.shape[1]
            outs.append(o)
        _lib.check(_lib.load().rga3_transpose16_many(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(dims, ctypes.c_void_p), n, _stream()), "transpose16_many")
    return outs


def segment_sum_rows(x, rows, offsets):
    _need_cuda(x, rows, offsets)
    assert x.dtype == torch.bfloat16 and x.stride(1) == 1 and rows.dtype == torch.int64 and offsets.dtype == torch.int64
    n = offsets.numel() - 1
    out = torch.empty((n, x.shape[1]), dtype=torch.bfloat16, device=x.device)
    if n > 0:
        _lib.check(_lib.load().rga3_segment_sum_rows(x.data_ptr(), rows.data_ptr(), offsets.data_ptr(), out.data_ptr(), n, x.shape[1], x.stride(0), _stream()),
                   "segment_sum_rows")
    return out


def adamw_step_(param, master, grad, m, v, lr, beta1, beta2, eps, weight_decay, step: int, grad_scale: float = 1.0):
    _need_cuda(param, master, grad, m, v)
    assert param.dtype == grad.dtype == torch.bfloat16 and master.dtype == m.dtype == v.dtype == torch.float32
    assert param.is_contiguous() and grad.is_contiguous() and master.is_contiguous()
    _lib.check(_lib.load().rga3_adamw_step(param.data_ptr(), master.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), param.numel(), float(lr),
                                           float(beta1), float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale), _stream()), "adamw_step")


def sumsq_accum_(g, out):
    _need_cuda(g, out)
    assert g.dtype == torch.bfloat16 and g.is_contiguous() and out.dtype == torch.float32
    _lib.check(_lib.load().rga3_sumsq_accum(g.data_ptr(), out.data_ptr(), g.numel(), _stream()), "sumsq_accum")


def sumsq_det_(g, partials, out, accumulate: bool):
    """out[0] = (accumulate ? out[0] : 0) + sum(g^2), bit-reproducible (fixed summation order)."""
    _need_cuda(g, partials, out)
    assert g.dtype == torch.bfloat16 and g.is_contiguous() and out.dtype == partials.dtype == torch.float32
    _lib.check(_lib.load().rga3_sumsq_det(g.data_ptr(), g.numel(), partials.data_ptr(), partials.numel(), out.data_ptr(), int(bool(accumulate)), _stream()), "sumsq_det")


def adamw_step_clip_(param, master, grad, m, v, lr, beta1, beta2, eps, weight_decay, step: int, sumsq=None, max_norm: float = 0.0):
    """AdamW with the clipping factor min(1, max_norm / (sqrt(sumsq) + 1e-6)) taken from device memory (sumsq None: no clipping)."""
    _need_cuda(param, master, grad, m, v, sumsq)
    assert param.dtype == grad.dtype == torch.bfloat16 and master.dtype == m.dtype == v.dtype == torch.float32
    assert param.is_contiguous() and grad.is_contiguous() and master.is_contiguous()
    _lib.check(_lib.load().rga3_adamw_step_clip(param.data_ptr(), master.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), param.numel(), float(lr),
                                                float(beta1), float(beta2), float(eps), float(weight_decay), int(step), _ptr(sumsq), float(max_norm), _stream()),
               "adamw_step_clip")


def adamw_step_clip_rows_(param, master, grad, m, v, row_active, lr, beta1, beta2, eps, step: int, sumsq=None, max_norm: float = 0.0):
    """adamw_step_clip_ (weight decay 0) on the rows r of a [rows, row_len] table with row_active[r] != 0; the other rows (g = m = v = 0) are exactly unchanged."""
    _need_cuda(param, master, grad, m, v, row_active, sumsq)
    assert param.dtype == grad.dtype == torch.bfloat16 and master.dtype == m.dtype == v.dtype == torch.float32 and row_active.dtype == torch.uint8
    assert param.dim() == 2 and param.is_contiguous() and grad.is_contiguous() and master.is_contiguous() and row_active.numel() == param.shape[0]
    _lib.check(_lib.load().rga3_adamw_step_clip_rows(param.data_ptr(), master.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), param.shape[0], param.shape[1],
                                                     row_active.data_ptr(), float(lr), float(beta1), float(beta2), float(eps), int(step), _ptr(sumsq), float(max_norm),
                                                     _stream()), "adamw_step_clip_rows")


def scatter_add_rows_(dst, idx, src, scale: float = 1.0):
    """dst[idx[i]] += scale * src[i] (bf16 rows, idx unique within the call)."""
    _need_cuda(dst, idx, src)
    assert dst.dtype == src.dtype == torch.bfloat16 and idx.dtype == torch.int64 and dst.dim() == 2 and src.dim() == 2
    assert dst.stride(1) == 1 and src.stride(1) == 1 and dst.shape[1] == src.shape[1] and idx.numel() == src.shape[0]
    if idx.numel():
        _lib.check(_lib.load().rga3_scatter_add_rows(dst.data_ptr(), idx.data_ptr(), src.data_ptr(), idx.numel(), dst.shape[1], dst.stride(0), src.stride(0),
                                                     float(scale), _stream()), "scatter_add_rows")
    return dst


def check_gemm_health(device=None):
    """Raise if a stream-K hand-off ever gave up in this process (a wrong C would have been written silently); synchronises the device.  Called at the
    end of bench.py's timed regions and by the training tests."""
    n = gemm_stream_k_timeouts(device)
    if n:
        for t in _gemm_ws.values():     # recovery: re-arm every flag word so later launches do not read a stale slab
            t[:4096].zero_()
        raise _lib.Rga3Error(f"stream-K GEMM hand-off timed out {n} time(s): results since the last check are not trustworthy (workspace flags re-zeroed)")


class GemmHealthWatch:
    """check_gemm_health without the device synchronisation, for the product path (FusedAdamW.step polls it): every `every` polls the give-up counter of each
    GEMM workspace is fetched by a non-blocking copy into pinned memory, and the fetch of the PREVIOUS round -- long finished by then -- is examined.  A
    stream-K hand-off that timed out is therefore reported within 2 * every optimizer steps instead of writing a wrong C silently for the rest of the run."""

    def __init__(self, every: int = 16):
        self.every, self.n, self.pending = max(1, int(every)), 0, []

    def poll(self):
        self.n += 1
        if self.n % self.every:
            return
        self.examine()
        for (dev, _), t in _gemm_ws.items():
            with torch.cuda.device(dev):
                off = int(_lib.load().rga3_gemm_timeout_counter_offset())
                if off < 0:
                    continue
                host = torch.empty(1, dtype=torch.int32).pin_memory()
                host.copy_(t[off:off + 4].view(torch.int32), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            self.pending.append((host, ev))

    def examine(self):
        pend, self.pending = self.pending, []
        for host, ev in pend:
            ev.synchronize()      # recorded `every` steps ago: already complete
            if int(host[0]) != 0:
                for t in _gemm_ws.values():
                    t[:4096].zero_()
                raise _lib.Rga3Error(f"stream-K GEMM hand-off timed out {int(host[0])} time(s) in the last {2 * self.every} optimizer steps: their results are not "
                                     "trustworthy (workspace flags re-zeroed)")


_inference_watch = GemmHealthWatch(every=1)


def poll_gemm_health():
    """End of evaluate() / generate(): issue a non-blocking fetch of every GEMM workspace's give-up counter and examine the fetch of the PREVIOUS call (no device
    synchronisation).  A stream-K owner that gave up has already poisoned its output tile with +inf (csrc/gemm_bf16.hip), so the product itself is never a finite wrong
    one; this turns the poisoned result into a raised Rga3Error at the next call at the latest."""
    if _gemm_ws:
        _inference_watch.poll()


# ------------------------------------------------------------------------------------------------ mask-path backward kernels
def layernorm_bwd(x, weight, dy, eps: float, want_param_grads=True):
    _need_cuda(x, weight, dy)
    assert x.is_contiguous() and dy.is_contiguous() and x.dtype == dy.dtype == torch.bfloat16
    L = _lib.load()
    dx = torch.empty_like(x)
    dw = db = ws = None
    nws = 0
    if want_param_grads:
        nws = int(L.rga3_layernorm_bwd_ws_floats(x.shape[0], x.shape[1]))
        if nws:      # deterministic path: gradients are written
            dw = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
            db = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
            ws = torch.empty(nws, dtype=torch.float32, device=x.device)
        else:        # generic widths accumulate with atomics
            dw = torch.zeros(x.shape[1], dtype=torch.float32, device=x.device)
            db = torch.zeros(x.shape[1], dtype=torch.float32, device=x.device)
    _lib.check(L.rga3_layernorm_bwd(x.data_ptr(), weight.data_ptr(), dy.data_ptr(), dx.data_ptr(), _ptr(dw), _ptr(db), x.shape[0], x.shape[1],
                                    float(eps), _ptr(ws), nws, _stream()), "layernorm_bwd")
    return dx, dw, db


def colsum(x, fused_finish: bool = False):
    """f32 column sums of a bf16 [rows, cols] tensor (row stride free)."""
    _need_cuda(x)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    L = _lib.load()
    rows, cols = x.shape
    if cols % 8 == 0 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0:   # two-stage deterministic sum
        nws = int(L.rga3_colsum_ws_floats(rows, cols))
        out = torch.empty(cols, dtype=torch.float32, device=x.device)
        ws = torch.empty(nws, dtype=torch.float32, device=x.device)
        _lib.check(L.rga3_colsum(x.data_ptr(), out.data_ptr(), rows, cols, x.stride(0), ws.data_ptr(), nws,
                                 _tn_counters(x.device)[128:].data_ptr() if fused_finish else None, _stream()), "colsum")   # default two launches: the fused finish lost
        return out
    out = torch.zeros(cols, dtype=torch.float32, device=x.device)
    _lib.check(L.rga3_colsum_accum(x.data_ptr(), out.data_ptr(), rows, cols, x.stride(0), _stream()), "colsum_accum")
    return out


def mask_product(hyper, up, P: int):
    """masks[b, m, p] = sum_c hyper[b, m, c] * up[b * P + p, c]; hyper [B, 4, C] bf16, up [B * P, C] bf16 -> [B, 4, P] f32 (all frames in one launch)."""
    _need_cuda(hyper, up)
    assert hyper.dtype == up.dtype == torch.bfloat16 and hyper.is_contiguous() and up.is_contiguous() and hyper.dim() == 3 and up.dim() == 2
    B, NM, C = hyper.shape
    assert up.shape == (B * P, C)
    masks = torch.empty((B, NM, P), dtype=torch.float32, device=up.device)
    _lib.check(_lib.load().rga3_mask_product(hyper.data_ptr(), up.data_ptr(), masks.data_ptr(), B, NM, P, C, _stream()), "mask_product")
    return masks


def mask_product_bwd(dmasks, hyper, up, P: int):
    """-> (dhyper [B, 4, C] bf16, dup [B * P, C] bf16) from f32 dmasks [B, 4, P]."""
    _need_cuda(dmasks, hyper, up)
    assert dmasks.dtype == torch.float32 and dmasks.is_contiguous() and hyper.is_contiguous() and up.is_contiguous()
    B, NM, C = hyper.shape
    L = _lib.load()
    nws = int(L.rga3_mask_product_bwd_ws_floats(B, NM, P, C))
    ws = torch.empty(nws, dtype=torch.float32, device=up.device)
    dup = torch.empty_like(up)
    dhyper = torch.empty_like(hyper)
    _lib.check(L.rga3_mask_product_bwd(dmasks.data_ptr(), hyper.data_ptr(), up.data_ptr(), dup.data_ptr(), dhyper.data_ptr(), B, NM, P, C, ws.data_ptr(), nws,
                                       _stream()), "mask_product_bwd")
    return dhyper, dup


def gelu(x):
    _need_cuda(x)
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    out = torch.empty_like(x)
    _lib.check(_lib.load().rga3_act(x.data_ptr(), 0, out.data_ptr(), x.numel(), 0, _stream()), "act")
    return out


def act_bwd(a, dy, kind: str):
    """kind 'gelu': a = pre-activation; 'relu': a = output."""
    _need_cuda(a, dy)
    assert a.dtype == dy.dtype == torch.bfloat16 and a.is_contiguous() and dy.is_contiguous()
    out = torch.empty_like(a)
    _lib.check(_lib.load().rga3_act(a.data_ptr(), dy.data_ptr(), out.data_ptr(), a.numel(), 1 if kind == "gelu" else 2, _stream()), "act")
    return out


def bilinear_bwd(dout, in_shape, plane_idx=None):
    _need_cuda(dout, plane_idx)
    assert dout.dtype == torch.float32 and dout.is_contiguous() and dout.dim() == 3
    # without plane_idx the gather kernel writes every element; with it the atomic form adds into zeros
    din = (torch.empty if plane_idx is None else torch.zeros)(in_shape, dtype=torch.float32, device=dout.device)
    _lib.check(_lib.load().rga3_bilinear_bwd(dout.data_ptr(), din.data_ptr(), _ptr(plane_idx), dout.shape[0], in_shape[1], in_shape[2], dout.shape[1],
                                             dout.shape[2], _stream()), "bilinear_bwd")
    return din


def pixel_shuffle2x_bwd(dout, F: int, H: int, W: int):
    _need_cuda(dout)
    assert dout.dtype == torch.bfloat16 and dout.is_contiguous()
    Co = dout.shape[1]
    dg = torch.empty((F * H * W, 4 * Co), dtype=torch.bfloat16, device=dout.device)
    _lib.check(_lib.load().rga3_pixel_shuffle2x_bwd(dout.data_ptr(), dg.data_ptr(), F, H, W, Co, _stream()), "pixel_shuffle2x_bwd")
    return dg


def bce_dice_grad(logits, targets, sums, coef_bce, coef_dice):
    """coef_*: python floats, or f32 device scalars (the upstream gradients as autograd delivers them: read on the device, no host sync)."""
    _need_cuda(logits, targets, sums)
    assert logits.dtype == targets.dtype == sums.dtype == torch.float32 and logits.is_contiguous() and targets.is_contiguous()
    d = torch.empty_like(logits)
    if isinstance(coef_bce, torch.Tensor):
        cb, cd = coef_bce.reshape(1).float().contiguous(), coef_dice.reshape(1).float().contiguous()
        _lib.check(_lib.load().rga3_bce_dice_grad_dev(logits.data_ptr(), targets.data_ptr(), sums.data_ptr(), d.data_ptr(), logits.shape[0], logits[0].numel(),
                                                      cb.data_ptr(), cd.data_ptr(), _stream()), "bce_dice_grad_dev")
        return d
    _lib.check(_lib.load().rga3_bce_dice_grad(logits.data_ptr(), targets.data_ptr(), sums.data_ptr(), d.data_ptr(), logits.shape[0], logits[0].numel(),
                                              float(coef_bce), float(coef_dice), _stream()), "bce_dice_grad")
    return d
