"""GPU: SAM-side input pipeline kernels vs the Pillow-pinned oracle -- bit-exact uint8 resize, bit-exact bf16 normalisation."""
import numpy as np
import pytest
import torch

from oracle import preproc as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("T,H,W,size", [(2, 480, 854, 1024), (1, 720, 1280, 1024), (1, 1080, 1920, 1024), (3, 256, 256, 1024), (1, 1500, 700, 1024),
                                        (2, 90, 130, 64), (1, 1024, 600, 1024), (1, 600, 1024, 1024), (1, 1024, 1024, 1024)])
def test_sam_preprocess_bit_exact(dev, T, H, W, size):
    from rga3.utils.preproc import sam_preprocess_frames

    rng = np.random.default_rng(T * 1000 + H + W)
    frames = rng.integers(0, 256, size=(T, H, W, 3), dtype=np.uint8)
    frames[:, : H // 3, :, 0] = 255
    frames[:, H // 2:, : W // 4, :] = 0
    want_u8, want_f = P.sam_preprocess(frames, size)
    out, u8 = sam_preprocess_frames(torch.from_numpy(frames).to(dev), size, return_u8=True)
    assert out.shape == (T, 3, size, size) and out.dtype == torch.bfloat16
    assert np.array_equal(u8.cpu().numpy(), want_u8)
    assert torch.equal(out.float().cpu(), want_f.to(torch.bfloat16).float())


def test_full_clip_properties(dev):
    """16 frames 480x854 -> [16,3,1024,1024]: constant frames stay constant (every row of the tables sums to 1.0 in fixed point up to
    rounding), and frame order / channel planes are not mixed."""
    from rga3.utils.preproc import sam_preprocess_frames

    T, H, W = 16, 480, 854
    frames = torch.empty((T, H, W, 3), dtype=torch.uint8)
    for t in range(T):
        frames[t, :, :, 0] = 10 * t
        frames[t, :, :, 1] = 255 - 10 * t
        frames[t, :, :, 2] = 7
    out, u8 = sam_preprocess_frames(frames.to(dev), 1024, return_u8=True)
    u8 = u8.cpu()
    for t in range(T):
        assert int((u8[t, :, :, 0].int() - 10 * t).abs().max()) <= 1 and int((u8[t, :, :, 1].int() - (255 - 10 * t)).abs().max()) <= 1
        assert int((u8[t, :, :, 2].int() - 7).abs().max()) <= 1
    ref = (u8.permute(0, 3, 1, 2).float() - torch.tensor(P.SAM_MEAN).view(1, 3, 1, 1)) / torch.tensor(P.SAM_STD).view(1, 3, 1, 1)
    assert torch.equal(out.float().cpu(), ref.to(torch.bfloat16).float())


# ---------------------------------------------------------------------------------------------- Qwen side
@pytest.mark.parametrize("T,H,W,max_pixels", [(3, 75, 130, 6 * 784), (4, 100, 60, 8 * 784), (2, 56, 84, 16384 * 784), (5, 480, 854, 384 * 784),
                                              (16, 360, 640, 336 * 784), (2, 448, 448, 384 * 784), (1, 720, 1280, 384 * 784)])
@pytest.mark.parametrize("fused", [False, True])
def test_qwen_preprocess_bit_exact(dev, T, H, W, max_pixels, fused):
    """uint8 frames -> pixel_values_videos: resized bytes, fp32 values and the bf16 cast all bit-identical to the oracle recipe."""
    from rga3.utils.preproc import qwen_preprocess_video

    rng = np.random.default_rng(T * 977 + H + 3 * W)
    frames = rng.integers(0, 256, size=(T, H, W, 3), dtype=np.uint8)
    frames[:, : H // 3, :, 1] = 255
    frames[:, H // 2:, : W // 4, :] = 0
    want, grid, res = P.qwen_video_preprocess(frames, max_pixels=max_pixels, fused=fused)
    d = torch.from_numpy(frames).to(dev)
    pv32, g32, u8 = qwen_preprocess_video(d, max_pixels=max_pixels, out_dtype=torch.float32, fused=fused, return_u8=True)
    assert g32.tolist() == [list(grid)] and np.array_equal(u8.cpu().numpy(), res)
    assert np.array_equal(pv32.cpu().numpy(), want)
    pv16, _ = qwen_preprocess_video(d, max_pixels=max_pixels, fused=fused)
    assert pv16.dtype == torch.bfloat16 and torch.equal(pv16.cpu(), torch.from_numpy(want).bfloat16())


def test_qwen_preprocess_fixture(dev):
    """Committed vectors made with Pillow + the installed transformers' smart_resize / patchify (tests/golden/make_qwen_preproc_fixtures.py)."""
    import os

    from rga3.utils.preproc import qwen_preprocess_video

    G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "qwen_preproc.npz"))
    for i in range(int(G["n"])):
        for fused, key in ((False, "pv49_"), (True, "pv5_")):
            pv, grid = qwen_preprocess_video(torch.from_numpy(G[f"frames{i}"]).to(dev), max_pixels=int(G[f"max_pixels{i}"]), out_dtype=torch.float32, fused=fused)
            assert grid[0].tolist() == G[f"grid{i}"].tolist()
            assert np.array_equal(pv.cpu().numpy(), G[f"{key}{i}"]), (i, fused)


def test_qwen_preprocess_feeds_the_configs1_shape(dev):
    """16 frames 448x448 under max_pixels 384*28*28 -> the BASELINE configs[1] operand: [8192, 1176] bf16, grid [8, 32, 32]; a constant
    clip gives the per-channel constants in every row (no frame / channel / patch mixing)."""
    from rga3.utils.preproc import qwen_norm_lut, qwen_preprocess_video

    frames = torch.empty((16, 448, 448, 3), dtype=torch.uint8, device=dev)
    frames[..., 0], frames[..., 1], frames[..., 2] = 10, 128, 250
    pv, grid = qwen_preprocess_video(frames, max_pixels=384 * 784)
    assert pv.shape == (8192, 1176) and grid.tolist() == [[8, 32, 32]]
    lut = qwen_norm_lut(dev)
    want = torch.stack([lut[0, 10], lut[1, 128], lut[2, 250]]).bfloat16().repeat_interleave(392)
    assert torch.equal(pv, want[None].expand(8192, -1))
