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
