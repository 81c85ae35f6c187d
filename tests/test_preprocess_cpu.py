"""CPU: the input-transform oracle (oracle/preprocess.py) is pinned against the third-party code the
reference calls (Pillow's resize, torch's nearest interpolate) -- live on random images and through the
committed vectors -- and the library's HOST coefficient function matches it.  No GPU work here."""
import os

import numpy as np
import pytest
import torch

from oracle import preprocess as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess_goldens.npz")


def test_oracle_matches_committed_pil_vectors():
    g = np.load(GOLD)
    for i, (H, W, S) in enumerate(g["cases"]):
        img, ids = g[f"img{i}"], g[f"ids{i}"]
        assert img.shape == (H, W, 3)
        assert np.array_equal(P.resize_bilinear_u8(img, S, S), g[f"resized{i}"]), i
        assert np.array_equal(P.image_transform(img, S).numpy(), g[f"tensor{i}"]), i       # bit-exact fp32
        binm, pm1 = P.mask_transform(ids, 1, S)
        assert np.array_equal(binm, g[f"mask{i}"]), i
        assert torch.equal(pm1, torch.from_numpy(g[f"mask{i}"].astype(np.float32))[None].repeat(3, 1, 1) * 2 - 1)


@pytest.mark.parametrize("H,W,S", [(480, 640, 512), (333, 500, 512), (640, 427, 512), (100, 80, 256), (1024, 768, 512),
                                   (512, 512, 512), (2, 3, 8), (1, 1, 4), (700, 512, 512)])
def test_oracle_matches_live_pillow_and_torch(H, W, S):
    from PIL import Image
    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img, "RGB").resize((S, S), Image.BILINEAR))
    assert np.array_equal(P.resize_bilinear_u8(img, S, S), ref)
    ids = rng.integers(0, 5, (H, W)).astype(np.uint8)
    r = torch.nn.functional.interpolate(torch.from_numpy((ids == 3).astype(np.float32))[None, None], (S, S),
                                        mode="nearest")[0, 0]
    binm, _ = P.mask_transform(ids, 2, S)
    assert torch.equal(r, torch.from_numpy(binm.astype(np.float32)))


def test_normalize_lut_is_torch_arithmetic():
    lut = P.normalize_lut()
    assert lut.shape == (256,) and lut[0] == -1.0 and lut[255] == 1.0
    x = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)
    assert torch.equal(lut, (x - 0.5) / 0.5)


def test_library_host_coefficients_match_oracle(hip_lib):
    """dfw_resample_coeffs is a host function of the C ABI (Pillow's precompute_coeffs in C doubles)."""
    lib = hip_lib
    for i, o in [(640, 512), (480, 512), (333, 512), (100, 512), (1024, 512), (512, 512), (37, 64), (2, 8), (5000, 512),
                 (1, 4)]:
        k = lib.dfw_resample_ksize(i, o)
        b, c = np.zeros((o, 2), np.int32), np.zeros((o, k), np.int32)
        assert lib.dfw_resample_coeffs(i, o, b.ctypes.data, c.ctypes.data) == 0
        rb, rk = P.pil_bilinear_coeffs(i, o)
        assert rk.shape[1] == k and np.array_equal(rb, b) and np.array_equal(rk, c), (i, o)
        assert abs(c.sum(1) - (1 << 22)).max() <= k                               # weights sum to 1 up to rounding
    assert lib.dfw_resample_ksize(0, 5) == 0
    assert lib.dfw_resample_coeffs(0, 5, None, None) != 0
