"""Golden vectors for the episode input transform, produced by the real third-party code the reference
calls: PIL.Image.resize(BILINEAR) (what torchvision.transforms.Resize does to a PIL image,
evaluation_util/data/dataset.py:37), torch's ToTensor/Normalize expressions (dataset.py:38-39) and
torch.nn.functional.interpolate(mode='nearest') (evaluation_util/data/coco.py:42,46).

    python tests/golden/make_preprocess_goldens.py   ->  tests/golden/preprocess_goldens.npz
"""
import os

import numpy as np
import torch
from PIL import Image

CASES = [(37, 41, 32), (50, 30, 32), (64, 64, 32), (20, 24, 48), (120, 90, 32), (33, 200, 40)]


def main():
    rng = np.random.default_rng(20240611)
    out = {"cases": np.array(CASES, np.int32)}
    for i, (H, W, S) in enumerate(CASES):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        if i == 2:
            img[:, :, :] = np.linspace(0, 255, W, dtype=np.uint8)[None, :, None]    # smooth ramp
        res = np.asarray(Image.fromarray(img, "RGB").resize((S, S), Image.BILINEAR))
        t = torch.from_numpy(res.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
        t = (t - 0.5) / 0.5
        ids = rng.integers(0, 4, (H, W)).astype(np.uint8)
        cls = 1
        m = torch.from_numpy((ids == cls + 1).astype(np.float32))
        mr = torch.nn.functional.interpolate(m[None, None], (S, S), mode="nearest")[0, 0]
        out[f"img{i}"], out[f"resized{i}"], out[f"tensor{i}"] = img, res, t.numpy()
        out[f"ids{i}"], out[f"mask{i}"] = ids, mr.numpy().astype(np.uint8)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "preprocess_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
