"""Generate tests/golden/episode_tiny.pt: inputs and expected outputs of the CPU oracle (oracle/)
for seeded tiny-config episodes.  There is no reference-produced golden for this path (the
reference cannot be imported here: diffusers is absent), so this fixture pins the ORACLE against
regressions and lets the GPU tests run without re-deriving it; its header says "parity unpinned".

    python tests/golden/make_episode_goldens.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from diffews_amd import config, weights  # noqa: E402
from diffews_amd.episodes import make_episode_batch  # noqa: E402
from oracle import pipeline as op  # noqa: E402
from oracle.unet import OracleUNet  # noqa: E402
from oracle.vae import OracleVAE  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "episode_tiny.pt")


def main():
    dt = torch.bfloat16  # weights rounded so that bf16 AND fp16 engines hold them exactly
    ucfg, vcfg = config.get("tiny_unet"), config.get("tiny_vae")
    kw = lambda c: {k: v for k, v in c.items() if not k.startswith("_")}
    usd = weights.synthetic_unet_state_dict(ucfg, round_to=dt)
    vsd = weights.synthetic_vae_state_dict(vcfg, round_to=dt)
    te = weights.synthetic_text_embed(ucfg).to(dt).float()
    ou = OracleUNet(**kw(ucfg)); ou.load_state_dict(usd); ou.eval()
    ov = OracleVAE(**kw(vcfg)); ov.load_state_dict(vsd); ov.eval()
    cases = []
    for (b, s, res, seed) in [(1, 1, 64, 11), (2, 1, 64, 12), (1, 3, 64, 13)]:
        bt = make_episode_batch(b, s, res, seed=seed)
        masks, out = op.pipeline_call(ou, ov, [bt["support_imgs"], bt["query_img"], bt["support_masks"]], te)
        cases.append(dict(b=b, nshot=s, res=res, seed=seed,
                          z0=out["z0"].half(), z_tag=out["z_tag"].half(),
                          seg_u8=torch.from_numpy(__import__("numpy").stack(masks))))
    torch.save(dict(header="oracle-generated (parity unpinned: no reference golden exists for this path)",
                    weights="synthetic_*_state_dict(tiny, seed default, round_to=bfloat16)", cases=cases), OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
