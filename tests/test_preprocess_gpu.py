"""GPU: the input-transform kernels (csrc/preprocess.hip via the C ABI) against the code the reference
runs on the host -- PIL.Image.resize(BILINEAR) + ToTensor + Normalize (evaluation_util/data/dataset.py:36-40),
F.interpolate(nearest) on the class mask (coco.py:42,46), mask expansion (main_oss.py:100-104).
Byte / integer work: BIT-EXACT.  Also the prefetching EpisodeLoader's tensor contract and its hand-off
to the pipeline."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def host_image(img_u8, S):
    """The reference transform on one decoded image, with the real third-party pieces."""
    from PIL import Image
    res = np.asarray(Image.fromarray(img_u8, "RGB").resize((S, S), Image.BILINEAR))
    t = torch.from_numpy(res.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    return (t - 0.5) / 0.5


def host_mask(ids, cls, S):
    m = torch.from_numpy(np.asarray(ids).astype(np.int64))
    m = (m == cls + 1).float()                                              # coco.py:74-75
    return F.interpolate(m[None, None], (S, S), mode="nearest")[0, 0]      # coco.py:42


@pytest.mark.parametrize("H,W,S", [(480, 640, 512), (333, 500, 512), (640, 427, 512), (100, 80, 256), (1024, 768, 512),
                                   (512, 512, 512), (2, 3, 8), (1, 1, 4), (37, 41, 64), (2000, 1500, 512)])
def test_image_transform_bit_exact(hip_lib, H, W, S):
    from diffews_amd.input_pipeline import DeviceImageTransform
    from oracle import preprocess as P
    rng = np.random.default_rng(H + 7 * W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    tf = DeviceImageTransform(S)
    out = tf.image(img).cpu()
    assert out.shape == (3, S, S) and out.dtype == torch.float32
    assert torch.equal(out, host_image(img, S))
    assert torch.equal(out, P.image_transform(img, S))
    assert float(out.min()) >= -1.0 and float(out.max()) <= 1.0             # P:309 range assert holds


@pytest.mark.parametrize("dtype", [np.uint8, np.int32, np.int64])
@pytest.mark.parametrize("H,W,S", [(480, 640, 512), (333, 500, 512), (700, 512, 512), (3, 2, 8), (64, 64, 64)])
def test_mask_transform_bit_exact(hip_lib, dtype, H, W, S):
    from diffews_amd.input_pipeline import DeviceImageTransform
    rng = np.random.default_rng(H * 3 + W)
    ids = rng.integers(0, 6, (H, W)).astype(dtype)
    tf = DeviceImageTransform(S)
    for cls in (0, 4):
        pm1, bn = tf.mask(ids, cls)
        ref = host_mask(ids, cls, S)
        assert torch.equal(bn.cpu().float(), ref)
        assert torch.equal(pm1.cpu(), ref[None].repeat(3, 1, 1) * 2 - 1)    # main_oss.py:100


def _jpeg_episodes(tmp_path, n, nshot, seed=0):
    """n episodes written as JPEG images + PNG class masks of assorted sizes, read back through PIL like
    DatasetCOCO.load_frame does."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    eps = []
    for e in range(n):
        cls = int(rng.integers(0, 5))

        def one(tag):
            H, W = int(rng.integers(90, 400)), int(rng.integers(90, 400))
            yy, xx = np.mgrid[0:H, 0:W]
            base = np.stack([(xx * 255 // W), (yy * 255 // H), ((xx + yy) * 255 // (H + W))], -1).astype(np.uint8)
            noise = rng.integers(0, 40, (H, W, 3), dtype=np.uint8)
            ip, mp = tmp_path / f"e{e}_{tag}.jpg", tmp_path / f"e{e}_{tag}.png"
            Image.fromarray(base // 2 + noise, "RGB").save(ip, quality=92)
            ids = np.zeros((H, W), np.uint8)
            ids[H // 5:H // 2, W // 4:3 * W // 4] = cls + 1
            ids[rng.integers(0, H, 50), rng.integers(0, W, 50)] = (cls + 2) % 6
            Image.fromarray(ids, "L").save(mp)
            return Image.open(ip).convert("RGB"), np.array(Image.open(mp))
        q = one("q")
        sup = [one(f"s{k}") for k in range(nshot)]
        eps.append(dict(query_img=q[0], query_mask=q[1], support_imgs=[s[0] for s in sup],
                        support_masks=[s[1] for s in sup], class_id=cls))
    return eps


@pytest.mark.parametrize("b,nshot,n", [(2, 1, 5), (1, 3, 2), (4, 2, 4)])
def test_episode_loader_contract(hip_lib, tmp_path, b, nshot, n):
    """Batches (incl. a ragged last one) == the reference's host-built tensors, shot-major folding
    episode*nshot + shot (main_oss.py:103-104); recycled buffers stay correct across batches."""
    from diffews_amd.input_pipeline import EpisodeLoader
    S = 128
    eps = _jpeg_episodes(tmp_path, n, nshot, seed=b * 10 + nshot)
    got = []
    for batch in EpisodeLoader(eps, S, b, nshot, depth=1):
        got.append({k: v.clone().cpu() for k, v in batch.items()})
    assert sum(g["query_img"].shape[0] for g in got) == n
    i = 0
    for g in got:
        nb = g["query_img"].shape[0]
        assert g["support_imgs"].shape == (nb * nshot, 3, S, S) and g["support_masks"].shape == (nb * nshot, 3, S, S)
        assert g["query_mask"].dtype == torch.uint8 and g["query_mask"].shape == (nb, S, S)
        for j in range(nb):
            e = eps[i + j]
            assert int(g["class_id"][j]) == e["class_id"]
            assert torch.equal(g["query_img"][j], host_image(np.asarray(e["query_img"]), S))
            assert torch.equal(g["query_mask"][j].float(), host_mask(e["query_mask"], e["class_id"], S))
            for k in range(nshot):
                assert torch.equal(g["support_imgs"][j * nshot + k], host_image(np.asarray(e["support_imgs"][k]), S))
                ref = host_mask(e["support_masks"][k], e["class_id"], S)[None].repeat(3, 1, 1) * 2 - 1
                assert torch.equal(g["support_masks"][j * nshot + k], ref)
        i += nb


def test_loader_feeds_pipeline(hip_lib, tmp_path):
    """Loader output drops into run_episodes exactly like episodes.make_episode_batch's tensors."""
    from diffews_amd import weights, config
    from diffews_amd.unet import MyUNet2DConditionModel
    from diffews_amd.vae import AutoencoderKL
    from diffews_amd.pipeline import MarigoldPipelineRGBLatentNoise
    from diffews_amd.scheduler import DDIMSchedulerCustomized
    from diffews_amd.input_pipeline import EpisodeLoader
    ucfg, vcfg = config.get("tiny_unet"), config.get("tiny_vae")
    dt = torch.bfloat16
    pipe = MarigoldPipelineRGBLatentNoise(
        MyUNet2DConditionModel(ucfg, weights.synthetic_unet_state_dict(ucfg, round_to=dt), torch_dtype=dt),
        AutoencoderKL(vcfg, weights.synthetic_vae_state_dict(vcfg, round_to=dt), torch_dtype=dt),
        DDIMSchedulerCustomized(**{k: v for k, v in config.get("scheduler").items() if not k.startswith("_")}),
        text_embeds=weights.synthetic_text_embed(ucfg))
    eps = _jpeg_episodes(tmp_path, 4, 1, seed=3)
    tot = 0
    for batch in EpisodeLoader(eps, 64, 2, 1):
        out = pipe.run_episodes(batch["support_imgs"], batch["query_img"], batch["support_masks"], batch["query_mask"])
        c = out["counts"].cpu()
        assert out["z0"].shape[0] == 2 and torch.isfinite(out["z0"]).all()
        assert (c[:, 2:] >= c[:, :2]).all() and int(c[:, 2:].sum()) >= 64 * 64   # union >= intersection, covers the image
        tot += 2
    assert tot == 4


def test_evaluation_loop_over_host_episodes(hip_lib, tmp_path):
    """test_diffusion(episodes=...) (main_oss.py:84-171 with the GPU input pipeline) == feeding the same
    episodes' host-built tensors through make_batch: identical integer counts, hence identical mIoU."""
    from diffews_amd import weights, config, evaluate
    from diffews_amd.unet import MyUNet2DConditionModel
    from diffews_amd.vae import AutoencoderKL
    from diffews_amd.pipeline import MarigoldPipelineRGBLatentNoise
    from diffews_amd.scheduler import DDIMSchedulerCustomized
    ucfg, vcfg = config.get("tiny_unet"), config.get("tiny_vae")
    dt = torch.float16
    pipe = MarigoldPipelineRGBLatentNoise(
        MyUNet2DConditionModel(ucfg, weights.synthetic_unet_state_dict(ucfg, round_to=dt), torch_dtype=dt),
        AutoencoderKL(vcfg, weights.synthetic_vae_state_dict(vcfg, round_to=dt), torch_dtype=dt),
        DDIMSchedulerCustomized(**{k: v for k, v in config.get("scheduler").items() if not k.startswith("_")}),
        text_embeds=weights.synthetic_text_embed(ucfg))
    S, n = 64, 5
    eps = _jpeg_episodes(tmp_path, n, 1, seed=11)

    def make_batch(idx):
        sup = torch.stack([host_image(np.asarray(eps[i]["support_imgs"][0]), S) for i in idx]).cuda()
        qry = torch.stack([host_image(np.asarray(eps[i]["query_img"]), S) for i in idx]).cuda()
        sm = torch.stack([host_mask(eps[i]["support_masks"][0], eps[i]["class_id"], S)[None].repeat(3, 1, 1) * 2 - 1
                          for i in idx]).cuda()
        qm = torch.stack([host_mask(eps[i]["query_mask"], eps[i]["class_id"], S) for i in idx]).to(torch.uint8).cuda()
        return dict(support_imgs=sup, query_img=qry, support_masks=sm, query_mask=qm,
                    class_id=torch.tensor([eps[i]["class_id"] for i in idx]))
    a = evaluate.test_diffusion(pipe, n, nshot=1, res=S, batch=2, episodes=eps)
    b = evaluate.test_diffusion(pipe, n, nshot=1, res=S, batch=2, make_batch=make_batch)
    assert a[0] == b[0] and a[1] == b[1]
    assert torch.equal(a[2].intersection_buf, b[2].intersection_buf) and torch.equal(a[2].union_buf, b[2].union_buf)


def test_loader_recycles_slots_without_consumer_sync(hip_lib, tmp_path):
    """More than 2*(depth+2) batches with a consumer that never synchronises and is slow on the GPU: the
    producer thread runs ahead of the device, so a slot's pinned staging buffer is refilled while its previous
    H2D copy could still be pending -- every batch must still equal the host-built tensors (the loader
    blocks on the slot's copy event before overwriting the host bytes)."""
    from diffews_amd.input_pipeline import EpisodeLoader
    S, b, depth = 96, 1, 1
    n = 4 * (depth + 2) + 2
    eps = _jpeg_episodes(tmp_path, n, 1, seed=77)
    slow = torch.randn(4096, 4096, device="cuda")
    keep = []
    for batch in EpisodeLoader(eps, S, b, 1, depth=depth):
        for _ in range(6):                       # ~ms of queued GPU work per batch, no host sync
            slow = (slow @ slow).clamp_(-1, 1)
        keep.append({k: v.clone() for k, v in batch.items()})   # device-side copies, still asynchronous
    torch.cuda.synchronize()
    assert len(keep) == n
    for i, g in enumerate(keep):
        e = eps[i]
        assert int(g["class_id"][0]) == e["class_id"]
        assert torch.equal(g["query_img"][0].cpu(), host_image(np.asarray(e["query_img"]), S))
        assert torch.equal(g["support_imgs"][0].cpu(), host_image(np.asarray(e["support_imgs"][0]), S))
        assert torch.equal(g["query_mask"][0].float().cpu(), host_mask(e["query_mask"], e["class_id"], S))
