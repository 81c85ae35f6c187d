"""CPU tests (no GPU) of the host-side logic and of the C-ABI library surface."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol():
    """hipcc cross-compiles for gfx950 without a GPU; every function declared in
    include/diffews_hip.h must be exported by libdiffews_hip.so and bound in _lib.SYMBOLS."""
    from diffews_amd import build, _lib
    lib = build.build()
    assert os.path.isfile(lib)
    hdr = open(os.path.join(ROOT, "include", "diffews_hip.h")).read()
    declared = set(re.findall(r"\b(dfw_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"dfw_stream_t"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    h = _lib.lib()
    for name in declared:
        assert getattr(h, name) is not None
    assert h.dfw_version() >= 100
    assert b"success" in h.dfw_error_string(0)
    # argument validation happens on the host before any launch: safe without a GPU
    a = _lib.GemmArgs()
    assert h.dfw_gemm(ctypes.byref(a), None) == -1            # DFW_EINVAL: null pointers
    dummy = ctypes.create_string_buffer(64)
    a.A = a.W = a.C = ctypes.addressof(dummy)
    a.M, a.N, a.K, a.taps, a.Cin, a.lda, a.ldc = 8, 8, 60, 1, 60, 64, 8
    a.a_elems, a.w_elems = 1 << 10, 1 << 10
    assert h.dfw_gemm(ctypes.byref(a), None) == -2            # DFW_ESHAPE: K not a multiple of 64
    f = _lib.FsaArgs()
    assert h.dfw_fsa_attention(ctypes.byref(f), None) == -1


def test_struct_layouts_match_header():
    """ctypes structures must mirror the C structs field for field (names and order)."""
    from diffews_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "diffews_hip.h")).read()
    for cname, cls in (("dfw_gemm_args", _lib.GemmArgs), ("dfw_fsa_args", _lib.FsaArgs), ("dfw_xattn_args", _lib.XattnArgs),
                       ("dfw_groupnorm_args", _lib.GroupNormArgs), ("dfw_layernorm_args", _lib.LayerNormArgs),
                       ("dfw_conv_small_args", _lib.ConvSmallArgs), ("dfw_gemm_tn_args", _lib.GemmTnArgs),
                       ("dfw_groupnorm_bwd_args", _lib.GroupNormBwdArgs), ("dfw_layernorm_bwd_args", _lib.LayerNormBwdArgs),
                       ("dfw_fsa_bwd_args", _lib.FsaBwdArgs), ("dfw_xattn_bwd_args", _lib.XattnBwdArgs),
                       ("dfw_attn_bwd_args", _lib.AttnBwdArgs), ("dfw_adamw_args", _lib.AdamWArgs), ("dfw_image_args", _lib.ImageArgs),
                       ("dfw_vattn_args", _lib.VattnArgs), ("dfw_config", _lib.Config)):
        body = re.search(r"typedef struct \{([^{}]*)\}\s*" + cname + ";", hdr).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            parts = decl.replace("*", " ").replace(",", " ").split()
            names += [p for p in parts if p not in ("const", "void", "float", "int32_t", "int64_t", "size_t", "uint8_t", "uint32_t")]
        assert names == [f[0] for f in cls._fields_], cname


def test_sizeof_structs_against_compiler(tmp_path):
    from diffews_amd import _lib
    src = tmp_path / "sz.c"
    names = ["dfw_gemm_args", "dfw_fsa_args", "dfw_xattn_args", "dfw_groupnorm_args", "dfw_layernorm_args",
             "dfw_conv_small_args", "dfw_gemm_tn_args", "dfw_groupnorm_bwd_args", "dfw_layernorm_bwd_args",
             "dfw_fsa_bwd_args", "dfw_xattn_bwd_args", "dfw_attn_bwd_args", "dfw_adamw_args", "dfw_image_args", "dfw_vattn_args",
             "dfw_config"]
    src.write_text('#include <stdio.h>\n#include "diffews_hip.h"\nint main(){' +
                   "".join(f'printf("%zu ", sizeof({n}));' for n in names) + 'return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [ctypes.sizeof(c) for c in (_lib.GemmArgs, _lib.FsaArgs, _lib.XattnArgs, _lib.GroupNormArgs,
                                                _lib.LayerNormArgs, _lib.ConvSmallArgs, _lib.GemmTnArgs,
                                                _lib.GroupNormBwdArgs, _lib.LayerNormBwdArgs, _lib.FsaBwdArgs,
                                                _lib.XattnBwdArgs, _lib.AttnBwdArgs, _lib.AdamWArgs, _lib.ImageArgs, _lib.VattnArgs,
                                                _lib.Config)]


def test_library_reads_nothing_from_the_environment():
    """include/diffews_hip.h: the only process-wide state is the record of dfw_configure(); no getenv in the kernels'
    host code (21 function-local environment switches lived there until round 3)."""
    import glob
    for f in glob.glob(os.path.join(ROOT, "diffews_amd", "csrc", "*")):
        with open(f) as fh:
            assert "getenv" not in fh.read(), f
    from diffews_amd import _lib as L
    d0 = L.configure()
    assert d0["conv_patch"] == 3 and d0["big_kernels"] == 1 and d0["fsa_key_split"] == 1 and d0["gemm_bm"] == 0
    assert L.configure(conv_patch=1, gemm_bm=128, gemm_bn=64)["conv_patch"] == 1
    with pytest.raises(RuntimeError):
        L.configure(gemm_bm=96, gemm_bn=96)
    assert L.configure() == d0


def test_product_never_imports_oracle():
    """The product path must not route through the oracle or any CPU fallback."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "diffews_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f


def test_ops_fail_loudly_without_library(monkeypatch, tmp_path):
    from diffews_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(RuntimeError, match="no CPU / PyTorch fallback"):
        _lib.lib()


def test_packing_roundtrips():
    from diffews_amd import packing
    w = torch.randn(8, 4, 3, 3)
    p = packing.pack_conv3x3(w)
    assert p.shape == (8, 36) and torch.equal(p[:, (1 * 3 + 2) * 4 + 3], w[:, 3, 1, 2])
    perm = packing.geglu_perm(128)
    assert sorted(perm.tolist()) == list(range(256))
    assert perm[:32].tolist() == list(range(32)) and perm[32:64].tolist() == list(range(128, 160))
    assert perm[64:96].tolist() == list(range(32, 64))
    ws = packing.pack_conv_small(torch.randn(16, 3, 3, 3))
    assert ws.shape == (16, 9, 3) and ws.dtype == torch.float32


def test_synthetic_weights_and_checkpoint_io(tmp_path):
    from diffews_amd import config, weights
    cfg = config.get("tiny_unet")
    sd = weights.synthetic_unet_state_dict(cfg)
    sd2 = weights.synthetic_unet_state_dict(cfg)
    assert all(torch.equal(sd[k], sd2[k]) for k in sd)                       # seeded
    assert torch.equal(sd["conv_in_ref.weight"], sd["conv_in.weight"].repeat(1, 2, 1, 1) / 2)
    weights.check_state_dict(sd, weights.unet_param_shapes(cfg), "unet")
    weights.save_pretrained(str(tmp_path / "ckpt"), cfg, sd, subfolder="unet")
    assert os.path.isfile(tmp_path / "ckpt" / "unet" / "diffusion_pytorch_model.safetensors")
    back = weights.load_state_dict(str(tmp_path / "ckpt"), "unet")
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    assert weights.load_config(str(tmp_path / "ckpt"), "unet")["in_channels_ref"] == 8
    bad = dict(sd)
    bad.pop("conv_in_ref.weight")
    with pytest.raises(ValueError, match="missing keys"):
        weights.check_state_dict(bad, weights.unet_param_shapes(cfg), "unet")
    # scheduler JSON at top level but loaded with subfolder="scheduler" (main_oss.py:367)
    from diffews_amd.scheduler import DDIMSchedulerCustomized
    import json
    d = tmp_path / "sched"
    d.mkdir()
    (d / "scheduler_config.json").write_text(json.dumps(config.get("scheduler")))
    s = DDIMSchedulerCustomized.from_pretrained(str(d), subfolder="scheduler")
    s.set_timesteps(1)
    assert s.timesteps.tolist() == [1]


def test_metrics_int64_matches_reference_fixtures():
    """Product metric (int64 buffers) against the fixtures generated from the reference's code."""
    import json
    from diffews_amd import metrics
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "metric_goldens.json")))
    for c in g["classify"]:
        ign = None if c["ignore"] is None else torch.tensor(c["ignore"])
        inter, union = metrics.classify_prediction(torch.tensor(c["pred"]), torch.tensor(c["gt"]), ign)
        assert inter.tolist() == [[int(v) for v in r] for r in c["inter"]]
        assert union.tolist() == [[int(v) for v in r] for r in c["union"]]
    m = g["meter"]
    meter = metrics.AverageMeter(m["benchmark"], m["class_ids"])
    for e in m["stream"]:
        meter.update(torch.tensor(e["inter"]), torch.tensor(e["union"]), torch.tensor(e["class_id"]))
    miou, fb, _ = meter.compute_iou()
    assert meter.intersection_buf.tolist() == [[int(v) for v in r] for r in m["intersection_buf"]]
    assert float(miou) == pytest.approx(m["miou"], rel=1e-6) and float(fb) == pytest.approx(m["fb_iou"], rel=1e-6)
    assert metrics.fold_class_ids("coco", 0) == m["class_ids"]


def test_episode_sharding_and_synthetic_contract():
    from diffews_amd import episodes as ep
    n = 1000
    seen = sorted(i for r in range(8) for i in ep.shard(n, r, 8))
    assert seen == list(range(n)) and len(ep.shard(n, 3, 8)) == 125
    b = ep.make_episode_batch(2, 3, 32, seed=1)
    assert b["support_imgs"].shape == (6, 3, 32, 32) and b["query_img"].shape == (2, 3, 32, 32)
    m = b["support_masks"]
    assert set(m.unique().tolist()) == {-1.0, 1.0} and torch.equal(m[:, 0], m[:, 1]) and torch.equal(m[:, 0], m[:, 2])
    assert float(b["query_img"].min()) >= -1 and float(b["query_img"].max()) <= 1
    assert b["query_mask"].dtype == torch.uint8 and set(b["query_mask"].unique().tolist()) <= {0, 1}
    assert ep.episode_class_ids([0, 1, 21]).tolist() == [0, 4, 4]


_WORKER = r'''
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from diffews_amd.metrics import AverageMeter, fold_class_ids
from diffews_amd import episodes as ep
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
meter = AverageMeter("coco", fold_class_ids("coco", 0))
g = torch.Generator().manual_seed(0)
allc = torch.randint(0, 262144, (40, 4), generator=g)          # same stream on every rank
allc[:, 2:] += allc[:, :2]
for i in ep.shard(40, rank, world):
    meter.update_from_counts(allc[i:i + 1], ep.episode_class_ids([i]))
meter.all_reduce()
miou, fb, _ = meter.compute_iou()
if rank == 0:
    print(json.dumps(dict(inter=meter.intersection_buf.tolist(), union=meter.union_buf.tolist(), miou=float(miou), fb=float(fb))))
dist.destroy_process_group()
'''


def test_two_rank_gloo_all_reduce_equals_single_process(tmp_path):
    """N>1 path on CPU: world_size-2 gloo run of the sharded meter == the single-process result,
    bit-identical (int64 sums are order independent)."""
    import json
    from diffews_amd import episodes as ep
    from diffews_amd.metrics import AverageMeter, fold_class_ids
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.check_output(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29571", str(script), ROOT], env=env, stderr=subprocess.STDOUT, timeout=300).decode()
    res = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    meter = AverageMeter("coco", fold_class_ids("coco", 0))
    g = torch.Generator().manual_seed(0)
    allc = torch.randint(0, 262144, (40, 4), generator=g)
    allc[:, 2:] += allc[:, :2]
    for i in range(40):
        meter.update_from_counts(allc[i:i + 1], ep.episode_class_ids([i]))
    miou, fb, _ = meter.compute_iou()
    assert res["inter"] == meter.intersection_buf.tolist() and res["union"] == meter.union_buf.tolist()
    assert res["miou"] == float(miou) and res["fb"] == float(fb)


def test_fold_class_ids_per_benchmark_match_reference_datasets():
    """`dataset.class_ids` of DatasetCOCO / DatasetPASCAL / DatasetFSS (coco.py:64-70, pascal.py:115-123,
    fss.py:100-107), captured by tests/golden/make_classid_goldens.py from the reference's own classes."""
    import json
    from diffews_amd import metrics
    with open(os.path.join(os.path.dirname(__file__), "golden", "classid_goldens.json")) as f:
        cases = json.load(f)
    assert {c["benchmark"] for c in cases} == {"coco", "pascal", "fss"}
    for c in cases:
        assert metrics.fold_class_ids(c["benchmark"], c["fold"], split=c["split"]) == c["ids"], c["benchmark"]
    # pascal folds are contiguous blocks, coco folds interleaved: they must differ
    assert metrics.fold_class_ids("pascal", 1) == [5, 6, 7, 8, 9] and metrics.fold_class_ids("coco", 1)[:3] == [1, 5, 9]
    # lvis: positions in the fold's category list (lvis.py:28-29, 83); list comes from the annotation file
    assert metrics.fold_class_ids("lvis", 2, cat_ids=list(range(100, 160))) == list(range(6))
    with pytest.raises(ValueError):
        metrics.fold_class_ids("lvis", 0)
    with pytest.raises(NotImplementedError):
        metrics.fold_class_ids("paco_part", 0)


def test_checkpoint_directory_and_empty_prompt_loading(tmp_path):
    """The launcher's checkpoint layout (main_oss.py:338-369): unet/ vae/ scheduler/ text_encoder/ tokenizer/.
    Host side of `MarigoldPipeline.from_pretrained(checkpoint, text_embeds=None)`: weights and configs load
    back unchanged, and CLIP("") is evaluated once from the directory exactly as P:585-601 does
    (tokenise "" unpadded -> [BOS, EOS] -> text_encoder(ids)[0])."""
    sys.path.insert(0, os.path.dirname(__file__))
    from ckpt_util import make_checkpoint_dir
    from diffews_amd import weights
    from diffews_amd.pipeline import load_empty_text_embed
    from diffews_amd.scheduler import DDIMSchedulerCustomized
    ck = make_checkpoint_dir(tmp_path / "ckpt")
    for sub, cfg, sd in (("unet", ck["ucfg"], ck["usd"]), ("vae", ck["vcfg"], ck["vsd"])):
        assert weights.load_config(ck["root"], sub) == cfg
        got = weights.load_state_dict(ck["root"], sub)
        assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    s = DDIMSchedulerCustomized.from_pretrained(ck["root"], subfolder="scheduler")
    s.set_timesteps(1)
    assert s.timesteps.tolist() == [1] and s.z0_is_neg_v(1)
    e = load_empty_text_embed(ck["root"])
    assert e.shape == (1, 2, ck["ucfg"]["cross_attention_dim"]) and e.dtype == torch.float32
    assert torch.allclose(e, ck["text_embed"], atol=1e-6)
    # the launcher hands over its own tokenizer (E:351-353): same ids, same embedding
    from transformers import CLIPTokenizer
    tok = CLIPTokenizer.from_pretrained(os.path.join(ck["root"], "tokenizer"))
    assert torch.equal(load_empty_text_embed(ck["root"], tok), e)
    with pytest.raises(FileNotFoundError):
        load_empty_text_embed(str(tmp_path))          # no text_encoder/ there
    with pytest.raises(ValueError):
        load_empty_text_embed(None)


def test_bench_refuses_to_report_fewer_gpus_than_asked():
    """`python bench.py --gpus N` (the driver's command line, no RANK in the environment) spawns its own
    ranks; where the node shows fewer than N GPUs it must fail loudly -- never print an n_gpus=1 line."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node could actually run 2 ranks")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DFW_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "n_gpus" not in r.stdout and "refusing" in r.stderr
    # under torchrun with a world size that contradicts --gpus: also an error, not a silent re-interpretation
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout


_GRAD_WORKER = r'''
import os, sys, hashlib, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from diffews_amd.train import allreduce_flat_gradient
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
n = 1_000_003                                            # not a multiple of the bucket size: ragged last bucket
g = torch.randn(n, generator=torch.Generator().manual_seed(100 + rank)) * 10.0 ** torch.randint(-6, 3, (n,), generator=torch.Generator().manual_seed(7)).float()
allreduce_flat_gradient(g, bucket_elems=65_536)
if rank == 0:
    print("SHA", hashlib.sha256(g.numpy().tobytes()).hexdigest())
dist.destroy_process_group()
'''


def test_two_rank_bucketed_gradient_all_reduce_is_bit_exact(tmp_path):
    """The training step's one collective (DDP's gradient all-reduce, T:1226-1228 / T:1391) rehearsed on CPU: the
    flat fp32 gradient summed over 2 gloo ranks in fixed buckets, then averaged == the single-process mean of the two
    micro-batch gradients, bit for bit (a + b is commutative in IEEE fp32, * 0.5 is exact)."""
    import hashlib
    script = tmp_path / "gw.py"
    script.write_text(_GRAD_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.check_output(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29573", str(script), ROOT], env=env, stderr=subprocess.STDOUT, timeout=300).decode()
    sha = [l.split()[1] for l in out.splitlines() if l.startswith("SHA")][-1]
    n = 1_000_003
    sc = 10.0 ** torch.randint(-6, 3, (n,), generator=torch.Generator().manual_seed(7)).float()
    g0 = torch.randn(n, generator=torch.Generator().manual_seed(100)) * sc
    g1 = torch.randn(n, generator=torch.Generator().manual_seed(101)) * sc
    want = (g0 + g1) * 0.5
    assert hashlib.sha256(want.numpy().tobytes()).hexdigest() == sha
    # single process / uninitialised process group: a no-op, not an error
    from diffews_amd.train import allreduce_flat_gradient, poly_lr
    g = g0.clone()
    assert allreduce_flat_gradient(g) is g and torch.equal(g, g0)
    assert poly_lr(1e-4, 0, 10) == pytest.approx(1e-4) and poly_lr(1e-4, 10, 10) == pytest.approx(1e-7)


def test_kernel_plans_of_the_baseline_shapes():
    """Host-side planning (no GPU call): which kernel the library picks for the shapes of BASELINE configs[1] / [4], and the
    key-split plan of the many-shot attention launches -- the choices DESIGN.md section 3 states."""
    import ctypes as C
    from diffews_amd import _lib
    lib = _lib.lib()

    def name(M, N, K, taps=1, H=0, W=0):
        a = _lib.GemmArgs()
        a.A = a.W = a.C = 4096                      # never dereferenced by the planner
        a.M, a.N, a.K, a.lda, a.ldc = M, N, K, K // taps, N
        a.a_elems, a.w_elems = M * (K // taps), N * K
        a.taps, a.Cin = taps, K // taps
        a.Hi = a.Ho = H
        a.Wi = a.Wo = W
        a.stride, a.pad = 1, 1 if taps == 9 else 0
        a.rows_per_img = H * W if taps == 9 else 0
        a.out_scale, a.batch, a.dtype = 1.0, 1, _lib.BF16
        buf = C.create_string_buffer(64)
        rc = lib.dfw_gemm_kernel_name(C.byref(a), buf, 64)
        assert rc == 0, rc
        return buf.value.decode()

    # VAE encoder, 12 images at 512^2: the stride-1 conv3x3 layers on the LDS-resident-patch kernel (512 x 128 tile for N = 128,
    # 256 x 256 for N % 256 == 0), the UNet's 64^2-level convs (N = 320: not a tile multiple) on gemm_big / gemm_kernel
    assert name(12 * 512 * 512, 128, 9 * 128, 9, 512, 512).startswith("conv_patch_kernel<bf16,512,128")
    # round 4: N % 256 == 0 on conv_patch8_kernel (64-channel K-tiles, eight-phase schedule), its 256 x 128 tile where 256 x 256
    # tiles would be too few for the chip (VAE decoder, 4 images at 64^2 x 512)
    assert name(12 * 256 * 256, 256, 9 * 256, 9, 256, 256) == "conv_patch8_kernel<bf16,256,256>"
    assert name(4 * 64 * 64, 512, 9 * 512, 9, 64, 64) == "conv_patch8_kernel<bf16,256,128>"
    # the UNet's 64^2 level (N = 320) on the lock-step batch of 8 latents: its 256 x 160 tile (128 x 2 = 256 tiles, one per CU);
    # on 4 latents (128 tiles) it stays on gemm_kernel
    assert name(8 * 64 * 64, 320, 9 * 320, 9, 64, 64) == "conv_patch8_kernel<bf16,256,160>"
    assert name(4 * 64 * 64, 320, 9 * 320, 9, 64, 64).startswith("gemm_kernel<bf16,")
    # UNet, lock-step batch of 8 latents: narrow tiles for the short-K linears, the ragged 960-column QKV on gemm_big
    assert name(8192, 640, 640) in ("gemm_kernel<bf16,64,64,lin>", "gemm_kernel<bf16,128,64,lin>")   # never 128 x 128 (+47 %)
    assert name(32768, 960, 320) == "gemm8_kernel<bf16,256,160,64,lin>"          # round 4: the 64-deep K-tile kernel, 256 x 160 tile (6 x 160)
    assert name(32768, 2560, 320) == "gemm8_kernel<bf16,256,256,64,lin>"
    assert name(32768, 320, 320) == "gemm8_kernel<bf16,256,160,64,lin>"          # 128 x 2 = 256 tiles, one per CU
    assert name(8192, 640, 640).startswith("gemm_kernel<bf16,")                  # too few rows for 256-row tiles

    def fsa_ws(batch, n_plain, nshot, heads, n):
        a = _lib.FsaArgs()
        a.batch, a.heads, a.n_q, a.n_kv, a.n_bank, a.nshot, a.n_plain = batch, heads, n, n, n, nshot, n_plain
        return lib.dfw_fsa_workspace_bytes(C.byref(a))

    assert fsa_ws(8, 4, 1, 5, 4096) == 0                        # configs[1] (1-shot): never split
    per_split = 5 * 4096 * 68 * 4                                # one bank-reading image, all heads: bytes per split
    assert fsa_ws(8, 7, 7, 5, 4096) == 4 * per_split            # configs[4] (7-shot): the query image's 8 segments -> 4 splits
    assert fsa_ws(12, 10, 5, 5, 4096) == 2 * 2 * per_split      # configs[2] (5-shot, b = 2): two query images x 2 splits
    assert fsa_ws(8, 7, 7, 20, 256) == 0                         # 16x16 level: rows too short to be worth a second kernel


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _rank_overlapped_reduce(rank, world, port, out_dir):
    """One rank of the overlapped-vs-serial gradient all-reduce test (gloo, CPU tensors)."""
    import random
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffews_amd.train import GradBucketReducer, ParamStore, allreduce_flat_gradient
    g = torch.Generator().manual_seed(100 + rank)
    spec, off = {}, 0
    sizes = [64 * random.Random(5 + i).randint(1, 400) for i in range(60)]
    for i, n in enumerate(sizes):
        spec[f"p{i}"] = (off, n)
        off += n
    numel = off
    grads = torch.randn(numel, generator=g)
    loss = torch.tensor([0.25 + rank])
    res = {}
    for mode in ("serial", "overlap_fp32", "overlap_bf16", "segments_fp32"):
        buf = torch.zeros(numel + ParamStore.TAIL)
        if mode == "serial":
            buf[:numel] = grads
            buf[numel] = loss[0]
            allreduce_flat_gradient(buf, bucket_elems=50_000)
            res[mode] = buf.clone()
            continue
        red = GradBucketReducer(buf, spec, bucket_elems=50_000,
                                comm_dtype=torch.bfloat16 if mode == "overlap_bf16" else torch.float32)
        if mode == "segments_fp32":
            # replay_segments: the captured-step form -- each "graph" writes the gradients of one cut of the backward, the
            # reducer fires that cut's buckets behind it (segments planned the way _SegmentRecorder cuts a capture)
            from diffews_amd.train import replay_segments
            red.reset_state()
            names = list(spec)[::-1]
            segs, cur = [], []
            for nme in names:
                cur.append(nme)
                ready = red.ready_after([nme])
                if ready:
                    for b in ready:
                        red.fired[b] = True
                    segs.append((list(cur), ready))
                    cur = []
            rest = [b for b in range(len(red.ranges) - 1, -1, -1) if not red.fired[b]]
            segs.append((list(cur), rest))
            red.active = False

            class _G:
                def __init__(self, ns, first):
                    self.ns, self.first = ns, first

                def replay(self):
                    if self.first:
                        buf[numel] = loss[0]
                    for nme in self.ns:
                        o, n = spec[nme]
                        buf[o:o + n] = grads[o:o + n]
            avg = replay_segments([(_G(ns, i == 0), bs) for i, (ns, bs) in enumerate(segs)], red)
            res[mode] = buf.clone()
            res[mode + "_loss"] = avg.clone()
            res[mode + "_order"] = list(red.fired_order)
            continue
        red.begin(loss)
        # the "backward": parameters become final from the end of the buffer towards its start, in uneven groups;
        # the same order on every rank (it is a property of the model, not of the data)
        names = list(spec)[::-1]
        rnd = random.Random(7)
        i = 0
        while i < len(names):
            k = rnd.randint(1, 5)
            for nme in names[i:i + k]:
                o, n = spec[nme]
                buf[o:o + n] = grads[o:o + n]
            red.mark(names[i:i + k])
            i += k
        avg = red.finish()
        res[mode] = buf.clone()
        res[mode + "_loss"] = avg.clone()
        res[mode + "_order"] = list(red.fired_order)
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_reducer_refuses_to_run_unbegun():
    """GradBucketReducer: mark() / finish() without begin() raise instead of silently leaving the local gradient in place."""
    import pytest as _pt
    from diffews_amd.train import GradBucketReducer, ParamStore
    buf = torch.zeros(1000 + ParamStore.TAIL)
    red = GradBucketReducer(buf, {"a": (0, 600), "b": (600, 400)}, bucket_elems=512, world_size=2,
                            collective=lambda t: t.mul_(2.0))
    with _pt.raises(RuntimeError):
        red.mark(["a"])
    with _pt.raises(RuntimeError):
        red.finish()
    buf[:1000] = 1.0
    buf[1000] = 123.0                   # stale tail value: begin(None) must clear it
    red.begin(None)
    red.mark(["b"])
    red.mark(["a"])
    avg = red.finish()
    assert float(avg) == 0.0 and torch.equal(buf[:1000], torch.ones(1000)) and sorted(red.fired_order) == [0, 1, 2][:len(red.ranges)]
    with _pt.raises(RuntimeError):
        red.finish()                    # the step is over


def test_two_rank_overlapped_gradient_reduce_equals_serial(tmp_path):
    """world_size 2, gloo: GradBucketReducer (buckets issued while the 'backward' still writes the earlier parameters,
    loss in the tail slot of the last range) == allreduce_flat_gradient (serial, after the backward) BIT FOR BIT in fp32;
    the bf16 wire format stays within bf16 rounding of it; both ranks hold the same result and fire the same bucket order."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_rank_overlapped_reduce, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["serial"], r1["serial"])
    assert torch.equal(r0["overlap_fp32"], r0["serial"]) and torch.equal(r1["overlap_fp32"], r0["serial"])
    assert float(r0["overlap_fp32_loss"]) == pytest.approx((0.25 + 1.25) / 2)
    assert r0["overlap_fp32_order"] == r1["overlap_fp32_order"] and r0["overlap_fp32_order"][0] == max(r0["overlap_fp32_order"])
    assert len(r0["overlap_fp32_order"]) >= 8
    d = (r0["overlap_bf16"][:-64] - r0["serial"][:-64]).norm() / r0["serial"][:-64].norm()
    assert float(d) < 8e-3 and torch.equal(r0["overlap_bf16"], r1["overlap_bf16"])
    # round 4: the captured-step form (a chain of segments cut at the bucket boundaries, replay_segments) is the same
    # reduction bit for bit, on both ranks, in the same firing order
    assert torch.equal(r0["segments_fp32"], r0["serial"]) and torch.equal(r1["segments_fp32"], r0["serial"])
    assert r0["segments_fp32_order"] == r0["overlap_fp32_order"] == r1["segments_fp32_order"]
    assert float(r0["segments_fp32_loss"]) == pytest.approx((0.25 + 1.25) / 2)
