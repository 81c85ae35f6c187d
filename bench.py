#!/usr/bin/env python
"""Headline benchmark: few-shot episodes / second of the DiffewS hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic episodes resident in HBM:
VAE-encode (2*nshot+1 images per episode) -> UNet support pass (fills the K/V banks) -> UNet query
pass (KV-fusion attention over [own ; bank]) -> z0 = -v -> VAE-decode -> uint8 mask -> dynamic
threshold + intersection/union counts -> AverageMeter update, all on device, captured in one HIP
graph.  Workload at N=1: BASELINE.json configs[1] -- SD-2.1 UNet + SD VAE, bf16, 512x512, 1-shot,
batch = 4 episodes per GPU.  N>1: one process per GPU (torchrun), each rank runs its own batch per
step (weak scaling), no data-path collective; the single RCCL sum-all-reduce of the [2, nclass]
inter/union buffers happens once after the timed steps (end of the evaluation stream).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel =
the MFMA implicit-GEMM conv3x3, measured live with events on the launch stream in an instrumented
eager pass) and `cpu_baseline` (the fp32 CPU oracle timed on a bounded sample on this box's host cores).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_pipeline(dtype, tiny=False, blobs=None, residual_dtype=None):
    """Seeded synthetic SD-2.1 UNet + SD VAE weights in the diffusers layout, values representable in fp16 (so the fp16
    engine, the fp32 CPU oracle of the cpu_baseline leg and -- after one more rounding -- the bf16 engine all hold the
    same checkpoint: the oracle's z0 of that leg is then a like-for-like reference for `fp16_fp32stream_z0_rel_err`)."""
    from diffews_amd import config, weights
    from diffews_amd.pipeline import MarigoldPipelineRGBLatentNoise
    from diffews_amd.scheduler import DDIMSchedulerCustomized
    from diffews_amd.unet import MyUNet2DConditionModel
    from diffews_amd.vae import AutoencoderKL
    t0 = time.time()
    if blobs is None:
        ucfg = config.get("tiny_unet" if tiny else "sd21_unet")
        vcfg = config.get("tiny_vae" if tiny else "sd_vae")
        usd = weights.synthetic_unet_state_dict(ucfg, round_to=torch.float16)
        vsd = weights.synthetic_vae_state_dict(vcfg, round_to=torch.float16)
        te = weights.synthetic_text_embed(ucfg).to(torch.float16).float()
        blobs = (ucfg, usd, vcfg, vsd, te)
    ucfg, usd, vcfg, vsd, te = blobs
    unet = MyUNet2DConditionModel(ucfg, usd, torch_dtype=dtype, residual_dtype=residual_dtype)
    vae = AutoencoderKL(vcfg, vsd, torch_dtype=dtype, residual_dtype=residual_dtype)
    sched = DDIMSchedulerCustomized(**{k: v for k, v in config.get("scheduler").items() if not k.startswith("_")})
    pipe = MarigoldPipelineRGBLatentNoise(unet, vae, sched, text_embeds=te.cuda())
    log(f"[bench] synthetic weights + packing ({dtype}, residual {residual_dtype or dtype}): {time.time() - t0:.1f}s")
    return pipe, blobs


ORACLE_EPISODE_SEED = 7     # the cpu_baseline leg's episode; the secondary leg reruns it on the GPU engine


def cpu_baseline(model_blobs, res, nshot, budget_s=20.0, ref_out=None):
    """fp32 CPU oracle (oracle/, the restatement of the reference's diffusers graph) on a bounded
    sample: single episodes of the bench workload's shape (>= 1 timed episode, `value`), and BASELINE.json
    configs[0] -- one 256x256 1-shot episode, fp32, 1 denoise step -- timed >= 3 times beside it.
    ref_out: file that receives the oracle's z0 (P:769) of the timed `res` episode -- the checker's output for the
    secondary leg's `fp16_fp32stream_z0_rel_err` (the oracle itself runs only here)."""
    from oracle import pipeline as op
    from oracle.unet import OracleUNet
    from oracle.vae import OracleVAE
    from diffews_amd import episodes
    ucfg, usd, vcfg, vsd, te = model_blobs
    # the GPU box exposes every host core in os.cpu_count() but schedules this job on its CPU share
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # the box's CPU share for a one-GPU job is 16 cores (a cgroup quota, invisible to sched_getaffinity): more threads than
    # that only thrash
    cores = max(1, min(cores, int(os.environ.get("DFW_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    kw = lambda c: {k: v for k, v in c.items() if not k.startswith("_")}
    ou = OracleUNet(**kw(ucfg)); ou.load_state_dict(usd); ou.eval()
    ov = OracleVAE(**kw(vcfg)); ov.load_state_dict(vsd); ov.eval()

    last = {}

    def timed(r, s, min_n, max_n, budget):
        bt = episodes.make_episode_batch(1, s, r, seed=ORACLE_EPISODE_SEED)
        op.single_infer(ou, ov, bt["support_imgs"], bt["query_img"], bt["support_masks"], te) if r <= 256 else None  # warm-up (cheap sizes only)
        ts = []
        while True:
            t0 = time.time()
            last["ref"] = op.single_infer(ou, ov, bt["support_imgs"], bt["query_img"], bt["support_masks"], te)
            ts.append(time.time() - t0)
            if len(ts) >= max_n or (len(ts) >= min_n and sum(ts) + ts[-1] > budget):
                break
        return ts
    t256 = timed(256, 1, 3, 5, 8.0)
    tb = timed(res, nshot, 2, 4, budget_s)
    if ref_out:
        torch.save({"z0": last["ref"]["z0"].float().cpu(), "seed": ORACLE_EPISODE_SEED, "res": res, "nshot": nshot}, ref_out)
    med = sorted(t256)[len(t256) // 2]
    # flat keys only: the driver's `parsed` keeps one level of nesting
    return dict(value=len(tb) / sum(tb), unit="episodes/s", cores=cores, kind="port",
                sample=f"{len(tb)} episode(s) {res}x{res} {nshot}-shot fp32, oracle/ on {cores} host threads, {sum(tb):.1f}s",
                configs0_256x256_1shot_value=round(1.0 / med, 4), configs0_256x256_1shot_median_s=round(med, 3),
                configs0_256x256_1shot_episodes=len(t256),
                configs0_note="BASELINE.json configs[0]: 256x256, 1-shot, fp32, 1 denoise step, median after 1 warm-up")


def roofline_pass(step):
    """Instrumented eager pass: every GEMM launch bracketed by events on its launch stream."""
    from diffews_amd import ops
    rec = []
    ops.gemm_hook = lambda name, flops, e0, e1, shape=None: rec.append((name, flops, e0, e1, shape))
    try:
        step()
        torch.cuda.synchronize()
    finally:
        ops.gemm_hook = None
    agg, shapes = {}, {}
    for name, flops, e0, e1, shape in rec:
        t = e0.elapsed_time(e1) * 1e-3
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += flops
        a[2] += t
        b = shapes.setdefault((name, shape), [0, 0.0, 0.0])
        b[0] += 1
        b[1] += flops
        b[2] += t
    if os.environ.get("DFW_BENCH_SHAPES"):
        for (name, shape), (n, fl, t) in sorted(shapes.items(), key=lambda kv: -kv[1][2])[:60]:
            log(f"[shape] {name:36s} M,N,K,taps,stride,ups,splitk,batch={shape}  x{n:3d}  {t * 1e3:8.3f} ms  "
                f"{fl / t / 1e12:7.1f} TF/s")
    return agg


def spawn_ranks(n):
    """`python bench.py --gpus N` with no RANK in the environment: start the N ranks as CHILD processes
    (python -m torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1), relay rank 0's JSON line
    (the children inherit stdout) and exit with their code.  Nothing here touches the GPU; the parent never
    re-execs itself."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()          # does not initialise the GPU on this image
    if ndev < n and not os.environ.get("DFW_ONE_DEVICE"):
        log(f"[bench] --gpus {n} requested but this node shows {ndev} GPU(s): refusing to report a smaller run")
        return 2
    from diffews_amd import build
    build.build()                              # before any rank starts (no hipcc race between ranks)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    log("[bench] spawning: " + " ".join(cmd))
    return subprocess.run(cmd, env=env).returncode


def _last_json_line(text):
    for ln in reversed((text or "").strip().splitlines()):
        ln = ln.strip()
        if ln.startswith("{"):
            try:
                return json.loads(ln)
            except ValueError:
                continue
    return None


def orchestrate(args):
    """Default 1-GPU run: this process never touches the GPU.  Child 1 (DFW_BENCH_WORKER=1) measures the headline value,
    the roofline and the cpu_baseline and prints its line; child 2 (--secondary-child) measures the secondary
    configurations.  The ONE line printed here is child 1's with child 2's dict under `secondary` -- a hang, abort or OOM
    kill in a secondary leg (bounded by --secondary-timeout) leaves the headline line intact."""
    import subprocess
    from diffews_amd import build
    build.build()
    tmp = tempfile.mkdtemp(prefix="dfw_bench_")
    ref_path = os.path.join(tmp, "oracle_z0.pt")
    me = os.path.abspath(__file__)
    env = dict(os.environ, DFW_BENCH_WORKER="1", DFW_BENCH_REF_OUT=ref_path)
    p1 = subprocess.run([sys.executable, me, *sys.argv[1:]], env=env, stdout=subprocess.PIPE, text=True)
    line = _last_json_line(p1.stdout)
    if p1.returncode != 0 or line is None:
        sys.stdout.write(p1.stdout or "")
        sys.stdout.flush()
        return p1.returncode or 1
    sec = None
    try:
        cmd = [sys.executable, me, "--secondary-child", "--dtype", args.dtype, "--batch", str(args.batch), "--res", str(args.res)]
        p2 = subprocess.run(cmd, env=dict(os.environ, DFW_BENCH_REF_IN=ref_path), stdout=subprocess.PIPE, text=True,
                            timeout=args.secondary_timeout)
        sec = _last_json_line(p2.stdout)
        if sec is None:
            sec = {"error": f"secondary child exited with code {p2.returncode} and no result"}
    except subprocess.TimeoutExpired:
        sec = {"error": f"secondary child killed after {args.secondary_timeout:.0f}s"}
    except Exception as e:      # never required for `value`
        sec = {"error": repr(e)}
    line["secondary"] = sec
    line["config"]["processes"] = "headline and secondary legs in separate child processes"
    print(json.dumps(line), flush=True)
    return 0


def secondary_child_main(args):
    from diffews_amd import build
    build.build()
    torch.cuda.set_device(0)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    pipe, blobs = build_pipeline(dtype)
    try:
        sec = secondary_measurements(pipe, blobs, args, args.res, dtype, ref_in=os.environ.get("DFW_BENCH_REF_IN"))
    except Exception as e:
        log(f"[bench] secondary measurements failed: {e!r}")
        sec = {"error": repr(e)}
    for k, v in sec.items():
        log(f"[secondary] {k}: {v}")
    print(json.dumps(sec), flush=True)
    return 0


def latest_pmc_summary():
    """Newest profiles/rNN_pmc_traffic.json (by round number): per-kernel HBM bytes and MFMA utilisation from
    the separate rocprofv3 --pmc passes of that round (profiles/run_pmc.sh)."""
    import glob
    import re
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")):
        m = re.match(r"r(\d+)_pmc_traffic\.json$", os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    if best is None:
        return None, {}
    with open(best[1]) as f:
        return os.path.basename(best[1]), json.load(f)


def ops_hook_off():
    from diffews_amd import ops
    return ops.gemm_hook is None      # the instrumented roofline pass needs the eager launches


def timed_steps(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def secondary_measurements(pipe, blobs, args, res, dtype, ref_in=None):
    """Beside the headline measurement, on a 1-GPU run: the other BASELINE.json GPU configurations through the
    same product code, so that the driver's record carries them (flat scalar keys):
      fp16_fp32stream_*  configs[1] in the mode that meets north_star's 1e-3 -- its OWN pipeline instance in fp16 storage
                     with residual_dtype=torch.float32 (the mode the reference launcher's default dtype selects, DESIGN
                     section 4) -- with the z0 error MEASURED in this run against the fp32 oracle of the cpu_baseline leg
      configs2_*     512x512 5-shot, batch 2 (24 576 keys at the 64x64 level), 5 steps, + its attention roofline
      configs4_*     the 7-shot training step (VAE-encode with sampling + UNet fwd + bwd + clip + AdamW), 3 steps, + the
                     roofline of its heaviest kernel pair, the KV-fusion attention backward (dQ + dK/dV)."""
    from diffews_amd import episodes
    from diffews_amd.metrics import AverageMeter, fold_class_ids
    out = {}
    meter = AverageMeter("coco", fold_class_ids("coco", 0), device="cuda")
    t_sec = time.time()

    def infer_config(pp, b, s, steps, warmup=2):
        bt = episodes.make_episode_batch(b, s, res, seed=300 + s, device="cuda")
        cls = episodes.episode_class_ids(list(range(b))).cuda()

        def step(captured=True):
            r = pp.run_episodes(bt["support_imgs"], bt["query_img"], bt["support_masks"], bt["query_mask"], captured=captured)
            meter.update_from_counts(r["counts"], cls)
            return r
        step(False)
        ms = timed_steps(step, warmup, steps) * 1e3
        agg = roofline_pass(lambda: step(False))
        attn = agg.get("fsa_attention")
        return ms, attn
    def leg_parity():
        # -- the parity mode: fp16 storage + fp32 residual stream, configs[1] shape, own pipeline instance
        p16, _ = build_pipeline(torch.float16, blobs=blobs, residual_dtype=torch.float32)
        ms, _ = infer_config(p16, args.batch, 1, 5)
        out["fp16_fp32stream_ms_per_step"] = round(ms, 3)
        out["fp16_fp32stream_value"] = round(args.batch / ms * 1e3, 3)
        out["fp16_fp32stream_note"] = ("configs[1] on a pipeline built in fp16 storage with residual_dtype=torch.float32 (fp32 residual "
                                       "stream, 16-bit MFMA operands): the mode torch_dtype=torch.float32 selects")
        if ref_in and os.path.isfile(ref_in):
            ref = torch.load(ref_in)
            bt = episodes.make_episode_batch(1, ref["nshot"], ref["res"], seed=ref["seed"], device="cuda")

            def z0_err(pp):
                z0 = pp.run_episodes(bt["support_imgs"], bt["query_img"], bt["support_masks"], captured=False)["z0"].float().cpu()
                return float((z0 - ref["z0"]).norm() / ref["z0"].norm())
            out["fp16_fp32stream_z0_rel_err"] = float(f"{z0_err(p16):.4e}")
            out["headline_mode_z0_rel_err"] = float(f"{z0_err(pipe):.4e}")
            out["z0_rel_err_note"] = (f"relative L2 on z0 (P:769) against the fp32 CPU oracle (cpu_baseline leg) on its {ref['res']}x{ref['res']} "
                                      f"{ref['nshot']}-shot episode, same fp16-representable checkpoint; headline mode = {args.dtype} storage + "
                                      f"{args.dtype} stream (its weights are that checkpoint rounded once more to {args.dtype})")
        del p16
        torch.cuda.empty_cache()

    def leg_configs2():
        # -- configs[2]
        ms, attn = infer_config(pipe, 2, 5, 5)
        out["configs2_ms_per_step"] = round(ms, 3)
        out["configs2_value"] = round(2 / ms * 1e3, 3)
        out["configs2_workload"] = f"SD-2.1 UNet + SD VAE, {res}x{res}, 5-shot, 2 episodes/GPU/step (BASELINE.json configs[2]), HIP graph"
        if attn:
            out["configs2_attention_tflops"] = round(attn[1] / attn[2] / 1e12, 2)
            out["configs2_attention_frac"] = round(attn[1] / attn[2] / 1e12 / MFMA_PEAK_TFLOPS, 4)
            out["configs2_attention_ms_per_step"] = round(attn[2] * 1e3, 3)
        pipe._graphs = {}
        torch.cuda.empty_cache()

    def leg_configs4():
        # -- configs[4]: the training step on this one GPU
        from diffews_amd.train import UNetTrainer, poly_lr
        ucfg, usd, vcfg, vsd, te = blobs
        s = 7
        tr = UNetTrainer(ucfg, usd, torch_dtype=dtype, loss_scale=1.0 if dtype == torch.bfloat16 else 1024.0)
        log(f"[secondary] trainer built at +{time.time() - t_sec:.1f}s")
        vae = pipe.vae
        bt = episodes.make_episode_batch(1, s, res, seed=200, device="cuda")
        qmask = (bt["query_mask"].float()[:, None].repeat(1, 3, 1, 1) * 2 - 1).contiguous()
        g = torch.Generator(device="cuda").manual_seed(1000)
        ehs = torch.randn(1, 77, ucfg["cross_attention_dim"], generator=torch.Generator().manual_seed(3)).cuda()
        srcs = [torch.cat([bt["support_imgs"], bt["query_img"]]).contiguous(), bt["support_masks"], qmask]
        st = {"step": 0, "loss": None}

        def encode():
            lat = vae.encode(srcs).latent_dist.sample(generator=g) * 0.18215
            return torch.cat([lat[:s], lat[s + 1:2 * s + 1]], 1), lat[s:s + 1], -lat[2 * s + 1:]

        def train_step():
            zc, zt, tgt = encode()
            if ops_hook_off():
                loss, _ = tr.forward_backward_captured(zc, zt, tgt, 1, ehs)
            else:
                loss, _ = tr.forward_backward(zc, zt, tgt, 1, ehs)
            tr.optimizer_step(poly_lr(1e-5, st["step"], 10000), max_grad_norm=1.0)
            st["step"] += 1
            st["loss"] = loss
        ms = timed_steps(train_step, 2, 3) * 1e3
        loss = float(st["loss"])
        if not (loss == loss and abs(loss) < 1e6):
            raise RuntimeError("configs[4]: non-finite loss")
        agg = roofline_pass(train_step)
        out["configs4_ms_per_step"] = round(ms, 3)
        out["configs4_value"] = round(1e3 / ms, 3)
        out["configs4_workload"] = (f"training step, SD-2.1 UNet 865.9 M + frozen SD VAE, {res}x{res}, 7-shot, 1 episode/GPU/step: 16 sampled "
                                    "VAE encodes (eager) + UNet fwd + bwd (one HIP graph) + clip_grad_norm_ + AdamW (BASELINE.json configs[4], one GPU)")
        out["configs4_loss"] = round(loss, 5)
        ab, af = agg.get("fsa_attention_bwd"), agg.get("fsa_attention")
        if ab:
            out["configs4_attention_bwd_kernels"] = "fsa_bwd_dq_kernel + fsa_bwd_dkv_kernel (+ delta)"
            out["configs4_attention_bwd_ms_per_step"] = round(ab[2] * 1e3, 3)
            out["configs4_attention_bwd_tflops"] = round(ab[1] / ab[2] / 1e12, 2)
            out["configs4_attention_bwd_frac"] = round(ab[1] / ab[2] / 1e12 / MFMA_PEAK_TFLOPS, 4)
        if af:
            out["configs4_attention_fwd_ms_per_step"] = round(af[2] * 1e3, 3)
            out["configs4_attention_fwd_frac"] = round(af[1] / af[2] / 1e12 / MFMA_PEAK_TFLOPS, 4)
        del tr
        torch.cuda.empty_cache()

    # every leg stands alone: a Python-level failure in one is recorded under `<leg>_error` and the others still run
    for name, leg in (("fp16_fp32stream", leg_parity), ("configs2", leg_configs2), ("configs4", leg_configs4)):
        try:
            leg()
        except Exception as e:
            log(f"[secondary] {name} failed: {e!r}")
            out[f"{name}_error"] = repr(e)
            torch.cuda.empty_cache()
        log(f"[secondary] {name} done at +{time.time() - t_sec:.1f}s")
    return out


def train_main(args):
    """`--train`: BASELINE configs[4] (train_icl_multitask_nocrop_nearest_nshot_v3.py:1320-1396).  One step per rank =
    VAE-encode with sampling of the episode's 2s+2 images (T:1347-1358, frozen VAE), lock-step UNet forward over
    [s support ; 1 query] latents, MSE against -z_mask_tag, backward, bucketed all-reduce of the flat fp32 gradient
    over the ranks (RCCL; DDP's collective, T:1226-1228), clip_grad_norm_(1) + AdamW + poly LR.  Weak scaling: one
    episode per GPU per step.  Prints one JSON line (metric: training episodes / second over all ranks)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}")
    from diffews_amd import build
    if rank == 0:
        build.build()
    import torch.distributed as dist
    backend = os.environ.get("DFW_DIST_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("DFW_ONE_DEVICE") else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"[bench] rank {rank} needs cuda:{dev_index} but this node shows {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
        dist.barrier()
        log(f"[bench] rank {rank}/{dist.get_world_size()} on cuda:{dev_index}, backend {dist.get_backend()}")
    from diffews_amd import config, episodes, weights
    from diffews_amd.train import UNetTrainer, poly_lr
    from diffews_amd.vae import AutoencoderKL
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    s = 7 if args.nshot is None else args.nshot       # configs[4] is 7-shot; an explicit --nshot is always honoured
    res = args.res
    ucfg, vcfg = config.get("tiny_unet" if args.tiny else "sd21_unet"), config.get("tiny_vae" if args.tiny else "sd_vae")
    t0 = time.time()
    tr = UNetTrainer(ucfg, weights.synthetic_unet_state_dict(ucfg), torch_dtype=dtype,
                     loss_scale=1.0 if dtype == torch.bfloat16 else 1024.0)
    vae = AutoencoderKL(vcfg, weights.synthetic_vae_state_dict(vcfg), torch_dtype=dtype)
    log(f"[bench] trainer: {tr.P.numel / 1e6:.1f} M parameters (flat fp32 master + grad + AdamW state), build {time.time() - t0:.1f}s")
    bt = episodes.make_episode_batch(1, s, res, seed=200 + rank, device="cuda")
    qmask = (bt["query_mask"].float()[:, None].repeat(1, 3, 1, 1) * 2 - 1).contiguous()       # T:1327-1334
    g = torch.Generator(device="cuda").manual_seed(1000 + rank)
    ehs = torch.randn(1, 77, ucfg["cross_attention_dim"], generator=torch.Generator().manual_seed(3)).cuda()   # 77-token prompt (T:1368)
    sf = 0.18215
    state = {"step": 0, "loss": None}

    # The frozen VAE's sampled encodes (T:1347-1358) of the episode's 2s + 2 images run as ONE batch (the encoder takes the
    # sources as a list): 90.1 -> 85.2 ms/step against four encode calls of 7 / 1 / 7 / 1 images.  `--prefetch` issues the
    # encodes of batch i + 1 on a side HIP stream while the UNet step of batch i runs (the VAE is frozen, nothing feeds
    # back); measured on MI355X it gains nothing (85.6 vs 85.2 ms: the step's kernels already fill the chip), so it is off
    # by default.  Every timed step contains exactly one set of encodes and one optimizer step either way.
    srcs = [torch.cat([bt["support_imgs"], bt["query_img"]]).contiguous(), bt["support_masks"], qmask]   # <= 3 sources

    def encode():
        lat = vae.encode(srcs).latent_dist.sample(generator=g) * sf
        z_ref, z_tag, z_mref, z_mtag = lat[:s], lat[s:s + 1], lat[s + 1:2 * s + 1], lat[2 * s + 1:]
        return torch.cat([z_ref, z_mref], 1), z_tag, -z_mtag                        # T:1360-1366

    # DDP's gradient all-reduce (T:1226-1228, T:1391), overlapped: the flat gradient's buckets are reduced from a side
    # stream while the backward of the earlier layers still runs; the loss rides in the last range (T:1387)
    comm_dt = torch.bfloat16 if args.grad_comm_dtype == "bf16" else torch.float32
    red = tr.make_reducer(bucket_elems=args.grad_bucket_mb * 250_000, comm_dtype=comm_dt) if (world > 1 or args.segmented) else None

    # one GPU: fwd + bwd as ONE HIP graph.  N GPUs: the same work as a chain of graphs cut where a gradient bucket becomes final,
    # the reducer issuing that bucket's all-reduce between two segments (round 4; --no-graph: the eager walk of rounds 2-3)
    use_graph = world == 1 and not args.no_graph and not args.segmented
    use_seg = (world > 1 or args.segmented) and not args.no_graph

    def train(lat):
        if use_seg and ops_hook_off() and red is not None:
            loss, _ = tr.forward_backward_segmented(lat[0], lat[1], lat[2], 1, ehs, red)   # rank-averaged loss, streams joined
            tr.optimizer_step(poly_lr(1e-5, state["step"], 10000), max_grad_norm=1.0)
            state["step"] += 1
            state["loss"] = loss
            return
        if use_graph and ops_hook_off():
            loss, _ = tr.forward_backward_captured(lat[0], lat[1], lat[2], 1, ehs)      # T:1367-1391, one graph replay
        else:
            loss, _ = tr.forward_backward(lat[0], lat[1], lat[2], 1, ehs, reducer=red)  # T:1367-1391
        if red is not None:
            loss = red.finish()                                                     # rank-averaged loss, streams joined
        tr.optimizer_step(poly_lr(1e-5, state["step"], 10000), max_grad_norm=1.0)  # T:1393-1395
        state["step"] += 1
        state["loss"] = loss

    prefetch = args.prefetch
    side = torch.cuda.Stream() if prefetch else None
    main_stream = torch.cuda.current_stream()

    def run(nsteps, lat):
        for _ in range(nsteps):
            if prefetch:
                side.wait_stream(main_stream)            # batch i + 1 is encoded beside step i, not beside step i - 1 too
                with torch.cuda.stream(side):
                    nxt = encode()
                train(lat)
                main_stream.wait_stream(side)
                for t in nxt:
                    t.record_stream(main_stream)
                lat = nxt
            else:
                train(lat)
                lat = encode()
        return lat

    lat = encode()
    lat = run(max(1, args.warmup), lat)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lat = run(args.steps, lat)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    loss = float(state["loss"])
    log(f"[bench] rank {rank}: {elapsed / args.steps * 1e3:.2f} ms/step, loss {loss:.5f}")
    if not (loss == loss and abs(loss) < 1e6):
        raise SystemExit("non-finite loss: invalid run")
    roof = None
    if rank == 0 and not args.no_roofline:
        # heaviest kernel pair of the step: the KV-fusion attention backward (dQ + dK/dV), live event timing on the launch
        # stream in one instrumented step; algorithmic FLOPs = 10 * 64 * heads * n_q * keys per image (5 GEMMs)
        saved = red
        red = None
        agg = roofline_pass(lambda: train(lat))
        red = saved
        ab, af = agg.get("fsa_attention_bwd"), agg.get("fsa_attention")
        if ab:
            roof = dict(bound="mfma", kernel="fsa_bwd_dq_kernel + fsa_bwd_dkv_kernel", achieved=round(ab[1] / ab[2] / 1e12, 2),
                        peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=round(ab[1] / ab[2] / 1e12 / MFMA_PEAK_TFLOPS, 4), traffic=None,
                        launches_per_step=ab[0], ms_per_step=round(ab[2] * 1e3, 3),
                        attention_fwd_frac=None if not af else round(af[1] / af[2] / 1e12 / MFMA_PEAK_TFLOPS, 4),
                        attention_fwd_ms_per_step=None if not af else round(af[2] * 1e3, 3))
    if rank == 0:
        line = {"metric": f"training episodes/sec ({res}x{res}, {s}-shot, SD-2 UNet fwd+bwd+AdamW)",
                "value": round(world * args.steps / elapsed, 3), "unit": "episodes/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": f"training step, SD-2.1 UNet 865.9 M + frozen SD VAE, {res}x{res}, {s}-shot, 1 episode/GPU/step "
                                       f"(BASELINE.json configs[4]){' TINY-DEBUG' if args.tiny else ''}",
                           "nshot": s, "resolution": res,
                           "parallelism": f"data-parallel x{world}, flat gradient all-reduce in 216 MB buckets ({args.grad_comm_dtype} on the wire) "
                                          "issued from a side stream during the backward, loss in the last bucket",
                           "optimizer": "clip_grad_norm_(1.0) + AdamW, fp32 master",
                           "hip_graph": ("UNet forward + backward (UNetTrainer.forward_backward_captured)" if use_graph else
                                         ("UNet forward + backward as a chain of graphs cut at the gradient buckets "
                                          "(UNetTrainer.forward_backward_segmented)" if use_seg else False)),
                           "vae_encode": "one batch of 2s+2 images per step" + (", next batch's encodes on a side stream during the UNet step" if prefetch else "")},
                "roofline": roof, "cpu_baseline": None}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4, help="episodes per GPU per step")
    ap.add_argument("--nshot", type=int, default=None,
                    help="shots per episode; default 1 for the inference metric (configs[1]), 7 for --train (configs[4])")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--tiny", action="store_true", help="tiny-width model (debug only; not a valid bench line)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the fp32-stream / configs[2] / configs[4] measurements that follow the headline one (N=1 only)")
    ap.add_argument("--cfg", action="append", default=[], metavar="FIELD=INT",
                    help="dfw_config field for this run (A/B of kernel plans, e.g. --cfg k8=0); not a default-changing flag")
    ap.add_argument("--vae-flash", default="auto", choices=["auto", "on", "off"],
                    help="VAE mid-block attention: flash kernel (no N x N scores), materialised scores, or per shape (default)")
    ap.add_argument("--inline", action="store_true",
                    help="run the headline and the secondary measurements in THIS process (default: two child processes, so "
                         "that a failure in a secondary leg cannot take the headline value with it); needed under rocprofv3")
    ap.add_argument("--secondary-timeout", type=float, default=600.0, help="seconds granted to the secondary child")
    ap.add_argument("--secondary-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--residual-dtype", default="storage", choices=["storage", "fp32"],
                    help="residual stream of the UNet / VAE: the storage dtype (default, fastest) or fp32 (parity mode)")
    ap.add_argument("--grad-comm-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="--train, N > 1: wire format of the gradient all-reduce (fp32 = DDP's; bf16 halves the xGMI bytes)")
    ap.add_argument("--prefetch", action="store_true", help="--train: encode the next batch on a side stream beside the step")
    ap.add_argument("--grad-bucket-mb", type=int, default=216, help="--train: size of a gradient all-reduce bucket in MB (fp32)")
    ap.add_argument("--segmented", action="store_true",
                    help="--train on ONE GPU: run the multi-GPU form of the step (forward + backward as a chain of HIP graphs cut at "
                         "the gradient-bucket boundaries, the reducer firing between segments) to measure it against the monolithic graph")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE configs[4] instead of the headline metric: training step (VAE-encode with sampling, UNet "
                         "fwd+bwd over a 7-shot episode per GPU, gradient all-reduce over the ranks, clip + AdamW)")
    args = ap.parse_args()

    if args.secondary_child:
        return secondary_child_main(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args.gpus)            # before ANY torch.cuda / HIP call in this process
    if args.train:
        return train_main(args)
    want_secondary = (args.gpus == 1 and "RANK" not in os.environ and not args.no_secondary and not args.tiny
                      and (args.res, args.nshot or 1, args.batch) == (512, 1, 4) and args.residual_dtype == "storage")
    worker = bool(os.environ.get("DFW_BENCH_WORKER"))
    if want_secondary and not args.inline and not worker:
        return orchestrate(args)                 # before ANY torch.cuda / HIP call in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch as `python bench.py --gpus N` "
                         f"(spawns its own ranks) or under torchrun with --nproc-per-node equal to --gpus")
    from diffews_amd import build, episodes
    if rank == 0:
        build.build()   # no-op when the in-tree .so is current; BEFORE the GPU is touched (hipcc children)
    import torch.distributed as dist
    # DFW_DIST_BACKEND=gloo + DFW_ONE_DEVICE=1: rehearse the N>1 control flow with every rank on
    # cuda:0 of a one-GPU box (RCCL refuses two ranks on one device); never used for a bench line.
    backend = os.environ.get("DFW_DIST_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("DFW_ONE_DEVICE") else local_rank
    ndev = torch.cuda.device_count()
    if dev_index >= ndev:
        raise SystemExit(f"[bench] rank {rank} needs cuda:{dev_index} but this node shows {ndev} GPU(s)")
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
        dist.barrier()          # the other ranks wait here for rank 0's build check
        log(f"[bench] rank {rank}/{dist.get_world_size()} on cuda:{dev_index}, backend {dist.get_backend()}")

    from diffews_amd.metrics import AverageMeter, fold_class_ids
    if args.cfg:
        from diffews_amd import _lib
        log("[bench] dfw_config:", _lib.configure(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.cfg}))
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    pipe, blobs = build_pipeline(dtype, tiny=args.tiny)
    if args.residual_dtype == "fp32":
        pipe.set_residual_dtype(torch.float32)
    if args.vae_flash != "auto":
        pipe.vae.encoder.mid.att.flash = pipe.vae.decoder.mid.att.flash = args.vae_flash == "on"

    b, s, res = args.batch, (1 if args.nshot is None else args.nshot), args.res
    bt = episodes.make_episode_batch(b, s, res, seed=100 + rank, device="cuda")
    cls = episodes.episode_class_ids(list(range(rank * b, rank * b + b))).cuda()
    meter = AverageMeter("coco", fold_class_ids("coco", 0), device="cuda")
    out = {}
    use_graph = not args.no_graph

    def step(captured=None):
        # the product's own step: one HIP-graph replay owned by the pipeline (run_episodes(captured=True)),
        # then the meter update (one small kernel)
        r = pipe.run_episodes(bt["support_imgs"], bt["query_img"], bt["support_masks"], bt["query_mask"],
                              captured=use_graph if captured is None else captured)
        meter.update_from_counts(r["counts"], cls)
        out["z0"] = r["z0"]
        out["r"] = r

    # warm-up: eager steps, then (graph mode) the capture + replays; afterwards the episode tensors ARE the
    # graph's static input buffers, so a step starts with its inputs resident in HBM and copies nothing
    for _ in range(max(1, args.warmup)):
        step(captured=False)
    torch.cuda.synchronize()
    if use_graph:
        step()
        stat = pipe.episode_input_buffers(b, s, res)
        bt = dict(support_imgs=stat["support_imgs"], query_img=stat["query_img"],
                  support_masks=stat["support_masks"], query_mask=stat["query_gt"])
        for _ in range(args.warmup):
            step()
    run = step

    meter.intersection_buf.zero_()
    meter.union_buf.zero_()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # end-of-stream metric reduction: the one collective of the evaluation path
    meter.all_reduce()
    miou, fb_iou, _ = meter.compute_iou()
    z0 = out["z0"]
    finite = bool(torch.isfinite(z0).all())
    log(f"[bench] rank {rank}: {elapsed / args.steps * 1e3:.2f} ms/step, mIoU {float(miou):.2f} FB-IoU {float(fb_iou):.2f}, "
        f"z0 finite={finite} |z0|={float(z0.abs().mean()):.4f}")
    if not finite:
        raise SystemExit("non-finite latents: invalid run")
    if os.environ.get("DFW_BENCH_DEBUG"):
        rr = out["r"]
        log("[debug] counts", rr["counts"].tolist(), "u8 max", rr["seg_u8"].flatten(1).max(1).values.tolist(),
            "dec min/max", float(rr["dec"].min()), float(rr["dec"].max()), "cls", cls.tolist(),
            "inter", meter.intersection_buf[:, [0, 4, 8, 12]].tolist(), "union", meter.union_buf[:, [0, 4, 8, 12]].tolist())

    eager_ms = None
    if rank == 0 and use_graph:                  # the same step launched kernel by kernel (host-bound): recorded, not `value`
        step(captured=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            step(captured=False)
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t1) / 3 * 1e3
        log(f"[bench] eager (no graph) {eager_ms:.2f} ms/step vs graph {elapsed / args.steps * 1e3:.2f} ms/step")
    roof = None
    if rank == 0 and not args.no_roofline:
        agg = roofline_pass(lambda: step(captured=False))
        tot_t = sum(v[2] for v in agg.values())
        for name, (n, fl, t) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
            log(f"[roofline] {name:40s} launches {n:4d}  {t * 1e3:8.3f} ms  {fl / t / 1e12:7.1f} TFLOP/s  "
                f"avg {t / n * 1e6:8.1f} us  ({100 * t / tot_t:4.1f}% of GEMM time)")
        attn = agg.pop("fsa_attention", None)   # reported beside the dominant GEMM kernel (north_star)
        tot_t = sum(v[2] for v in agg.values())
        dom = max(agg.items(), key=lambda kv: kv[1][2])
        n, fl, t = dom[1]
        ach = fl / t / 1e12
        # HBM bytes per launch and MFMA-pipe utilisation of that kernel come from separate rocprofv3 --pmc
        # passes (FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE cannot share a pass with
        # each other or with timing) summarised by profiles/summarize_pmc.py, gfx950 correction
        # (2*FETCH_SIZE + WRITE_SIZE) KiB; the newest round's file is used and named in the line.
        pmc_name, pmc = latest_pmc_summary()
        traffic = mfma_util = None
        if dom[0] in pmc:
            traffic = round(pmc[dom[0]]["hbm_bytes_per_launch"])
            mfma_util = pmc[dom[0]].get("mfma_util")
        roof = dict(bound="mfma", kernel=dom[0], achieved=round(ach, 2), peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(ach / MFMA_PEAK_TFLOPS, 4), traffic=traffic, mfma_util=mfma_util,
                    pmc_source=pmc_name, launches_per_step=n,
                    avg_launch_us=round(t / n * 1e6, 2), flops_per_launch=fl / n,
                    gemm_time_share_of_step=round(tot_t / (elapsed / args.steps), 3))
        if attn is not None:
            an, afl, at = attn
            # KV-fusion self-attention launches (QK^T + PV on MFMA, head_dim 64): the roofline north_star
            # quotes; same live HIP-event timing, algorithmic flops = 4 * 64 * heads * n_q * keys per image
            roof["attention"] = dict(kernel="fsa_ring_kernel", achieved=round(afl / at / 1e12, 2),
                                     peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                                     frac=round(afl / at / 1e12 / MFMA_PEAK_TFLOPS, 4), launches_per_step=an,
                                     ms_per_step=round(at * 1e3, 3),
                                     mfma_util=max([v.get("mfma_util") or 0.0 for k, v in pmc.items() if k.startswith("fsa_ring_kernel")],
                                                   default=None))
    cpu = None
    ref_path = os.environ.get("DFW_BENCH_REF_OUT") or (os.path.join(tempfile.mkdtemp(prefix="dfw_bench_"), "oracle_z0.pt")
                                                       if want_secondary and args.inline else None)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(blobs, res, s, ref_out=ref_path)
        except Exception as e:  # the baseline is reported, never required for `value`
            log(f"[bench] cpu_baseline failed: {e!r}")

    secondary = None
    if want_secondary and args.inline:
        try:
            secondary = secondary_measurements(pipe, blobs, args, res, dtype, ref_in=ref_path)
            for k, v in secondary.items():
                log(f"[secondary] {k}: {v}")
        except Exception as e:   # reported beside the headline value, never required for it
            log(f"[bench] secondary measurements failed: {e!r}")
            secondary = {"error": repr(e)}

    if rank == 0:
        n_gpus = world
        eps = n_gpus * b * args.steps / elapsed
        line = {
            "metric": f"few-shot episodes/sec ({res}x{res}, {s}-shot, SD-2 UNet)", "value": round(eps, 3),
            "unit": "episodes/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"SD-2.1 UNet + SD VAE, {res}x{res}, {s}-shot, {b} episodes/GPU/step "
                                   f"({'BASELINE.json configs[1]' if (res, s, b) == (512, 1, 4) else 'non-default configuration'})"
                                   f"{' TINY-DEBUG' if args.tiny else ''}",
                       "episodes_per_gpu_per_step": b, "nshot": s, "resolution": res,
                       "parallelism": f"episode-sharded x{n_gpus}", "hip_graph": use_graph,
                       "residual_dtype": "fp32" if args.residual_dtype == "fp32" else args.dtype,
                       "graph_owner": "pipeline.run_episodes(captured=True)" if use_graph else None,
                       "eager_ms_per_step": None if eager_ms is None else round(eager_ms, 3)},
            "roofline": roof, "cpu_baseline": cpu, "secondary": secondary,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
