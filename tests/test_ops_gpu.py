"""GPU parity of every HIP op against a plain PyTorch fp32 reference of the same op.

Inputs are rounded to the storage dtype first, so the reference sees exactly the operands the
kernel sees; the kernels accumulate in fp32, so the only differences are accumulation order and
the final rounding of the output to the storage dtype (tolerances below state this).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16]
# relative L2 tolerance: output rounding (2^-9 bf16 / 2^-12 fp16, rms ~ 0.6x) + fp32 accumulation order
TOL = {torch.bfloat16: 4e-3, torch.float16: 6e-4}


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


@pytest.fixture(scope="module")
def ops(hip_lib):
    from diffews_amd import ops
    return ops


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (200, 320, 320), (4096, 960, 320), (130, 64, 1280),
                                   (64, 1280, 2560), (8, 256, 1024), (1000, 192, 128)])
def test_linear(ops, dtype, M, N, K):
    x, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
    bias = torch.randn(N)
    res = rnd((M, N), dtype, 3)
    ref = x.float() @ w.float().t() + bias + res.float()
    y = ops.linear(x.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda())
    assert rel(y, ref) < TOL[dtype]
    y32 = ops.linear(x.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda(), out_f32=True)
    assert rel(y32, ref) < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("splitk", [1, 2, 5])
def test_linear_splitk_rowbias_silu(ops, dtype, splitk):
    M, N, K, rpi = 96, 128, 640, 32
    x, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
    bias, rb = torch.randn(N), torch.randn(M // rpi, N)
    ref = F.silu((x.float() @ w.float().t() + bias + rb.repeat_interleave(rpi, 0)) * 0.5)
    y = ops.linear(x.cuda(), w.cuda(), bias=bias.cuda(), rowbias=rb.cuda(), rows_per_img=rpi, act=1,
                   out_scale=0.5, splitk=splitk)
    assert rel(y, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_strided_view(ops, dtype):
    """A operand as a column slice of a wider buffer (q/k/v views of the fused QKV output)."""
    M, K, N = 300, 128, 192
    big = rnd((M, 3 * K), dtype, 5)
    w = rnd((N, K), dtype, 6, K ** -0.5)
    y = ops.linear(big.cuda()[:, K:2 * K], w.cuda())
    assert rel(y, big[:, K:2 * K].float() @ w.float().t()) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_geglu(ops, dtype):
    from diffews_amd.packing import pack_geglu
    M, C = 260, 128
    x = rnd((M, C), dtype, 1)
    w, b = rnd((8 * C, C), dtype, 2, C ** -0.5), torch.randn(8 * C) * 0.1
    h = x.float() @ w.float().t() + b
    a, g = h.chunk(2, dim=-1)
    ref = a * F.gelu(g)
    wp, bp = pack_geglu(w, b)
    y = ops.linear(x.cuda(), wp.cuda(), bias=bp.cuda(), geglu=True)
    assert y.shape == (M, 4 * C)
    assert rel(y, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,pad,ups", [
    (2, 16, 16, 64, 64, 1, 1, False), (1, 12, 20, 128, 320, 1, 1, False), (2, 16, 16, 64, 128, 2, 1, False),
    (2, 16, 16, 64, 64, 2, 0, False), (1, 8, 8, 128, 64, 1, 1, True), (3, 8, 8, 1280, 1280, 1, 1, False),
    (1, 64, 64, 128, 128, 1, 1, False)])
def test_conv3x3(ops, dtype, B, H, W, Cin, Cout, stride, pad, ups):
    from diffews_amd.packing import pack_conv3x3
    x = rnd((B, Cin, H, W), dtype, 1)
    w = rnd((Cout, Cin, 3, 3), dtype, 2, (9 * Cin) ** -0.5)
    bias = torch.randn(Cout)
    xin = x.float()
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    if stride == 2 and pad == 0:
        xin = F.pad(xin, (0, 1, 0, 1))
    ref = F.conv2d(xin, w.float(), bias, stride=stride, padding=pad)
    rb = torch.randn(B, Cout)
    ref = ref + rb[:, :, None, None]
    y = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().cuda(), pack_conv3x3(w).cuda(), Cout, bias=bias.cuda(),
                    stride=stride, pad=pad, ups=ups, rowbias=rb.cuda())
    assert rel(y.permute(0, 3, 1, 2), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout,ups", [(3, 128, 128, 64, 256, False), (3, 128, 128, 64, 128, False),
                                               (1, 112, 240, 128, 256, False), (2, 80, 80, 64, 128, True)])
def test_conv3x3_big_tiles(ops, dtype, B, H, W, Cin, Cout, ups):
    """Shapes large enough for the 256-row LDS-DMA ring kernel (gemm_big.hip), incl. residual,
    per-image bias and the fused nearest-2x upsample."""
    from diffews_amd.packing import pack_conv3x3
    x = rnd((B, Cin, H, W), dtype, 1)
    w = rnd((Cout, Cin, 3, 3), dtype, 2, (9 * Cin) ** -0.5)
    bias, rb = torch.randn(Cout), torch.randn(B, Cout)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
    ref = F.conv2d(xin, w.float(), bias, padding=1) + rb[:, :, None, None]
    res = rnd(tuple(ref.shape), dtype, 3)
    ref = (ref + res.float()) * 0.5
    y = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().cuda(), pack_conv3x3(w).cuda(), Cout, bias=bias.cuda(),
                    ups=ups, rowbias=rb.cuda(), residual=res.permute(0, 2, 3, 1).contiguous().cuda(), out_scale=0.5)
    assert rel(y.permute(0, 3, 1, 2), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(1, 512, 512, 128, 128), (2, 256, 256, 64, 128), (1, 320, 336, 64, 128),
                                            (3, 512, 256, 64, 384)])
def test_conv3x3_patch_tiles(ops, dtype, B, H, W, Cin, Cout):
    """Shapes the LDS-resident-patch kernel (conv_patch.hip, 512 x 128 tile) takes by default: result against torch
    fp32 on the same 16-bit inputs (first and last image, all four borders), with residual, per-image bias and
    output scale in the epilogue, and its fused GroupNorm sums against a separate pass."""
    from diffews_amd.packing import pack_conv3x3
    names = []
    ops.gemm_hook = lambda name, *a: names.append(name)
    try:
        x = rnd((B, H, W, Cin), dtype, 1).cuda()
        w = rnd((Cout, Cin, 3, 3), dtype, 2, (9 * Cin) ** -0.5).cuda()
        bias, rb = torch.randn(Cout).cuda(), torch.randn(B, Cout).cuda()
        res = rnd((B, H, W, Cout), dtype, 3).cuda()
        y = ops.conv3x3(x, pack_conv3x3(w.cpu()).cuda(), Cout, bias=bias, rowbias=rb, residual=res, out_scale=0.5,
                        gn_groups=32)
    finally:
        ops.gemm_hook = None
    assert names and names[0].startswith("conv_patch_kernel"), names
    for i in {0, B - 1}:
        ref = F.conv2d(x[i:i + 1].float().permute(0, 3, 1, 2), w.float(), bias, padding=1) + rb[i][None, :, None, None]
        ref = ((ref.permute(0, 2, 3, 1) + res[i:i + 1].float()) * 0.5).cpu()
        assert rel(y[i:i + 1].cpu(), ref) < TOL[dtype]
    if Cout // 32 & (Cout // 32 - 1):                  # 12 channels per group: no fused sums, groupnorm() takes its own
        assert getattr(y, "_gn_stats", None) is None
        return
    part, chunks, groups = y._gn_stats
    g, b = (torch.randn(Cout) * 0.2 + 1).cuda(), (torch.randn(Cout) * 0.2).cuda()
    fused = ops.groupnorm(y, g, b, 32, 1e-6, silu=True)
    plain = ops.groupnorm(y.clone(), g, b, 32, 1e-6, silu=True)
    assert rel(fused.cpu(), plain.cpu()) < 1e-3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 128, 128, 64, 256), (3, 128, 128, 64, 128), (2, 128, 192, 128, 512)])
def test_conv_fused_groupnorm_stats(ops, dtype, B, H, W, Cin, Cout):
    """GroupNorm statistics emitted by the producing conv's epilogue == a separate statistics pass."""
    from diffews_amd.packing import pack_conv3x3
    x = rnd((B, H, W, Cin), dtype, 1).cuda()
    w = pack_conv3x3(rnd((Cout, Cin, 3, 3), dtype, 2, (9 * Cin) ** -0.5)).cuda()
    bias = torch.randn(Cout).cuda()
    g, b = (torch.randn(Cout) * 0.2 + 1).cuda(), (torch.randn(Cout) * 0.2).cuda()
    y = ops.conv3x3(x, w, Cout, bias=bias, gn_groups=32)
    assert getattr(y, "_gn_stats", None) is not None, "fused statistics expected for this shape"
    fused = ops.groupnorm(y, g, b, 32, 1e-6, silu=True)
    plain = ops.groupnorm(y.clone(), g, b, 32, 1e-6, silu=True)          # clone drops the attribute
    ref = F.silu(F.group_norm(y.float().permute(0, 3, 1, 2), 32, g, b, eps=1e-6)).permute(0, 2, 3, 1)
    assert rel(fused, ref) < TOL[dtype] and rel(plain, ref) < TOL[dtype]
    assert rel(fused, plain) < 1e-3
    # fp32 residual stream (round 4): the fp32-output epilogues emit the sums of the fp32 values they store, from registers
    res = torch.randn(B, H, W, Cout).cuda()
    y32 = ops.conv3x3(x, w, Cout, bias=bias, residual=res, out_f32=True, gn_groups=32)
    assert y32.dtype == torch.float32
    if Cout == 128:       # 96 tiles of 512 x 128: below the big kernels' tile threshold in the fp32-output plan -> no fused sums
        return
    assert getattr(y32, "_gn_stats", None) is not None, "fused statistics expected (fp32 output)"
    fused = ops.groupnorm(y32, g, b, 32, 1e-6, silu=True, out_dtype=dtype)
    plain = ops.groupnorm(y32.clone(), g, b, 32, 1e-6, silu=True, out_dtype=dtype)
    ref = F.silu(F.group_norm(y32.permute(0, 3, 1, 2), 32, g, b, eps=1e-6)).permute(0, 2, 3, 1)
    assert rel(fused, ref) < TOL[dtype] and rel(fused, plain) < 1e-3
    st2 = ops.conv3x3(x, w, Cout, bias=bias, residual=res, out_f32=True, gn_groups=32)._gn_stats[0]
    assert torch.equal(st2, y32._gn_stats[0])          # fixed summation order


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("silu", [True, False])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(4, 160, 160, 64, 128), (3, 128, 128, 128, 256), (10, 96, 112, 128, 128)])
def test_conv_groupnorm_input(ops, dtype, silu, B, H, W, Cin, Cout):
    """conv3x3(gn_in=...) = dfw_groupnorm(+SiLU) followed by the conv (ResnetBlock2D norm + nonlinearity + conv as one call;
    zero padding of the NORMALISED tensor), and the conv's output statistics feed the next GroupNorm."""
    from diffews_amd.packing import pack_conv3x3
    x = (rnd((B, H, W, Cin), dtype, 1) * 2 + 0.3).cuda()
    w = pack_conv3x3(rnd((Cout, Cin, 3, 3), dtype, 2, (9 * Cin) ** -0.5)).cuda()
    bias = torch.randn(Cout).cuda()
    res = rnd((B, H, W, Cout), dtype, 3).cuda()
    g, b = (torch.randn(Cin) * 0.2 + 1).cuda(), (torch.randn(Cin) * 0.2).cuda()
    g2, b2 = (torch.randn(Cout) * 0.2 + 1).cuda(), (torch.randn(Cout) * 0.2).cuda()
    fused = ops.conv3x3(x, w, Cout, bias=bias, residual=res, gn_groups=32, gn_in=(g, b, 32, 1e-6, silu))
    xn = ops.groupnorm(x, g, b, 32, 1e-6, silu=silu)
    plain = ops.conv3x3(xn, w, Cout, bias=bias, residual=res)
    ref = F.conv2d(xn.float().permute(0, 3, 1, 2), unpack3x3(w, Cin), bias, padding=1).permute(0, 2, 3, 1) + res.float()
    assert rel(plain, ref) < TOL[dtype] and rel(fused, ref) < TOL[dtype]
    assert torch.equal(fused, plain)
    assert getattr(fused, "_gn_stats", None) is not None
    n_f = ops.groupnorm(fused, g2, b2, 32, 1e-6, silu=True)
    n_p = ops.groupnorm(plain, g2, b2, 32, 1e-6, silu=True)
    assert rel(n_f, n_p) < 1e-3


def unpack3x3(wp, Cin):
    """[Cout, 9*Cin] (ky, kx, cin) -> [Cout, Cin, 3, 3] fp32."""
    return wp.float().view(wp.shape[0], 3, 3, Cin).permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("B,H,Cin,Cout", [(12, 128, 256, 256), (3, 512, 128, 128), (12, 64, 512, 512)])
def test_conv_ring_kernels_race_screen(ops, B, H, Cin, Cout):
    """The ping-pong conv kernels keep three LDS-DMA stages in flight across two barriers per K-step; an
    ordering slip there shows up as run-to-run differences long before it fails a tolerance.  Same problem
    launched repeatedly with unrelated work in between: outputs and fused statistics must be bit-identical."""
    from diffews_amd.packing import pack_conv3x3
    dtype = torch.bfloat16
    x = rnd((B, H, H, Cin), dtype, 1).cuda()
    w = pack_conv3x3(rnd((Cout, Cin, 3, 3), dtype, 2, (9 * Cin) ** -0.5)).cuda()
    bias = torch.randn(Cout).cuda()
    res = rnd((B, H, H, Cout), dtype, 3).cuda()
    junk = torch.randn(2048, 2048, device="cuda")
    first = ops.conv3x3(x, w, Cout, bias=bias, residual=res, gn_groups=32)
    st0 = first._gn_stats[0].clone()
    for it in range(12):
        if it % 3 == 0:
            junk = junk @ junk * 1e-4
        y = ops.conv3x3(x, w, Cout, bias=bias, residual=res, gn_groups=32)
        assert torch.equal(y, first) and torch.equal(y._gn_stats[0], st0), it


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_big_tiles(ops, dtype):
    from diffews_amd.packing import pack_geglu
    for M, N, K in [(49152, 256, 128), (24576, 384, 320), (50000, 128, 64 * 3)]:
        x, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
        bias, res = torch.randn(N), rnd((M, N), dtype, 3)
        ref = x.float() @ w.float().t() + bias + res.float()
        assert rel(ops.linear(x.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda()), ref) < TOL[dtype], (M, N, K)
    M, C = 12288, 128
    x = rnd((M, C), dtype, 1)
    w, b = rnd((8 * C, C), dtype, 2, C ** -0.5), torch.randn(8 * C) * 0.1
    a, g = (x.float() @ w.float().t() + b).chunk(2, dim=-1)
    wp, bp = pack_geglu(w, b)
    assert rel(ops.linear(x.cuda(), wp.cuda(), bias=bp.cuda(), geglu=True), a * F.gelu(g)) < TOL[dtype]


def _kernel_name(ops, fn):
    """Name of the GEMM kernel the library plans for the call made inside fn (ops.gemm_hook sees every dfw_gemm)."""
    names = []
    ops.gemm_hook = lambda name, flops, e0, e1, shape=None: names.append(name)
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        ops.gemm_hook = None
    return names


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm8_linear(ops, dtype):
    """gemm8_kernel (round 4: 64-deep K-tiles, half-tile LDS-DMA staging, one instruction stream for both wave groups) on
    Linear shapes: against the fp32 reference, and BIT-EQUAL to gemm_big_kernel (dfw_config.k8 = 0) -- both accumulate the
    same 32-deep MFMA steps in the same order.  Ragged M (the buffer bounds check zero-fills the tail rows), several tiles
    per workgroup (the K-tile stream runs on across tiles), odd and even K-tile counts, bias / residual / row bias /
    column scale, GEGLU, batched (VAE attention) launches."""
    from diffews_amd import _lib
    from diffews_amd.packing import pack_geglu
    try:
        # N % 256 == 0: the 256 x 256 tile; N = 384 / 960 / 128: the 256 x 128 tile (dfw_config.k8 = 3 plans it where gemm_big's
        # 512 x 128 or 256 x 128 tiles were planned; 960 = a half-empty last column tile, K = 128 = two K-tiles per tile)
        for M, N, K in [(49152, 256, 256), (50000, 512, 320), (131072, 256, 448), (70000, 768, 1280), (49152, 384, 320),
                        (33000, 960, 320), (200000, 128, 256), (786432, 256, 128)]:
            x, w = rnd((M, K), dtype, 1).cuda(), rnd((N, K), dtype, 2, K ** -0.5).cuda()
            bias, res = torch.randn(N).cuda(), rnd((M, N), dtype, 3).cuda()
            rpi = 1000
            rb = torch.randn((M + rpi - 1) // rpi, N).cuda()
            kw = dict(bias=bias, residual=res, rowbias=rb, rows_per_img=rpi, out_scale=0.5, colscale=(64, 0.25))
            _lib.configure(k8=3)
            assert any(n.startswith("gemm8_kernel") for n in _kernel_name(ops, lambda: ops.linear(x, w, **kw))), (M, N, K)
            y8 = ops.linear(x, w, **kw)
            _lib.configure(k8=0)
            y0 = ops.linear(x, w, **kw)
            ref = x.float() @ w.float().t() + bias + res.float() + rb.repeat_interleave(rpi, 0)[:M]
            ref = torch.cat([ref[:, :64] * 0.25, ref[:, 64:] * 0.5], 1)
            assert rel(y8, ref) < TOL[dtype], (M, N, K)
            assert torch.equal(y8, y0), (M, N, K, float((y8.float() - y0.float()).abs().max()))
        # GEGLU (the 64^2-level FF projection shape class)
        M, C = 24576, 128
        x = rnd((M, C * 2), dtype, 1).cuda()
        w, b = rnd((8 * C, 2 * C), dtype, 2, (2 * C) ** -0.5), torch.randn(8 * C) * 0.1
        a, g = (x.float().cpu() @ w.float().t() + b).chunk(2, dim=-1)
        wp, bp = pack_geglu(w, b)
        _lib.configure(k8=3)
        y8 = ops.linear(x, wp.cuda(), bias=bp.cuda(), geglu=True)
        _lib.configure(k8=0)
        y0 = ops.linear(x, wp.cuda(), bias=bp.cuda(), geglu=True)
        assert rel(y8, a * F.gelu(g)) < TOL[dtype] and torch.equal(y8, y0)
        # batched: [Bt, M, K] x [Bt, N, K]^T (VAE mid-block attention products)
        q, k = rnd((3, 4096, 512), dtype, 5).cuda(), rnd((3, 4096, 512), dtype, 6, 512 ** -0.5).cuda()
        _lib.configure(k8=3)
        s8 = ops.bmm_nt(q, k)
        _lib.configure(k8=0)
        s0 = ops.bmm_nt(q, k)
        assert rel(s8, torch.bmm(q.float(), k.float().transpose(1, 2))) < TOL[dtype] and torch.equal(s8, s0)
    finally:
        _lib.configure()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(32768, 320, 320), (33000, 320, 1280), (49152, 320, 128), (70000, 320, 640), (40000, 960, 320)])
def test_gemm8_linear_n160(ops, dtype, M, N, K):
    """gemm8_kernel's 256 x 160 tile (round 4: the N = 320 linears of the UNet's 64^2 level; 64 x 80 wave tiles, one 160-row W stage
    per K-tile, 160-byte staging rows): against the fp32 reference and against gemm_kernel (dfw_config.k8 = 0: same products,
    another summation order); ragged M, two K-tiles per tile, one and several tiles per workgroup, bias / residual / row bias /
    column scale (N = 960: the fused QKV projection)."""
    from diffews_amd import _lib
    x, w = rnd((M, K), dtype, 1).cuda(), rnd((N, K), dtype, 2, K ** -0.5).cuda()
    bias, res = torch.randn(N).cuda(), rnd((M, N), dtype, 3).cuda()
    rpi = 4096
    rb = torch.randn((M + rpi - 1) // rpi, N).cuda()
    kw = dict(bias=bias, residual=res, rowbias=rb, rows_per_img=rpi, out_scale=0.5, colscale=(64, 0.25))
    try:
        names = _kernel_name(ops, lambda: ops.linear(x, w, **kw))
        assert names == ["gemm8_kernel<%s,256,160,64,lin>" % ("bf16" if dtype == torch.bfloat16 else "f16")], names
        y8 = ops.linear(x, w, **kw)
        y8b = ops.linear(x, w, **kw)
        _lib.configure(k8=0)
        y0 = ops.linear(x, w, **kw)
    finally:
        _lib.configure()
    ref = x.float() @ w.float().t() + bias + res.float() + rb.repeat_interleave(rpi, 0)[:M]
    ref = torch.cat([ref[:, :64] * 0.25, ref[:, 64:] * 0.5], 1)
    assert rel(y8, ref) < TOL[dtype]
    assert rel(y8, y0) < (2e-3 if dtype == torch.bfloat16 else 3e-4)
    assert torch.equal(y8, y8b)
    # a plain call (no epilogue operands)
    assert rel(ops.linear(x, w), x.float() @ w.float().t()) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,pad,ups", [(3, 128, 128, 64, 256, 1, 1, True), (4, 256, 256, 128, 256, 2, 0, False),
                                                          (12, 128, 128, 320, 256, 2, 1, False), (2, 96, 160, 192, 512, 1, 1, True),
                                                          (4, 256, 256, 128, 128, 2, 0, False), (3, 64, 64, 640, 128, 1, 1, True), (3, 64, 64, 320, 384, 1, 1, True)])
def test_gemm8_conv_gather(ops, dtype, B, H, W, Cin, Cout, stride, pad, ups):
    """gemm8_kernel on the conv3x3 shapes that gather A per tap: fused nearest-2x upsampling (Upsample2D), stride 2 with
    symmetric (UNet Downsample2D) and asymmetric (VAE encoder: F.pad(0,1,0,1), padding 0) padding; residual, per-image bias,
    fused GroupNorm statistics; odd K-tile counts (Cin = 320: 45) and several tiles per workgroup.  Against F.conv2d and
    against gemm_big_kernel (same products, another summation order of the 32-deep steps: fp32 rounding only)."""
    from diffews_amd import _lib
    from diffews_amd.packing import pack_conv3x3
    x = rnd((B, Cin, H, W), dtype, 1)
    w = rnd((Cout, Cin, 3, 3), dtype, 2, (9 * Cin) ** -0.5)
    bias, rb = torch.randn(Cout), torch.randn(B, Cout)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
    if stride == 2 and pad == 0:
        ref = F.conv2d(F.pad(xin, (0, 1, 0, 1)), w.float(), bias, stride=2)
    else:
        ref = F.conv2d(xin, w.float(), bias, stride=stride, padding=1)
    ref = ref + rb[:, :, None, None]
    res = rnd(tuple(ref.shape), dtype, 3)
    ref = ref + res.float()
    xc, wc = x.permute(0, 2, 3, 1).contiguous().cuda(), pack_conv3x3(w).cuda()
    kw = dict(bias=bias.cuda(), stride=stride, pad=pad, ups=ups, rowbias=rb.cuda(),
              residual=res.permute(0, 2, 3, 1).contiguous().cuda(), gn_groups=32)
    try:
        _lib.configure(k8=3)
        names = _kernel_name(ops, lambda: ops.conv3x3(xc, wc, Cout, **kw))
        assert any(n.startswith("gemm8_kernel") for n in names), names
        y8 = ops.conv3x3(xc, wc, Cout, **kw)
        y8b = ops.conv3x3(xc, wc, Cout, **kw)
        _lib.configure(k8=0)
        y0 = ops.conv3x3(xc, wc, Cout, **kw)
    finally:
        _lib.configure()
    assert rel(y8.permute(0, 3, 1, 2), ref) < TOL[dtype]
    assert rel(y8, y0) < (2e-3 if dtype == torch.bfloat16 else 3e-4)
    assert torch.equal(y8, y8b)                                           # run-to-run: no race in the ring
    if Cout % 32 == 0 and ((Cout // 32) & (Cout // 32 - 1)) == 0:        # channel groups that tile the 64-column wave tiles
        assert torch.equal(y8._gn_stats[0], y8b._gn_stats[0])
        assert y0._gn_stats[1:] == y8._gn_stats[1:] and rel(y8._gn_stats[0], y0._gn_stats[0]) < 1e-3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(12, 128, 128, 256, 256), (3, 512, 512, 128, 128), (2, 256, 256, 64, 128), (3, 128, 128, 320, 256),
                                            (1, 320, 336, 64, 320), (4, 64, 64, 320, 320), (2, 96, 160, 192, 512), (1, 256, 256, 128, 384),
                                            (4, 64, 64, 512, 512), (8, 64, 64, 320, 320), (8, 64, 64, 640, 320), (12, 64, 64, 192, 320),
                                            (2, 128, 128, 64, 320)])
def test_conv_patch8(ops, dtype, B, H, W, Cin, Cout):
    """conv_patch8_kernel (round 4: the LDS-resident 18 x 18 x 64-channel patch on the eight-phase schedule; dfw_config.conv_patch = 4
    plans its 256 x 256 tile for N % 256 == 0 and its 256 x 128 tile for the other N % 64 == 0 layers): against F.conv2d on the
    same 16-bit inputs (first and last image: all four borders), against conv_patch_kernel / gemm_kernel (conv_patch = 2: same
    products, another summation order), run-to-run bit-equal, fused GroupNorm sums.  One chunk per tile (Cin = 64: the staging
    area and the next tile's patch alternate every tile), odd chunk counts (Cin = 320, 192), several tiles per workgroup,
    ragged N (320 = 2.5 column tiles), N % 256 == 0 with too few 256 x 256 tiles (4 x 64^2 x 512: the 256 x 128 tile), N = 320 with enough rows for
    its 256 x 160 tile (64 x 80 wave tiles, 160-byte staging rows; one and several tiles per workgroup, one and ten chunks)."""
    from diffews_amd import _lib
    from diffews_amd.packing import pack_conv3x3
    x = rnd((B, H, W, Cin), dtype, 1).cuda()
    w = rnd((Cout, Cin, 3, 3), dtype, 2, (9 * Cin) ** -0.5).cuda()
    wp = pack_conv3x3(w.cpu()).cuda()
    bias, rb = torch.randn(Cout).cuda(), torch.randn(B, Cout).cuda()
    res = rnd((B, H, W, Cout), dtype, 3).cuda()
    kw = dict(bias=bias, rowbias=rb, residual=res, out_scale=0.5, gn_groups=32)
    try:
        _lib.configure(conv_patch=4)
        names = _kernel_name(ops, lambda: ops.conv3x3(x, wp, Cout, **kw))
        assert names and names[0].startswith("conv_patch8_kernel"), names
        y8 = ops.conv3x3(x, wp, Cout, **kw)
        junk = torch.randn(2048, 2048, device="cuda")
        junk = junk @ junk
        y8b = ops.conv3x3(x, wp, Cout, **kw)
        _lib.configure(conv_patch=2)
        y0 = ops.conv3x3(x, wp, Cout, **kw)
    finally:
        _lib.configure()
    for i in {0, B - 1}:
        ref = F.conv2d(x[i:i + 1].float().permute(0, 3, 1, 2), w.float(), bias, padding=1) + rb[i][None, :, None, None]
        ref = ((ref.permute(0, 2, 3, 1) + res[i:i + 1].float()) * 0.5).cpu()
        assert rel(y8[i:i + 1].cpu(), ref) < TOL[dtype]
    assert rel(y8, y0) < (2e-3 if dtype == torch.bfloat16 else 3e-4)
    assert torch.equal(y8, y8b)                                           # run-to-run: no race in the rings
    if ((Cout // 32) & (Cout // 32 - 1)) == 0:                           # channel groups that tile the 64-column wave tiles
        assert y8._gn_stats is not None and torch.equal(y8._gn_stats[0], y8b._gn_stats[0])
        g, b = (torch.randn(Cout) * 0.2 + 1).cuda(), (torch.randn(Cout) * 0.2).cuda()
        fused = ops.groupnorm(y8, g, b, 32, 1e-6, silu=True)
        plain = ops.groupnorm(y8.clone(), g, b, 32, 1e-6, silu=True)
        assert rel(fused, plain) < 1e-3
    else:
        assert getattr(y8, "_gn_stats", None) is None


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Cout", [3, 4, 8])
def test_conv3x3_small_cout_nchw(ops, dtype, Cout):
    from diffews_amd.packing import pack_conv3x3
    B, H, W, Cin = 2, 16, 24, 128
    x = rnd((B, Cin, H, W), dtype, 1)
    w = rnd((Cout, Cin, 3, 3), dtype, 2, (9 * Cin) ** -0.5)
    bias = torch.randn(Cout)
    ref = -F.conv2d(x.float(), w.float(), bias, padding=1)
    y = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().cuda(), pack_conv3x3(w).cuda(), Cout, bias=bias.cuda(),
                    out_nchw_f32=True, out_scale=-1.0)
    assert y.shape == ref.shape and rel(y, ref) < 2e-5


def _prescale(q, pre):
    """pre: (q handed to the kernel already multiplied by attn.scale * log2 e and rounded once, the q the
    reference must then be computed from) -- the kernel's input is what it is judged on."""
    if not pre:
        return q, q.float()
    from diffews_amd.ops import FSA_QSCALE
    q_in = (q.float() * FSA_QSCALE).to(q.dtype)
    return q_in, q_in.float() / FSA_QSCALE


PRE = [False, True]


@pytest.mark.parametrize("pre", PRE, ids=["scale_in_kernel", "q_prescaled"])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,heads,N,nshot", [(1, 1, 64, 0), (2, 2, 256, 0), (2, 2, 256, 1), (1, 5, 200, 2),
                                             (2, 1, 16, 3), (1, 2, 1024, 1), (2, 3, 100, 0)])
def test_fsa_attention(ops, dtype, B, heads, N, nshot, pre):
    """Empty bank == plain SDPA; with a bank == SDPA over [own ; shot0 ; shot1 ...] per episode
    (attention_processor.py:256-258: ref image = episode*nshot + shot)."""
    C = heads * 64
    qkv = rnd((B, N, 3 * C), dtype, 1)
    bank = rnd((max(B * nshot, 1), N, 3 * C), dtype, 2)
    q, k, v = qkv.float().split(C, dim=-1)
    q_in, q = _prescale(qkv[..., :C], pre)
    if nshot:
        kb, vb = bank.float()[..., C:2 * C], bank.float()[..., 2 * C:]
        k = torch.cat([k, kb.reshape(B, nshot * N, C)], dim=1)
        v = torch.cat([v, vb.reshape(B, nshot * N, C)], dim=1)
    sh = lambda t: t.reshape(B, -1, heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sh(q), sh(k), sh(v)).transpose(1, 2).reshape(B, N, C)
    qg, bg = qkv.cuda(), bank.cuda()
    y = ops.fsa_attention(q_in.cuda(), qg[..., C:2 * C], qg[..., 2 * C:], heads,
                          k_bank=bg[..., C:2 * C] if nshot else None, v_bank=bg[..., 2 * C:] if nshot else None,
                          nshot=nshot, q_prescaled=pre)
    assert rel(y, ref) < 1.5 * TOL[dtype]  # + P rounded to the storage dtype before PV


@pytest.mark.parametrize("pre", PRE, ids=["scale_in_kernel", "q_prescaled"])
@pytest.mark.parametrize("dtype", DTYPES)
def test_fsa_attention_online_softmax_rescale(ops, dtype, pre):
    """Force the running-max rescale branch: one late key dominates each query row."""
    B, heads, N = 1, 1, 256
    C = 64
    q, k, v = rnd((B, N, C), dtype, 1), rnd((B, N, C), dtype, 2), rnd((B, N, C), dtype, 3)
    k[0, 200] = q[0, 17] * 4  # spike in the 4th key tile
    k[0, 70] = q[0, 100] * 3
    q, qr = _prescale(q, pre)
    ref = F.scaled_dot_product_attention(qr[:, None], k.float()[:, None], v.float()[:, None])[:, 0]
    y = ops.fsa_attention(q.cuda(), k.cuda(), v.cuda(), heads, q_prescaled=pre)
    assert rel(y, ref) < 1.5 * TOL[dtype]
    assert rel(y[0, 17], ref[0, 17]) < 3 * TOL[dtype]


@pytest.mark.parametrize("pre", PRE, ids=["scale_in_kernel", "q_prescaled"])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("step", [0.5, 2.0, 5.0, 12.0, -3.0])
def test_fsa_attention_deferred_rescale_ramp(ops, dtype, step, pre):
    """The kernel rescales O / l only when a row's maximum grew by more than 2^8 since the last rescale
    (deferred rescale).  A bounded random test never takes either side deliberately, so build rows whose
    score maximum climbs by `step` log2-units per 64-key tile over 16 tiles: 0.5 never fires after the
    first tile (P up to 2^8 at the old scale), 2.0 fires every 5th tile, 5.0 every 2nd, 12.0 every tile,
    -3.0 is a FALLING ramp (the first tile holds the maximum, later tiles sink 45 log2-units below the
    reference) -- and per row, every element must still match (a mis-ordered rescale corrupts only the rows
    that grew).  q_prescaled exercises the accumulator-initialised reference maximum of the same kernel."""
    B, heads, N, C = 1, 1, 1024, 64
    g = torch.Generator().manual_seed(int(abs(step) * 10))
    q = torch.randn(B, N, C, generator=g)
    q = q / q.norm(dim=-1, keepdim=True) * 8.0
    k = torch.randn(B, N, C, generator=g) * 0.05
    v = torch.randn(B, N, C, generator=g)
    # key j of tile t carries a component along a common direction u that lifts every row's score by
    # t * step (log2 units) when q has a unit component along u
    u = torch.zeros(C); u[0] = 1.0
    q[..., 0] = 4.0                                    # q.u = 4
    c = (64 ** -0.5) * 1.4426950408889634             # score -> log2 units
    tile = (torch.arange(N) // 64).float()
    k[0, :, 0] = tile * step / (4.0 * c)
    half = (N // 2)
    k[0, half:, 0] = k[0, half:, 0].flip(0) if step == 5.0 else k[0, half:, 0]   # also a falling tail
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    q, qr = _prescale(q, pre)
    ref = F.scaled_dot_product_attention(qr[:, None], k.float()[:, None], v.float()[:, None])[:, 0]
    y = ops.fsa_attention(q.cuda(), k.cuda(), v.cuda(), heads, q_prescaled=pre).float().cpu()
    assert rel(y, ref) < 1.5 * TOL[dtype]
    row_err = (y - ref).norm(dim=-1) / (ref.norm(dim=-1) + 1e-6)
    assert float(row_err.max()) < 6 * TOL[dtype], float(row_err.max())


@pytest.mark.parametrize("pre", PRE, ids=["scale_in_kernel", "q_prescaled"])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,nshot,heads,N", [(2, 1, 2, 256), (1, 3, 1, 192), (3, 2, 2, 100)])
def test_fsa_attention_lockstep_launch(ops, dtype, b, nshot, heads, N, pre):
    """n_plain: ONE launch over [support images ; query images] == the bank-fill launch on the support
    images followed by the bank-reading launch on the queries (A:251-267), bit for bit, and == SDPA."""
    C = heads * 64
    n_ref = b * nshot
    qkv = rnd((n_ref + b, N, 3 * C), dtype, 11).cuda()
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    q, qr = _prescale(q, pre)
    q, qr = q.contiguous(), qr.cuda()
    two = torch.empty(n_ref + b, N, C, dtype=dtype, device="cuda")
    ops.fsa_attention(q[:n_ref], k[:n_ref], v[:n_ref], heads, out=two[:n_ref], q_prescaled=pre)
    ops.fsa_attention(q[n_ref:], k[n_ref:], v[n_ref:], heads, k[:n_ref], v[:n_ref], nshot=nshot, out=two[n_ref:],
                      q_prescaled=pre)
    one = ops.fsa_attention(q, k, v, heads, k[:n_ref], v[:n_ref], nshot=nshot, n_plain=n_ref, q_prescaled=pre)
    assert torch.equal(one, two)
    q = qr
    sh = lambda t: t.float().reshape(t.shape[0], -1, heads, 64).transpose(1, 2)
    kq = torch.cat([k[n_ref:], k[:n_ref].reshape(b, nshot * N, C)], dim=1)
    vq = torch.cat([v[n_ref:], v[:n_ref].reshape(b, nshot * N, C)], dim=1)
    ref_q = F.scaled_dot_product_attention(sh(q[n_ref:]), sh(kq), sh(vq)).transpose(1, 2).reshape(b, N, C)
    ref_s = F.scaled_dot_product_attention(sh(q[:n_ref]), sh(k[:n_ref]), sh(v[:n_ref])).transpose(1, 2).reshape(n_ref, N, C)
    assert rel(one[n_ref:], ref_q) < 1.5 * TOL[dtype] and rel(one[:n_ref], ref_s) < 1.5 * TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,nshot,heads,N,n_plain_mode", [(1, 7, 2, 1024, True), (2, 5, 1, 2048, True), (2, 3, 2, 2100, False)])
def test_fsa_attention_key_split(ops, dtype, b, nshot, heads, N, n_plain_mode):
    """Many shots: the bank-reading images' key range is split over several workgroups and the partial softmax results
    merged (dfw_fsa_args.workspace).  Same result as the unsplit launch up to the merge's fp32 arithmetic, == SDPA over the
    materialised concat, and the row log-sum-exp the backward reads is the merged one."""
    import ctypes
    from diffews_amd import _lib
    C = heads * 64
    n_ref = b * nshot
    qkv = rnd((n_ref + b, N, 3 * C), dtype, 21).cuda()
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    q, qr = _prescale(q, True)
    q, qr = q.contiguous(), qr.cuda()
    if n_plain_mode:
        args = (q, k, v, heads, k[:n_ref], v[:n_ref])
        kw = dict(nshot=nshot, n_plain=n_ref, q_prescaled=True)
        Bt = n_ref + b
    else:
        args = (q[n_ref:], k[n_ref:], v[n_ref:], heads, k[:n_ref], v[:n_ref])
        kw = dict(nshot=nshot, q_prescaled=True)
        Bt = b
    lse1 = torch.empty(Bt, heads, N, dtype=torch.float32, device="cuda")
    lse0 = torch.empty_like(lse1)
    split = ops.fsa_attention(*args, lse=lse1, **kw)
    plain = ops.fsa_attention(*args, lse=lse0, key_split=False, **kw)
    a = _lib.FsaArgs()
    a.batch, a.heads, a.n_q, a.n_kv, a.n_bank, a.nshot, a.n_plain = Bt, heads, N, N, N, nshot, (n_ref if n_plain_mode else 0)
    assert _lib.lib().dfw_fsa_workspace_bytes(ctypes.byref(a)) > 0, "this shape is expected to take the split path"
    assert rel(split, plain) < TOL[dtype]      # two 16-bit roundings of P against different running maxima, one of the output
    assert float((lse1 - lse0).abs().max()) < 1e-3
    sh = lambda t: t.float().reshape(t.shape[0], -1, heads, 64).transpose(1, 2)
    kq = torch.cat([k[n_ref:], k[:n_ref].reshape(b, nshot * N, C)], dim=1)
    vq = torch.cat([v[n_ref:], v[:n_ref].reshape(b, nshot * N, C)], dim=1)
    ref_q = F.scaled_dot_product_attention(sh(qr[n_ref:]), sh(kq), sh(vq)).transpose(1, 2).reshape(b, N, C)
    assert rel(split[-b:], ref_q) < 1.5 * TOL[dtype]


def _vattn_ref(q, k, v):
    s = torch.bmm(q.double(), k.double().transpose(1, 2)) * 512 ** -0.5
    return torch.bmm(torch.softmax(s, -1), v.double()).float()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N", [(2, 4096), (1, 1024), (3, 1680), (1, 100), (2, 33)])
def test_vae_attention(ops, dtype, B, N):
    """Flash attention of the VAE mid-block (one head of dim 512, csrc/vae_attention.hip) against an fp64 softmax attention
    over the materialised scores: whole and ragged key tiles (N % 32 != 0: masked keys, unstored query rows), column-slice
    views of one fused QKV buffer, q pre-scaled as the fused projection leaves it."""
    qkv = rnd((B, N, 1536), dtype, 1).cuda()
    q, k, v = qkv[..., :512], qkv[..., 512:1024], qkv[..., 1024:]
    qs = (q.float() * ops.VATTN_QSCALE).to(dtype)               # what linear(..., colscale=(512, VATTN_QSCALE)) hands over
    buf = torch.cat([qs, k, v], -1).contiguous()
    y = ops.vae_attention(buf[..., :512], buf[..., 512:1024], buf[..., 1024:])
    ref = _vattn_ref(qs.float() / ops.VATTN_QSCALE, k.float(), v.float())
    assert y.shape == (B, N, 512) and rel(y, ref) < (6e-3 if dtype == torch.bfloat16 else 1e-3)
    assert torch.equal(y, ops.vae_attention(buf[..., :512], buf[..., 512:1024], buf[..., 1024:]))     # deterministic


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("spike_at", [40, 700, 1023, 2047])
def test_vae_attention_reference_restart(ops, dtype, spike_at):
    """The flash kernel takes the first key tile's row maximum as its reference and never rescales its accumulators inside
    the tile loop: a later score more than 2^8 above the reference makes the WORKGROUP leave the loop, compute the exact row
    maxima in one extra pass and start over.  Forced here: one key far along the sequence (also in the LAST tile) matches a
    block of queries ~30 log2 units better than anything before it, in one workgroup only (the other workgroups of the launch
    take the plain path); checked against the fp64 reference on the whole tensor."""
    B, N = 1, 2048
    g = torch.Generator().manual_seed(7)
    q = torch.randn(B, N, 512, generator=g) * 0.3
    k = torch.randn(B, N, 512, generator=g) * 0.3
    v = torch.randn(B, N, 512, generator=g)
    u = torch.randn(512, generator=g)
    u = u / u.norm()
    q[0, 300:420] += 25.0 * u          # rows 300..419 (workgroups 2 and 3) ...
    k[0, spike_at] += 25.0 * u          # ... meet this key: + 625 / sqrt(512) * log2(e) ~ 40 log2 units
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    qs = (q.float() * ops.VATTN_QSCALE).to(dtype).cuda()
    y = ops.vae_attention(qs, k.cuda(), v.cuda())
    ref = _vattn_ref(qs.float().cpu() / ops.VATTN_QSCALE, k.float(), v.float())
    assert torch.isfinite(y).all()
    assert rel(y, ref) < (6e-3 if dtype == torch.bfloat16 else 1e-3)
    assert rel(y[0, 300:420], ref[0, 300:420]) < (8e-3 if dtype == torch.bfloat16 else 1.5e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("L", [2, 77])
def test_cross_attention(ops, dtype, L):
    B, heads, N = 2, 3, 300
    C = heads * 64
    q, kv = rnd((B, N, C), dtype, 1), rnd((B, L, 2 * C), dtype, 2)
    sh = lambda t: t.reshape(B, -1, heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sh(q.float()), sh(kv.float()[..., :C]), sh(kv.float()[..., C:]))
    ref = ref.transpose(1, 2).reshape(B, N, C)
    kvg = kv.cuda()
    y = ops.cross_attention(q.cuda(), kvg[..., :C], kvg[..., C:], heads)
    assert rel(y, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,HW,C,silu", [(2, 256, 64, True), (1, 1000, 320, False), (3, 64, 2560, True),
                                         (2, 4096, 128, True), (1, 16, 1920, True)])
def test_groupnorm(ops, dtype, B, HW, C, silu):
    x = rnd((B, HW, C), dtype, 1) * 2 + 0.5
    g, b = torch.randn(C) * 0.2 + 1, torch.randn(C) * 0.2
    ref = F.group_norm(x.float().transpose(1, 2), 32, g, b, eps=1e-5).transpose(1, 2)
    if silu:
        ref = F.silu(ref)
    y = ops.groupnorm(x.cuda(), g.cuda(), b.cuda(), 32, 1e-5, silu)
    assert rel(y, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,C", [(100, 64), (1000, 320), (257, 640), (64, 1280)])
def test_layernorm(ops, dtype, rows, C):
    x = rnd((rows, C), dtype, 1) * 3 + 1
    g, b = torch.randn(C) * 0.2 + 1, torch.randn(C) * 0.2
    ref = F.layer_norm(x.float(), (C,), g, b, 1e-5)
    assert rel(ops.layernorm(x.cuda(), g.cuda(), b.cuda()), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Cin,Cout,taps,nchw", [(3, 64, 9, False), (8, 320, 9, False), (4, 512, 9, False),
                                                (8, 4, 1, True), (4, 4, 1, True)])
def test_conv_small(ops, dtype, Cin, Cout, taps, nchw):
    B, H, W = 2, 20, 12
    x = torch.randn(B, Cin, H, W)
    k = 3 if taps == 9 else 1
    w = torch.randn(Cout, Cin, k, k) * (taps * Cin) ** -0.5
    bias = torch.randn(Cout)
    ref = (F.conv2d(x * 0.5, w, bias, padding=k // 2)) * 2.0
    wp = w.permute(0, 2, 3, 1).reshape(Cout, taps, Cin).contiguous()
    y = ops.conv_small(x.cuda(), wp.cuda(), bias.cuda(), Cout, taps, dtype, nchw_f32_out=nchw, in_scale=0.5,
                       out_scale=2.0)
    if nchw:
        assert rel(y, ref) < 1e-5
    else:
        assert rel(y.permute(0, 3, 1, 2), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Cin,H,W,Cout", [(3, 3, 64, 64, 128), (2, 4, 32, 32, 512), (2, 8, 32, 64, 256), (1, 3, 128, 128, 128)])
def test_conv_small_fused_groupnorm_stats(ops, dtype, B, Cin, H, W, Cout):
    """conv_in's epilogue emits the first GroupNorm's partial sums (of the stored, rounded values):
    groupnorm(y) with them == groupnorm of the same tensor with its own statistics pass."""
    from diffews_amd.packing import pack_conv_small
    g = torch.Generator().manual_seed(B * 100 + Cout)
    x = (torch.rand(B, Cin, H, W, generator=g) * 2 - 1).cuda()
    w = pack_conv_small(torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    gm, bt = (torch.randn(Cout, generator=g) * 0.2 + 1).cuda(), (torch.randn(Cout, generator=g) * 0.2).cuda()
    y = ops.conv_small(x, w, bias, Cout, 9, dtype, gn_groups=32)
    assert getattr(y, "_gn_stats", None) is not None, "fused statistics expected for this shape"
    y0 = ops.conv_small(x, w, bias, Cout, 9, dtype)
    assert torch.equal(y, y0)
    fused = ops.groupnorm(y, gm, bt, 32, 1e-6, silu=True)
    plain = ops.groupnorm(y0, gm, bt, 32, 1e-6, silu=True)
    ref = F.silu(F.group_norm(y0.float().permute(0, 3, 1, 2), 32, gm, bt, eps=1e-6)).permute(0, 2, 3, 1)
    assert rel(fused, ref) < TOL[dtype] and rel(fused, plain) < 1e-3


@pytest.mark.parametrize("dtype", DTYPES)
def test_softmax_transpose_concat_bmm(ops, dtype):
    s = torch.randn(3, 50, 264) * 4
    y = ops.softmax_rows(s.cuda(), dtype, scale=0.3)
    assert rel(y, torch.softmax(s * 0.3, -1)) < TOL[dtype]
    x = rnd((3, 70, 136), dtype, 1)
    assert torch.equal(ops.transpose(x.cuda()).cpu(), x.transpose(1, 2).contiguous())
    a, b = rnd((2, 5, 7, 64), dtype, 2), rnd((2, 5, 7, 128), dtype, 3)
    assert torch.equal(ops.concat_channels(a.cuda(), b.cuda()).cpu(), torch.cat([a, b], -1))
    p, q = rnd((3, 100, 128), dtype, 4), rnd((3, 72, 128), dtype, 5)
    assert rel(ops.bmm_nt(p.cuda(), q.cuda(), out_f32=True), p.float() @ q.float().transpose(1, 2)) < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,groups,L", [(1000, 5, 2), (64, 20, 2), (300, 10, 4), (7, 1, 64)])
def test_softmax_groups(ops, dtype, rows, groups, L):
    """Per-(row, group) softmax over L fp32 scores, zeroed padding (folded prompt attention)."""
    x = (torch.randn(rows, 64, generator=torch.Generator().manual_seed(rows)) * 3).cuda()
    y = ops.softmax_groups(x, groups, L, dtype).float().cpu()
    ref = torch.zeros(rows, 64)
    ref[:, :groups * L] = torch.softmax(x.cpu()[:, :groups * L].view(rows, groups, L), dim=-1).view(rows, -1)
    assert rel(y, ref) < TOL[dtype] and float(y[:, groups * L:].abs().max() if groups * L < 64 else 0) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_timestep_embedding(ops, dtype):
    t = torch.tensor([1.0, 500.0, 999.0])
    half = 160
    freq = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    e = t[:, None] * freq[None]
    ref = torch.cat([torch.cos(e), torch.sin(e)], -1)
    y = ops.timestep_embedding(t.cuda(), 320, dtype)
    assert (y.float().cpu() - ref).abs().max() < (1e-2 if dtype == torch.bfloat16 else 2e-3)


def test_seg_postprocess_bit_exact(ops):
    """uint8 image and inter/union counts are integer work: bit-exact against the reference formulas
    (pipeline P:790-795,534; main_oss.py:128-137; evaluation.py:24-38)."""
    import numpy as np
    g = torch.Generator().manual_seed(0)
    B, H, W = 3, 40, 56
    x = (torch.rand(B, 3, H, W, generator=g) * 2.4 - 1.2)
    gt = (torch.rand(B, H, W, generator=g) > 0.5).to(torch.uint8)
    gt[1][torch.rand(H, W, generator=g) > 0.9] = 255
    u8, counts = ops.seg_postprocess(x.cuda(), gt.cuda(), 0.25)
    seg = (x.clip(-1, 1).clip(-1.0, 1.0) * 0.5 + 0.5) * 255
    ref_u8 = seg.clip(0, 255).numpy().astype(np.uint8)
    assert np.array_equal(u8.cpu().numpy(), ref_u8)
    for b in range(B):
        pred = torch.from_numpy(ref_u8[b]).float().div(255)[None]
        pm = (pred.mean(dim=1) > pred.max() * 0.25).float()[0]
        g_ = gt[b].float()
        pm[g_ == 255] = 255
        same = pm[pm == g_]
        inter = torch.histc(same, bins=2, min=0, max=1)
        union = torch.histc(pm, bins=2, min=0, max=1) + torch.histc(g_, bins=2, min=0, max=1) - inter
        assert counts[b].cpu().tolist() == [int(inter[0]), int(inter[1]), int(union[0]), int(union[1])]


def test_seg_postprocess_threshold_modes(ops):
    """The launcher's other threshold choices (main_oss.py:128-135) for B > 1: `pred_mask.max()` over the whole
    batch tensor (batch_max=True) and the fixed --threshold (r_threshold <= 0), bit-exact against a host
    restatement; with neither flag the reference's shape assert fails -- rejected here."""
    import numpy as np
    g = torch.Generator().manual_seed(3)
    B, H, W = 4, 24, 40
    x = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    x[1] *= 0.3                                     # images with different maxima: per-image != batch-global
    x[2] *= 0.6
    gt = (torch.rand(B, H, W, generator=g) > 0.5).to(torch.uint8)
    u8_ref = ((x.clip(-1, 1) * 0.5 + 0.5) * 255).clip(0, 255).numpy().astype(np.uint8)
    pred_t = torch.from_numpy(u8_ref).float().div(255)            # to_tensor of the uint8 image(s)

    def counts_of(pm):
        out = []
        for b in range(B):
            p_, g_ = pm[b].float(), gt[b].float()
            inter = torch.histc(p_[p_ == g_], bins=2, min=0, max=1)
            union = torch.histc(p_, bins=2, min=0, max=1) + torch.histc(g_, bins=2, min=0, max=1) - inter
            out.append([int(inter[0]), int(inter[1]), int(union[0]), int(union[1])])
        return out
    _, c_glob = ops.seg_postprocess(x.cuda(), gt.cuda(), 0.25, batch_max=True)
    assert c_glob.cpu().tolist() == counts_of(pred_t.mean(dim=1) > pred_t.max() * 0.25)
    _, c_img = ops.seg_postprocess(x.cuda(), gt.cuda(), 0.25)
    per_img = torch.stack([pred_t[b].mean(dim=0) > pred_t[b].max() * 0.25 for b in range(B)])
    assert c_img.cpu().tolist() == counts_of(per_img) and c_img.cpu().tolist() != c_glob.cpu().tolist()
    _, c_fix = ops.seg_postprocess(x.cuda(), gt.cuda(), 0.0, threshold=0.4)
    assert c_fix.cpu().tolist() == counts_of(pred_t.mean(dim=1) > 0.4)
    with pytest.raises(RuntimeError):
        ops.seg_postprocess(x.cuda(), gt.cuda(), 0.0, threshold=0.0)


def test_meter_update_kernel_equals_index_add(ops):
    """dfw_meter_update == AverageMeter.update's index_add_ (logger.py:35-37), repeated classes included."""
    from diffews_amd.metrics import AverageMeter, fold_class_ids
    g = torch.Generator().manual_seed(1)
    a = AverageMeter("coco", fold_class_ids("coco", 0), device="cuda")
    b = AverageMeter("coco", fold_class_ids("coco", 0), device="cpu")
    for _ in range(5):
        counts = torch.randint(0, 300000, (6, 4), generator=g, dtype=torch.int64)
        cls = torch.tensor([0, 4, 4, 76, 8, 0])
        a.update_from_counts(counts.cuda(), cls.cuda())
        b.update(counts[:, 0:2].t(), counts[:, 2:4].t(), cls)
    assert torch.equal(a.intersection_buf.cpu(), b.intersection_buf) and torch.equal(a.union_buf.cpu(), b.union_buf)
    assert float(a.compute_iou()[0]) == float(b.compute_iou()[0])


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_column_scale(ops, dtype):
    """colscale: the first n output columns (the q third of the fused [Wq;Wk;Wv] projection) leave the epilogue
    multiplied by s in fp32 BEFORE the single rounding; the other columns are untouched.  Every GEMM kernel
    family that serves a QKV projection: 64x64-level rows (gemm_big) and short / ragged row counts (gemm.hip)."""
    from diffews_amd.ops import FSA_QSCALE
    for M, C in [(32768, 320), (8192, 640), (2048, 1280), (300, 64), (512, 1280)]:
        x, w = rnd((M, C), dtype, 1), rnd((3 * C, C), dtype, 2, C ** -0.5)
        ref = x.float() @ w.float().t()
        ref[:, :C] *= FSA_QSCALE
        y = ops.linear(x.cuda(), w.cuda(), colscale=(C, FSA_QSCALE))
        assert rel(y[:, :C], ref[:, :C]) < TOL[dtype] and rel(y[:, C:], ref[:, C:]) < TOL[dtype], (M, C)
        # same bits as scaling a wider-precision result: compare with the fp32-output GEMM scaled on the host
        y32 = ops.linear(x.cuda(), w.cuda(), out_f32=True).cpu()
        y32[:, :C] *= FSA_QSCALE
        assert torch.equal(y.cpu(), y32.to(dtype)), (M, C)


# ---------------------------------------------------------------------------------------------------------------
# fp32 residual stream (residual_dtype=torch.float32): fp32 residual / fp32 output epilogues, fp32-input norms,
# fp32 NHWC conv_in, fp32 concat, fp32 -> storage copy.  The stream tensor is never rounded to 16 bits, so the
# fp32 outputs must match the fp32 reference to accumulation-order precision (2e-5), not to storage rounding.
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (200, 320, 320), (4096, 640, 2560), (8192, 320, 1280), (64, 1280, 5120)])
def test_linear_fp32_residual_stream(ops, dtype, M, N, K):
    x, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
    bias = torch.randn(N)
    res = torch.randn(M, N, generator=torch.Generator().manual_seed(3)) * 3   # fp32, NOT representable in 16 bits
    ref = x.float() @ w.float().t() + bias + res
    y32 = ops.linear(x.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda(), out_f32=True)
    assert y32.dtype == torch.float32 and rel(y32, ref) < 2e-5
    y16 = ops.linear(x.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda())     # ff.net.2 -> proj_out operand
    assert y16.dtype == dtype and rel(y16, ref) < TOL[dtype]
    # forced split-K: the residual is added by the reduce pass
    y32s = ops.linear(x.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda(), out_f32=True, splitk=2)
    assert rel(y32s, ref) < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,pad,ups", [(2, 32, 32, 128, 128, 1, 1, False), (1, 64, 64, 256, 256, 1, 1, False),
                                                           (2, 16, 16, 320, 640, 1, 1, False), (1, 32, 32, 128, 128, 2, 0, False),
                                                           (1, 16, 16, 64, 64, 1, 1, True), (3, 8, 8, 1280, 1280, 1, 1, False)])
def test_conv3x3_fp32_residual_stream(ops, dtype, B, H, W, Cin, Cout, stride, pad, ups):
    from diffews_amd.packing import pack_conv3x3
    x = rnd((B, H, W, Cin), dtype, 1)
    w = rnd((Cout, Cin, 3, 3), dtype, 2, (9 * Cin) ** -0.5)
    bias = torch.randn(Cout)
    xin = x.float().permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    if stride == 2 and pad == 0:
        xin = F.pad(xin, (0, 1, 0, 1))
    ref = F.conv2d(xin, w.float(), bias, stride=stride, padding=pad).permute(0, 2, 3, 1)
    res = torch.randn(ref.shape, generator=torch.Generator().manual_seed(5)) * 3
    wp = pack_conv3x3(w).cuda()
    y = ops.conv3x3(x.cuda(), wp, Cout, bias=bias.cuda(), stride=stride, pad=pad, ups=ups, residual=res.cuda().contiguous(),
                    out_f32=True, gn_groups=32)
    assert y.dtype == torch.float32 and rel(y, ref + res) < 2e-5
    y0 = ops.conv3x3(x.cuda(), wp, Cout, bias=bias.cuda(), stride=stride, pad=pad, ups=ups, out_f32=True)
    assert rel(y0, ref) < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,HW,C,silu", [(2, 256, 64, True), (1, 1000, 320, False), (3, 64, 2560, True), (2, 4096, 128, True)])
def test_groupnorm_layernorm_fp32_input(ops, dtype, B, HW, C, silu):
    g0 = torch.Generator().manual_seed(11)
    x = torch.randn(B, HW, C, generator=g0) * 2 + 0.5       # fp32 stream
    g, b = torch.randn(C) * 0.2 + 1, torch.randn(C) * 0.2
    ref = F.group_norm(x.transpose(1, 2), 32, g, b, eps=1e-5).transpose(1, 2)
    if silu:
        ref = F.silu(ref)
    y = ops.groupnorm(x.cuda(), g.cuda(), b.cuda(), 32, 1e-5, silu, out_dtype=dtype)
    assert y.dtype == dtype and rel(y, ref) < TOL[dtype]
    with pytest.raises(TypeError):
        ops.groupnorm(x.cuda(), g.cuda(), b.cuda(), 32, 1e-5, silu)
    if C <= 2048:       # LayerNorm rows of the UNet are at most 1280 wide (kernel limit 2048)
        x2 = x.reshape(-1, C)[:300].contiguous() * 3 + 1
        refl = F.layer_norm(x2, (C,), g, b, 1e-5)
        yl = ops.layernorm(x2.cuda(), g.cuda(), b.cuda(), out_dtype=dtype)
        assert yl.dtype == dtype and rel(yl, refl) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_fp32_stream_glue(ops, dtype):
    from diffews_amd.packing import pack_conv_small
    g = torch.Generator().manual_seed(2)
    # conv_in with an fp32 NHWC output (8w kernel: W % 8 == 0; generic kernels: W = 12, 10)
    for (B, Cin, H, W, Cout) in ((2, 4, 16, 16, 320), (1, 3, 20, 12, 128), (1, 8, 6, 10, 64)):
        x = torch.rand(B, Cin, H, W, generator=g) * 2 - 1
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2
        bias = torch.randn(Cout, generator=g)
        ref = F.conv2d(x, w, bias, padding=1).permute(0, 2, 3, 1)
        y = ops.conv_small(x.cuda(), pack_conv_small(w).cuda(), bias.cuda(), Cout, 9, dtype, out_f32=True, gn_groups=32)
        assert y.dtype == torch.float32 and rel(y, ref) < 1e-5 and getattr(y, "_gn_stats", None) is None
    a, b = torch.randn(3, 7, 5, 64, generator=g), torch.randn(3, 7, 5, 128, generator=g)
    assert torch.equal(ops.concat_channels(a.cuda(), b.cuda()).cpu(), torch.cat([a, b], -1))
    c = ops.to_storage(a.cuda(), dtype)
    assert c.dtype == dtype and torch.equal(c.cpu(), a.to(dtype))
    assert ops.to_storage(c, dtype) is c


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_stream_operand_split(ops, dtype):
    """split_storage: hi + lo reproduces the fp32 tensor far below the storage rounding; linear_stream / conv3x3_stream on an
    fp32 input therefore match the fp32 reference of the UNROUNDED input to accumulation precision."""
    from diffews_amd.packing import pack_conv3x3
    g = torch.Generator().manual_seed(3)
    x = torch.randn(512, 256, generator=g) * 3
    hi, lo = ops.split_storage(x.cuda(), dtype)
    assert hi.dtype == dtype and torch.equal(hi.cpu(), x.to(dtype))
    err = rel(hi.float() + lo.float(), x)
    assert err < (1e-6 if dtype == torch.float16 else 3e-5), err           # 2^-22 fp16, 2^-16 bf16
    w = rnd((128, 256), dtype, 4, 256 ** -0.5)
    bias = torch.randn(128)
    res = torch.randn(512, 128, generator=g)
    y = ops.linear_stream(x.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda())
    ref = x @ w.float().t() + bias + res
    assert y.dtype == torch.float32 and rel(y, ref) < (3e-6 if dtype == torch.float16 else 6e-5)
    plain = ops.linear(x.to(dtype).cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda(), out_f32=True)
    assert rel(plain, ref) > 5 * rel(y, ref)                               # the one-operand form carries the input rounding
    xc = torch.randn(2, 32, 32, 128, generator=g) * 2
    wc = rnd((128, 128, 3, 3), dtype, 5, (9 * 128) ** -0.5)
    yc = ops.conv3x3_stream(xc.cuda(), pack_conv3x3(wc).cuda(), 128, bias=bias.cuda(), stride=2, pad=0)
    refc = F.conv2d(F.pad(xc.permute(0, 3, 1, 2), (0, 1, 0, 1)), wc.float(), bias, stride=2).permute(0, 2, 3, 1)
    assert yc.dtype == torch.float32 and rel(yc, refc) < (3e-6 if dtype == torch.float16 else 6e-5)
    # 16-bit stream: the plain ops
    y16 = ops.linear_stream(x.to(dtype).cuda(), w.cuda(), bias=bias.cuda())
    assert y16.dtype == dtype
