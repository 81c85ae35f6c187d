"""Tests of the prototype scratch/proto/fsa_pipe_kernel.hip as they ran while it was part of the library (round 3):\nparametrisations to reuse when it returns.  Not collected by pytest (lives outside tests/)."""\n@pytest.fixture
def pipelined(hip_lib):
    """fsa_pipe_kernel is opt-in (dfw_config.fsa_pipelined): switch it on for the test, restore the defaults after."""
    from diffews_amd import _lib
    _lib.configure(fsa_pipelined=1)
    yield
    _lib.configure()


# fsa_pipe_kernel: the software-pipelined forward (q pre-scaled, more than 1024 query rows).  Three tiles are in flight
# per wave (S(t+2) on the matrix pipe, P(t+1) on the vector pipe, P(t).V(t) on the matrix pipe), so the cases that matter
# are tile counts 1, 2, 3, odd / even, ragged last tiles of the own and bank segments, and the deferred rescale firing in
# the pipelined body (its fix-ups touch O, l and the already-started next score tile).
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,heads,N,nkv,nshot", [(1, 2, 2048, 2048, 0), (2, 1, 1100, 1100, 1), (1, 1, 1088, 1088, 2),
                                                 (1, 1, 1100, 64, 0), (1, 2, 1100, 77, 0), (1, 1, 1100, 128, 0),
                                                 (1, 1, 1100, 192, 0), (1, 1, 1100, 200, 0), (1, 1, 1100, 320, 0),
                                                 (1, 1, 1100, 333, 0), (2, 2, 1300, 1300, 3)])
def test_fsa_attention_pipelined_kernel(ops, pipelined, dtype, B, heads, N, nkv, nshot):
    C = heads * 64
    g = torch.Generator().manual_seed(N + nkv + nshot)
    q = (torch.randn(B, N, C, generator=g)).to(dtype)
    kv = (torch.randn(B, nkv, 2 * C, generator=g)).to(dtype)
    bank = (torch.randn(max(B * nshot, 1), nkv, 2 * C, generator=g)).to(dtype)
    q_in, qf = _prescale(q, True)
    k, v = kv.float()[..., :C], kv.float()[..., C:]
    if nshot:
        k = torch.cat([k, bank.float()[..., :C].reshape(B, nshot * nkv, C)], 1)
        v = torch.cat([v, bank.float()[..., C:].reshape(B, nshot * nkv, C)], 1)
    sh = lambda t: t.reshape(B, -1, heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sh(qf), sh(k), sh(v)).transpose(1, 2).reshape(B, N, C)
    sc = torch.einsum("bhqd,bhkd->bhqk", sh(qf), sh(k)) * (64 ** -0.5) * math.log2(math.e)
    lse_ref = torch.logsumexp(sc * math.log(2.0), dim=-1) / math.log(2.0)
    kg, bg = kv.cuda(), bank.cuda()
    lse = torch.empty(B, heads, N, dtype=torch.float32, device="cuda")
    y = ops.fsa_attention(q_in.cuda(), kg[..., :C], kg[..., C:], heads, k_bank=bg[..., :C] if nshot else None,
                          v_bank=bg[..., C:] if nshot else None, nshot=nshot, q_prescaled=True, lse=lse, key_split=False)
    assert rel(y, ref) < 1.5 * TOL[dtype]
    assert float((lse.cpu() - lse_ref).abs().max()) < 2e-2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("step", [0.5, 2.0, 5.0, 12.0, -3.0, 40.0])
def test_fsa_attention_pipelined_deferred_rescale_ramp(ops, pipelined, dtype, step):
    """The ramp of test_fsa_attention_deferred_rescale_ramp on the pipelined kernel (N = 2048: 32 tiles): 0.5 never moves
    the reference after the first tile, 2.0 every 5th tile, 5.0 every 2nd, 12.0 / 40.0 every tile (consecutive pipelined
    iterations both take the fix-up path), -3.0 falls.  Every row must match: a fix-up applied to the wrong one of O, l,
    P(t), S(t+1), S(t+2) corrupts exactly the rows whose maximum moved."""
    B, heads, N, C = 1, 1, 2048, 64
    g = torch.Generator().manual_seed(int(abs(step) * 10) + 1)
    q = torch.randn(B, N, C, generator=g)
    q = q / q.norm(dim=-1, keepdim=True) * 8.0
    k = torch.randn(B, N, C, generator=g) * 0.05
    v = torch.randn(B, N, C, generator=g)
    q[..., 0] = 4.0
    c = (64 ** -0.5) * 1.4426950408889634
    tile = (torch.arange(N) // 64).float()
    k[0, :, 0] = tile * step / (4.0 * c) / (2.0 if step == 40.0 else 1.0)
    # rows differ in how strongly they see the ramp: some rows' maxima move, others' do not (per-lane d, wave-uniform branch)
    q[0, ::3, 0] = 1.0
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    q_in, qr = _prescale(q, True)
    ref = F.scaled_dot_product_attention(qr[:, None], k.float()[:, None], v.float()[:, None])[:, 0]
    y = ops.fsa_attention(q_in.cuda(), k.cuda(), v.cuda(), heads, q_prescaled=True).float().cpu()
    assert rel(y, ref) < 1.5 * TOL[dtype]
    row_err = (y - ref).norm(dim=-1) / (ref.norm(dim=-1) + 1e-6)
    assert float(row_err.max()) < 6 * TOL[dtype], float(row_err.max())


@pytest.mark.parametrize("dtype", DTYPES)
def test_fsa_attention_pipelined_lockstep_equals_two_launches(ops, pipelined, dtype):
    b, nshot, heads, N = 2, 2, 2, 1100
    C = heads * 64
    n_ref = b * nshot
    qkv = rnd((n_ref + b, N, 3 * C), dtype, 13).cuda()
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    q, _ = _prescale(q, True)
    q = q.contiguous()
    two = torch.empty(n_ref + b, N, C, dtype=dtype, device="cuda")
    ops.fsa_attention(q[:n_ref], k[:n_ref], v[:n_ref], heads, out=two[:n_ref], q_prescaled=True)
    ops.fsa_attention(q[n_ref:], k[n_ref:], v[n_ref:], heads, k[:n_ref], v[:n_ref], nshot=nshot, out=two[n_ref:],
                      q_prescaled=True, key_split=False)
    one = ops.fsa_attention(q, k, v, heads, k[:n_ref], v[:n_ref], nshot=nshot, n_plain=n_ref, q_prescaled=True, key_split=False)
    assert torch.equal(one, two)


