// PROTOTYPE, not built into libdiffews_hip.so (round 3): the software-pipelined KV-fusion attention forward.
// Parity-green when it lived in attention.hip (tests/test_ops_gpu.py::test_fsa_attention_pipelined_* at commit 8ffb950..),
// measured 438-465 us against the ring kernel's 352 us on the 64x64-level lock-step launch, so it was
// taken out of the library again.  Kept here as the starting point of the next attempt: it plugs into attention.hip after
// fsa_ring_kernel (uses FsaP, dma16, make_srd, lds_tr_read, half_swap_* of that file / attention_common.h) and is launched with
// 512 threads, one workgroup per CU, 6 * 16 KiB of dynamic LDS.
// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined variant (q pre-scaled, one 32-row query block per wave, 8 waves, ONE workgroup per CU).
//
// Why: in fsa_ring_kernel every wave runs  QK^T(t) -> softmax(t) -> P.V(t)  as one dependency chain and the per-tile
// barrier keeps the workgroup's waves in phase, so a SIMD alternates between "all its waves in MFMA chains" and "all its
// waves in softmax VALU": the counters show SQ_ACTIVE_INST_VALU + MFMA-busy = 87 % of the elapsed cycles -- the two pipes
// run back to back, not side by side (profiles/r02_pmc_attention.txt).  At head_dim 64 a tile has 512 MFMA cycles beside
// ~650 VALU issue cycles per wave, so the pipes must overlap INSIDE a wave's instruction stream.
//
// Here the three stages of consecutive tiles are independent work in one basic block of iteration t:
//     matrix pipe :  S(t+2) = K(t+2) . Q^T  (accumulators start at -m_ref)      and   O += V(t)^T . P(t)^T
//     vector pipe :  P(t+1) = exp2(S(t+1)), row sums, 16-bit conversion
// (register cost: two score tiles + two P tiles + O = 170 VGPRs => two waves per SIMD).  The row maximum of S(t+1) and the
// (rare, deferred) rescale decision sit in front of the block: when the reference maximum moves by d, O, l and the
// already-started S(t+2) are corrected together (O *= 2^-d, l *= 2^-d, S(t+2) -= d), so P, O and l always share one scale.
// K/V stages: ring of 6 (tile t's V, tile t+2's K live; three tiles in flight), one barrier per tile as before.
// VAR (experiments, dfw_config.fsa_pipelined = 1 + VAR): 0 = one-slot read-ahead, scheduling barrier per slot;
// 1 = the same source without the per-slot barriers (the compiler schedules the 16-MFMA block); 2 = three-slot read-ahead
// (four fragment buffers per operand), barrier per slot.
template <typename T, int NW, int VAR = 0>
__global__ __launch_bounds__(NW * 64, 1) void fsa_pipe_kernel(const FsaP p) {
  constexpr int KT = 64, S = 6;
  constexpr float kDefer = 8.0f;
  constexpr int TILE = KT * 128;            // bytes of one K (or V) tile
  constexpr int STAGE = 2 * TILE;
  constexpr int DPS = 16 / NW;              // DMA wave-instructions per stage per wave (K + V)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  int head, b, qblk;
  {
    const int X = (int)gridDim.x, H = (int)gridDim.y, P = H * (int)gridDim.z;
    const int v = (int)blockIdx.x + X * ((int)blockIdx.y + H * (int)blockIdx.z);
    if ((P & 7) == 0 && p.xcd_remap) {
      const int c = v & 7, k = v >> 3;
      const int pr = c + 8 * (k / X);
      qblk = k - (k / X) * X;
      head = pr % H;
      b = (int)gridDim.z - 1 - pr / H;
    } else {
      qblk = (int)blockIdx.x;
      head = (int)blockIdx.y;
      b = (int)gridDim.z - 1 - (int)blockIdx.z;
    }
  }
  int split = 0;
  if (p.nsplit > 1 && b >= p.n_plain) {
    const int v = b - p.n_plain;
    split = v % p.nsplit;
    b = p.n_plain + v / p.nsplit;
  }
  const int bank_b = b - p.n_plain;
  const int q0 = qblk * (NW * 32) + wave * 32;
  const uint32_t lds0 = lds_addr(smem);

  const __amdgpu_buffer_rsrc_t rq = make_rsrc(p.q, p.q_bytes);
  const u32x4 rk = make_srd(p.k, p.k_bytes), rv = make_srd(p.v, p.v_bytes);
  const u32x4 rkb = make_srd(p.kb ? p.kb : p.k, p.kb ? p.kb_bytes : 0u);
  const u32x4 rvb = make_srd(p.vb ? p.vb : p.v, p.vb ? p.vb_bytes : 0u);

  typename Tr<T>::v8 qf[4];
  {
    const int qrow = q0 + lr;
    const uint32_t base = qrow < p.n_q
        ? (uint32_t)(((size_t)b * p.q_bs + (size_t)qrow * p.ldq + head * 64 + lh * 8) * sizeof(T)) : kOOB;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = as_v8<T>(buf_load16(rq, base + (uint32_t)(s * 32)));
  }

  const int tiles_own = (p.n_kv + KT - 1) / KT;
  const int tiles_bank = (p.nshot > 0 && bank_b >= 0) ? (p.n_bank + KT - 1) / KT : 0;
  const int nseg = 1 + (tiles_bank ? p.nshot : 0);
  const bool parted = p.nsplit > 1 && bank_b >= 0;
  const int seg0 = parted ? split * nseg / p.nsplit : 0, seg1 = parted ? (split + 1) * nseg / p.nsplit : nseg;
  const int ntiles = (seg0 == 0 ? tiles_own : 0) + (seg1 - (seg0 == 0 ? 1 : seg0)) * tiles_bank;
  int ld_seg = seg0, ld_tt = 0;
  auto issue = [&](int st) {
    const uint32_t dst = lds0 + (uint32_t)st * STAGE;
    const int key0 = ld_tt * KT;
    const bool own = ld_seg == 0;
    const int nkeys = own ? p.n_kv : p.n_bank;
    const size_t img = own ? (size_t)b : (size_t)bank_b * p.nshot + (ld_seg - 1);
    const size_t kbase = img * (own ? p.k_bs : p.kb_bs) + head * 64;
    const size_t vbase = img * (own ? p.v_bs : p.vb_bs) + head * 64;
    const int ldk = own ? p.ldk : p.ldkb, ldv = own ? p.ldv : p.ldvb;
    u32x4 srk, srv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      srk[e] = __builtin_amdgcn_readfirstlane(own ? rk[e] : rkb[e]);
      srv[e] = __builtin_amdgcn_readfirstlane(own ? rv[e] : rvb[e]);
    }
    // lane-derived indices recomputed behind an opaque copy of the lane id: hoisted to kernel entry they live across the tile
    // loop, get spilled at this kernel's 256-register budget, and the reload's compiler-inserted s_waitcnt vmcnt(0) then drains
    // the hand-counted DMA ring in EVERY iteration (found in the first build's ISA: 475 us instead of the ring kernel's 353)
    int lane_i = lane;
    asm volatile("" : "+v"(lane_i));
    const int lrow = lane_i >> 3, slot = lane_i & 7;
#pragma unroll
    for (int j = 0; j < DPS / 2; ++j) {
      const int row = (j * NW + wave) * 8 + lrow;
      const int key = key0 + row;
      const bool ok = key < nkeys;
      const int ck = slot ^ ((row >> 1) & 7);
      const int cv = slot ^ (((row >> 1) & 1) << 2);
      const uint32_t ko = ok ? (uint32_t)((kbase + (size_t)key * ldk + ck * 8) * sizeof(T)) : kOOB;
      const uint32_t vo = ok ? (uint32_t)((vbase + (size_t)key * ldv + cv * 8) * sizeof(T)) : kOOB;
      dma16(srk, ko, dst + (uint32_t)(j * NW + wave) * 1024u);
      dma16(srv, vo, dst + TILE + (uint32_t)(j * NW + wave) * 1024u);
    }
    const int lim = own ? tiles_own : tiles_bank;
    if (++ld_tt == lim) { ld_tt = 0; ++ld_seg; }
  };

  uint32_t kq[4];
#pragma unroll
  for (int ss = 0; ss < 4; ++ss) kq[ss] = (uint32_t)(lr * 128 + ((lh ^ ((lr >> 1) & 7)) << 4)) ^ (uint32_t)(ss << 5);
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  uint32_t vq[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int dcol = d * 32 + 16 * tg + 4 * tp, row0 = 4 * lh + tq;
    vq[d] = (uint32_t)(row0 * 128 + (((dcol >> 3) ^ (((row0 >> 1) & 1) << 2)) << 4) + ((dcol & 7) << 1));
  }

  f32x16 o[2];
  float m_run = -1e30f, l_run = 0.f;
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;

  int issued = 0;
#pragma unroll
  for (int i = 0; i < S - 1; ++i)
    if (issued < ntiles) { issue(i); ++issued; }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): Q fragments (and the first stages) -- see fsa_ring_kernel
  int c_tt = 0, c_own = seg0 == 0 ? 1 : 0;
  int seg_nkv = p.n_kv, seg_nbank = p.n_bank;
  asm volatile("" : "+s"(seg_nkv), "+s"(seg_nbank));
  auto next_nvalid = [&]() __attribute__((always_inline)) {
    const int nv = (c_own ? seg_nkv : seg_nbank) - c_tt * KT;
    if (++c_tt == (c_own ? tiles_own : tiles_bank)) { c_tt = 0; c_own = 0; }
    return nv;
  };
  auto bar = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto stage_of = [&](int t) __attribute__((always_inline)) -> const char* { return smem + (t % S) * STAGE; };

  // S^T = K . Q^T into s (accumulators start at init)
  auto qk = [&](const char* kbuf, f32x16 (&s)[2], float init) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = init;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        typename Tr<T>::v8 kf = as_v8<T>(*(const i32x4*)(kbuf + kq[ss] + kb * 4096));
        s[kb] = Tr<T>::mfma(kf, qf[ss], s[kb]);
      }
  };
  auto mask_tail = [&](f32x16 (&s)[2], int nvalid) __attribute__((always_inline)) {
    if (nvalid < KT) {
      asm volatile("" ::: "memory");     // keep the ragged-tile mask a real (scalar) branch: see fsa_ring_kernel
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= nvalid) s[kb][r] = -INFINITY;
        }
    }
  };
  auto rowmax = [&](const f32x16 (&s)[2]) __attribute__((always_inline)) -> float {
    float mt = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[kb][r]);
    return half_swap_max(mt);
  };
  // P = exp2(s - d) (d = 0 on the common path), row sum into l_run, 16-bit fragments into pf
  auto exp_pack = [&](f32x16 (&s)[2], float d, typename Tr<T>::v8 (&pf)[4]) __attribute__((always_inline)) {
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[kb][r] - d);
        s[kb][r] = e;
        psum += e;
      }
    l_run += psum;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[kb * 2 + t2][j] = (T)s[kb][8 * t2 + j];
  };
  auto exp_pack0 = [&](f32x16 (&s)[2], typename Tr<T>::v8 (&pf)[4]) __attribute__((always_inline)) {
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[kb][r]);
        s[kb][r] = e;
        psum += e;
      }
    l_run += psum;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[kb * 2 + t2][j] = (T)s[kb][8 * t2 + j];
  };
  auto pv = [&](const char* vbuf, const typename Tr<T>::v8 (&pf)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const char* vd = vbuf + vq[d] + (kb * 32 + 16 * t2) * 128;
          typename Tr<T>::v4 lo = lds_tr_read<T>(vd);
          typename Tr<T>::v4 hi = lds_tr_read<T>(vd + 8 * 128);
          typename Tr<T>::v8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi[j]; }
          o[d] = Tr<T>::mfma(vf, pf[kb * 2 + t2], o[d]);
        }
  };
  auto wait_tile = [&](int t) __attribute__((always_inline)) {   // this wave's DMA pieces of tile t have landed
    const int younger = issued - t - 1;
    if (younger >= 3) wait_vm<3 * DPS>();
    else if (younger == 2) wait_vm<2 * DPS>();
    else if (younger == 1) wait_vm<DPS>();
    else wait_vm<0>();
  };

  f32x16 sA[2], sB[2];
  typename Tr<T>::v8 pA[4], pB[4];
  f32x16 minit;                      // -m_ref in every register: the C operand of the first MFMA of each S^T chain
  float mt_next = 0.f;               // row maximum of the pending score tile S(t+1), relative to m_ref
  auto set_minit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) minit[r] = -m_run;
  };
  // ---- prologue: P(0) from S(0) (sets the reference maximum); S(1) relative to it, masked, with its row maximum
  if (ntiles > 0) {
    wait_tile(0);
    bar();
    const int nv0 = next_nvalid();
    qk(stage_of(0), sA, 0.f);
    mask_tail(sA, nv0);
    m_run = rowmax(sA);
    exp_pack(sA, m_run, pA);
  }
  set_minit();
  if (ntiles > 1) {
    wait_tile(1);
    bar();
    const int nv1 = next_nvalid();
    qk(stage_of(1), sA, -m_run);
    mask_tail(sA, nv1);
    mt_next = rowmax(sA);
  }
  int st_v = 0, st_k = 2, st_i = S - 1;     // ring slots of tile t (V), tile t+2 (K) and of the next tile to issue
  auto adv = [&](int& x) __attribute__((always_inline)) { x = x + 1 == S ? 0 : x + 1; };
  // One pipelined iteration t (0 <= t < ntiles - 2): consumes P(t) [pc] and S(t+1) [sc, masked, row maximum in mt_next],
  // produces S(t+2) [sn] and P(t+1) [pn].  16 slots, each = the LDS reads of the NEXT slot's MFMA, one MFMA (slots 0-7:
  // S(t+2) = K(t+2).Q^T starting from minit; slots 8-15: O += V(t)^T.P(t)^T) and a slice of the vector work on S(t+1): two
  // exp2, two row-sum adds, one 16-bit pack -- plus, in slots 8-15, the row maximum of the finished S(t+2), four scores per
  // slot.  The scheduling barriers pin each slice beside its MFMA (~28 VALU issue cycles in a 32-cycle MFMA shadow).
  // Rare fix-ups around the slots: a moving reference maximum (deferred rescale) and a ragged tile t+2.
  auto body = [&](int t, f32x16 (&sc)[2], f32x16 (&sn)[2], typename Tr<T>::v8 (&pc)[4], typename Tr<T>::v8 (&pn)[4])
      __attribute__((always_inline)) {
    wait_tile(t + 2);
    bar();                                            // tile t+2 visible to every wave; the slot of tile t-1 is free
    if (issued < ntiles) { issue(st_i); adv(st_i); ++issued; }
    const int nv2 = next_nvalid();
    const char* kbuf = smem + st_k * STAGE;
    const char* vbuf = smem + st_v * STAGE + TILE;
    adv(st_k);
    adv(st_v);
    float alpha = 1.f, dmove = 0.f;
    const bool moved = __builtin_amdgcn_ballot_w64(mt_next > kDefer) != 0;
    if (moved) {
      // Deferred rescale, decided before S(t+1) is exponentiated: S(t+1) moves to the new reference now; O (which still has
      // to take P(t).V(t), a product at the OLD reference, in this iteration's slots), l and the S(t+2) started from the old
      // minit follow right after the slots -- P, O and l share one scale whenever they meet.
      dmove = fmaxf(mt_next, 0.f);
      alpha = __builtin_amdgcn_exp2f(-dmove);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[kb][r] -= dmove;
    }
    float ps0 = 0.f, ps1 = 0.f;
    float mx0, mx1;
    constexpr int AH = VAR == 2 ? 3 : 1;           // read-ahead distance in slots
    constexpr int NBUF = VAR == 2 ? 4 : 2;
    typename Tr<T>::v8 kf[NBUF];
    typename Tr<T>::v8 vf[NBUF];
    auto frag_read = [&](int j) __attribute__((always_inline)) {     // operands of slot j
      if (j < 8) {
        const int kb = j >> 2, ss = j & 3;
        kf[j % NBUF] = as_v8<T>(*(const i32x4*)(kbuf + kq[ss] + kb * 4096));
      } else if (j < 16) {
        const int jj = j - 8, kb = jj >> 2, t2 = (jj >> 1) & 1, dd = jj & 1;
        const char* vd = vbuf + vq[dd] + (kb * 32 + 16 * t2) * 128;
        typename Tr<T>::v4 lo = lds_tr_read<T>(vd);
        typename Tr<T>::v4 hi = lds_tr_read<T>(vd + 8 * 128);
#pragma unroll
        for (int e = 0; e < 4; ++e) { vf[j % NBUF][e] = lo[e]; vf[j % NBUF][4 + e] = hi[e]; }
      }
    };
#pragma unroll
    for (int j = 0; j < AH; ++j) frag_read(j);
    if constexpr (VAR != 1) __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      frag_read(i + AH);                             // operand reads AH slots ahead
      // -- the slot's MFMA
      if (i < 8) {
        const int kb = i >> 2, ss = i & 3;
        sn[kb] = Tr<T>::mfma(kf[i % NBUF], qf[ss], ss == 0 ? minit : sn[kb]);
      } else {
        const int j = i - 8, kb = j >> 2, t2 = (j >> 1) & 1, dd = j & 1;
        o[dd] = Tr<T>::mfma(vf[i % NBUF], pc[kb * 2 + t2], o[dd]);
      }
      // -- vector slice: scores 2i, 2i+1 of S(t+1) (block kb = i / 8, registers r0, r0 + 1)
      {
        const int kb = i >> 3, r0 = 2 * (i & 7);
        const float e0 = __builtin_amdgcn_exp2f(sc[kb][r0]);
        const float e1 = __builtin_amdgcn_exp2f(sc[kb][r0 + 1]);
        ps0 += e0;
        ps1 += e1;
        const int f = kb * 2 + (r0 >> 3), j = r0 & 7;
        pn[f][j] = (T)e0;
        pn[f][j + 1] = (T)e1;
      }
      if (i >= 8) {     // S(t+2) is complete (its last MFMA was slot 7): four of its 32 scores per slot into the row maximum
        const int j = i - 8, kb = j >> 2, r0 = 4 * (j & 3);
        const float a4 = fmaxf(fmaxf(sn[kb][r0], sn[kb][r0 + 1]), fmaxf(sn[kb][r0 + 2], sn[kb][r0 + 3]));
        if (j == 0) mx0 = a4;
        else if (j == 4) mx1 = a4;
        else if (j < 4) mx0 = fmaxf(mx0, a4);
        else mx1 = fmaxf(mx1, a4);
      }
      if constexpr (VAR != 1) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    l_run = l_run * alpha + (ps0 + ps1);
    if (moved) {
      m_run += dmove;
#pragma unroll
      for (int dd = 0; dd < 2; ++dd)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dd][r] *= alpha;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sn[kb][r] -= dmove;
      set_minit();
    }
    if (nv2 < KT) {        // ragged last tile of a key segment (out-of-range keys were zero-filled: finite scores)
      mask_tail(sn, nv2);
      mt_next = rowmax(sn);
    } else {
      mt_next = half_swap_max(fmaxf(mx0, mx1)) - dmove;
    }
  };
  // Tail: the last two tiles have no S(t+2) to start.  t = ntiles - 2: P(t).V(t), then P(t+1) from S(t+1) (with the same
  // rescale rule), then P(t+1).V(t+1).
  auto tail = [&](f32x16 (&sc)[2], typename Tr<T>::v8 (&pc)[4], typename Tr<T>::v8 (&pn)[4]) __attribute__((always_inline)) {
    if (ntiles == 0) return;
    pv(smem + st_v * STAGE + TILE, pc);
    adv(st_v);
    if (ntiles < 2) return;
    float d = 0.f;
    if (__builtin_amdgcn_ballot_w64(mt_next > kDefer) != 0) {
      d = fmaxf(mt_next, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-d);
      m_run += d;
      l_run *= alpha;
#pragma unroll
      for (int dd = 0; dd < 2; ++dd)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dd][r] *= alpha;
    }
    exp_pack(sc, d, pn);
    pv(smem + st_v * STAGE + TILE, pn);
  };
  {
    const int nmain = ntiles - 2;
    int t = 0;
    for (; t + 1 < nmain; t += 2) {
      body(t, sA, sB, pA, pB);
      body(t + 1, sB, sA, pB, pA);
    }
    if (t < nmain) {
      body(t, sA, sB, pA, pB);
      tail(sB, pB, pA);
    } else {
      tail(sA, pA, pB);
    }
  }

  {
    const float l_tot = half_swap_sum(l_run);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + lr;
    if (parted) {
      if (qrow < p.n_q) {
        float* pr = p.part + ((((size_t)bank_b * p.nsplit + split) * p.heads + head) * p.n_q + qrow) * 68;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int gg = 0; gg < 4; ++gg) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = o[d][4 * gg + e];
            *(f32x4*)(pr + d * 32 + 8 * gg + 4 * lh) = v;
          }
        if (lh == 0) {
          pr[64] = m_run;
          pr[65] = l_tot;
        }
      }
      return;
    }
    if (qrow < p.n_q) {
      if (p.lse && lh == 0)
        p.lse[((size_t)b * p.heads + head) * p.n_q + qrow] = m_run + __builtin_amdgcn_logf(l_tot);
      char* ob = p.out + ((size_t)b * p.o_bs + (size_t)qrow * p.ldo + head * 64) * sizeof(T);
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = o[d][4 * gg + e] * inv;
          *(i32x2*)(ob + (d * 32 + 8 * gg + 4 * lh) * sizeof(T)) = pack4<T>(v);
        }
    }
  }
}

