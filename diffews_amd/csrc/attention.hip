// KV-fusion flash attention for gfx950 (head_dim 64): softmax(q [k_own ; k_bank]^T * scale) [v_own ; v_bank].
//
// Keys come from two base pointers (the query pass' own K/V and the per-layer bank written by the
// support pass); the reference's torch.cat (attention_processor.py:258,267) is never materialised.
// Key order is the reference's: [own ; shot 0 ; shot 1 ; ...] with bank image = episode*nshot+shot.
//
// Workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 query rows.
// Per 64-key tile (K and V tiles double-buffered in LDS, register-staged issue-early/write-late):
//   S^T[key][q]  = K . Q^T      8 x v_mfma_f32_32x32x16  (K fragment from LDS, Q fragment in VGPRs)
//   online softmax: computing S transposed puts a whole score row (one q) in ONE lane (32 values,
//                   the other 32 keys of the row in lane^32), so the row max / row sum are
//                   in-register reductions + one v_permlane32_swap; exp2 with a folded scale.
//   O^T[d][q]   += V^T . P^T    8 x MFMA; the S^T accumulator tile is reused directly as the B
//                   operand (rows of S^T are the reduction index, no lane movement), V^T
//                   fragments come from the row-major V tile via ds_read_b64_tr_b16.
// fp32 accumulation and fp32 softmax statistics throughout.
#include "common.h"
#include "attention_common.h"
#include <type_traits>
#include <stdlib.h>
#include <stdio.h>

// Measured and removed: exponentiating half a tile inside the P.V MFMA gaps of the SAME tile (-17 %),
// an anti-phase two-barrier schedule of the two wave groups (-4.5 %), a first kernel with register-staged K/V tiles.

namespace dfw {

struct FsaP {
  const char* q; const char* k; const char* v; const char* kb; const char* vb; char* out;
  uint32_t q_bytes, k_bytes, v_bytes, kb_bytes, vb_bytes;
  int batch, heads, n_q, n_kv, n_bank, nshot, n_plain, xcd_remap;
  int ldq, ldk, ldv, ldkb, ldvb, ldo;
  long long q_bs, k_bs, v_bs, kb_bs, vb_bs, o_bs;
  float c;  // scale * log2(e)
  int pre;  // q already carries c (dfw_fsa_args.q_prescaled)
  float* lse;  // optional [batch][heads][n_q]: log2-sum-exp2 of the (scaled) scores, for the backward
  // Key split of the bank-reading images (lock-step launches with many shots: a query row walks (1 + nshot) x as many keys
  // as a support row, and its few workgroups would be the critical path): such an image appears nsplit times in the grid,
  // each instance walks a contiguous range of the key segments [own ; shot 0 ; ...] and leaves its un-normalised
  // accumulator, running maximum and sum in `part`; fsa_combine_kernel merges them.  nsplit = 1: nothing of this.
  int nsplit;
  float* part;   // [(batch - n_plain) * nsplit][heads][n_q][68]: o[64] (un-normalised), m (log2 units), l, 2 pad
};

// ------------------------------------------------------------------------------------------------
// v2: same mathematics and register layout as fsa_kernel, different data movement:
//   * NW = 8 (or 4) waves share every K/V tile: 256 (128) query rows per workgroup;
//   * K/V tiles arrive by LDS-DMA into a ring of S = 4 stages (K 8 KB + V 8 KB each), issued from
//     inline asm so three tiles stay in flight across the barriers (counted s_waitcnt vmcnt);
//   * the bank swizzles are applied on the DMA source address (LDS is written linearly);
//   * one barrier per key tile, no register staging, no ds_write.
// PRE: q arrives multiplied by scale * log2(e) (the QKV projection's epilogue did it in fp32 before its single
// rounding), so a score needs no multiply, and the running reference maximum rides in as the INITIAL VALUE of
// the S^T accumulators (S'' = q.k - m_ref comes straight out of the MFMA chain): on a tile that does not move
// the reference the softmax is max3 + exp2 + add + cvt per element pair -- no fma, no subtract.
template <typename T, int NW, int QB, bool PRE>
__global__ __launch_bounds__(NW * 64, (QB == 2 ? 2 : 4) * NW / 8 > 0 ? (QB == 2 ? 2 : 4) * NW / 8 : 1) void fsa_ring_kernel(const FsaP p) {
  // QB = 32-row query blocks per wave: with QB = 2 the two blocks are independent dependency chains
  // in one instruction stream, so one block's softmax VALU work overlaps the other's MFMAs, and
  // every K / V fragment read from LDS feeds two MFMAs.
  constexpr int KT = 64, S = 4;
  constexpr float kDefer = 8.0f;            // see the online softmax below
  constexpr int TILE = KT * 128;            // bytes of one K (or V) tile
  constexpr int STAGE = 2 * TILE;
  constexpr int DPS = 16 / NW;              // DMA wave-instructions per stage per wave (K + V)
  __shared__ __attribute__((aligned(16))) char smem[S * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  // Longest work first: images [n_plain, batch) also walk their episode's bank (the query images of a
  // lock-step [support ; query] launch) and sit at the END of the batch, so the z walk is reversed --
  // their workgroups are dispatched first and the short support-image ones fill in behind them.
  // XCD-aware placement: workgroups are dispatched round-robin over the 8 XCDs in launch order, so with
  // the plain (x, head, image) grid the 16 query blocks that stream the SAME K/V of one (image, head) land
  // on 8 different L2s and every L2 fetches that K/V from beyond it.  When (heads * images) % 8 == 0 the
  // launch-order index is re-read as (XCD c, k-th workgroup on it) -> pair c + 8*(k / X), query block
  // k % X: all query blocks of a pair share one XCD's L2, pairs still go out longest first.
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
  if (p.nsplit > 1 && b >= p.n_plain) {       // grid.z lists every bank-reading image nsplit times (after the plain ones)
    const int v = b - p.n_plain;
    split = v % p.nsplit;
    b = p.n_plain + v / p.nsplit;
  }
  const int bank_b = b - p.n_plain;          // episode index into the bank (< 0: own keys only)
  const int q0 = qblk * (NW * 32 * QB) + wave * (32 * QB);
  const uint32_t lds0 = lds_addr(smem);

  const __amdgpu_buffer_rsrc_t rq = make_rsrc(p.q, p.q_bytes);
  const u32x4 rk = make_srd(p.k, p.k_bytes), rv = make_srd(p.v, p.v_bytes);
  const u32x4 rkb = make_srd(p.kb ? p.kb : p.k, p.kb ? p.kb_bytes : 0u);
  const u32x4 rvb = make_srd(p.vb ? p.vb : p.v, p.vb ? p.vb_bytes : 0u);

  typename Tr<T>::v8 qf[QB][4];
#pragma unroll
  for (int g = 0; g < QB; ++g) {
    const int qrow = q0 + g * 32 + lr;
    const uint32_t base = qrow < p.n_q
        ? (uint32_t)(((size_t)b * p.q_bs + (size_t)qrow * p.ldq + head * 64 + lh * 8) * sizeof(T)) : kOOB;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[g][s] = as_v8<T>(buf_load16(rq, base + (uint32_t)(s * 32)));
  }

  // ---- loader: wave-instruction j of a wave covers tile rows (j*NW + wave)*8 .. +8 of K, then of V;
  // lane -> row + (lane>>3), LDS slot lane&7; source chunk = slot ^ swizzle(row)
  const int lrow = lane >> 3, slot = lane & 7;
  const int tiles_own = (p.n_kv + KT - 1) / KT;
  const int tiles_bank = (p.nshot > 0 && bank_b >= 0) ? (p.n_bank + KT - 1) / KT : 0;
  // this instance's key segments [seg0, seg1) of [own ; shot 0 ; ...]
  const int nseg = 1 + (tiles_bank ? p.nshot : 0);
  const bool parted = p.nsplit > 1 && bank_b >= 0;
  const int seg0 = parted ? split * nseg / p.nsplit : 0, seg1 = parted ? (split + 1) * nseg / p.nsplit : nseg;
  const int ntiles = (seg0 == 0 ? tiles_own : 0) + (seg1 - (seg0 == 0 ? 1 : seg0)) * tiles_bank;
  int ld_seg = seg0, ld_tt = 0;   // segment / tile-in-segment of the next tile to load
  auto issue = [&](int st) {
    const uint32_t dst = lds0 + (uint32_t)st * STAGE;
    const int key0 = ld_tt * KT;
    const bool own = ld_seg == 0;
    const int nseg = own ? p.n_kv : p.n_bank;
    const size_t img = own ? (size_t)b : (size_t)bank_b * p.nshot + (ld_seg - 1);
    const size_t kbase = img * (own ? p.k_bs : p.kb_bs) + head * 64;
    const size_t vbase = img * (own ? p.v_bs : p.vb_bs) + head * 64;
    const int ldk = own ? p.ldk : p.ldkb, ldv = own ? p.ldv : p.ldvb;
    u32x4 srk, srv;   // descriptor of this tile's source, forced back into SGPRs after the select
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      srk[e] = __builtin_amdgcn_readfirstlane(own ? rk[e] : rkb[e]);
      srv[e] = __builtin_amdgcn_readfirstlane(own ? rv[e] : rvb[e]);
    }
#pragma unroll
    for (int j = 0; j < DPS / 2; ++j) {
      const int row = (j * NW + wave) * 8 + lrow;
      const int key = key0 + row;
      const bool ok = key < nseg;
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

  // LDS fragment addresses as (lane-constant base) + (compile-time immediate): the swizzles depend on
  // the row only through bits that the +32 (K block), +16 / +8 (V^T row groups) steps leave unchanged,
  // so 4 K bases and 2 V bases replace 8 + 16 per-fragment address registers (which spilled at the
  // 128-VGPR budget of 4 waves per SIMD, and cost a v_or each per tile).
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

  f32x16 o[QB][2];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int g = 0; g < QB; ++g) {
    m_run[g] = -1e30f;
    l_run[g] = 0.f;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[g][d][r] = 0.f;
  }

  int issued = 0;
#pragma unroll
  for (int i = 0; i < S - 1; ++i)
    if (issued < ntiles) { issue(i); ++issued; }
  // The Q fragments were fetched with compiler-visible buffer loads; without an explicit wait the
  // compiler cannot prove them complete across the loop back-edge and plants vmcnt(5..2) in front of
  // every tile's QK^T MFMAs -- which also drains the hand-counted K/V DMA ring early.  One full wait
  // here (the first tiles are needed by the first iteration anyway) removes them from the loop.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
  int c_tt = 0, c_own = seg0 == 0 ? 1 : 0;  // compute-side tile-in-segment / own-segment flag
  typename Tr<T>::v8 pf[QB][4];   // P^T fragments of the tile between its softmax and its P.V
  auto qk_softmax = [&](const char* kbuf, int nvalid, bool first) __attribute__((always_inline)) {
    // ---- S^T = K . Q^T  (each K fragment feeds QB MFMAs)
    f32x16 s[QB][2];
#pragma unroll
    for (int g = 0; g < QB; ++g) {
      // PRE: accumulators start at -m_ref (0 on the first tile, whose maximum becomes the reference)
      const float init = (PRE && !first) ? -m_run[g] : 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[g][kb][r] = init;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        typename Tr<T>::v8 kf = as_v8<T>(*(const i32x4*)(kbuf + kq[ss] + kb * 4096));
#pragma unroll
        for (int g = 0; g < QB; ++g) s[g][kb] = Tr<T>::mfma(kf, qf[g][ss], s[g][kb]);
      }
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int g = 0; g < QB; ++g) {
      if (nvalid < KT) {
        // ragged last tile of a key segment only.  The empty asm keeps this a REAL (scalar) branch: hipcc
        // otherwise if-converts the body into 32 v_cmp + 32 v_cndmask + the key-index arithmetic executed on
        // EVERY tile -- ~100 VALU instructions, as many issue cycles as the whole softmax.
        asm volatile("" ::: "memory");
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (key >= nvalid) s[g][kb][r] = -INFINITY;
          }
      }
      // ---- online softmax (row = this lane's q; its other 32 keys live in lane^32)
      float mt = s[g][0][0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[g][kb][r]);
      mt = half_swap_max(mt);
      float psum = 0.f;
      if constexpr (PRE) {
        // mt = (row maximum of this tile) - m_ref, in log2 units.  Deferred rescale as below: the reference
        // moves only when some row's maximum has grown past it by more than kDefer; the first tile sets it.
        // Decided before this tile's P exists and after the previous tile's P.V: O, l and P share one scale.
        if (first || __builtin_amdgcn_ballot_w64(mt > kDefer) != 0) {
          const float d = first ? mt : fmaxf(mt, 0.f);       // every lane moves to its own true maximum
          m_run[g] = first ? mt : m_run[g] + d;
          if (!first) {
            const float alpha = __builtin_amdgcn_exp2f(-d);
            l_run[g] *= alpha;
#pragma unroll
            for (int dd = 0; dd < 2; ++dd)
#pragma unroll
              for (int r = 0; r < 16; ++r) o[g][dd][r] *= alpha;
          }
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float e = __builtin_amdgcn_exp2f(s[g][kb][r] - d);
              s[g][kb][r] = e;
              psum += e;
            }
        } else {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float e = __builtin_amdgcn_exp2f(s[g][kb][r]);
              s[g][kb][r] = e;
              psum += e;
            }
        }
      } else {
      // Deferred rescale: O and l keep their reference maximum m_run until some row's maximum has
      // grown by more than kDefer (log2 units), so the O-wide multiply runs on a few tiles instead of
      // every tile; until then this tile's P is simply bounded by 2^kDefer instead of 1 (exact in
      // fp32 sums, same relative rounding in the 16-bit P).  The decision is taken BEFORE this tile's
      // P is exponentiated and after the previous tile's P.V completed, so O, l and P always share
      // one scale.  Wave-uniform branch (ballot): every lane then moves to its own true maximum.
      if (__builtin_amdgcn_ballot_w64((mt - m_run[g]) * p.c > kDefer) != 0) {
        const float m_new = fmaxf(m_run[g], mt);
        const float alpha = __builtin_amdgcn_exp2f((m_run[g] - m_new) * p.c);
        m_run[g] = m_new;
        l_run[g] *= alpha;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[g][d][r] *= alpha;
      }
      const float mc = m_run[g] * p.c;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(s[g][kb][r] * p.c - mc);
          s[g][kb][r] = e;
          psum += e;
        }
      }
      l_run[g] += psum;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[g][kb * 2 + t2][j] = (T)s[g][kb][8 * t2 + j];
      }
    }

  };
  auto pv = [&](const char* vbuf) __attribute__((always_inline)) {
    // ---- O^T += V^T . P^T  (each V^T fragment feeds QB MFMAs)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          // rows (kb*32 + 16*t2 + 4*lh + tq) and +8 of the V tile
          const char* vd = vbuf + vq[d] + (kb * 32 + 16 * t2) * 128;
          typename Tr<T>::v4 lo = lds_tr_read<T>(vd);
          typename Tr<T>::v4 hi = lds_tr_read<T>(vd + 8 * 128);
          typename Tr<T>::v8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi[j]; }
#pragma unroll
          for (int g = 0; g < QB; ++g) o[g][d] = Tr<T>::mfma(vf, pf[g][kb * 2 + t2], o[g][d]);
        }
      }
    __builtin_amdgcn_s_setprio(0);
  };
  auto wait_tile = [&](int t) __attribute__((always_inline)) {
    const int younger = issued - t - 1;
    if (younger >= 2) wait_vm<2 * DPS>();
    else if (younger == 1) wait_vm<DPS>();
    else wait_vm<0>();
  };
  auto bar = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // segment lengths pinned in SGPRs: left to the compiler, `c_own ? p.n_kv : p.n_bank` became a scalar LOAD from
  // a selected kernarg address plus s_waitcnt lgkmcnt(0) at the head of every tile
  int seg_nkv = p.n_kv, seg_nbank = p.n_bank;
  asm volatile("" : "+s"(seg_nkv), "+s"(seg_nbank));
  auto next_nvalid = [&]() __attribute__((always_inline)) {
    const int nv = (c_own ? seg_nkv : seg_nbank) - c_tt * KT;
    if (++c_tt == (c_own ? tiles_own : tiles_bank)) { c_tt = 0; c_own = 0; }
    return nv;
  };
  {
    for (int t = 0; t < ntiles; ++t) {
      wait_tile(t);
      bar();
      if (issued < ntiles) { issue(issued & (S - 1)); ++issued; }
      const char* kbuf = smem + (t & (S - 1)) * STAGE;
      const int nvalid = next_nvalid();
      qk_softmax(kbuf, nvalid, t == 0);
      pv(kbuf + TILE);
    }
  }

#pragma unroll
  for (int g = 0; g < QB; ++g) {
  const float l_tot = half_swap_sum(l_run[g]);
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + g * 32 + lr;
  if (parted) {
    if (qrow < p.n_q) {
      float* pr = p.part + ((((size_t)bank_b * p.nsplit + split) * p.heads + head) * p.n_q + qrow) * 68;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = o[g][d][4 * gg + e];
          *(f32x4*)(pr + d * 32 + 8 * gg + 4 * lh) = v;
        }
      if (lh == 0) {
        pr[64] = PRE ? m_run[g] : m_run[g] * p.c;
        pr[65] = l_tot;
      }
    }
    continue;
  }
  if (qrow < p.n_q) {
    if (p.lse && lh == 0)   // exp2(s' - lse) is the row's probability (s' = scaled score in log2 units)
      p.lse[((size_t)b * p.heads + head) * p.n_q + qrow] = (PRE ? m_run[g] : m_run[g] * p.c) + __builtin_amdgcn_logf(l_tot);
    char* ob = p.out + ((size_t)b * p.o_bs + (size_t)qrow * p.ldo + head * 64) * sizeof(T);
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o[g][d][4 * gg + e] * inv;
        *(i32x2*)(ob + (d * 32 + 8 * gg + 4 * lh) * sizeof(T)) = pack4<T>(v);
      }
  }
  }
}

// Merge of the key-split partial results: out = sum_s 2^(m_s - M) o_s / sum_s 2^(m_s - M) l_s, lse = M + log2(that sum).
// One thread per (row, 4 output columns); splits visited in order (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void fsa_combine_kernel(const FsaP p) {
  const int nq_img = p.batch - p.n_plain;
  const long long rows = (long long)nq_img * p.heads * p.n_q;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long row = e >> 4;
  const int c4 = (int)(e & 15);
  if (row >= rows) return;
  const int qrow = (int)(row % p.n_q);
  const long long t = row / p.n_q;
  const int head = (int)(t % p.heads), bq = (int)(t / p.heads);
  const float* base = p.part + ((((size_t)bq * p.nsplit) * p.heads + head) * p.n_q + qrow) * 68;
  const size_t sstride = (size_t)p.heads * p.n_q * 68;
  float M = -INFINITY;
  for (int s2 = 0; s2 < p.nsplit; ++s2) M = fmaxf(M, base[s2 * sstride + 64]);
  float L = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s2 = 0; s2 < p.nsplit; ++s2) {
    const float* ps = base + s2 * sstride;
    const float w = __builtin_amdgcn_exp2f(ps[64] - M);
    L += w * ps[65];
    const f32x4 v = *(const f32x4*)(ps + c4 * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += w * v[i];
  }
  const float inv = 1.0f / L;
  const int b = p.n_plain + bq;
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = acc[i] * inv;
  *(i32x2*)(p.out + ((size_t)b * p.o_bs + (size_t)qrow * p.ldo + head * 64 + c4 * 4) * sizeof(T)) = pack4<T>(v);
  if (p.lse && c4 == 0) p.lse[((size_t)b * p.heads + head) * p.n_q + qrow] = M + __builtin_amdgcn_logf(L);
}

// ------------------------------------------------------------------------------------------------
// Cross-attention over a short context (L keys): one thread per query row, K/V of the (batch, head)
// staged in LDS as fp32, online softmax in registers.
struct XaP {
  const char* q; const char* k; const char* v; char* out;
  int batch, heads, n_q, L, ldq, ldk, ldv, ldo;
  long long q_bs, k_bs, v_bs, o_bs;
  float c;
};

template <typename T>
__global__ __launch_bounds__(256) void xattn_kernel(const XaP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_x[];
  float* Ks = (float*)smem_x;          // [L][64]
  float* Vs = Ks + (size_t)p.L * 64;   // [L][64]
  const int head = blockIdx.y, b = blockIdx.z;
  for (int e = threadIdx.x; e < p.L * 8; e += 256) {
    const int j = e >> 3, c8 = e & 7;
    float f[8];
    unpack8<T>(*(const i32x4*)(p.k + ((size_t)b * p.k_bs + (size_t)j * p.ldk + head * 64 + c8 * 8) * sizeof(T)), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) Ks[j * 64 + c8 * 8 + i] = f[i];
    unpack8<T>(*(const i32x4*)(p.v + ((size_t)b * p.v_bs + (size_t)j * p.ldv + head * 64 + c8 * 8) * sizeof(T)), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) Vs[j * 64 + c8 * 8 + i] = f[i];
  }
  __syncthreads();
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= p.n_q) return;
  float qv[64];
  const char* qp = p.q + ((size_t)b * p.q_bs + (size_t)row * p.ldq + head * 64) * sizeof(T);
#pragma unroll
  for (int c8 = 0; c8 < 8; ++c8) unpack8<T>(*(const i32x4*)(qp + c8 * 16), qv + c8 * 8);
  float o[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) o[i] = 0.f;
  float m = -1e30f, l = 0.f;
  for (int j = 0; j < p.L; ++j) {
    float sdot = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) sdot += qv[i] * Ks[j * 64 + i];
    const float mn = fmaxf(m, sdot);
    const float al = __builtin_amdgcn_exp2f((m - mn) * p.c), e = __builtin_amdgcn_exp2f((sdot - mn) * p.c);
    m = mn;
    l = l * al + e;
#pragma unroll
    for (int i = 0; i < 64; ++i) o[i] = o[i] * al + e * Vs[j * 64 + i];
  }
  const float inv = 1.0f / l;
  char* op = p.out + ((size_t)b * p.o_bs + (size_t)row * p.ldo + head * 64) * sizeof(T);
#pragma unroll
  for (int c8 = 0; c8 < 8; ++c8) {
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = o[c8 * 8 + i] * inv;
    *(i32x4*)(op + c8 * 16) = pack8<T>(f);
  }
}

}  // namespace dfw

using namespace dfw;

static int64_t extent(int batch, int64_t bs, int n, int ld, int heads) {
  return (int64_t)(batch - 1) * bs + (int64_t)(n - 1) * ld + (int64_t)heads * 64;
}

// Key split of the bank-reading images (see FsaP::nsplit): the smallest split count whose longest workgroup is no longer
// the launch's critical path.  Work in key tiles per workgroup column: plain images 1 segment, bank readers 1 + nshot.
static int fsa_split_count(const dfw_fsa_args* a) {
  if (!cfg().fsa_key_split || a->nshot < 2 || a->n_q > 65536) return 1;
  const int nq_img = a->batch - a->n_plain, nseg = 1 + a->nshot;
  if (nq_img <= 0) return 1;
  if ((long long)a->n_kv + (long long)a->nshot * a->n_bank < 8192) return 1;    // short rows: nothing worth a second kernel
  const int rows_per_wg = a->n_q <= 1024 ? 128 : 256;
  const long long wg_per_img = (long long)a->heads * ((a->n_q + rows_per_wg - 1) / rows_per_wg);
  const long long slots = a->n_q <= 1024 ? 1024 : 512;            // resident workgroups (4 x 256-thread / 2 x 512-thread per CU)
  const double total = (double)wg_per_img * ((double)a->n_plain + (double)nq_img * nseg);   // in units of one segment's tiles
  const double fair = total / slots > 1.0 ? total / slots : 1.0;
  if (cfg().fsa_force_splits) { const int f = cfg().fsa_force_splits; return f > nseg ? nseg : f; }   // sweeps
  if ((double)nseg <= 1.5 * fair) return 1;
  for (int ns = 2; ns <= nseg; ++ns)
    if ((double)((nseg + ns - 1) / ns) <= 1.25 * fair) return ns;
  return nseg;
}

extern "C" size_t dfw_fsa_workspace_bytes(const dfw_fsa_args* a) {
  if (!a || a->batch <= 0 || a->heads <= 0 || a->n_q <= 0) return 0;
  const int ns = fsa_split_count(a);
  if (ns <= 1) return 0;
  return (size_t)(a->batch - a->n_plain) * ns * a->heads * a->n_q * 68 * sizeof(float);
}

extern "C" int dfw_fsa_attention(const dfw_fsa_args* a, dfw_stream_t stream) {
  if (!a || !a->q || !a->k || !a->v || !a->out) return DFW_EINVAL;
  if (a->batch <= 0 || a->heads <= 0 || a->n_q <= 0 || a->n_kv <= 0 || a->nshot < 0) return DFW_EINVAL;
  if (a->nshot > 0 && (!a->k_bank || !a->v_bank || a->n_bank <= 0)) return DFW_EINVAL;
  if (a->n_plain < 0 || a->n_plain > a->batch || (a->n_plain > 0 && a->nshot == 0)) return DFW_EINVAL;
  if (a->dtype != DFW_BF16 && a->dtype != DFW_F16) return DFW_EINVAL;
  if ((a->ldq | a->ldk | a->ldv | a->ldo) % 8 != 0) return DFW_ESHAPE;
  if (a->nshot > 0 && (a->ldkb | a->ldvb) % 8 != 0) return DFW_ESHAPE;
  if ((a->q_bs | a->k_bs | a->v_bs | a->o_bs) % 8 != 0) return DFW_ESHAPE;
  FsaP p;
  p.q = (const char*)a->q; p.k = (const char*)a->k; p.v = (const char*)a->v;
  p.kb = (const char*)a->k_bank; p.vb = (const char*)a->v_bank; p.out = (char*)a->out;
  const int64_t qe = extent(a->batch, a->q_bs, a->n_q, a->ldq, a->heads);
  const int64_t ke = extent(a->batch, a->k_bs, a->n_kv, a->ldk, a->heads);
  const int64_t ve = extent(a->batch, a->v_bs, a->n_kv, a->ldv, a->heads);
  int64_t kbe = 0, vbe = 0;
  if (a->nshot > 0) {
    const int nb = (a->batch - a->n_plain) * a->nshot;   // bank images
    if (nb > 0) {
      kbe = extent(nb, a->kb_bs, a->n_bank, a->ldkb, a->heads);
      vbe = extent(nb, a->vb_bs, a->n_bank, a->ldvb, a->heads);
    }
  }
  const int64_t lim = (1ll << 30);  // elements (2 bytes each)
  if (qe >= lim || ke >= lim || ve >= lim || kbe >= lim || vbe >= lim) return DFW_ERANGE;
  p.q_bytes = (uint32_t)(qe * 2); p.k_bytes = (uint32_t)(ke * 2); p.v_bytes = (uint32_t)(ve * 2);
  p.kb_bytes = (uint32_t)(kbe * 2); p.vb_bytes = (uint32_t)(vbe * 2);
  p.batch = a->batch; p.heads = a->heads; p.n_q = a->n_q; p.n_kv = a->n_kv;
  p.n_bank = a->n_bank; p.nshot = a->nshot; p.n_plain = a->n_plain;
  p.xcd_remap = 1;
  p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldkb = a->ldkb; p.ldvb = a->ldvb; p.ldo = a->ldo;
  p.q_bs = a->q_bs; p.k_bs = a->k_bs; p.v_bs = a->v_bs; p.kb_bs = a->kb_bs; p.vb_bs = a->vb_bs; p.o_bs = a->o_bs;
  p.c = a->scale * 1.4426950408889634f;
  p.pre = a->q_prescaled ? 1 : 0;
  p.lse = a->lse;
  // key split: only with a caller-provided workspace of dfw_fsa_workspace_bytes(); without one the launch is unsplit
  p.nsplit = 1; p.part = nullptr;
  {
    const int ns = fsa_split_count(a);
    if (ns > 1 && a->workspace && a->workspace_bytes >= dfw_fsa_workspace_bytes(a) && (((uintptr_t)a->workspace) & 15) == 0) {
      p.nsplit = ns;
      p.part = (float*)a->workspace;
    }
  }
  const int grid_z = a->n_plain + (a->batch - a->n_plain) * p.nsplit;
  hipStream_t st = (hipStream_t)stream;
  const bool bf = a->dtype == DFW_BF16;
  {
    // 8 waves x 32 query rows per workgroup; short rows (n_q <= 1024): 128-query workgroups balance the grid better
    const int nw = a->n_q <= 1024 ? 4 : 8;
    dim3 grid((a->n_q + nw * 32 - 1) / (nw * 32), a->heads, grid_z);
    const bool pre = a->q_prescaled != 0;
    if (nw == 8) {
      if (bf) { if (pre) hipLaunchKernelGGL((fsa_ring_kernel<__bf16, 8, 1, true>), grid, dim3(512), 0, st, p);
                else hipLaunchKernelGGL((fsa_ring_kernel<__bf16, 8, 1, false>), grid, dim3(512), 0, st, p); }
      else { if (pre) hipLaunchKernelGGL((fsa_ring_kernel<_Float16, 8, 1, true>), grid, dim3(512), 0, st, p);
             else hipLaunchKernelGGL((fsa_ring_kernel<_Float16, 8, 1, false>), grid, dim3(512), 0, st, p); }
    } else {
      if (bf) { if (pre) hipLaunchKernelGGL((fsa_ring_kernel<__bf16, 4, 1, true>), grid, dim3(256), 0, st, p);
                else hipLaunchKernelGGL((fsa_ring_kernel<__bf16, 4, 1, false>), grid, dim3(256), 0, st, p); }
      else { if (pre) hipLaunchKernelGGL((fsa_ring_kernel<_Float16, 4, 1, true>), grid, dim3(256), 0, st, p);
             else hipLaunchKernelGGL((fsa_ring_kernel<_Float16, 4, 1, false>), grid, dim3(256), 0, st, p); }
    }
  }
  DFW_CHECK_LAUNCH();
  if (p.nsplit > 1) {
    const long long threads = (long long)(a->batch - a->n_plain) * a->heads * a->n_q * 16;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    if (bf) hipLaunchKernelGGL((fsa_combine_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((fsa_combine_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, p);
    DFW_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int dfw_cross_attention(const dfw_xattn_args* a, dfw_stream_t stream) {
  if (!a || !a->q || !a->k || !a->v || !a->out) return DFW_EINVAL;
  if (a->batch <= 0 || a->heads <= 0 || a->n_q <= 0 || a->L <= 0) return DFW_EINVAL;
  if (a->dtype != DFW_BF16 && a->dtype != DFW_F16) return DFW_EINVAL;
  if ((a->ldq | a->ldk | a->ldv | a->ldo) % 8 != 0) return DFW_ESHAPE;
  if ((a->q_bs | a->k_bs | a->v_bs | a->o_bs) % 8 != 0) return DFW_ESHAPE;
  const size_t lds = (size_t)a->L * 64 * 2 * sizeof(float);
  if (lds > 64 * 1024) return DFW_ESHAPE;  // L <= 128
  XaP p;
  p.q = (const char*)a->q; p.k = (const char*)a->k; p.v = (const char*)a->v; p.out = (char*)a->out;
  p.batch = a->batch; p.heads = a->heads; p.n_q = a->n_q; p.L = a->L;
  p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
  p.q_bs = a->q_bs; p.k_bs = a->k_bs; p.v_bs = a->v_bs; p.o_bs = a->o_bs;
  p.c = a->scale * 1.4426950408889634f;
  dim3 grid((a->n_q + 255) / 256, a->heads, a->batch);
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == DFW_BF16) hipLaunchKernelGGL((xattn_kernel<__bf16>), grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL((xattn_kernel<_Float16>), grid, dim3(256), lds, st, p);
  DFW_CHECK_LAUNCH();
  return 0;
}
