// Flash attention for the VAE mid-block (diffusers AutoencoderKL `Attention`: ONE head of dim 512 over N = h*w tokens,
// P:852-853 / P:901-902; the reference runs it through xformers memory_efficient_attention, E:374-376, and never
// materialises the N x N scores).  Round 4: replaces  bmm_nt (fp32 [B, N, N] scores) -> softmax_rows -> transpose ->
// bmm_nt  of rounds 1-3 (805 MB of scores for 12 images at 512^2).
//
// head_dim 512 turns the usual structure around: the output accumulator of 32 queries is 512 x 32 fp32 = 256 registers
// per lane, and the query fragments another 128, so a wave owns 32 queries and the WHOLE 512-register file of its SIMD:
// workgroup = 4 waves = one per SIMD = 128 queries (the 4-wave, 512-register structure of the guide's fast attention
// kernel; with 256-register waves every wave would have to re-read the whole K and V tile for 16 queries and the LDS
// port, not the matrix pipe, would bound the kernel).  Per 32-key tile and wave:
//   S^T[32 keys][32 q]  = K . Q^T        32 x v_mfma_f32_32x32x16 (one accumulator chain; K fragment = one ds_read_b128,
//                                         Q fragments resident in 128 VGPRs)
//   online softmax on 16 scores per lane (a score row = one lane + lane^32), deferred rescale (threshold 2^8),
//   q arrives pre-scaled by scale * log2(e) (the fused QKV projection's colscale), the running maximum rides in as the
//   accumulators' initial value
//   O^T[512 d][32 q]   += V^T . P^T      32 MFMA: 16 d-blocks x 2 k-steps, V^T fragments by ds_read_b64_tr_b16, P^T
//                                         straight from the S^T accumulators (keys on the registers)
// 64 MFMAs (2048 matrix cycles) against 16 exponentials per lane: the softmax is ~10 % of a tile, unlike head_dim 64.
// K / V tiles (32 keys x 1 KiB each) arrive by LDS-DMA, one whole key row per wave-instruction, into two 64 KiB stages;
// one barrier per tile.  K rows are swizzled (16-byte chunk ^ (key & 15)) for the b128 fragment reads, V rows
// (chunk ^ ((key & 3) << 2)) for the transposed reads; both on the DMA's source address.
// The output is normalised, staged through the (then idle) LDS stages and stored as whole 1 KiB rows.
#include "common.h"
#include "attention_common.h"
#include <type_traits>

namespace dfw {

struct VattnP {
  const char* q; const char* k; const char* v; char* out;
  uint32_t q_bytes, k_bytes, v_bytes;
  int batch, n, ldq, ldk, ldv, ldo;
  long long q_bs, k_bs, v_bs, o_bs;
};

// O^T += V^T . P^T with the accumulator PINNED to the accumulation registers ("+a"): 16 such blocks are the wave's 256 AGPRs,
// everything else (Q fragments, scores, addresses) lives in its 256 architectural VGPRs.  Left to the register allocator the
// builtin form spilled the Q fragments at their loads and moved O through scratch (904 bytes per lane).  The leading s_nop
// covers the VALU-write (v_cvt_pk of P) -> MFMA-operand hazard, which hipcc does not pad inside an asm statement; two
// consecutive MFMAs on one accumulator are an accumulate chain and need no wait states.
template <typename T> __device__ __forceinline__ void mfma_acc(f32x16& acc, typename Tr<T>::v8 a, typename Tr<T>::v8 b);
template <> __device__ __forceinline__ void mfma_acc<__bf16>(f32x16& acc, bf16x8 a, bf16x8 b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
template <> __device__ __forceinline__ void mfma_acc<_Float16>(f32x16& acc, f16x8 a, f16x8 b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// S^T += K . Q^T with the accumulator pinned to ARCHITECTURAL registers ("+v"): the builtin form put the score tile into
// a[0:15] -- all 256 AGPRs belong to O -- and shuttled the sixteenth O block through VGPRs around every P.V MFMA, reading it
// back before the (asm, hence invisible to the hazard recogniser) MFMA had written it.  mfma_done(): the wait states between
// the chain's last MFMA and the first vector instruction that reads the scores (8-pass XDL: 12; hipcc pads nothing around asm).
template <typename T> __device__ __forceinline__ void mfma_vreg(f32x16& acc, typename Tr<T>::v8 a, typename Tr<T>::v8 b);
template <> __device__ __forceinline__ void mfma_vreg<__bf16>(f32x16& acc, bf16x8 a, bf16x8 b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <> __device__ __forceinline__ void mfma_vreg<_Float16>(f32x16& acc, f16x8 a, f16x8 b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_done(f32x16& acc) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc)); }

template <typename T>
__global__ __launch_bounds__(256, 1) void vattn_kernel(const VattnP p) {
  constexpr int KT = 32, D = 512;
  constexpr int TILE = KT * D * 2;          // 32 KiB: one K (or V) tile
  constexpr int STAGE = 2 * TILE;
  constexpr float kDefer = 8.0f;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  // XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs, and the nqb query blocks of one image stream the SAME
  // 8 MB of K / V -- image (xcd + 8 j) runs entirely on XCD `xcd` (32 CUs = the 32 query blocks of a 64 x 64 latent), so its
  // L2 serves 31 of 32 tile fetches; with the plain (q-block, image) grid every XCD fetched every image from beyond its L2
  // and the kernel ran at the Infinity-Cache gather rate, 4 us per tile instead of 1.
  const int nqb = (p.n + 127) >> 7;
  const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
  const int b = xcd + 8 * (kx / nqb);
  if (b >= p.batch) return;
  const int q0 = (kx % nqb) * 128 + wave * 32;
  const uint32_t lds0 = lds_addr(smem);

  const __amdgpu_buffer_rsrc_t rq = make_rsrc(p.q, p.q_bytes);
  const u32x4 rk = make_srd(p.k, p.k_bytes), rv = make_srd(p.v, p.v_bytes);

  // ---- Q fragments: B operand of 32x32x16, lane (q = lr, k = 16 s + 8 lh + j)
  typename Tr<T>::v8 qf[32];
  {
    const int qrow = q0 + lr;
    const uint32_t base = qrow < p.n ? (uint32_t)(((size_t)b * p.q_bs + (size_t)qrow * p.ldq + lh * 8) * sizeof(T)) : kOOB;
#pragma unroll
    for (int s = 0; s < 32; ++s) qf[s] = as_v8<T>(buf_load16(rq, base + (uint32_t)(s * 32)));
  }

  // ---- loader: wave-instruction i (0..7) of this wave writes key row (i * 4 + wave) of the K tile, then of the V tile;
  // lane = 16-byte slot of the row, source chunk = slot ^ swizzle(row)
  const int ntiles = (p.n + KT - 1) / KT;
  // One wave per SIMD: an LDS-DMA instruction the wave is stuck issuing is time its matrix pipe idles (a first build issued the
  // next tile's sixteen pieces in one burst behind the barrier: ~3 000 of a tile's 8 000 cycles).  The pieces are therefore
  // issued ONE AT A TIME between MFMAs: K row i of the next tile after the (4 i + 1)-th MFMA of S^T = K Q^T, V row i after the
  // (4 i + 1)-th MFMA of O^T += V^T P^T.
  auto issue_k = [&](int t, int i) __attribute__((always_inline)) {
    const int row = i * 4 + wave, key = t * KT + row;
    const uint32_t ko = key < p.n ? (uint32_t)(((size_t)b * p.k_bs + (size_t)key * p.ldk + (lane ^ (row & 15)) * 8) * sizeof(T)) : kOOB;
    dma16(rk, ko, lds0 + (uint32_t)(t & 1) * STAGE + (uint32_t)row * 1024u);
  };
  auto issue_v = [&](int t, int i) __attribute__((always_inline)) {
    const int row = i * 4 + wave, key = t * KT + row;
    const uint32_t vo = key < p.n ? (uint32_t)(((size_t)b * p.v_bs + (size_t)key * p.ldv + (lane ^ ((row & 3) << 2)) * 8) * sizeof(T)) : kOOB;
    dma16(rv, vo, lds0 + (uint32_t)(t & 1) * STAGE + TILE + (uint32_t)row * 1024u);
  };
  auto issue = [&](int t) __attribute__((always_inline)) {      // a whole tile at once: the first tile of a pass only
#pragma unroll
    for (int i = 0; i < 8; ++i) { issue_k(t, i); issue_v(t, i); }
  };

  // ---- fragment addresses.  K: row lr, k-step s: chunk (2 s + lh) ^ (lr & 15) -- s enters above bit 0 of the chunk index,
  // (lr & 15) below bit 4: (2 s + lh) ^ x = 2 s ^ (lh ^ x) only for the low bits; keep it simple: two lane bases per parity
  // are not enough either, so the K address is base + ((2 s + lh) ^ (lr & 15)) * 16 computed from 16 precomputed... no:
  // (2 s + lh) ^ m with m = lr & 15 < 16 = (2 s & ~15) + (((2 s & 15) + lh) ^ m): s = 8 a + c (c = 0..7) gives
  // 256 a [immediate] + 16 * ((2 c + lh) ^ m): eight lane-constant bases (one per c), a = 0..3 as immediates.
  // (round-4 note: eight precomputed bases cost six registers this kernel does not have; (2 c + lh) ^ m splits into
  // (2 c ^ (m & 14)) | (lh ^ (m & 1)): one lane base + one lane mask, one v_xor + one v_add per read)
  const uint32_t kbase = (uint32_t)(lr * 1024 + ((lh ^ (lr & 1)) << 4)), kmask = (uint32_t)((lr & 14) << 4);
  // V^T (A operand [32 d][16 keys]): 16-lane groups, lane 4 tq + tp of a group addresses row tq, columns 4 tp .. + 3;
  // group tg covers d columns 16 tg .. + 15 of the block; half lh takes rows 4 lh + tq (and + 8): chunk of column
  // dcol = 32 db + 16 tg + 4 tp is 4 db + 2 tg + (tp >> 1), swizzled by ^ (tq << 2) -> 4 (db ^ tq) + (2 tg + (tp >> 1)):
  // four lane-constant bases (db & 3), db >> 2 as a 256-byte immediate.
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  uint32_t vb4[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
    vb4[c] = (uint32_t)((4 * lh + tq) * 1024 + (((4 * (c ^ tq) + 2 * tg + (tp >> 1))) << 4) + ((tp & 1) << 3));

  // The O accumulators never see a vector instruction inside the tile loop (a rescale `o *= alpha` there made the register
  // allocator keep O in VGPRs across the loop: 904 bytes of scratch per lane).  The reference maximum is the row maximum of
  // the FIRST tile; a later tile whose maximum exceeds it by more than 2^kDefer does not rescale O -- it raises a workgroup flag,
  // every wave leaves the tile loop at the next barrier, ONE extra pass over the keys computes the exact row maxima (QK^T
  // only), and the tile loop restarts with them (it cannot trigger again).  Rare: the flag needs a score 2^8 above everything
  // the first 32 keys produced; the result does not depend on the reference.
  f32x16 o[16];
  float m_run = 0.f, l_run = 0.f;
  volatile int* flag = (volatile int*)(smem + 2 * STAGE);
  if (tid == 0) *flag = 0;
  bool exact = false;

  // Fragment reads run PD MFMAs ahead of their consumer, in source order pinned by scheduling fences: the MFMAs are asm
  // statements, and left to itself hipcc placed every ds_read directly in front of the MFMA that needs it -- one exposed LDS
  // latency per MFMA, 4.8 us per 32-key tile where the 64 MFMAs take 1.1 (first build: 334 TFLOP/s at 12 x 4096 tokens).
  constexpr int PD = 3, NBUF = PD + 1;
  // KIN: issue the next tile's K rows inside the chain (the exact-maximum pass, which has no P.V phase).  The tile loop issues
  // them in its P.V phase instead: between the DEPENDENT MFMAs of this chain a piece cost ~340 cycles of matrix-pipe idle,
  // between the independent ones of P.V ~90 (ablation on MI355X, 8 x 4096 tokens: QK^T + DMA 255 us where QK^T alone takes 82).
  auto qk = [&](const char* kbuf, float init, int tnext, auto KINT) __attribute__((always_inline)) -> f32x16 {
    constexpr bool KIN = decltype(KINT)::value;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = init;
    typename Tr<T>::v8 kf[NBUF];
    uint32_t km = kmask;
    asm volatile("" : "+v"(km));        // opaque per tile: hoisted out of the tile loop the eight xor results are eight registers
    auto rd = [&](int ss) __attribute__((always_inline)) {
      kf[ss % NBUF] = as_v8<T>(*(const i32x4*)(kbuf + kbase + ((uint32_t)((ss & 7) << 5) ^ km) + (ss >> 3) * 256));
    };
#pragma unroll
    for (int i = 0; i < PD; ++i) rd(i);
    // (fully unrolled: the Q fragments and the fragment buffers are statically indexed registers)
#pragma unroll
    for (int ss = 0; ss < 32; ++ss) {
      if (ss + PD < 32) rd(ss + PD);
      __builtin_amdgcn_sched_barrier(0);
      mfma_vreg<T>(s, kf[ss % NBUF], qf[ss]);
      if (KIN && (ss & 3) == 1 && tnext >= 0) issue_k(tnext, ss >> 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_done(s);
    return s;
  };
  auto mask_tail = [&](f32x16& s, int t) __attribute__((always_inline)) {
    const int nvalid = p.n - t * KT;
    if (nvalid < KT) {
      asm volatile("" ::: "memory");      // a real branch (ragged last tile only)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (key >= nvalid) s[r] = -INFINITY;
      }
    }
  };
  auto row_max = [&](const f32x16& s) __attribute__((always_inline)) -> float {
    float mt = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
    return half_swap_max(mt);
  };

  __builtin_amdgcn_s_waitcnt(0x0F70);   // the Q loads (compiler-visible) must not leave counted waits inside the loops; vmcnt(0)
  for (;;) {
#pragma unroll
    for (int d = 0; d < 16; ++d) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
      asm volatile("" : "+a"(o[d]));
    }
    l_run = 0.f;
    issue(0);
#pragma unroll 1
    for (int t = 0; t < ntiles; ++t) {
      wait_vm<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // a flag written in the previous iteration is in LDS before the barrier
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (*flag) break;                                        // workgroup-uniform: every wave reads it behind the same barrier
      const int tnext = t + 1 < ntiles ? t + 1 : -1;
      const char* kbuf = smem + (t & 1) * STAGE;
      const char* vbuf = kbuf + TILE;
      const bool first = t == 0 && !exact;
      // ---- S^T = K . Q^T, accumulators start at -m_ref (0 on the tile that sets the reference)
      f32x16 s = qk(kbuf, first ? 0.f : -m_run, tnext, std::false_type{});
      mask_tail(s, t);
      // ---- online softmax: this lane's 16 keys, the row's other 16 in lane ^ 32
      const float mt = row_max(s);
      float d = 0.f;
      if (first) {
        d = mt;
        m_run = mt;
      } else if (__builtin_amdgcn_ballot_w64(mt > kDefer) != 0) {
        if (lane == 0) *flag = 1;
        continue;                                              // no P.V with an unusable reference (nobody will read the next tile)
      }
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[r] - d);
        s[r] = e;
        psum += e;
      }
      l_run += psum;
      typename Tr<T>::v8 pf[2];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[t2][j] = (T)s[8 * t2 + j];
      // ---- O^T += V^T . P^T
      {
        typename Tr<T>::v4 vlo[NBUF], vhi[NBUF];
        auto rdv = [&](int u) __attribute__((always_inline)) {     // u = 2 db + t2
          const char* vd = vbuf + vb4[(u >> 1) & 3] + (u >> 3) * 256 + (16 * (u & 1)) * 1024;
          vlo[u % NBUF] = lds_tr_read<T>(vd);
          vhi[u % NBUF] = lds_tr_read<T>(vd + 8 * 1024);
        };
#pragma unroll
        for (int i = 0; i < PD; ++i) rdv(i);
#pragma unroll
        for (int u = 0; u < 32; ++u) {
          if (u + PD < 32) rdv(u + PD);
          __builtin_amdgcn_sched_barrier(0);
          typename Tr<T>::v8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) { vf[j] = vlo[u % NBUF][j]; vf[4 + j] = vhi[u % NBUF][j]; }
          mfma_acc<T>(o[u >> 1], vf, pf[u & 1]);
          if ((u & 3) == 1 && tnext >= 0) issue_k(tnext, u >> 2);
          if ((u & 3) == 3 && tnext >= 0) issue_v(tnext, u >> 2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // common exit of the tile loop (natural end or flag seen): drain, rendezvous, read the flag once more (a flag raised on the
    // LAST tile is seen only here)
    wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const bool redo = *flag != 0;
    if (!redo) break;
    __builtin_amdgcn_s_barrier();          // everyone has read the flag
    asm volatile("" ::: "memory");
    if (tid == 0) *flag = 0;
    // ---- exact row maxima: one pass of QK^T over all keys
    float mx = -INFINITY;
    issue(0);
#pragma unroll 1
    for (int t = 0; t < ntiles; ++t) {
      wait_vm<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      f32x16 s = qk(smem + (t & 1) * STAGE, 0.f, t + 1 < ntiles ? t + 1 : -1, std::true_type{});     // (K rows only)
      mask_tail(s, t);
      mx = fmaxf(mx, row_max(s));
    }
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();          // the stages are idle again before the restart's first issue
    asm volatile("" ::: "memory");
    m_run = mx;
    exact = true;
  }

  // ---- epilogue: normalise, stage the wave's 32 x 512 block through LDS (every tile has been read: one barrier), whole rows out
  const float inv = 1.0f / half_swap_sum(l_run);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  char* stg = smem + wave * 32768;
#pragma unroll
  for (int db = 0; db < 16; ++db) {
    asm volatile("" : "+a"(o[db]));
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = o[db][4 * gg + e] * inv;
      const int chunk = db * 4 + gg;                        // 16-byte chunk of the row holding d = 32 db + 8 gg + 4 lh .. + 3
      *(i32x2*)(stg + lr * 1024 + ((chunk ^ (lr & 15)) << 4) + lh * 8) = pack4<T>(v);
    }
  }
  // same wave wrote and reads its region: program order + the compiler's lgkmcnt wait suffice
#pragma unroll 4
  for (int r = 0; r < 32; ++r) {
    const int qrow = q0 + r;
    const i32x4 val = *(const i32x4*)(stg + r * 1024 + ((lane ^ (r & 15)) << 4));
    if (qrow < p.n) *(i32x4*)(p.out + ((size_t)b * p.o_bs + (size_t)qrow * p.ldo + lane * 8) * sizeof(T)) = val;
  }
}

}  // namespace dfw

using namespace dfw;

extern "C" int dfw_vae_attention(const dfw_vattn_args* a, dfw_stream_t stream) {
  if (!a || !a->q || !a->k || !a->v || !a->out) return DFW_EINVAL;
  if (a->batch <= 0 || a->n <= 0) return DFW_EINVAL;
  if (a->dtype != DFW_BF16 && a->dtype != DFW_F16) return DFW_EINVAL;
  if (a->head_dim != 512 || !a->q_prescaled) return DFW_ESHAPE;       // the one shape of the SD VAE; q * scale * log2(e) by the projection
  if ((a->ldq | a->ldk | a->ldv | a->ldo) % 8 != 0 || a->ldq < 512 || a->ldk < 512 || a->ldv < 512 || a->ldo < 512) return DFW_ESHAPE;
  if (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->out) & 15) return DFW_ESHAPE;
  const long long qe = (long long)(a->batch - 1) * a->q_bs + (long long)(a->n - 1) * a->ldq + 512;
  const long long ke = (long long)(a->batch - 1) * a->k_bs + (long long)(a->n - 1) * a->ldk + 512;
  const long long ve = (long long)(a->batch - 1) * a->v_bs + (long long)(a->n - 1) * a->ldv + 512;
  if (qe >= (1ll << 30) || ke >= (1ll << 30) || ve >= (1ll << 30)) return DFW_ERANGE;
  VattnP p;
  p.q = (const char*)a->q; p.k = (const char*)a->k; p.v = (const char*)a->v; p.out = (char*)a->out;
  p.q_bytes = (uint32_t)(qe * 2); p.k_bytes = (uint32_t)(ke * 2); p.v_bytes = (uint32_t)(ve * 2);
  p.batch = a->batch; p.n = a->n; p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
  p.q_bs = a->q_bs; p.k_bs = a->k_bs; p.v_bs = a->v_bs; p.o_bs = a->o_bs;
  const dim3 grid(8 * ((a->n + 127) / 128) * ((a->batch + 7) / 8));
  constexpr int lds = 131072 + 16;      // two stages + the restart flag
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == DFW_BF16) {
    auto kfn = vattn_kernel<__bf16>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, p);
  } else {
    auto kfn = vattn_kernel<_Float16>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, p);
  }
  DFW_CHECK_LAUNCH();
  return 0;
}
