// Backward of the KV-fusion self-attention (attention_processor.py:247-271 under autograd: the banks keep their
// graph, so the query loss reaches the support pass through k_bank / v_bank, T:1374-1375), head_dim 64, for the
// lock-step batch [support images ; query images] of dfw_fsa_args.n_plain (or plain self-attention, nshot = 0).
//
// Flash-style: the N x N_k probabilities are recomputed from q, k and the forward's per-row log-sum-exp instead
// of being stored.  Two kernels, each without any cross-workgroup sum (deterministic, no atomics):
//   fsa_bwd_dq_kernel   one workgroup = 128 query rows of one (image, head); walks that row's keys
//                       [own ; shot 0 ; shot 1 ...] exactly like the forward:  S^T = K Q^T - lse (the row constant
//                       rides in as the accumulator's initial value), P^T = exp2(S^T), dP^T = V dO^T - delta,
//                       dS^T = P^T o dP^T, dQ^T += K^T dS^T.
//   fsa_bwd_dkv_kernel  one workgroup = 128 keys of one (image, head); walks every query row that attends to them
//                       (the image's own queries and, for a support image, its episode's query image):
//                       S = Q K^T - lse, P = exp2(S), dP = dO V^T - delta, dS = P o dP, dV^T += dO^T P,
//                       dK^T += Q^T dS.  K / V fragments stay in registers; the key sits on the MFMA lane, so P and
//                       dS are used as B operands straight from the accumulators.
// q is the PRE-SCALED query (q * scale * log2 e, dfw_gemm_args.colscale): scores are exp2 exponents.  dq is returned
// with respect to the UNSCALED projection output (d(x Wq) = scale * dS K), dk = ln2 * dS^T q_pre, so the caller
// backpropagates through the fused QKV Linear without special cases.
#include "common.h"
#include "attention_common.h"

// Round 3 (counter-driven, profiles/r03_attention_bwd_*): (1) amdgpu_waves_per_eu(2) -- with the 512-register budget of a
// 256-thread kernel hipcc kept the MFMA accumulators in AGPRs and copied them around the vector work (256 copies per tile);
// (2) the streamed tiles arrive by LDS-DMA into a double buffer (one barrier per tile, no staging registers), the DMA
// descriptors rebuilt per tile from scalars so that the lane offsets are loop constants; (3) no per-element masks in full
// tiles and no s_setprio fences between the stages of a tile, so the compiler overlaps one block's exponentials with the
// other's MFMAs; (4) a query split of the dK/dV kernel when the key axis is short.  1509 -> 1244 us on the 64x64-level 7-shot
// launch, 9.2 -> 7.3 ms of attention backward per training step.

namespace dfw {

struct FsaBwdP {
  const char* q; const char* k; const char* v; const char* dout; const float* lse; const float* delta;
  char* dq; char* dk; char* dv;
  uint32_t qkv_bytes, do_bytes, dqkv_bytes;
  int batch, heads, n, nshot, n_plain;
  int ld, ldo, ldd;                       // token strides of q, dout, dq
  long long bs, obs, dbs;                 // image strides
  float scale;
  // keys / values: n_kv rows per image at token stride ldkv, image stride kvbs, addressed from p.k (v = k + voff bytes
  // inside the same buffer of kv_bytes); their gradients at lddkv / dkvbs from p.dk / p.dv.  Self-attention over the fused
  // qkv buffer: n_kv = n, ldkv = ld, kvbs = bs, lddkv = ldd, dkvbs = dbs.  Cross-attention: its own K/V tensors.
  int n_kv, ldkv, lddkv;
  long long kvbs, dkvbs;
  uint32_t kv_bytes, voff;
  // dQ key split (the forward's FsaP::nsplit): a bank-reading image appears nsplit times in the dQ grid, each instance
  // walks a contiguous range of the key segments and leaves its partial dQ (fp32, already scaled) in `part`
  // [(batch - n_plain) * nsplit][n][heads * 64]; fsa_dq_combine_kernel sums them in order.  dQ is linear in the keys.
  int nsplit;
  float* part;
  // row constants as the kernels read them: stat[0 .. total) = -delta, stat[total .. 2 total) = -lse (total = batch * heads * n),
  // written by fsa_delta_kernel into the caller's `delta` scratch (2 * total floats)
  const float* stat; long long stat_half; uint32_t stat_bytes;
  // dK/dV query split (short key axis: the 77 prompt tokens are ONE key block per image and head, 40 workgroups on the 64x64
  // level): grid.x lists every key block qsplit times, instance c walks the c-th run of the query tiles and leaves fp32
  // partials in kvpart [2 (dK | dV)][qsplit][batch][n_kv][heads * 64]; fsa_dkv_fold_kernel sums them in order.
  int qsplit;
  float* kvpart;
};

// A 256-thread kernel is allowed 512 registers per wave, and with that budget hipcc 7.2 puts MFMA accumulators in AGPRs and
// copies them to VGPRs and back around the vector work: 256 v_accvgpr_read / _write per tile beside 32 MFMAs in the dK/dV
// kernel (SQ_INSTS_VALU 353 per wave-tile, the kernels were vector-issue-bound on copies).  Asking for two waves per SIMD
// caps the budget at 256, every MFMA takes the VGPR form and the copies disappear.
#define DFW_TWO_WAVES __attribute__((amdgpu_waves_per_eu(2)))

__device__ __forceinline__ uint32_t row_off(int row, int chunk) {   // K-style image: b128 row reads
  return (uint32_t)(row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
__device__ __forceinline__ uint32_t tr_off(int row, int dcol) {     // V-style image: transposed reads
  return (uint32_t)(row * 128 + ((((dcol >> 3) ^ (((row >> 1) & 1) << 2))) << 4) + ((dcol & 7) << 1));
}

// stat[e] = -delta[b][h][q] = -sum_d dO * O,  stat[total + e] = -lse[b][h][q]   (fp32): the two row constants in the form
// the dQ / dK-dV kernels start their accumulators from (the dK-dV kernel fetches them by LDS-DMA: no arithmetic on the way)
template <typename T>
__global__ __launch_bounds__(256) void fsa_delta_kernel(const char* o, const char* dout, const float* lse, float* delta, int batch,
                                                        int heads, int n, int ldo, long long obs, int ldd, long long dbs) {
  // 8 adjacent lanes per row, 16 bytes each: a wave-instruction reads 8 whole 128-byte rows (one thread per row made it
  // touch 64 different cache lines per load: 18 us where the bytes take 8)
  const long long total = (long long)batch * heads * n;
  const int c = threadIdx.x & 7;
  for (long long e0 = (long long)blockIdx.x * 32; e0 < total; e0 += (long long)gridDim.x * 32) {
    const long long e = e0 + (threadIdx.x >> 3);
    float acc = 0.f;
    if (e < total) {
      const int qi = (int)(e % n);
      const long long bh = e / n;
      const int h = (int)(bh % heads), b = (int)(bh / heads);
      float a[8], d[8];
      unpack8<T>(*(const i32x4*)(o + ((size_t)b * obs + (size_t)qi * ldo + h * 64) * sizeof(T) + c * 16), a);
      unpack8<T>(*(const i32x4*)(dout + ((size_t)b * dbs + (size_t)qi * ldd + h * 64) * sizeof(T) + c * 16), d);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += a[i] * d[i];
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (c == 0 && e < total) {
      delta[e] = -acc;
      delta[total + e] = -lse[e];
    }
  }
}

// ------------------------------------------------------------------------------------------------ dQ
template <typename T>
__global__ __launch_bounds__(256) DFW_TWO_WAVES void fsa_bwd_dq_kernel(const FsaBwdP p) {
  constexpr int KT = 64, TILE = KT * 128;
  __shared__ __attribute__((aligned(16))) char smem[2 * 3 * TILE];   // [buf][K rows | K tr | V rows]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y;
  int b = (int)gridDim.z - 1 - (int)blockIdx.z, split = 0;      // longest rows (the bank readers, at the end) first
  if (p.nsplit > 1 && b >= p.n_plain) {
    const int v = b - p.n_plain;
    split = v % p.nsplit;
    b = p.n_plain + v / p.nsplit;
  }
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int bank_b = b - p.n_plain;
  const __amdgpu_buffer_rsrc_t rqkv = make_rsrc(p.q, p.qkv_bytes);
  const __amdgpu_buffer_rsrc_t rdo = make_rsrc(p.dout, p.do_bytes);
  const uint32_t voff = p.voff;

  const int qrow = q0 + lr;
  const bool qok = qrow < p.n;
  typename Tr<T>::v8 qf[4], dof[4];
  {
    const uint32_t bq = qok ? (uint32_t)(((size_t)b * p.bs + (size_t)qrow * p.ld + head * 64 + lh * 8) * sizeof(T)) : kOOB;
    const uint32_t bd = qok ? (uint32_t)(((size_t)b * p.obs + (size_t)qrow * p.ldo + head * 64 + lh * 8) * sizeof(T)) : kOOB;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = as_v8<T>(buf_load16(rqkv, bq == kOOB ? kOOB : bq + (uint32_t)(s * 32)));
      dof[s] = as_v8<T>(buf_load16(rdo, bd == kOOB ? kOOB : bd + (uint32_t)(s * 32)));
    }
  }
  const size_t stat = ((size_t)b * p.heads + head) * p.n + (qok ? qrow : 0);
  const float nlse = qok ? -p.lse[stat] : 0.f;           // rows past the end: Q = dO = 0, P = exp2(0) = 1, dP = dS = 0
  const float ndelta = qok ? p.stat[stat] : 0.f;

  const int tiles_own = (p.n_kv + KT - 1) / KT;
  const int tiles_bank = (p.nshot > 0 && bank_b >= 0) ? tiles_own : 0;
  const int ntiles = tiles_own + (tiles_bank ? p.nshot * tiles_bank : 0);
  // this instance's key segments [seg0, seg1) of [own ; shot 0 ; ...] as a range of the global tile numbering
  const int nseg = 1 + (tiles_bank ? p.nshot : 0);
  const bool parted = p.nsplit > 1 && bank_b >= 0;
  const int seg0 = parted ? split * nseg / p.nsplit : 0, seg1 = parted ? (split + 1) * nseg / p.nsplit : nseg;
  const int t_begin = seg0 == 0 ? 0 : tiles_own + (seg0 - 1) * tiles_bank;
  const int t_end = parted ? tiles_own + (seg1 - 1) * tiles_bank : ntiles;
  // Key-side tiles by LDS-DMA into the double buffer, as in the dK/dV kernel below: per 64-key tile three 8 KiB images
  // [K rows | K tr | V rows], wave w issues pieces w and w + 4 of each (6 instructions), swizzles applied on the source side.
  // (per-tile buffer descriptors and loop-constant lane offsets: see the dK/dV kernel)
  const uint32_t lds0 = lds_addr(smem);
  uint32_t vk[2][2];
  {
    const int lrow = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (wave + 4 * j) * 8 + lrow;
      vk[j][0] = (uint32_t)((row * p.ldkv + (slot ^ ((row >> 1) & 7)) * 8) * sizeof(T));            // row image
      vk[j][1] = (uint32_t)((row * p.ldkv + (slot ^ (((row >> 1) & 1) << 2)) * 8) * sizeof(T));     // transposed-read image
    }
  }
  auto issue = [&](int t, int buf) {
    int img = b, tt = t;
    if (t >= tiles_own) { img = bank_b * p.nshot + (t - tiles_own) / tiles_bank; tt = (t - tiles_own) % tiles_bank; }
    const uint32_t dst = lds0 + (uint32_t)buf * (3 * TILE);
    const int rows = min(KT, p.n_kv - tt * KT);                  // valid keys of this tile (>= 1)
    const uint32_t bytes = (uint32_t)(((rows - 1) * p.ldkv + 64) * sizeof(T));
    const char* kb = p.k + ((size_t)img * p.kvbs + (size_t)tt * KT * p.ldkv + head * 64) * sizeof(T);
    const u32x4 rk = make_srd(kb, bytes), rv = make_srd(kb + voff, bytes);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t piece = (uint32_t)(wave + 4 * j) * 1024u;
      dma16(rk, vk[j][0], dst + piece);
      dma16(rk, vk[j][1], dst + TILE + piece);
      dma16(rv, vk[j][0], dst + 2 * TILE + piece);
    }
  };
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  f32x16 o[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;

  // the Q / dO fragments and the row constants came by compiler-visible loads: one full wait the compiler understands, so that
  // it plants no vmcnt of its own inside the loop (which would drain the DMA of the next tile)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  issue(t_begin, 0);
  int cur = 0;
  for (int t = t_begin; t < t_end; ++t) {
    wait_vm<0>();                      // this wave's pieces of tile t have landed
    __builtin_amdgcn_s_barrier();      // tile t visible to every wave; every wave is done reading tile t-1
    asm volatile("" ::: "memory");
    if (t + 1 < t_end) issue(t + 1, cur ^ 1);
    const char* kbuf = smem + cur * 3 * TILE;
    const char* ktr = kbuf + TILE;
    const char* vbuf = kbuf + 2 * TILE;
    int tt = t;
    if (t >= tiles_own) tt = (t - tiles_own) % tiles_bank;
    const int nvalid = p.n_kv - tt * KT;
    // S^T = K Q^T - lse ;  dP^T = V dO^T - delta   (row constants as initial accumulators)
    f32x16 s[2], dp[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[kb][r] = nlse; dp[kb][r] = ndelta; }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        const int row = kb * 32 + lr;
        const typename Tr<T>::v8 kf = as_v8<T>(*(const i32x4*)(kbuf + row_off(row, 2 * ss + lh)));
        const typename Tr<T>::v8 vf = as_v8<T>(*(const i32x4*)(vbuf + row_off(row, 2 * ss + lh)));
        s[kb] = Tr<T>::mfma(kf, qf[ss], s[kb]);
        dp[kb] = Tr<T>::mfma(vf, dof[ss], dp[kb]);
      }
    }
    // dS^T = P^T o dP^T ; rows of S^T are keys: (r & 3) + 8 (r >> 2) + 4 lh within the 32-key block.
    // No per-element mask in the full tiles (it was 68 of the loop's ~190 vector instructions): a query row past the end has
    // Q = dO = 0 and constants 0, so P = 1 and dS = 0; keys past the end exist only in an image's last tile, where their
    // scores are sent to -inf under a wave-uniform branch (their K rows are zero anyway, this keeps P finite).
    if (nvalid < KT) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh >= nvalid) s[kb][r] = -INFINITY;
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r]) * dp[kb][r];
    // dQ^T += K^T dS^T   (no s_setprio fences between the stages: the compiler interleaves block 1's exponentials with block
    // 0's MFMAs, which the fences prevented)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        typename Tr<T>::v8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (T)s[kb][8 * t2 + j];
        const int row0 = kb * 32 + 16 * t2 + 4 * lh + tq, row1 = row0 + 8;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int dcol = d * 32 + 16 * tg + 4 * tp;
          const typename Tr<T>::v4 lo = lds_tr_read<T>(ktr + tr_off(row0, dcol));
          const typename Tr<T>::v4 hi = lds_tr_read<T>(ktr + tr_off(row1, dcol));
          typename Tr<T>::v8 kf;
#pragma unroll
          for (int j = 0; j < 4; ++j) { kf[j] = lo[j]; kf[4 + j] = hi[j]; }
          o[d] = Tr<T>::mfma(kf, pf, o[d]);
        }
      }
    cur ^= 1;
  }
  if (qok && parted) {
    float* pr = p.part + (((size_t)bank_b * p.nsplit + split) * p.n + qrow) * ((size_t)p.heads * 64) + head * 64;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o[d][4 * g + e] * p.scale;
        *(f32x4*)(pr + d * 32 + 8 * g + 4 * lh) = v;
      }
  } else if (qok) {
    char* ob = p.dq + ((size_t)b * p.dbs + (size_t)qrow * p.ldd + head * 64) * sizeof(T);
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o[d][4 * g + e] * p.scale;
        *(i32x2*)(ob + (d * 32 + 8 * g + 4 * lh) * sizeof(T)) = pack4<T>(v);
      }
  }
}

// dq[b][row][c] = sum over the splits (in order) of the partial dQ; one thread per 4 columns
template <typename T>
__global__ __launch_bounds__(256) void fsa_dq_combine_kernel(const FsaBwdP p) {
  const int C = p.heads * 64, c4n = C / 4;
  const long long total = (long long)(p.batch - p.n_plain) * p.n * c4n;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int c4 = (int)(e % c4n);
  const long long r = e / c4n;
  const int row = (int)(r % p.n), bq = (int)(r / p.n);
  const float* base = p.part + (((size_t)bq * p.nsplit) * p.n + row) * (size_t)C + c4 * 4;
  const size_t sstride = (size_t)p.n * C;
  f32x4 acc = *(const f32x4*)base;
  for (int s2 = 1; s2 < p.nsplit; ++s2) {
    const f32x4 v = *(const f32x4*)(base + s2 * sstride);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += v[i];
  }
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  *(i32x2*)(p.dq + ((size_t)(p.n_plain + bq) * p.dbs + (size_t)row * p.ldd + c4 * 4) * sizeof(T)) = pack4<T>(v);
}

// ------------------------------------------------------------------------------------------------ dK, dV
// Round 3: the query-side tiles arrive by LDS-DMA (buffer_load ... lds) into a double buffer -- no register staging, no
// ds_write phase, ONE barrier per tile (the register-staged single buffer needed two and held 16 + 1 staging registers):
// per 64-row tile four 8 KiB images [Q rows | Q tr | dO rows | dO tr] (Q and dO are fetched twice, once per swizzle: the row
// image serves the b128 row reads of S / dP, the transposed image the ds_read_b64_tr_b16 of dV^T / dK^T; the second fetch is
// an L2 hit) and the two row constants (-lse, -delta: 2 x 256 B, dword LDS-DMA from the negated copies fsa_delta_kernel
// wrote).  Wave w issues pieces w and w + 4 of every image (8 instructions), wave 0 the constants too.  Schedule of tile t:
// vmcnt(0) (tile t was issued a whole tile ago) -> barrier (publishes tile t; every wave is done with tile t-1, whose buffer
// is free) -> issue tile t+1 into the other buffer -> the MFMA / softmax work of tile t.  The loop contains no compiler-visible
// vector-memory operation, so nothing but the explicit vmcnt(0) ever waits on the DMA.  Rows past the end of the image are
// zero-filled by the bounds check (Q = dO = 0, constants 0): P = 1, dP = dS = 0, and both products receive exact zeros.
__device__ __forceinline__ void dma4(u32x4 srd, uint32_t voff, uint32_t lds_byte) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
               :: "s"(__builtin_amdgcn_readfirstlane(lds_byte)), "v"(voff), "s"(srd) : "memory");
}

template <typename T>
__global__ __launch_bounds__(256) DFW_TWO_WAVES void fsa_bwd_dkv_kernel(const FsaBwdP p) {
  constexpr int QT = 64, TILE = QT * 128, BUF = 4 * TILE + 2 * QT * 4;   // four images + [-lse | -delta]
  extern __shared__ __attribute__((aligned(16))) char smem[];            // 2 * BUF = 65 KiB (dynamic: above the static limit)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y, kimg = blockIdx.z;
  const int kblk = (int)blockIdx.x / p.qsplit, qchunk = (int)blockIdx.x - kblk * p.qsplit;
  const int key = kblk * 128 + wave * 32 + lr;            // the key this lane owns (column of S)
  const bool kok = key < p.n_kv;
  const __amdgpu_buffer_rsrc_t rkv = make_rsrc(p.k, p.kv_bytes);
  const uint32_t voff = p.voff;
  // B operands: lane holds K[key][16 s + 8 lh + 0..7] (= K^T[k = d][col = key]), same for V
  typename Tr<T>::v8 kf[4], vf[4];
  {
    const uint32_t base = kok ? (uint32_t)(((size_t)kimg * p.kvbs + (size_t)key * p.ldkv + head * 64 + lh * 8) * sizeof(T)) : kOOB;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = as_v8<T>(buf_load16(rkv, base == kOOB ? kOOB : base + (uint32_t)(s * 32)));
      vf[s] = as_v8<T>(buf_load16(rkv, base == kOOB ? kOOB : base + voff + (uint32_t)(s * 32)));
    }
  }
  // query sources: the image's own rows, plus (support image of a lock-step batch) its episode's query image
  const int nsrc = (p.nshot > 0 && kimg < p.n_plain) ? 2 : 1;
  const int src1 = p.nshot > 0 ? p.n_plain + kimg / p.nshot : 0;
  const int tiles_q = (p.n + QT - 1) / QT;
  const int ntiles_all = nsrc * tiles_q;
  const int t_begin = (int)((long long)qchunk * ntiles_all / p.qsplit), ntiles = (int)((long long)(qchunk + 1) * ntiles_all / p.qsplit);

  // DMA addressing without per-tile vector arithmetic (it was a third of the loop's vector instructions): the buffer
  // descriptors are rebuilt PER TILE from scalars -- base = the tile's first row in this head's 64 columns, size = up to the
  // last valid row -- so the lane offsets (row * ld + swizzled chunk) are loop constants and rows past the end of the image
  // fall outside the descriptor and are zero-filled by the bounds check.
  const uint32_t lds0 = lds_addr(smem);
  uint32_t vq[2][2], vd[2][2];
  {
    const int lrow = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (wave + 4 * j) * 8 + lrow;
      const int cr = slot ^ ((row >> 1) & 7);                    // row image: chunk of LDS slot `slot`
      const int ct = slot ^ (((row >> 1) & 1) << 2);             // transposed-read image
      vq[j][0] = (uint32_t)((row * p.ld + cr * 8) * sizeof(T));
      vq[j][1] = (uint32_t)((row * p.ld + ct * 8) * sizeof(T));
      vd[j][0] = (uint32_t)((row * p.ldo + cr * 8) * sizeof(T));
      vd[j][1] = (uint32_t)((row * p.ldo + ct * 8) * sizeof(T));
    }
  }
  const uint32_t vst = (uint32_t)lane * 4u;
  auto issue = [&](int t) {
    const int img = t < tiles_q ? kimg : src1, tt = t < tiles_q ? t : t - tiles_q;
    const uint32_t dst = lds0 + (uint32_t)(t & 1) * BUF;
    const int rows = min(QT, p.n - tt * QT);                     // valid rows of this tile (>= 1)
    const u32x4 rq = make_srd(p.q + ((size_t)img * p.bs + (size_t)tt * QT * p.ld + head * 64) * sizeof(T),
                              (uint32_t)(((rows - 1) * p.ld + 64) * sizeof(T)));
    const u32x4 rdo = make_srd(p.dout + ((size_t)img * p.obs + (size_t)tt * QT * p.ldo + head * 64) * sizeof(T),
                               (uint32_t)(((rows - 1) * p.ldo + 64) * sizeof(T)));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t piece = (uint32_t)(wave + 4 * j) * 1024u;
      dma16(rq, vq[j][0], dst + piece);
      dma16(rq, vq[j][1], dst + TILE + piece);
      dma16(rdo, vd[j][0], dst + 2 * TILE + piece);
      dma16(rdo, vd[j][1], dst + 3 * TILE + piece);
    }
    if (wave == 0) {                         // [-lse | -delta] of the tile's 64 rows: one dword per lane each
      const float* st = p.stat + ((size_t)img * p.heads + head) * p.n + (size_t)tt * QT;
      dma4(make_srd(st + p.stat_half, (uint32_t)rows * 4u), vst, dst + 4 * TILE);
      dma4(make_srd(st, (uint32_t)rows * 4u), vst, dst + 4 * TILE + QT * 4);
    }
  };
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  f32x16 dk[2], dv[2];     // dK^T / dV^T [d block][rows d, col = key]
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }

  // The K / V fragments were fetched with compiler-visible buffer loads: without a wait the COMPILER understands it cannot
  // prove them complete across the loop back-edge and plants vmcnt(1) / vmcnt(0) in front of the tile's MFMAs -- which would
  // also drain the DMA of tile t+1 issued just before (seen in the first build's ISA).  One full wait here removes them.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
  if (t_begin < ntiles) issue(t_begin);
  for (int t = t_begin; t < ntiles; ++t) {
    wait_vm<0>();                      // this wave's pieces of tile t have landed
    __builtin_amdgcn_s_barrier();      // tile t visible to every wave; every wave is done reading tile t-1
    asm volatile("" ::: "memory");
    if (t + 1 < ntiles) issue(t + 1);  // flies during this tile's MFMAs
    const char* qrow_t = smem + (t & 1) * BUF;
    const char* qtr = qrow_t + TILE;
    const char* drow = qrow_t + 2 * TILE;
    const char* dtr = qrow_t + 3 * TILE;
    const float* stats = (const float*)(qrow_t + 4 * TILE);      // [0..63] -lse, [64..127] -delta
    // One tile = two 32-row query blocks qb, three stages each: A(qb) S = Q K^T - lse and dP = dO V^T - delta (8 MFMAs),
    // B(qb) P = exp2(S), dS = P o dP, 16-bit pack (48 vector instructions), C(qb) dV^T += dO^T P and dK^T += Q^T dS (8 MFMAs).
    // Written as A(0) A(1) B(0) C(0) B(1) C(1) in ONE scheduling region without s_setprio fences, so that the compiler puts
    // the vector work of one block into the MFMA gaps of the other: a wave that runs the stages back to back leaves the matrix
    // pipe idle during B and the vector pipe idle during A / C, and with two waves per SIMD the partner covers only part of it
    // (counters before: MFMA busy 40 %, 34 % of wave time parked on waits; 1403 -> 1287 us on the 64x64-level 7-shot launch.
    // An explicit sched_group_barrier pipeline of the same stages was 3.5 % slower than the compiler's own interleave; an
    // 8-wave workgroup whose two wave groups split the query rows -- equal tile counts for support and query images -- was
    // 3-8 % slower on every level: the shared barrier costs more than the balance gains).
    f32x16 sa[2], da2[2];
    typename Tr<T>::v8 pf[2][2], sf[2][2];     // [qb][k-step]: P and dS as B operands
    auto stage_a = [&](int qb) {
      // rows = queries: (r & 3) + 8 (r >> 2) + 4 lh of the 32-row block
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        sa[qb][r] = stats[qi];
        da2[qb][r] = stats[QT + qi];
      }
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        const int row = qb * 32 + lr;
        const typename Tr<T>::v8 qa = as_v8<T>(*(const i32x4*)(qrow_t + row_off(row, 2 * ss + lh)));
        const typename Tr<T>::v8 da = as_v8<T>(*(const i32x4*)(drow + row_off(row, 2 * ss + lh)));
        sa[qb] = Tr<T>::mfma(qa, kf[ss], sa[qb]);
        da2[qb] = Tr<T>::mfma(da, vf[ss], da2[qb]);
      }
    };
    auto stage_b = [&](int qb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(sa[qb][r]);     // a key past the end: K = V = 0 (bounds check), P finite, its column is never stored
        pf[qb][r >> 3][r & 7] = (T)pr;
        sf[qb][r >> 3][r & 7] = (T)(pr * da2[qb][r]);
      }
    };
    auto stage_c = [&](int qb) {
      // A fragments: transposed reads, rows in the accumulator's k order
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const int row0 = qb * 32 + 16 * t2 + 4 * lh + tq, row1 = row0 + 8;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int dcol = d * 32 + 16 * tg + 4 * tp;
          const typename Tr<T>::v4 dlo = lds_tr_read<T>(dtr + tr_off(row0, dcol)), dhi = lds_tr_read<T>(dtr + tr_off(row1, dcol));
          const typename Tr<T>::v4 qlo = lds_tr_read<T>(qtr + tr_off(row0, dcol)), qhi = lds_tr_read<T>(qtr + tr_off(row1, dcol));
          typename Tr<T>::v8 da, qa;
#pragma unroll
          for (int j = 0; j < 4; ++j) { da[j] = dlo[j]; da[4 + j] = dhi[j]; qa[j] = qlo[j]; qa[4 + j] = qhi[j]; }
          dv[d] = Tr<T>::mfma(da, pf[qb][t2], dv[d]);
          dk[d] = Tr<T>::mfma(qa, sf[qb][t2], dk[d]);
        }
      }
    };
    stage_a(0);
    stage_a(1);
    stage_b(0);
    stage_c(0);
    stage_b(1);
    stage_c(1);
  }
  if (kok && p.qsplit > 1) {
    const size_t Cc = (size_t)p.heads * 64, slab = (size_t)p.qsplit * p.batch * p.n_kv * Cc;
    float* pk = p.kvpart + (((size_t)qchunk * p.batch + kimg) * p.n_kv + key) * Cc + head * 64;
    const float ln2 = 0.6931471805599453f;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 a, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = dk[d][4 * g + e] * ln2; c[e] = dv[d][4 * g + e]; }
        *(f32x4*)(pk + d * 32 + 8 * g + 4 * lh) = a;
        *(f32x4*)(pk + slab + d * 32 + 8 * g + 4 * lh) = c;
      }
  } else if (kok) {
    // D layout of dK^T / dV^T: col = key (this lane), rows d = (r & 3) + 8 (r >> 2) + 4 lh of the 32-d block
    char* kb = p.dk + ((size_t)kimg * p.dkvbs + (size_t)key * p.lddkv + head * 64) * sizeof(T);
    char* vb = p.dv + ((size_t)kimg * p.dkvbs + (size_t)key * p.lddkv + head * 64) * sizeof(T);
    const float ln2 = 0.6931471805599453f;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float a[4], c[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = dk[d][4 * g + e] * ln2; c[e] = dv[d][4 * g + e]; }
        *(i32x2*)(kb + (d * 32 + 8 * g + 4 * lh) * sizeof(T)) = pack4<T>(a);
        *(i32x2*)(vb + (d * 32 + 8 * g + 4 * lh) * sizeof(T)) = pack4<T>(c);
      }
  }
}

// dk / dv [img][key][c] = sum over the query chunks (in order) of the fp32 partials; one thread per 4 columns
template <typename T>
__global__ __launch_bounds__(256) void fsa_dkv_fold_kernel(const FsaBwdP p) {
  const int Cc = p.heads * 64, c4n = Cc / 4;
  const long long total = (long long)p.batch * p.n_kv * c4n;
  const size_t slab = (size_t)p.qsplit * p.batch * p.n_kv * Cc, chunk = (size_t)p.batch * p.n_kv * Cc;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c4 = (int)(e % c4n);
    const long long rk = e / c4n;
    const int key = (int)(rk % p.n_kv), img = (int)(rk / p.n_kv);
    const float* src = p.kvpart + ((size_t)img * p.n_kv + key) * Cc + c4 * 4;
    f32x4 a = *(const f32x4*)src, c = *(const f32x4*)(src + slab);
    for (int q = 1; q < p.qsplit; ++q) {
      const f32x4 a2 = *(const f32x4*)(src + q * chunk), c2 = *(const f32x4*)(src + slab + q * chunk);
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] += a2[i]; c[i] += c2[i]; }
    }
    float av[4] = {a[0], a[1], a[2], a[3]}, cv[4] = {c[0], c[1], c[2], c[3]};
    const size_t o = ((size_t)img * p.dkvbs + (size_t)key * p.lddkv + c4 * 4) * sizeof(T);
    *(i32x2*)(p.dk + o) = pack4<T>(av);
    *(i32x2*)(p.dv + o) = pack4<T>(cv);
  }
}

// ------------------------------------------------------------------------------------------------ attn2 backward
// Cross-attention over a short context (L prompt tokens; 77 in training, T:1368): VALU kernel like the forward.
// Workgroup = 64 query rows of one (image, head).  Phase 1: one thread per row recomputes p = softmax(q k^T scale),
// dp = do v^T, ds = p (dp - sum p dp) scale and writes dq; p and ds stay in LDS.  Phase 2: all threads fold the
// 64 rows into the tile's partial dK[j][d] = sum_r ds[r][j] q[r][d], dV[j][d] = sum_r p[r][j] do[r][d]
// (fp32, [chunk][L][64] per (image, head)); xattn_bwd_fold_kernel sums the chunks in order and writes the 16-bit
// gradient of the layer's prompt K / V projection output.
struct XabP {
  const char* q; const char* k; const char* v; const char* dout; char* dq; float* part; char* dk; char* dv;
  int batch, heads, n_q, L, ldq, ldk, ldv, ldo, lddq, lddkv, chunks;
  long long q_bs, k_bs, v_bs, o_bs, dq_bs, dkv_bs;
  float scale;
};

template <typename T>
__global__ __launch_bounds__(256) void xattn_bwd_kernel(const XabP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_xb[];
  const int L = p.L, LP = L + 1;
  float* Ks = (float*)smem_xb;            // [L][64]
  float* Vs = Ks + (size_t)L * 64;        // [L][64]
  float* Qs = Vs + (size_t)L * 64;        // [64][65]
  float* Ds = Qs + 64 * 65;               // [64][65]
  float* Ps = Ds + 64 * 65;               // [64][LP]
  float* Ss = Ps + 64 * LP;               // [64][LP]
  const int chunk = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  for (int e = threadIdx.x; e < L * 8; e += 256) {
    const int j = e >> 3, c8 = e & 7;
    float f[8];
    unpack8<T>(*(const i32x4*)(p.k + ((size_t)b * p.k_bs + (size_t)j * p.ldk + head * 64 + c8 * 8) * sizeof(T)), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) Ks[j * 64 + c8 * 8 + i] = f[i];
    unpack8<T>(*(const i32x4*)(p.v + ((size_t)b * p.v_bs + (size_t)j * p.ldv + head * 64 + c8 * 8) * sizeof(T)), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) Vs[j * 64 + c8 * 8 + i] = f[i];
  }
  for (int e = threadIdx.x; e < 64 * 8; e += 256) {
    const int r = e >> 3, c8 = e & 7, row = chunk * 64 + r;
    float fq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < p.n_q) {
      unpack8<T>(*(const i32x4*)(p.q + ((size_t)b * p.q_bs + (size_t)row * p.ldq + head * 64 + c8 * 8) * sizeof(T)), fq);
      unpack8<T>(*(const i32x4*)(p.dout + ((size_t)b * p.o_bs + (size_t)row * p.ldo + head * 64 + c8 * 8) * sizeof(T)), fd);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { Qs[r * 65 + c8 * 8 + i] = fq[i]; Ds[r * 65 + c8 * 8 + i] = fd[i]; }
  }
  __syncthreads();
  {
    // phase 1: 4 threads per query row, each owning 16 of the 64 head dims: dot products are 16 FMAs + two lane
    // exchanges (the 4 threads are adjacent lanes), dq needs no exchange at all
    const int r = threadIdx.x >> 2, part = threadIdx.x & 3, d0 = part * 16, row = chunk * 64 + r;
    const bool rok = row < p.n_q;
    float qv[16], dv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { qv[i] = Qs[r * 65 + d0 + i]; dv[i] = Ds[r * 65 + d0 + i]; }
    float m = -1e30f;
    for (int j = 0; j < L; ++j) {
      float sd = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) sd += qv[i] * Ks[j * 64 + d0 + i];
      sd += __shfl_xor(sd, 1, 64);
      sd += __shfl_xor(sd, 2, 64);
      sd *= p.scale;
      if (part == 0) Ps[r * LP + j] = sd;
      m = fmaxf(m, sd);
    }
    float l = 0.f;
    for (int j = 0; j < L; ++j) l += __expf(Ps[r * LP + j] - m);     // every part reads the row written by part 0 (same wave)
    const float inv = 1.0f / l;
    float delta = 0.f;
    for (int j = 0; j < L; ++j) {
      float dp = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) dp += dv[i] * Vs[j * 64 + d0 + i];
      dp += __shfl_xor(dp, 1, 64);
      dp += __shfl_xor(dp, 2, 64);
      const float pr = __expf(Ps[r * LP + j] - m) * inv;
      if (part == 0) Ss[r * LP + j] = dp;
      delta += pr * dp;
    }
    float dq[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[i] = 0.f;
    for (int j = 0; j < L; ++j) {
      const float pr = rok ? __expf(Ps[r * LP + j] - m) * inv : 0.f;
      const float ds = pr * (Ss[r * LP + j] - delta) * p.scale;
#pragma unroll
      for (int i = 0; i < 16; ++i) dq[i] += ds * Ks[j * 64 + d0 + i];
      // the 4 threads of a row must all have read score / dp of key j before part 0 overwrites them: they are lanes
      // of one wave executing in lockstep, and the stores below come after this iteration's loads in program order
      if (part == 0) { Ps[r * LP + j] = pr; Ss[r * LP + j] = ds; }
    }
    if (rok) {
      char* op = p.dq + ((size_t)b * p.dq_bs + (size_t)row * p.lddq + head * 64 + d0) * sizeof(T);
      *(i32x4*)(op) = pack8<T>(dq);
      *(i32x4*)(op + 16) = pack8<T>(dq + 8);
    }
  }
  __syncthreads();
  // phase 2: thread = 5 keys x 4 head dims of the tile's partial dK / dV (11 LDS reads per 40 FMAs)
  float* out = p.part + ((((size_t)b * p.heads + head) * p.chunks + chunk) * 2) * (size_t)L * 64;
  {
    const int jb = (threadIdx.x >> 4) * 5, db = (threadIdx.x & 15) * 4;
    float ak[5][4], av[5][4];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) { ak[a][c] = 0.f; av[a][c] = 0.f; }
    for (int r = 0; r < 64; ++r) {
      float q4[4], d4[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) { q4[c] = Qs[r * 65 + db + c]; d4[c] = Ds[r * 65 + db + c]; }
#pragma unroll
      for (int a = 0; a < 5; ++a) {
        const int jj = jb + a < L ? jb + a : L - 1;        // keys past L: computed on a valid address, not stored
        const float ds = Ss[r * LP + jj], pr = Ps[r * LP + jj];
#pragma unroll
        for (int c = 0; c < 4; ++c) { ak[a][c] += ds * q4[c]; av[a][c] += pr * d4[c]; }
      }
    }
#pragma unroll
    for (int a = 0; a < 5; ++a)
      if (jb + a < L) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          out[(jb + a) * 64 + db + c] = ak[a][c];
          out[(size_t)L * 64 + (jb + a) * 64 + db + c] = av[a][c];
        }
      }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void xattn_bwd_fold_kernel(const XabP p) {
  const long long total = (long long)p.batch * p.heads * p.L * 64;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int d = (int)(e & 63);
    const long long t = e >> 6;
    const int j = (int)(t % p.L);
    const long long bh = t / p.L;
    const int head = (int)(bh % p.heads), b = (int)(bh / p.heads);
    float ak = 0.f, av = 0.f;
    for (int c = 0; c < p.chunks; ++c) {
      const float* o = p.part + (((size_t)bh * p.chunks + c) * 2) * (size_t)p.L * 64 + j * 64 + d;
      ak += o[0];
      av += o[(size_t)p.L * 64];
    }
    const size_t off = (size_t)b * p.dkv_bs + (size_t)j * p.lddkv + head * 64 + d;
    ((T*)p.dk)[off] = (T)ak;
    ((T*)p.dv)[off] = (T)av;
  }
}

// y = silu(a) (b == nullptr) or y = b * silu'(a): the SiLU of the timestep-embedding MLP (TimestepEmbedding.act and the
// F.silu in front of every time_emb_proj), a few thousand elements
template <typename T>
__global__ __launch_bounds__(256) void silu_kernel(const T* a, const T* b, T* y, long long n) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const float z = (float)a[e];
    const float sg = 1.0f / (1.0f + __expf(-z));
    y[e] = b ? (T)((float)b[e] * sg * (1.0f + z * (1.0f - sg))) : (T)(z * sg);
  }
}

}  // namespace dfw

using namespace dfw;

extern "C" size_t dfw_cross_attention_bwd_workspace_bytes(int32_t batch, int32_t heads, int32_t n_q, int32_t L) {
  if (batch <= 0 || heads <= 0 || n_q <= 0 || L <= 0) return 0;
  return (size_t)batch * heads * ((n_q + 63) / 64) * 2 * L * 64 * sizeof(float);
}

extern "C" int dfw_cross_attention_bwd(const dfw_xattn_bwd_args* a, dfw_stream_t stream) {
  if (!a || !a->q || !a->k || !a->v || !a->dout || !a->dq || !a->dk || !a->dv || !a->workspace) return DFW_EINVAL;
  if (a->batch <= 0 || a->heads <= 0 || a->n_q <= 0 || a->L <= 0) return DFW_EINVAL;
  if (a->dtype != DFW_BF16 && a->dtype != DFW_F16) return DFW_EINVAL;
  if ((a->ldq | a->ldk | a->ldv | a->ldo | a->lddq) % 8 != 0) return DFW_ESHAPE;
  if (a->L > 80) return DFW_ESHAPE;      // phase 2 tiles the keys as 16 x 5 (77 prompt tokens, T:1368)
  const size_t lds = ((size_t)a->L * 64 * 2 + 2 * 64 * 65 + 2 * 64 * (a->L + 1)) * sizeof(float);
  if (lds > 150 * 1024) return DFW_ESHAPE;
  const int chunks = (a->n_q + 63) / 64;
  if (a->workspace_bytes < dfw_cross_attention_bwd_workspace_bytes(a->batch, a->heads, a->n_q, a->L)) return DFW_EWORKSPACE;
  XabP p;
  p.q = (const char*)a->q; p.k = (const char*)a->k; p.v = (const char*)a->v; p.dout = (const char*)a->dout;
  p.dq = (char*)a->dq; p.part = (float*)a->workspace; p.dk = (char*)a->dk; p.dv = (char*)a->dv;
  p.batch = a->batch; p.heads = a->heads; p.n_q = a->n_q; p.L = a->L;
  p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo; p.lddq = a->lddq; p.lddkv = a->lddkv; p.chunks = chunks;
  p.q_bs = a->q_bs; p.k_bs = a->k_bs; p.v_bs = a->v_bs; p.o_bs = a->o_bs; p.dq_bs = a->dq_bs; p.dkv_bs = a->dkv_bs;
  p.scale = a->scale;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(chunks, a->heads, a->batch);
  const long long total = (long long)a->batch * a->heads * a->L * 64;
  int fg = (int)((total + 255) / 256);
  if (fg > 2048) fg = 2048;
  if (a->dtype == DFW_BF16) {
    auto kfn = xattn_bwd_kernel<__bf16>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, p);
    DFW_CHECK_LAUNCH();
    hipLaunchKernelGGL((xattn_bwd_fold_kernel<__bf16>), dim3(fg), dim3(256), 0, st, p);
  } else {
    auto kfn = xattn_bwd_kernel<_Float16>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, p);
    DFW_CHECK_LAUNCH();
    hipLaunchKernelGGL((xattn_bwd_fold_kernel<_Float16>), dim3(fg), dim3(256), 0, st, p);
  }
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_silu(const void* a, const void* dy, void* y, int64_t n, int32_t dtype, dfw_stream_t stream) {
  if (!a || !y || n <= 0) return DFW_EINVAL;
  if (dtype != DFW_BF16 && dtype != DFW_F16) return DFW_EINVAL;
  int g = (int)((n + 255) / 256);
  if (g > 2048) g = 2048;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DFW_BF16) hipLaunchKernelGGL((silu_kernel<__bf16>), dim3(g), dim3(256), 0, st, (const __bf16*)a, (const __bf16*)dy, (__bf16*)y, (long long)n);
  else hipLaunchKernelGGL((silu_kernel<_Float16>), dim3(g), dim3(256), 0, st, (const _Float16*)a, (const _Float16*)dy, (_Float16*)y, (long long)n);
  DFW_CHECK_LAUNCH();
  return 0;
}

static void launch_dkv(bool bf, dim3 grid, hipStream_t st, const dfw::FsaBwdP& p) {
  constexpr int kLds = 2 * (4 * 64 * 128 + 2 * 64 * 4);
  if (bf) {
    auto kfn = dfw::fsa_bwd_dkv_kernel<__bf16>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    hipLaunchKernelGGL(kfn, grid, dim3(256), kLds, st, p);
  } else {
    auto kfn = dfw::fsa_bwd_dkv_kernel<_Float16>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    hipLaunchKernelGGL(kfn, grid, dim3(256), kLds, st, p);
  }
}

// Same rule as the forward's fsa_split_count (attention.hip), for the dQ grid: 128-row workgroups, up to 3 per CU.
static int fsa_bwd_split_count(const dfw_fsa_bwd_args* a) {
  if (!cfg().fsa_key_split || a->nshot < 2) return 1;
  const int nq_img = a->batch - a->n_plain, nseg = 1 + a->nshot;
  if (nq_img <= 0 || (long long)a->n * nseg < 8192) return 1;
  const long long wg_per_img = (long long)a->heads * ((a->n + 127) / 128);
  const double total = (double)wg_per_img * ((double)a->n_plain + (double)nq_img * nseg);
  const double fair = total / 768.0 > 1.0 ? total / 768.0 : 1.0;
  if (cfg().fsa_force_splits) { const int f = cfg().fsa_force_splits; return f > nseg ? nseg : f; }   // sweeps
  if ((double)nseg <= 1.5 * fair) return 1;
  for (int ns = 2; ns <= nseg; ++ns)
    if ((double)((nseg + ns - 1) / ns) <= 1.25 * fair) return ns;
  return nseg;
}

extern "C" size_t dfw_fsa_attention_bwd_workspace_bytes(const dfw_fsa_bwd_args* a) {
  if (!a || a->batch <= 0 || a->heads <= 0 || a->n <= 0) return 0;
  const int ns = fsa_bwd_split_count(a);
  if (ns <= 1) return 0;
  return (size_t)(a->batch - a->n_plain) * ns * a->n * a->heads * 64 * sizeof(float);
}

extern "C" int dfw_fsa_attention_bwd(const dfw_fsa_bwd_args* a, dfw_stream_t stream) {
  if (!a || !a->qkv || !a->out || !a->dout || !a->lse || !a->delta || !a->dqkv) return DFW_EINVAL;
  if (a->batch <= 0 || a->heads <= 0 || a->n <= 0 || a->nshot < 0 || a->n_plain < 0 || a->n_plain > a->batch) return DFW_EINVAL;
  if (a->delta_bytes < 2ull * (size_t)a->batch * (size_t)a->heads * (size_t)a->n * sizeof(float)) return DFW_EWORKSPACE;
  if (a->nshot > 0 && (a->n_plain <= 0 || (a->batch - a->n_plain) * a->nshot != a->n_plain)) return DFW_EINVAL;
  if (a->dtype != DFW_BF16 && a->dtype != DFW_F16) return DFW_EINVAL;
  const int C = a->heads * 64;
  if ((a->ld % 8) || (a->ldo % 8) || (a->ldd % 8) || a->ld < 3 * C || a->ldd < 3 * C || a->ldo < C) return DFW_ESHAPE;
  const int64_t qe = (int64_t)a->batch * a->n * a->ld, oe = (int64_t)a->batch * a->n * a->ldo, de = (int64_t)a->batch * a->n * a->ldd;
  if (qe >= (1ll << 30) || oe >= (1ll << 30) || de >= (1ll << 30)) return DFW_ERANGE;
  hipStream_t st = (hipStream_t)stream;
  const bool bf = a->dtype == DFW_BF16;
  const size_t es = 2;
  // delta = rowsum(dO o O)
  {
    const long long total = (long long)a->batch * a->heads * a->n;
    int g = (int)((total + 31) / 32);
    if (g > 8192) g = 8192;
    if (bf) hipLaunchKernelGGL((fsa_delta_kernel<__bf16>), dim3(g), dim3(256), 0, st, (const char*)a->out, (const char*)a->dout, a->lse, a->delta, a->batch, a->heads, a->n, a->ldo, (long long)a->n * a->ldo, a->ldo, (long long)a->n * a->ldo);
    else hipLaunchKernelGGL((fsa_delta_kernel<_Float16>), dim3(g), dim3(256), 0, st, (const char*)a->out, (const char*)a->dout, a->lse, a->delta, a->batch, a->heads, a->n, a->ldo, (long long)a->n * a->ldo, a->ldo, (long long)a->n * a->ldo);
    DFW_CHECK_LAUNCH();
  }
  FsaBwdP p;
  p.q = (const char*)a->qkv; p.k = p.q + (size_t)C * es; p.v = p.q + (size_t)2 * C * es;
  p.dout = (const char*)a->dout; p.lse = a->lse; p.delta = a->delta;
  p.stat = a->delta; p.stat_half = (long long)a->batch * a->heads * a->n;
  if (p.stat_half * 8 >= (1ll << 31)) return DFW_ERANGE;
  p.stat_bytes = (uint32_t)(p.stat_half * 8);
  p.dq = (char*)a->dqkv; p.dk = p.dq + (size_t)C * es; p.dv = p.dq + (size_t)2 * C * es;
  p.qkv_bytes = (uint32_t)(qe * es); p.do_bytes = (uint32_t)(oe * es); p.dqkv_bytes = (uint32_t)(de * es);
  p.batch = a->batch; p.heads = a->heads; p.n = a->n; p.nshot = a->nshot; p.n_plain = a->n_plain;
  p.ld = a->ld; p.ldo = a->ldo; p.ldd = a->ldd;
  p.bs = (long long)a->n * a->ld; p.obs = (long long)a->n * a->ldo; p.dbs = (long long)a->n * a->ldd;
  p.scale = a->scale;
  p.n_kv = a->n; p.ldkv = a->ld; p.lddkv = a->ldd; p.kvbs = p.bs; p.dkvbs = p.dbs;
  p.kv_bytes = p.qkv_bytes - (uint32_t)(C * es); p.voff = (uint32_t)(C * es);
  p.nsplit = 1; p.part = nullptr; p.qsplit = 1; p.kvpart = nullptr;
  {
    const int ns = fsa_bwd_split_count(a);
    if (ns > 1 && a->workspace && a->workspace_bytes >= dfw_fsa_attention_bwd_workspace_bytes(a) && (((uintptr_t)a->workspace) & 15) == 0) {
      p.nsplit = ns;
      p.part = (float*)a->workspace;
    }
  }
  dim3 grid((a->n + 127) / 128, a->heads, a->batch);
  dim3 gridq((a->n + 127) / 128, a->heads, a->n_plain + (a->batch - a->n_plain) * p.nsplit);
  if (bf) hipLaunchKernelGGL((fsa_bwd_dq_kernel<__bf16>), gridq, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((fsa_bwd_dq_kernel<_Float16>), gridq, dim3(256), 0, st, p);
  DFW_CHECK_LAUNCH();
  if (p.nsplit > 1) {
    const long long threads = (long long)(a->batch - a->n_plain) * a->n * (C / 4);
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    if (bf) hipLaunchKernelGGL((fsa_dq_combine_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((fsa_dq_combine_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, p);
    DFW_CHECK_LAUNCH();
  }
  launch_dkv(bf, grid, st, p);
  DFW_CHECK_LAUNCH();
  return 0;
}

// Query split of the dK/dV kernel for a short key axis: as many instances per key block as it takes to put about two
// workgroups on every CU, each walking at least two query tiles.  1 when the launch fills the chip by itself.
static int attn_bwd_qsplit(const dfw_attn_bwd_args* a) {
  const long long blocks = (long long)((a->n_kv + 127) / 128) * a->heads * a->batch;
  const int tiles = (a->n_q + 63) / 64;
  if (blocks >= 256 || tiles < 4) return 1;
  long long qs = (512 + blocks - 1) / blocks;
  if (qs > tiles / 2) qs = tiles / 2;
  return qs < 1 ? 1 : (int)qs;
}

extern "C" size_t dfw_attention_bwd_workspace_bytes(const dfw_attn_bwd_args* a) {
  if (!a || a->batch <= 0 || a->heads <= 0 || a->n_q <= 0 || a->n_kv <= 0) return 0;
  const int qs = attn_bwd_qsplit(a);
  if (qs <= 1) return 0;
  return (size_t)2 * qs * a->batch * a->n_kv * a->heads * 64 * sizeof(float);
}

// General form: queries and keys / values in their own tensors (attn2 of the training step on the MFMA path: the 77 prompt
// tokens are two 64-key tiles, the ragged one masked).  Same kernels, same conventions (q pre-scaled, lse from the forward).
extern "C" int dfw_attention_bwd(const dfw_attn_bwd_args* a, dfw_stream_t stream) {
  if (!a || !a->q || !a->k || !a->v || !a->out || !a->dout || !a->lse || !a->delta || !a->dq || !a->dk || !a->dv) return DFW_EINVAL;
  if (a->batch <= 0 || a->heads <= 0 || a->n_q <= 0 || a->n_kv <= 0) return DFW_EINVAL;
  if (a->delta_bytes < 2ull * (size_t)a->batch * (size_t)a->heads * (size_t)a->n_q * sizeof(float)) return DFW_EWORKSPACE;
  if (a->dtype != DFW_BF16 && a->dtype != DFW_F16) return DFW_EINVAL;
  const int C = a->heads * 64;
  if ((a->ldq | a->ldkv | a->ldo | a->lddq | a->lddkv) % 8 != 0) return DFW_ESHAPE;
  if (a->ldq < C || a->ldkv < C || a->ldo < C || a->lddq < C || a->lddkv < C) return DFW_ESHAPE;
  const char* kc = (const char*)a->k;
  const char* vc = (const char*)a->v;
  if (vc < kc || ((vc - kc) & 15)) return DFW_ESHAPE;        // v rides in k's buffer descriptor at a 16-byte-aligned offset
  const size_t es = 2;
  auto extent = [&](int64_t bs, int n, int ld) { return (int64_t)(a->batch - 1) * bs + (int64_t)(n - 1) * ld + C; };
  const int64_t qe = extent(a->q_bs, a->n_q, a->ldq), oe = extent(a->o_bs, a->n_q, a->ldo);
  const int64_t ke = extent(a->kv_bs, a->n_kv, a->ldkv) + (int64_t)((vc - kc) / es);
  if (qe >= (1ll << 30) || oe >= (1ll << 30) || ke >= (1ll << 30)) return DFW_ERANGE;
  hipStream_t st = (hipStream_t)stream;
  const bool bf = a->dtype == DFW_BF16;
  {
    const long long total = (long long)a->batch * a->heads * a->n_q;
    int g = (int)((total + 31) / 32);
    if (g > 8192) g = 8192;
    if (bf) hipLaunchKernelGGL((fsa_delta_kernel<__bf16>), dim3(g), dim3(256), 0, st, (const char*)a->out, (const char*)a->dout, a->lse, a->delta, a->batch, a->heads, a->n_q, a->ldo, (long long)a->o_bs, a->ldo, (long long)a->o_bs);
    else hipLaunchKernelGGL((fsa_delta_kernel<_Float16>), dim3(g), dim3(256), 0, st, (const char*)a->out, (const char*)a->dout, a->lse, a->delta, a->batch, a->heads, a->n_q, a->ldo, (long long)a->o_bs, a->ldo, (long long)a->o_bs);
    DFW_CHECK_LAUNCH();
  }
  FsaBwdP p;
  p.q = (const char*)a->q; p.k = kc; p.v = vc;
  p.dout = (const char*)a->dout; p.lse = a->lse; p.delta = a->delta;
  p.stat = a->delta; p.stat_half = (long long)a->batch * a->heads * a->n_q;
  if (p.stat_half * 8 >= (1ll << 31)) return DFW_ERANGE;
  p.stat_bytes = (uint32_t)(p.stat_half * 8);
  p.dq = (char*)a->dq; p.dk = (char*)a->dk; p.dv = (char*)a->dv;
  p.qkv_bytes = (uint32_t)(qe * es); p.do_bytes = (uint32_t)(oe * es); p.dqkv_bytes = 0;
  p.batch = a->batch; p.heads = a->heads; p.n = a->n_q; p.nshot = 0; p.n_plain = a->batch;
  p.ld = a->ldq; p.ldo = a->ldo; p.ldd = a->lddq;
  p.bs = a->q_bs; p.obs = a->o_bs; p.dbs = a->dq_bs;
  p.scale = a->scale;
  p.n_kv = a->n_kv; p.ldkv = a->ldkv; p.lddkv = a->lddkv; p.kvbs = a->kv_bs; p.dkvbs = a->dkv_bs;
  p.kv_bytes = (uint32_t)(ke * es); p.voff = (uint32_t)(vc - kc);
  p.nsplit = 1; p.part = nullptr; p.qsplit = 1; p.kvpart = nullptr;
  {
    const int qs = attn_bwd_qsplit(a);
    if (qs > 1 && a->workspace && a->workspace_bytes >= dfw_attention_bwd_workspace_bytes(a) && (((uintptr_t)a->workspace) & 15) == 0) {
      p.qsplit = qs;
      p.kvpart = (float*)a->workspace;
    }
  }
  dim3 gq((a->n_q + 127) / 128, a->heads, a->batch), gk((a->n_kv + 127) / 128 * p.qsplit, a->heads, a->batch);
  if (bf) hipLaunchKernelGGL((fsa_bwd_dq_kernel<__bf16>), gq, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((fsa_bwd_dq_kernel<_Float16>), gq, dim3(256), 0, st, p);
  DFW_CHECK_LAUNCH();
  launch_dkv(bf, gk, st, p);
  DFW_CHECK_LAUNCH();
  if (p.qsplit > 1) {
    const long long total = (long long)a->batch * a->n_kv * (C / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (bf) hipLaunchKernelGGL((fsa_dkv_fold_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((fsa_dkv_fold_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, p);
    DFW_CHECK_LAUNCH();
  }
  return 0;
}
