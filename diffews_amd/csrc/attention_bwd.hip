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

#ifndef DFW_FSA_BWD_PRIO
#define DFW_FSA_BWD_PRIO 1
#endif
#if DFW_FSA_BWD_PRIO
#define DFW_BWD_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define DFW_BWD_PRIO(x) ((void)0)
#endif

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
};

__device__ __forceinline__ uint32_t row_off(int row, int chunk) {   // K-style image: b128 row reads
  return (uint32_t)(row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}
__device__ __forceinline__ uint32_t tr_off(int row, int dcol) {     // V-style image: transposed reads
  return (uint32_t)(row * 128 + ((((dcol >> 3) ^ (((row >> 1) & 1) << 2))) << 4) + ((dcol & 7) << 1));
}

// delta[b][h][q] = sum_d dO * O   (fp32)
template <typename T>
__global__ __launch_bounds__(256) void fsa_delta_kernel(const char* o, const char* dout, float* delta, int batch, int heads,
                                                        int n, int ldo, long long obs, int ldd, long long dbs) {
  const long long total = (long long)batch * heads * n;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int qi = (int)(e % n);
    const long long bh = e / n;
    const int h = (int)(bh % heads), b = (int)(bh / heads);
    const char* po = o + ((size_t)b * obs + (size_t)qi * ldo + h * 64) * sizeof(T);
    const char* pd = dout + ((size_t)b * dbs + (size_t)qi * ldd + h * 64) * sizeof(T);
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float a[8], d[8];
      unpack8<T>(*(const i32x4*)(po + c * 16), a);
      unpack8<T>(*(const i32x4*)(pd + c * 16), d);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += a[i] * d[i];
    }
    delta[e] = acc;
  }
}

// ------------------------------------------------------------------------------------------------ dQ
template <typename T>
__global__ __launch_bounds__(256) void fsa_bwd_dq_kernel(const FsaBwdP p) {
  constexpr int KT = 64, TILE = KT * 128;
  __shared__ __attribute__((aligned(16))) char smem[2 * 3 * TILE];   // [buf][K rows | K tr | V rows]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  const __amdgpu_buffer_rsrc_t rkv = make_rsrc(p.k, p.kv_bytes);
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
  const float nlse = qok ? -p.lse[stat] : -INFINITY;     // invalid rows: P = exp2(s + (-inf)) = 0... (see below)
  const float ndelta = qok ? -p.delta[stat] : 0.f;

  const int srow0 = tid >> 3, sc = tid & 7;
  const int tiles_own = (p.n_kv + KT - 1) / KT;
  const int tiles_bank = (p.nshot > 0 && bank_b >= 0) ? tiles_own : 0;
  const int ntiles = tiles_own + (tiles_bank ? p.nshot * tiles_bank : 0);
  // this instance's key segments [seg0, seg1) of [own ; shot 0 ; ...] as a range of the global tile numbering
  const int nseg = 1 + (tiles_bank ? p.nshot : 0);
  const bool parted = p.nsplit > 1 && bank_b >= 0;
  const int seg0 = parted ? split * nseg / p.nsplit : 0, seg1 = parted ? (split + 1) * nseg / p.nsplit : nseg;
  const int t_begin = seg0 == 0 ? 0 : tiles_own + (seg0 - 1) * tiles_bank;
  const int t_end = parted ? tiles_own + (seg1 - 1) * tiles_bank : ntiles;
  i32x4 gk[2], gv[2];
  auto issue = [&](int t) {
    int img = b, tt = t;
    if (t >= tiles_own) { img = bank_b * p.nshot + (t - tiles_own) / tiles_bank; tt = (t - tiles_own) % tiles_bank; }
    const size_t base = (size_t)img * p.kvbs + head * 64 + sc * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = tt * KT + srow0 + 32 * i;
      const uint32_t o = key < p.n_kv ? (uint32_t)((base + (size_t)key * p.ldkv) * sizeof(T)) : kOOB;
      gk[i] = buf_load16(rkv, o);
      gv[i] = buf_load16(rkv, o == kOOB ? kOOB : o + voff);
    }
  };
  auto write_lds = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = srow0 + 32 * i;
      *(i32x4*)(buf + row_off(row, sc)) = gk[i];
      *(i32x4*)(buf + TILE + tr_off(row, sc * 8)) = gk[i];
      *(i32x4*)(buf + 2 * TILE + row_off(row, sc)) = gv[i];
    }
  };
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  f32x16 o[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;

  issue(t_begin);
  write_lds(smem);
  __syncthreads();
  int cur = 0;
  for (int t = t_begin; t < t_end; ++t) {
    const bool more = t + 1 < t_end;
    if (more) issue(t + 1);
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
      for (int r = 0; r < 16; ++r) { s[kb][r] = qok ? nlse : 0.f; dp[kb][r] = ndelta; }
    DFW_BWD_PRIO(1);           // the MFMA chains outrank the other waves' softmax VALU (forward: +9 %)
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
    DFW_BWD_PRIO(0);
    // dS^T = P^T o dP^T ; rows of S^T are keys: (r & 3) + 8 (r >> 2) + 4 lh within the 32-key block
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float pr = (qok && key < nvalid) ? __builtin_amdgcn_exp2f(s[kb][r]) : 0.f;
        s[kb][r] = pr * dp[kb][r];
      }
    // dQ^T += K^T dS^T
    DFW_BWD_PRIO(1);
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
    DFW_BWD_PRIO(0);
    if (more) write_lds(smem + (cur ^ 1) * 3 * TILE);
    __syncthreads();
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
template <typename T>
__global__ __launch_bounds__(256) void fsa_bwd_dkv_kernel(const FsaBwdP p) {
  constexpr int QT = 64, TILE = QT * 128;
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE];   // [Q rows | Q tr | dO rows | dO tr] (32 KiB, single buffer)
  __shared__ float stats[2][QT];                                  // [-lse | -delta][q]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y, kimg = blockIdx.z;
  const int key = blockIdx.x * 128 + wave * 32 + lr;      // the key this lane owns (column of S)
  const bool kok = key < p.n_kv;
  const __amdgpu_buffer_rsrc_t rqkv = make_rsrc(p.q, p.qkv_bytes);
  const __amdgpu_buffer_rsrc_t rkv = make_rsrc(p.k, p.kv_bytes);
  const __amdgpu_buffer_rsrc_t rdo = make_rsrc(p.dout, p.do_bytes);
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
  const int ntiles = nsrc * tiles_q;

  const int srow0 = tid >> 3, sc = tid & 7;
  i32x4 gq[2], gd[2];
  float gst = 0.f;
  auto issue = [&](int t) {
    const int img = t < tiles_q ? kimg : src1, tt = t < tiles_q ? t : t - tiles_q;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int qi = tt * QT + srow0 + 32 * i;
      const bool ok = qi < p.n;
      gq[i] = buf_load16(rqkv, ok ? (uint32_t)(((size_t)img * p.bs + (size_t)qi * p.ld + head * 64 + sc * 8) * sizeof(T)) : kOOB);
      gd[i] = buf_load16(rdo, ok ? (uint32_t)(((size_t)img * p.obs + (size_t)qi * p.ldo + head * 64 + sc * 8) * sizeof(T)) : kOOB);
    }
    if (tid < 2 * QT) {       // threads 0..63: -lse, 64..127: -delta of the tile's rows
      const int qi = tt * QT + (tid & 63);
      const size_t st = ((size_t)img * p.heads + head) * p.n + qi;
      if (qi < p.n) gst = tid < QT ? -p.lse[st] : -p.delta[st];
      else gst = tid < QT ? -INFINITY : 0.f;        // rows past the end: P = exp2(-inf) = 0
    }
  };
  auto write_lds = [&]() {
    char* base = smem;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = srow0 + 32 * i;
      *(i32x4*)(base + row_off(row, sc)) = gq[i];
      *(i32x4*)(base + TILE + tr_off(row, sc * 8)) = gq[i];
      *(i32x4*)(base + 2 * TILE + row_off(row, sc)) = gd[i];
      *(i32x4*)(base + 3 * TILE + tr_off(row, sc * 8)) = gd[i];
    }
    if (tid < 2 * QT) stats[tid >> 6][tid & 63] = gst;
  };
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  f32x16 dk[2], dv[2];     // dK^T / dV^T [d block][rows d, col = key]
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }

  issue(0);
  write_lds();
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const bool more = t + 1 < ntiles;
    if (more) issue(t + 1);          // next tile's global loads fly during this tile's MFMAs
    const char* qrow_t = smem;
    const char* qtr = qrow_t + TILE;
    const char* drow = qrow_t + 2 * TILE;
    const char* dtr = qrow_t + 3 * TILE;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      // S = Q K^T - lse ; dP = dO V^T - delta   (rows = queries: (r & 3) + 8 (r >> 2) + 4 lh of the 32-row block)
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        s[r] = stats[0][qi];
        dp[r] = stats[1][qi];
      }
      DFW_BWD_PRIO(1);
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        const int row = qb * 32 + lr;
        const typename Tr<T>::v8 qa = as_v8<T>(*(const i32x4*)(qrow_t + row_off(row, 2 * ss + lh)));
        const typename Tr<T>::v8 da = as_v8<T>(*(const i32x4*)(drow + row_off(row, 2 * ss + lh)));
        s = Tr<T>::mfma(qa, kf[ss], s);
        dp = Tr<T>::mfma(da, vf[ss], dp);
      }
      DFW_BWD_PRIO(0);
      typename Tr<T>::v8 pf[2], sf[2];       // P and dS as B operands: k-step t2 <- registers 8 t2 .. 8 t2 + 7
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = kok ? __builtin_amdgcn_exp2f(s[r]) : 0.f;
        pf[r >> 3][r & 7] = (T)pr;
        sf[r >> 3][r & 7] = (T)(pr * dp[r]);
      }
      // dV^T += dO^T P ; dK^T += Q^T dS   (A fragments: transposed reads, rows in the accumulator's k order)
      DFW_BWD_PRIO(1);
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
          dv[d] = Tr<T>::mfma(da, pf[t2], dv[d]);
          dk[d] = Tr<T>::mfma(qa, sf[t2], dk[d]);
        }
      }
      DFW_BWD_PRIO(0);
    }
    __syncthreads();                  // every wave is done reading this tile
    if (more) write_lds();
    __syncthreads();
  }
  if (kok) {
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
    int g = (int)((total + 255) / 256);
    if (g > 4096) g = 4096;
    if (bf) hipLaunchKernelGGL((fsa_delta_kernel<__bf16>), dim3(g), dim3(256), 0, st, (const char*)a->out, (const char*)a->dout, a->delta, a->batch, a->heads, a->n, a->ldo, (long long)a->n * a->ldo, a->ldo, (long long)a->n * a->ldo);
    else hipLaunchKernelGGL((fsa_delta_kernel<_Float16>), dim3(g), dim3(256), 0, st, (const char*)a->out, (const char*)a->dout, a->delta, a->batch, a->heads, a->n, a->ldo, (long long)a->n * a->ldo, a->ldo, (long long)a->n * a->ldo);
    DFW_CHECK_LAUNCH();
  }
  FsaBwdP p;
  p.q = (const char*)a->qkv; p.k = p.q + (size_t)C * es; p.v = p.q + (size_t)2 * C * es;
  p.dout = (const char*)a->dout; p.lse = a->lse; p.delta = a->delta;
  p.dq = (char*)a->dqkv; p.dk = p.dq + (size_t)C * es; p.dv = p.dq + (size_t)2 * C * es;
  p.qkv_bytes = (uint32_t)(qe * es); p.do_bytes = (uint32_t)(oe * es); p.dqkv_bytes = (uint32_t)(de * es);
  p.batch = a->batch; p.heads = a->heads; p.n = a->n; p.nshot = a->nshot; p.n_plain = a->n_plain;
  p.ld = a->ld; p.ldo = a->ldo; p.ldd = a->ldd;
  p.bs = (long long)a->n * a->ld; p.obs = (long long)a->n * a->ldo; p.dbs = (long long)a->n * a->ldd;
  p.scale = a->scale;
  p.n_kv = a->n; p.ldkv = a->ld; p.lddkv = a->ldd; p.kvbs = p.bs; p.dkvbs = p.dbs;
  p.kv_bytes = p.qkv_bytes - (uint32_t)(C * es); p.voff = (uint32_t)(C * es);
  p.nsplit = 1; p.part = nullptr;
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
  if (bf) hipLaunchKernelGGL((fsa_bwd_dkv_kernel<__bf16>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((fsa_bwd_dkv_kernel<_Float16>), grid, dim3(256), 0, st, p);
  DFW_CHECK_LAUNCH();
  return 0;
}

// General form: queries and keys / values in their own tensors (attn2 of the training step on the MFMA path: the 77 prompt
// tokens are two 64-key tiles, the ragged one masked).  Same kernels, same conventions (q pre-scaled, lse from the forward).
extern "C" int dfw_attention_bwd(const dfw_attn_bwd_args* a, dfw_stream_t stream) {
  if (!a || !a->q || !a->k || !a->v || !a->out || !a->dout || !a->lse || !a->delta || !a->dq || !a->dk || !a->dv) return DFW_EINVAL;
  if (a->batch <= 0 || a->heads <= 0 || a->n_q <= 0 || a->n_kv <= 0) return DFW_EINVAL;
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
    int g = (int)((total + 255) / 256);
    if (g > 4096) g = 4096;
    if (bf) hipLaunchKernelGGL((fsa_delta_kernel<__bf16>), dim3(g), dim3(256), 0, st, (const char*)a->out, (const char*)a->dout, a->delta, a->batch, a->heads, a->n_q, a->ldo, (long long)a->o_bs, a->ldo, (long long)a->o_bs);
    else hipLaunchKernelGGL((fsa_delta_kernel<_Float16>), dim3(g), dim3(256), 0, st, (const char*)a->out, (const char*)a->dout, a->delta, a->batch, a->heads, a->n_q, a->ldo, (long long)a->o_bs, a->ldo, (long long)a->o_bs);
    DFW_CHECK_LAUNCH();
  }
  FsaBwdP p;
  p.q = (const char*)a->q; p.k = kc; p.v = vc;
  p.dout = (const char*)a->dout; p.lse = a->lse; p.delta = a->delta;
  p.dq = (char*)a->dq; p.dk = (char*)a->dk; p.dv = (char*)a->dv;
  p.qkv_bytes = (uint32_t)(qe * es); p.do_bytes = (uint32_t)(oe * es); p.dqkv_bytes = 0;
  p.batch = a->batch; p.heads = a->heads; p.n = a->n_q; p.nshot = 0; p.n_plain = a->batch;
  p.ld = a->ldq; p.ldo = a->ldo; p.ldd = a->lddq;
  p.bs = a->q_bs; p.obs = a->o_bs; p.dbs = a->dq_bs;
  p.scale = a->scale;
  p.n_kv = a->n_kv; p.ldkv = a->ldkv; p.lddkv = a->lddkv; p.kvbs = a->kv_bs; p.dkvbs = a->dkv_bs;
  p.kv_bytes = (uint32_t)(ke * es); p.voff = (uint32_t)(vc - kc);
  p.nsplit = 1; p.part = nullptr;
  dim3 gq((a->n_q + 127) / 128, a->heads, a->batch), gk((a->n_kv + 127) / 128, a->heads, a->batch);
  if (bf) hipLaunchKernelGGL((fsa_bwd_dq_kernel<__bf16>), gq, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((fsa_bwd_dq_kernel<_Float16>), gq, dim3(256), 0, st, p);
  DFW_CHECK_LAUNCH();
  if (bf) hipLaunchKernelGGL((fsa_bwd_dkv_kernel<__bf16>), gk, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((fsa_bwd_dkv_kernel<_Float16>), gk, dim3(256), 0, st, p);
  DFW_CHECK_LAUNCH();
  return 0;
}
