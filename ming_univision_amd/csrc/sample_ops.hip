// sample_ops.hip — next-token sampling on one row of fp32 logits: the warpers HF `GenerationMixin.generate` applies when the
// checkpoint's generation config asks for `do_sample` (the reference forwards **generate_kwargs to it, modeling_bailingmm.py:249-262;
// transformers/generation/logits_process.py: TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper -> softmax -> multinomial).
//
//   scores = logits / temperature
//   top-k:  keep every score >= the k-th largest one (ties at the threshold stay, as `scores < topk[-1]` removes only smaller ones)
//   top-p:  softmax over what top-k kept; a token stays iff the probability mass of the tokens ranked ABOVE it is < top_p
//           (the ascending-cumsum form `cumulative_probs <= 1 - top_p` of the reference implementation; the best token always stays)
//   draw:   inverse CDF over the kept tokens in descending-score order (ties: lower id first) at the caller's uniform u in [0, 1)
//           — the caller owns the random stream, so a run is reproducible and testable (torch.multinomial's stream is not portable).
//
// One 1024-thread workgroup per row.  The k-th largest score is found exactly by a three-pass radix select over order-preserving
// 32-bit keys (11 + 11 + 10 bits, LDS histograms, the 126 k logits stay in L2), the survivors (<= 2048) are sorted by a bitonic
// network in LDS.  top_k = 0 with top_p < 1: the nucleus is cut at top_p of the FULL vocabulary's softmax mass (one more pass over the
// row) and searched among the 2048 largest scores — HF's result exactly whenever its nucleus holds <= 2048 tokens; a larger nucleus is
// truncated to the 2048 best and reported (status bit MN_SAMPLE_NUCLEUS_TRUNCATED).  top_k > 2048 is refused by the entry point.
// Ties at the k-th score that do not fit the 2048 candidates are kept in ascending id order (deterministic, status bit
// MN_SAMPLE_TIES_TRUNCATED).  top_k = 0 with top_p >= 1 (pure temperature sampling) draws from the FULL vocabulary in id order.
#include "common.h"

namespace {

constexpr int CAP = 2048;     // candidate capacity (two per thread)

__device__ __forceinline__ uint32_t okey(float f) {           // order-preserving: a < b  <=>  okey(a) < okey(b)
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// inclusive block scan (1024 threads) of one float per thread
__device__ __forceinline__ float block_scan_f(float v, float* wsum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  __syncthreads();
  if (lane == 63) wsum[wave] = v;
  __syncthreads();
  float base = 0.f;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  return v + base;
}
__device__ __forceinline__ int block_scan_i(int v, int* wsum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  __syncthreads();
  if (lane == 63) wsum[wave] = v;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  return v + base;
}

__global__ __launch_bounds__(1024) void sample_logits_kernel(const float* __restrict__ logits, int64_t ld, int V, float inv_temp, int top_k,
                                                             float top_p, const float* __restrict__ u, int64_t vocab_offset,
                                                             int64_t* __restrict__ idx, int32_t* __restrict__ status) {
  __shared__ int hist[2048];
  __shared__ float sval[CAP];
  __shared__ int sidx[CAP];
  __shared__ float red[32];
  __shared__ float wsf[16];
  __shared__ float cdf[1024];
  __shared__ int wsi[16];
  __shared__ int s_bin, s_need, s_cnt, s_pick;
  const int tid = threadIdx.x, m = blockIdx.x;
  const float* row = logits + (int64_t)m * ld;
  const float uu = u[m];
  float mx = -INFINITY;
  for (int j = tid; j < V; j += 1024) mx = fmaxf(mx, row[j]);
  mx = block_max(mx, red);
  if (status && tid == 0) status[m] = 0;

  if (top_k <= 0 && top_p >= 1.0f) {
    // ---- pure temperature sampling over the whole vocabulary, inverse CDF in id order: contiguous chunk per thread
    const int per = (V + 1023) / 1024, j0 = tid * per, j1 = min(V, j0 + per);
    float loc = 0.f;
    for (int j = j0; j < j1; ++j) loc += __expf((row[j] - mx) * inv_temp);
    const float incl = block_scan_f(loc, wsf);
    cdf[tid] = incl;
    if (tid == 0) s_pick = 0x7fffffff;
    __syncthreads();
    const float excl = tid ? cdf[tid - 1] : 0.f;                   // the previous thread's inclusive sum: the intervals tile [0, total)
    const float target = uu * cdf[1023];
    if (excl <= target && target < incl && j1 > j0) {
      float acc = excl;
      int pick = j1 - 1;
      for (int j = j0; j < j1; ++j) {
        acc += __expf((row[j] - mx) * inv_temp);
        if (acc > target) { pick = j; break; }
      }
      atomicMin(&s_pick, pick);
    }
    __syncthreads();
    if (tid == 0 && s_pick == 0x7fffffff) s_pick = V - 1;         // target rounded up to the total (u within 2^-24 of 1)
    __syncthreads();
    if (tid == 0) idx[m] = (int64_t)s_pick + vocab_offset;
    return;
  }

  // top-p without top-k: HF's softmax runs over the whole vocabulary, so the nucleus limit is top_p of the FULL mass
  float z_full = 0.f;
  if (top_k <= 0) {
    for (int j = tid; j < V; j += 1024) z_full += __expf((row[j] - mx) * inv_temp);
    z_full = block_sum(z_full, red);
  }
  // ---- the K-th largest key, exactly: radix select from the top bits down
  int K = top_k > 0 ? min(top_k, CAP) : CAP;
  K = min(K, V);
  uint32_t prefix = 0, mask = 0;
  int need = K;
  const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
  for (int pass = 0; pass < 3; ++pass) {
    const int sh = shifts[pass], nb = 1 << nbits[pass];
    for (int b = tid; b < 2048; b += 1024) hist[b] = 0;
    __syncthreads();
    for (int j = tid; j < V; j += 1024) {
      const uint32_t key = okey(row[j]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> sh) & (nb - 1)], 1);
    }
    __syncthreads();
    // suffix counts from the top bin: thread t owns bins nb-1-2t and nb-2-2t
    const int b0 = nb - 1 - 2 * tid, b1 = b0 - 1;
    const int c0 = b0 >= 0 ? hist[b0] : 0, c1 = b1 >= 0 ? hist[b1] : 0;
    const int incl = block_scan_i(c0 + c1, wsi);
    const int before = incl - c0 - c1;                            // elements in bins above b0
    if (before < need && incl >= need) {                          // the need-th largest lies in b0 or b1
      if (before + c0 >= need) { s_bin = b0; s_need = need - before; }
      else { s_bin = b1; s_need = need - before - c0; }
    }
    __syncthreads();
    prefix |= (uint32_t)s_bin << sh;
    mask |= (uint32_t)(nb - 1) << sh;
    need = s_need;
    __syncthreads();
  }
  const uint32_t thr = prefix;                                     // key of the K-th largest score
  // ---- survivors: every score > the K-th largest in any order (fewer than K <= CAP of them; the sort below is a total order), then
  // the ties AT the K-th score in ascending id order while they fit — which ones stay never depends on atomics' arrival order
  if (tid == 0) s_cnt = 0;
  for (int i = tid; i < CAP; i += 1024) { sval[i] = -INFINITY; sidx[i] = 0x7fffffff; }
  __syncthreads();
  const int per_t = (V + 1023) / 1024, t0 = tid * per_t, t1 = min(V, t0 + per_t);
  int my_ties = 0;
  for (int j = tid; j < V; j += 1024) {
    const float v = row[j];
    if (okey(v) > thr) {
      const int p = atomicAdd(&s_cnt, 1);
      sval[p] = v; sidx[p] = j;
    }
  }
  for (int j = t0; j < t1; ++j) my_ties += okey(row[j]) == thr;
  const int ties_incl = block_scan_i(my_ties, wsi);               // (its barriers also publish s_cnt)
  {
    int p = s_cnt + ties_incl - my_ties;                          // contiguous id chunk per thread: ties land in id order
    for (int j = t0; j < t1 && p < CAP; ++j)
      if (okey(row[j]) == thr) { sval[p] = row[j]; sidx[p] = j; ++p; }
  }
  __shared__ int s_total;
  if (tid == 1023) s_total = s_cnt + ties_incl;
  __syncthreads();
  const int n = min(s_total, CAP);
  if (status && tid == 0 && top_k > 0 && s_total > CAP) atomicOr(&status[m], 2);   // (top_k = 0: only a truncated nucleus matters, below)
  // ---- bitonic sort, descending by score, ties by ascending id (a total order: the result does not depend on arrival order)
  for (int k2 = 2; k2 <= CAP; k2 <<= 1) {
    for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
      for (int i = tid; i < CAP; i += 1024) {
        const int l = i ^ j2;
        if (l > i) {
          const float a = sval[i], b = sval[l];
          const int ia = sidx[i], ib = sidx[l];
          const bool a_first = a > b || (a == b && ia < ib);       // a ranks before b
          const bool up = (i & k2) == 0;                          // this pair's direction: "first" element at the lower index
          if (up != a_first) { sval[i] = b; sval[l] = a; sidx[i] = ib; sidx[l] = ia; }
        }
      }
      __syncthreads();
    }
  }
  // ---- weights, top-p prefix, inverse CDF (two consecutive ranks per thread)
  const int r0 = 2 * tid, r1 = r0 + 1;
  const float w0 = r0 < n ? __expf((sval[r0] - mx) * inv_temp) : 0.f, w1 = r1 < n ? __expf((sval[r1] - mx) * inv_temp) : 0.f;
  const float incl = block_scan_f(w0 + w1, wsf);
  cdf[tid] = incl;
  if (tid == 0) { s_pick = 0x7fffffff; s_need = 0; }
  __syncthreads();
  // mass ranked above r0 / r1 / r1 + 1; the previous thread's inclusive sum is this thread's start, so the intervals tile [0, zk)
  const float ex0 = tid ? cdf[tid - 1] : 0.f, ex1 = ex0 + w0, ex2 = incl;
  const float lim = top_p * (top_k > 0 ? cdf[1023] : z_full);     // top-k first: its softmax is over what top-k kept
  const bool keep0 = r0 < n && (r0 == 0 || ex0 < lim), keep1 = r1 < n && ex1 < lim, keep2 = r1 + 1 < n && ex2 < lim;
  // the kept set is a prefix of the ranking (ex is monotone); its last rank and its mass
  __shared__ float s_zp;
  if (keep0 && !keep1) { s_zp = ex1; s_need = r0; }
  if (keep1 && !keep2) { s_zp = ex2; s_need = r1; }
  // the nucleus wants more than the candidates hold (top_k = 0, near-uniform row): truncated to the CAP best, reported
  if (status && tid == 0 && top_k <= 0 && n == CAP && cdf[1023] < lim) atomicOr(&status[m], 1);
  __syncthreads();
  const float target = uu * s_zp;
  if (keep0 && ex0 <= target && target < ex1) atomicMin(&s_pick, r0);
  if (keep1 && ex1 <= target && target < ex2) atomicMin(&s_pick, r1);
  __syncthreads();
  if (tid == 0) s_pick = min(sidx[s_pick == 0x7fffffff ? s_need : s_pick], V - 1);      // no owner: target rounded up to the kept mass -> last kept rank; (NaN logits: any valid id)
  __syncthreads();
  if (tid == 0) idx[m] = (int64_t)s_pick + vocab_offset;
}

}  // namespace

// idx[m] = vocab_offset + a token drawn from row m of logits [M, V] (fp32, row stride ld) under HF's temperature / top-k / top-p
// warpers at the uniform u[m] in [0, 1).  top_k = 0: off; top_p >= 1: off.  temperature > 0.
// status (optional, int32 [M]): 0, or MN_SAMPLE_NUCLEUS_TRUNCATED | MN_SAMPLE_TIES_TRUNCATED for rows whose kept set was cut at 2048.
extern "C" int mn_sample_logits(const float* logits, int64_t ld, int M, int V, float temperature, int top_k, float top_p, const float* u,
                                int64_t vocab_offset, int64_t* idx, int32_t* status, void* stream) {
  MN_CHECK_ARG(logits && u && idx && M >= 1 && V >= 1 && ld >= V && temperature > 0.f && top_k >= 0 && top_p > 0.f,
               "mn_sample_logits: bad args (temperature > 0, top_k >= 0, top_p > 0)");
  MN_CHECK_ARG(top_k <= CAP, "mn_sample_logits: top_k = %d exceeds the %d candidates the kernel ranks", top_k, CAP);
  hipLaunchKernelGGL(sample_logits_kernel, dim3(M), dim3(1024), 0, mn_stream(stream), logits, ld, V, 1.0f / temperature, top_k, top_p, u,
                     vocab_offset, idx, status);
  MN_CHECK_LAUNCH("mn_sample_logits");
  return MN_OK;
}
