// Experiment: what does the weight-streaming access pattern of stream_mfma.hip reach with NO compute?
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_ceiling stream_ceiling.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(err_), __LINE__); exit(1); } } while (0)

// tile = 16 rows x ks k (bf16); chunk = 16 rows x 256 k = 8 KiB; a wave streams its tiles chunk by chunk
template <int DEPTH, bool NTL, int MODE>   // MODE 0: registers only; 1: + park in LDS (ds_write) and read one fragment back
__global__ __launch_bounds__(1024) void k(const uint16_t* __restrict__ W, uint32_t* out, int Ntot, int K, int ks) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int z = blockIdx.y, k0 = z * ks, klen = min(ks, K - k0);
  const int r8 = lane >> 3, c8 = lane & 7;
  const int ntiles = Ntot / 16, nch = klen / 256, twaves = gridDim.x * nw, t0 = blockIdx.x * nw + wave;
  const int mytiles = t0 < ntiles ? (ntiles - t0 + twaves - 1) / twaves : 0;
  const int total = mytiles * nch;
  char* wbuf = lds + wave * 8192;
  u32x4 ring[DEPTH][8];
  int it = t0, ich = 0;
  auto issue = [&](u32x4 (&dst)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (i & 1) * 8 + r8;
      const uint16_t* p = W + (int64_t)(it * 16 + row) * K + k0 + ich * 256 + ((i >> 1) * 8 + c8) * 8;
      if (NTL) dst[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
      else dst[i] = *reinterpret_cast<const u32x4*>(p);
    }
    if (++ich == nch) { ich = 0; it += twaves; }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) if (d < total) issue(ring[d]);
  u32x4 acc = {0, 0, 0, 0};
  for (int q0 = 0; q0 < total; q0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int q = q0 + d;
      if (q < total) {
        if (MODE == 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(wbuf + ((i * 64 + lane) ^ (lane & 15)) * 16 % 8192) = ring[d][i];
          if (q + DEPTH < total) issue(ring[d]);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc ^= *reinterpret_cast<const u32x4*>(wbuf + ((i * 64 + (lane ^ 21)) & 511) * 16);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc ^= ring[d][i];
          if (q + DEPTH < total) issue(ring[d]);
        }
      }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x] = 1;
}

template <int DEPTH, bool NTL, int MODE>
void run(const char* name, std::vector<uint16_t*>& bufs, uint32_t* out, int Ntot, int K, int ks, int nw) {
  const int nz = K / ks, gx = 256 / nz;
  CK(hipFuncSetAttribute((const void*)&k<DEPTH, NTL, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const size_t lds = 64 * 1024 + nw * 8192;   // same footprint as the real kernel: one workgroup per CU
  hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
  for (int i = 0; i < 6; ++i) hipLaunchKernelGGL((k<DEPTH, NTL, MODE>), dim3(gx, nz), dim3(nw * 64), lds, 0, bufs[i % bufs.size()], out, Ntot, K, ks);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(s));
    for (int i = 0; i < 24; ++i) hipLaunchKernelGGL((k<DEPTH, NTL, MODE>), dim3(gx, nz), dim3(nw * 64), lds, 0, bufs[i % bufs.size()], out, Ntot, K, ks);
    CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
    float ms; CK(hipEventElapsedTime(&ms, s, e));
    if (ms < best) best = ms;
  }
  const double us = best * 1e3 / 24, gb = (double)Ntot * K * 2 / us * 1e-3;
  printf("%-34s N=%d K=%d ks=%d nw=%d grid=%dx%d: %.1f us  %.0f GB/s\n", name, Ntot, K, ks, nw, gx, nz, us, gb);
}

int main() {
  const int N = 16384, K = 3072;
  std::vector<uint16_t*> bufs(6);
  for (auto& b : bufs) { CK(hipMalloc(&b, (size_t)N * K * 2)); CK(hipMemset(b, 1, (size_t)N * K * 2)); }
  uint32_t* out; CK(hipMalloc(&out, 4096));
  for (int nw : {8, 16}) {
    run<1, false, 0>("regs depth1", bufs, out, N, K, 768, nw);
    run<1, true, 0>("regs depth1 nt", bufs, out, N, K, 768, nw);
    run<2, true, 0>("regs depth2 nt", bufs, out, N, K, 768, nw);
    run<4, true, 0>("regs depth4 nt", bufs, out, N, K, 768, nw);
    run<1, true, 1>("lds depth1 nt", bufs, out, N, K, 768, nw);
    run<2, true, 1>("lds depth2 nt", bufs, out, N, K, 768, nw);
  }
  run<2, true, 0>("regs depth2 nt ks=3072", bufs, out, N, K, 3072, 8);
  run<4, true, 0>("regs depth4 nt ks=3072", bufs, out, N, K, 3072, 8);
  run<4, true, 0>("regs depth4 nt ks=3072 nw16", bufs, out, N, K, 3072, 16);
  return 0;
}
