// Probe of ds_read_b64_tr_b16 semantics: LDS holds u16 element indices; every lane passes a byte address; print what comes back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(int mode, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  uint32_t addr;
  if (mode == 0) addr = 8 * l;                                   // lane l -> chunk l (4 contiguous elements)
  else if (mode == 1) addr = 2 * ((l & 15) / 4 * 16 + (l & 3) * 4 + (l >> 4) * 64);   // [4 rows][16 cols] block per 16-lane group, lane -> (row = (l&15)/4, chunk = l&3)
  else if (mode == 2) addr = 2 * ((l & 3) * 16 + ((l & 15) / 4) * 4 + (l >> 4) * 64); // lane -> (row = l&3, chunk = (l&15)/4)
  else addr = 64;                                                 // uniform
  addr += (uint32_t)(uintptr_t)lds;
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("  l%2d: %4d %4d %4d %4d", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); if (l % 4 == 3) printf("\n"); }
  }
  return 0;
}
