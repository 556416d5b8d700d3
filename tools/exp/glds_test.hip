// experiment: where does global_load_lds_dwordx4 land for various LDS bases?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const u32x4* g, u32x4* out, int base, int lds_total) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < lds_total / 16; i += blockDim.x) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0xdeadbeef, 0, 0, 0};
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32x4* src = g + threadIdx.x;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(smem + base + wave * 1024), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < lds_total / 16; i += blockDim.x) out[i] = reinterpret_cast<u32x4*>(smem)[i];
}
int main() {
  const int nthr = 512;
  std::vector<unsigned> h(nthr * 4);
  for (int i = 0; i < nthr; ++i) { h[4*i] = 0x1000 + i; h[4*i+1] = h[4*i+2] = h[4*i+3] = i; }
  u32x4 *g, *o; hipMalloc(&g, nthr * 16); hipMemcpy(g, h.data(), nthr * 16, hipMemcpyHostToDevice);
  const int tests[][2] = {{0, 65536}, {0x8000, 65536}, {0xC000, 65536}, {0xE000, 65536}, {0xC000, 98304}, {0x10000, 98304}, {0x18000, 131072}};
  for (auto& t : tests) {
    int base = t[0], total = t[1];
    hipMalloc(&o, total);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, total);
    hipLaunchKernelGGL(k, dim3(1), dim3(nthr), total, 0, g, o, base, total);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned> r(total / 4);
    hipMemcpy(r.data(), o, total, hipMemcpyDeviceToHost);
    int good = 0, misplaced = 0; int first_bad = -1;
    for (int i = 0; i < nthr; ++i) {
      unsigned v = r[(base + i * 16) / 4];
      if (v == 0x1000u + i) ++good; else if (first_bad < 0) first_bad = i;
    }
    for (int j = 0; j < total / 16; ++j) { unsigned v = r[j * 4]; if (v != 0xdeadbeef && (j * 16 < base || j * 16 >= base + nthr * 16)) ++misplaced; }
    printf("base 0x%05x total %6d: %s good %d/%d first_bad lane %d misplaced chunks %d\n", base, total, hipGetErrorString(e), good, nthr, first_bad, misplaced);
    if (first_bad >= 0) { for (int j = 0; j < total / 16; ++j) { unsigned v = r[j*4]; if (v == 0x1000u + first_bad) printf("   lane %d data found at LDS 0x%x (expected 0x%x)\n", first_bad, j * 16, base + first_bad * 16); } }
    hipFree(o);
  }
  return 0;
}
