// Which SIMD does wave w of a 1024-thread workgroup run on?  (HW_REG_HW_ID: wave_id [3:0],
// simd_id [5:4], cu_id [11:8], se_id [15:13] on gfx9.)  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/hwid_probe.hip -o /tmp/hwid && /tmp/hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(1024) hwid_kernel(unsigned* out) {
  extern __shared__ float smem[];
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // all 32 bits of HW_ID
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
  if (threadIdx.x == 5000) smem[0] = 1.f;
}

int main() {
  for (int threads : {1024, 512}) {
    const int blocks = 8, waves = threads / 64;
    unsigned* d;
    hipMalloc(&d, blocks * 16 * sizeof(unsigned));
    hipMemset(d, 0, blocks * 16 * sizeof(unsigned));
    hipFuncSetAttribute(reinterpret_cast<const void*>(hwid_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 133 * 1024);
    hipLaunchKernelGGL(hwid_kernel, dim3(blocks), dim3(threads), 133 * 1024, 0, d);
    std::vector<unsigned> h(blocks * 16);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int b = 0; b < blocks; ++b) {
      printf("threads %d block %d: simd of wave 0..%d =", threads, b, waves - 1);
      for (int w = 0; w < waves; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
      printf("   (cu %u se %u, wave slots", (h[b * 16] >> 8) & 15, (h[b * 16] >> 13) & 7);
      for (int w = 0; w < waves; ++w) printf(" %u", h[b * 16 + w] & 15);
      printf(")\n");
    }
    hipFree(d);
  }
  return 0;
}
