// valu_rate_probe.hip — issue cost (cycles per wave-instruction on one SIMD) of the VALU
// instructions the bf16 split conversion is built from.  One wave per SIMD-sized workgroup;
// independent chains, s_memtime around an unrolled loop.
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_probe valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(64) probe(float* out, uint64_t* cycles, int iters) {
  float a0 = threadIdx.x * 1.0001f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  uint32_t u0 = __float_as_uint(a0), u1 = __float_as_uint(a1), u2 = __float_as_uint(a2),
           u3 = __float_as_uint(a3), u4 = 0, u5 = 1, u6 = 2, u7 = 3;
  f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) {  // v_cvt_pk_bf16_f32: 8 independent
      asm volatile(REP8("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %5, %6\n"
                        "v_cvt_pk_bf16_f32 %2, %6, %7\n v_cvt_pk_bf16_f32 %3, %7, %4\n")
                   : "=v"(u4), "=v"(u5), "=v"(u6), "=v"(u7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    } else if (OP == 1) {  // v_perm_b32
      asm volatile(REP8("v_perm_b32 %0, %4, %5, %8\n v_perm_b32 %1, %5, %6, %8\n"
                        "v_perm_b32 %2, %6, %7, %8\n v_perm_b32 %3, %7, %4, %8\n")
                   : "=v"(u4), "=v"(u5), "=v"(u6), "=v"(u7) : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "s"(0x07060302u));
    } else if (OP == 2) {  // v_and_b32
      asm volatile(REP8("v_and_b32 %0, 0xffff0000, %4\n v_and_b32 %1, 0xffff0000, %5\n"
                        "v_and_b32 %2, 0xffff0000, %6\n v_and_b32 %3, 0xffff0000, %7\n")
                   : "=v"(u4), "=v"(u5), "=v"(u6), "=v"(u7) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
    } else if (OP == 3) {  // v_sub_f32
      asm volatile(REP8("v_sub_f32 %0, %4, %5\n v_sub_f32 %1, %5, %6\n"
                        "v_sub_f32 %2, %6, %7\n v_sub_f32 %3, %7, %4\n")
                   : "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    } else if (OP == 4) {  // v_pk_add_f32
      asm volatile(REP8("v_pk_add_f32 %0, %2, %3\n v_pk_add_f32 %1, %3, %2\n"
                        "v_pk_add_f32 %0, %3, %2\n v_pk_add_f32 %1, %2, %3\n")
                   : "=v"(p2), "=v"(p3) : "v"(p0), "v"(p1));
    } else if (OP == 5) {  // v_lshlrev_b32
      asm volatile(REP8("v_lshlrev_b32 %0, 16, %4\n v_lshlrev_b32 %1, 16, %5\n"
                        "v_lshlrev_b32 %2, 16, %6\n v_lshlrev_b32 %3, 16, %7\n")
                   : "=v"(u4), "=v"(u5), "=v"(u6), "=v"(u7) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + threadIdx.x] = a4 + a5 + a6 + a7 + __uint_as_float(u4 ^ u5 ^ u6 ^ u7) + p2[0] + p3[1];
}

template <int OP>
static void run(const char* name, int waves_per_simd) {
  float* out; uint64_t* cyc;
  const int blocks = 1;  // one workgroup: its waves land on the SIMDs of one CU
  hipMalloc(&out, 64 * 64 * 4); hipMalloc(&cyc, 64 * 8);
  const int iters = 2000;
  hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(64 * waves_per_simd * 4 > 1024 ? 1024 : 64), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  uint64_t h = 0; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-22s %7.2f counter ticks per instruction (one wave, 32 instr per iteration)\n", name,
         static_cast<double>(h) / (iters * 32.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<3>("v_sub_f32", 1);
  run<2>("v_and_b32", 1);
  run<5>("v_lshlrev_b32", 1);
  run<1>("v_perm_b32", 1);
  run<4>("v_pk_add_f32", 1);
  run<0>("v_cvt_pk_bf16_f32", 1);
  return 0;
}
