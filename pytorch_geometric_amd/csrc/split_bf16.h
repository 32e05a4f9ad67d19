// split_bf16.h — fp32 values as exact sums of three bf16 terms, shared by gemm.hip (operands split
// in registers next to the matrix instructions) and sage_fused.hip (operands split ONCE per element
// where they are produced: the aggregated / root rows on their way into LDS, the weights by a
// pre-pass).
//
// x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (both residuals are
// exact in fp32; three 8-bit significands cover the 24 of an fp32 value).  A product a * b is then
// the sum of the six cross terms of weight >= 2^-16 (a1 b3, a3 b1, a2 b2, a1 b2, a2 b1, a1 b1 — the
// three dropped ones are below 2^-24 of the product), each an EXACT bf16 x bf16 product summed in
// the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  Six instructions of 32 cycles replace eight
// fp32 instructions of 64 cycles for the same 16 k values.  Measured error against fp64
// (profiles/r02_split_bf16_accuracy_probe.txt): at or below that of the fp32 fmaf chain for
// K = 256 .. 2048 on normal, all-positive and wide-dynamic-range inputs.  Differences from the
// exact mode: results are not bitwise those of an fmaf chain, and an Inf operand gives NaN
// (Inf - Inf in the residual) where IEEE arithmetic would give Inf.
#pragma once
#include "common.h"

namespace pygamd {

typedef float sb_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const sb_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));  // v_cvt_pk_bf16_f32
}

// two fp32 values -> one packed bf16 pair per term (term 0 = leading)
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t (&t)[3]) {
  const uint32_t a = pack_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(a << 16);
  const float r1 = x1 - __uint_as_float(a & 0xffff0000u);
  const uint32_t b = pack_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(b << 16);
  const float s1 = r1 - __uint_as_float(b & 0xffff0000u);
  t[0] = a;
  t[1] = b;
  t[2] = pack_bf16(s0, s1);
}

// The same with the residual subtractions pinned to single v_sub_f32.  hipcc packs the two
// subtractions of a pair into one v_pk_add_f32, and a packed fp32 instruction stalls the matrix
// instructions of the OTHER waves on its SIMD: in the weight-gradient kernel, where one wave group
// converts while its partner multiplies, that serialised the two (CHANGELOG.md §5c).  The kernels
// that convert and multiply in the same wave, or whose products hide under a gather (gemm_nt,
// the one-kernel layer), measured the same with either form and keep the compiler's.
__device__ __forceinline__ float sub_f32_single(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void split_pair_single(float x0, float x1, uint32_t (&t)[3]) {
  const uint32_t a = pack_bf16(x0, x1);
  const float r0 = sub_f32_single(x0, __uint_as_float(a << 16));
  const float r1 = sub_f32_single(x1, __uint_as_float(a & 0xffff0000u));
  const uint32_t b = pack_bf16(r0, r1);
  const float s0 = sub_f32_single(r0, __uint_as_float(b << 16));
  const float s1 = sub_f32_single(r1, __uint_as_float(b & 0xffff0000u));
  t[0] = a;
  t[1] = b;
  t[2] = pack_bf16(s0, s1);
}

// The six cross terms in the order they are accumulated (small ones first): term t multiplies
// part kSplitTa[t] of a with part kSplitTb[t] of b.
constexpr int kSplitTerms = 6;
constexpr int kSplitTa[kSplitTerms] = {0, 2, 1, 0, 1, 0};
constexpr int kSplitTb[kSplitTerms] = {2, 0, 1, 1, 0, 0};

}  // namespace pygamd
