// common.h — shared device/host helpers for the gfx950 kernels (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pyg_amd.h"

namespace pygamd {

constexpr int kWave = 64;           // CDNA wavefront
constexpr int kBlock = 256;         // 4 waves per workgroup, one per SIMD
constexpr int kWavesPerBlock = kBlock / kWave;

extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) {
  g_last_hip_error = static_cast<int>(e);
  return PYGAMD_ERR_HIP;
}

#define PYGAMD_HIP_CHECK(expr)                      \
  do {                                              \
    hipError_t _e = (expr);                         \
    if (_e != hipSuccess) return ::pygamd::hip_fail(_e); \
  } while (0)

#define PYGAMD_LAUNCH_CHECK() PYGAMD_HIP_CHECK(hipGetLastError())

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// Grid for "one wave per work item" kernels: 4 items per 256-thread block, grid rounded up to a
// multiple of 8 so the XCD remap below is a bijection.
inline unsigned wave_grid(int64_t n_items) {
  int64_t blocks = ceil_div(n_items, kWavesPerBlock);
  blocks = round_up(blocks < 1 ? 1 : blocks, 8);
  return static_cast<unsigned>(blocks);
}

// Workgroup b is observed to run on XCD b % 8 (MI355X: 8 XCDs, private 4 MiB L2 each).  Map the
// hardware block id to a logical id so that each XCD walks one contiguous eighth of the rows:
// rowptr/col/out lines are then touched by one L2 only, and graphs with locality keep their
// neighbour rows in the same L2.  Speed only — correctness never depends on placement.
__device__ __forceinline__ int64_t xcd_logical_block() {
  const int64_t b = blockIdx.x;
  const int64_t per_xcd = gridDim.x >> 3;  // gridDim.x is a multiple of 8
  return (b & 7) * per_xcd + (b >> 3);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// wave index inside the block as a scalar (SGPR) value, so everything derived from it
// (row id, rowptr loads, loop bounds) stays on the scalar unit.
__device__ __forceinline__ int wave_in_block() {
  return __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
}

// Broadcast lane `k`'s value to the whole wave when k is wave-uniform: v_readlane_b32 -> SGPR.
__device__ __forceinline__ int32_t bcast_uniform(int32_t v, int k) {
  return __builtin_amdgcn_readlane(v, k);
}
__device__ __forceinline__ int64_t bcast_uniform(int64_t v, int k) {
  const int32_t lo = __builtin_amdgcn_readlane(static_cast<int32_t>(v), k);
  const int32_t hi = __builtin_amdgcn_readlane(static_cast<int32_t>(v >> 32), k);
  return (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
}
__device__ __forceinline__ float bcast_uniform(float v, int k) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k));
}

// Per-lane source lane (ds_bpermute through the LDS crossbar).
__device__ __forceinline__ int32_t bcast_lane(int32_t v, int k) { return __shfl(v, k, kWave); }
__device__ __forceinline__ int64_t bcast_lane(int64_t v, int k) {
  const int32_t lo = __shfl(static_cast<int32_t>(v), k, kWave);
  const int32_t hi = __shfl(static_cast<int32_t>(v >> 32), k, kWave);
  return (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
}
__device__ __forceinline__ float bcast_lane(float v, int k) { return __shfl(v, k, kWave); }

// The inverse of bcast_lane: lane l's value lands in lane `to` (ds_permute, a push through the
// LDS crossbar).  `to` must be a permutation of 0..63 over the wave.
__device__ __forceinline__ int32_t push_lane(int32_t v, int to) {
  return __builtin_amdgcn_ds_permute(to << 2, v);
}
__device__ __forceinline__ int64_t push_lane(int64_t v, int to) {
  const int32_t lo = __builtin_amdgcn_ds_permute(to << 2, static_cast<int32_t>(v));
  const int32_t hi = __builtin_amdgcn_ds_permute(to << 2, static_cast<int32_t>(v >> 32));
  return (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
}

template <int VW>
struct Vec;
template <>
struct Vec<1> {
  float v[1];
};
template <>
struct alignas(8) Vec<2> {
  float v[2];
};
template <>
struct alignas(16) Vec<4> {
  float v[4];
};

template <int VW>
__device__ __forceinline__ Vec<VW> load_vec(const float* p) {
  return *reinterpret_cast<const Vec<VW>*>(p);
}
// rows that are read exactly once (epilogue operands): keep them out of the way of the gathered
// feature rows in L2
template <int VW>
__device__ __forceinline__ Vec<VW> load_vec_streamed(const float* p) {
  typedef float vt __attribute__((ext_vector_type(VW)));
  Vec<VW> r;
  if constexpr (VW == 1) {
    r.v[0] = __builtin_nontemporal_load(p);
  } else {
    const vt t = __builtin_nontemporal_load(reinterpret_cast<const vt*>(p));
#pragma unroll
    for (int i = 0; i < VW; ++i) r.v[i] = t[i];
  }
  return r;
}
template <int VW>
__device__ __forceinline__ void store_vec(float* p, const Vec<VW>& v) {
  *reinterpret_cast<Vec<VW>*>(p) = v;
}

// float atomic max/min through the sign-split integer trick: non-negative floats order like
// their int bits, negative floats order inversely to their unsigned bits.  NaN propagates like
// torch's amax/amin: for max it is stored as +qNaN (int-larger than +inf, and unsigned-smaller
// than every negative), for min as -qNaN (unsigned-larger than -inf, int-smaller than every
// non-negative) — both absorbing under the respective pair of integer atomics.
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  int bits = (v != v) ? 0x7FC00000 : __float_as_int(v);
  if (bits >= 0) {
    atomicMax(reinterpret_cast<int*>(addr), bits);
  } else {
    atomicMin(reinterpret_cast<unsigned int*>(addr), static_cast<unsigned int>(bits));
  }
}
__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
  int bits = (v != v) ? static_cast<int>(0xFFC00000u) : __float_as_int(v);
  if (bits >= 0) {
    atomicMin(reinterpret_cast<int*>(addr), bits);
  } else {
    atomicMax(reinterpret_cast<unsigned int*>(addr), static_cast<unsigned int>(bits));
  }
}
__device__ __forceinline__ void atomic_mul_f32(float* addr, float v) {
  unsigned int* a = reinterpret_cast<unsigned int*>(addr);
  unsigned int old = *a, assumed;
  do {
    assumed = old;
    old = atomicCAS(a, assumed, __float_as_uint(__uint_as_float(assumed) * v));
  } while (assumed != old);
}

#define PYGAMD_DISPATCH_IDX(idx_dtype, ...)                 \
  [&]() -> int {                                            \
    if ((idx_dtype) == PYGAMD_IDX_I64) {                    \
      using IdxT = int64_t;                                 \
      return __VA_ARGS__();                                 \
    } else if ((idx_dtype) == PYGAMD_IDX_I32) {             \
      using IdxT = int32_t;                                 \
      return __VA_ARGS__();                                 \
    }                                                       \
    return PYGAMD_ERR_INVALID_ARG;                          \
  }()

}  // namespace pygamd
