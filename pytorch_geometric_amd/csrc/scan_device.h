// scan_device.h — prefix sums of an integer array in global memory (uint32 / int32 / int64), any
// length: three launches (chunk sums, one workgroup over the chunk sums, chunk scans).  Used by
// the radix sort's digit tables and the hub-row compaction of csrc/graph.hip (exclusive, in
// place) and by pygamd_cumsum (inclusive: the samplers' offsets, torch.cumsum's place).
#pragma once
#include "common.h"

namespace pygamd {

constexpr int kScanItems = 16;                    // elements per thread
constexpr int kScanChunk = kBlock * kScanItems;   // 4096 elements per workgroup

// inclusive scan of one value per thread across the workgroup; returns the thread's inclusive
// value, `total` = the workgroup's sum (wave shuffles + one LDS hop; two barriers)
template <typename T>
__device__ __forceinline__ T block_inclusive_scan(T v, T* wave_tot, T& total) {
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
  T s = v;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const T o = __shfl_up(s, off, kWave);
    if (lane >= off) s += o;
  }
  if (lane == kWave - 1) wave_tot[w] = s;
  __syncthreads();
  T before = 0, all = 0;
#pragma unroll
  for (int i = 0; i < kWavesPerBlock; ++i) {
    const T t = wave_tot[i];
    if (i < w) before += t;
    all += t;
  }
  __syncthreads();  // wave_tot may be reused by the caller's next round
  total = all;
  return s + before;
}

// sums[b] = sum of chunk b
template <typename T>
__global__ void __launch_bounds__(kBlock)
    scan_chunk_sums_kernel(const T* __restrict__ data, int64_t n, T* __restrict__ sums) {
  __shared__ T wave_tot[kWavesPerBlock];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanChunk;
  T v = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int64_t i = base + static_cast<int64_t>(j) * kBlock + threadIdx.x;
    if (i < n) v += data[i];
  }
  T total;
  block_inclusive_scan(v, wave_tot, total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// one workgroup: sums[0 .. n_chunks) -> exclusive prefix, sums[n_chunks] = grand total
template <typename T>
__global__ void __launch_bounds__(kBlock) scan_sums_kernel(T* __restrict__ sums, int64_t n_chunks) {
  __shared__ T wave_tot[kWavesPerBlock];
  T carry = 0;
  for (int64_t base = 0; base < n_chunks; base += kBlock) {
    const int64_t i = base + threadIdx.x;
    const T v = i < n_chunks ? sums[i] : static_cast<T>(0);
    T total;
    const T incl = block_inclusive_scan(v, wave_tot, total);
    if (i < n_chunks) sums[i] = carry + incl - v;
    carry += total;
  }
  if (threadIdx.x == 0) sums[n_chunks] = carry;
}

// out[chunk b] = prefix of in[chunk b] (exclusive or inclusive) + sums[b]; `out` may be `in`.  A
// thread owns kScanItems CONSECUTIVE elements (blocked arrangement), so the order of the array is
// the scan order.
template <typename T, bool INCLUSIVE>
__global__ void __launch_bounds__(kBlock)
    scan_chunks_kernel(const T* in, int64_t n, const T* __restrict__ sums, T* out) {
  __shared__ T wave_tot[kWavesPerBlock];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanChunk +
                       static_cast<int64_t>(threadIdx.x) * kScanItems;
  T v[kScanItems];
  T mine = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    v[j] = base + j < n ? in[base + j] : static_cast<T>(0);
    mine += v[j];
  }
  T total;
  const T incl = block_inclusive_scan(mine, wave_tot, total);
  T run = sums[blockIdx.x] + incl - mine;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    if (INCLUSIVE) run += v[j];
    if (base + j < n) out[base + j] = run;
    if (!INCLUSIVE) run += v[j];
  }
}

// short arrays (a sampler's per-node counts): ONE workgroup walks the chunks with a carry
template <typename T, bool INCLUSIVE>
__global__ void __launch_bounds__(kBlock) scan_single_kernel(const T* in, int64_t n, T* out) {
  __shared__ T wave_tot[kWavesPerBlock];
  T carry = 0;
  for (int64_t c0 = 0; c0 < n; c0 += kScanChunk) {
    const int64_t base = c0 + static_cast<int64_t>(threadIdx.x) * kScanItems;
    T v[kScanItems];
    T mine = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
      v[j] = base + j < n ? in[base + j] : static_cast<T>(0);
      mine += v[j];
    }
    T total;
    const T incl = block_inclusive_scan(mine, wave_tot, total);
    T run = carry + incl - mine;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
      if (INCLUSIVE) run += v[j];
      if (base + j < n) out[base + j] = run;
      if (!INCLUSIVE) run += v[j];
    }
    carry += total;
  }
}
constexpr int64_t kScanSingleMax = 8 * kScanChunk;  // up to 32 k elements in one launch

static inline int64_t scan_chunks_of(int64_t n) { return ceil_div(n, kScanChunk); }
// scratch the scan needs next to the data: one element per chunk + the grand total
static inline size_t scan_scratch_bytes(int64_t n, size_t elem = sizeof(uint32_t)) {
  return static_cast<size_t>(scan_chunks_of(n) + 1) * elem;
}

// out[0 .. n) = prefix sums of in[0 .. n) (out may be in); sums[scan_chunks_of(n)] = the grand
// total (device memory).  The caller guarantees that the total fits T.
template <typename T, bool INCLUSIVE>
static inline int prefix_sum(const T* in, int64_t n, T* out, T* sums, hipStream_t st) {
  if (n <= 0) return PYGAMD_OK;
  const int64_t chunks = scan_chunks_of(n);
  hipLaunchKernelGGL((scan_chunk_sums_kernel<T>), dim3(static_cast<unsigned>(chunks)),
                     dim3(kBlock), 0, st, in, n, sums);
  PYGAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL((scan_sums_kernel<T>), dim3(1), dim3(kBlock), 0, st, sums, chunks);
  PYGAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL((scan_chunks_kernel<T, INCLUSIVE>), dim3(static_cast<unsigned>(chunks)),
                     dim3(kBlock), 0, st, in, n, sums, out);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

// torch.cumsum(in, 0): one launch up to kScanSingleMax elements (no grand total kept)
template <typename T>
static inline int cumsum_device(const T* in, int64_t n, T* out, T* sums, hipStream_t st) {
  if (n <= 0) return PYGAMD_OK;
  if (n <= kScanSingleMax) {
    hipLaunchKernelGGL((scan_single_kernel<T, true>), dim3(1), dim3(kBlock), 0, st, in, n, out);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  }
  return prefix_sum<T, true>(in, n, out, sums, st);
}

static inline int exclusive_scan_u32(uint32_t* data, int64_t n, uint32_t* sums, hipStream_t st) {
  return prefix_sum<uint32_t, false>(data, n, data, sums, st);
}

}  // namespace pygamd
