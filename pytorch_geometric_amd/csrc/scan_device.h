// scan_device.h — exclusive prefix sum of a uint32 array in global memory, in place, any length
// below 2^32 total: three launches (chunk sums, one workgroup over the chunk sums, chunk scans).
// Used by the radix sort's digit tables and by the hub-row compaction of csrc/graph.hip.
#pragma once
#include "common.h"

namespace pygamd {

constexpr int kScanItems = 16;                    // elements per thread
constexpr int kScanChunk = kBlock * kScanItems;   // 4096 elements per workgroup

// inclusive scan of one value per thread across the workgroup; returns the thread's inclusive
// value, `total` = the workgroup's sum (wave shuffles + one LDS hop; two barriers)
__device__ __forceinline__ uint32_t block_inclusive_scan_u32(uint32_t v, uint32_t* wave_tot,
                                                             uint32_t& total) {
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
  uint32_t s = v;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const uint32_t o = __shfl_up(s, off, kWave);
    if (lane >= off) s += o;
  }
  if (lane == kWave - 1) wave_tot[w] = s;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int i = 0; i < kWavesPerBlock; ++i) {
    const uint32_t t = wave_tot[i];
    if (i < w) before += t;
    all += t;
  }
  __syncthreads();  // wave_tot may be reused by the caller's next round
  total = all;
  return s + before;
}

// sums[b] = sum of chunk b
__global__ void __launch_bounds__(kBlock)
    scan_chunk_sums_kernel(const uint32_t* __restrict__ data, int64_t n,
                           uint32_t* __restrict__ sums) {
  __shared__ uint32_t wave_tot[kWavesPerBlock];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanChunk;
  uint32_t v = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int64_t i = base + static_cast<int64_t>(j) * kBlock + threadIdx.x;
    if (i < n) v += data[i];
  }
  uint32_t total;
  block_inclusive_scan_u32(v, wave_tot, total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// one workgroup: sums[0 .. n_chunks) -> exclusive prefix, sums[n_chunks] = grand total
__global__ void __launch_bounds__(kBlock)
    scan_sums_kernel(uint32_t* __restrict__ sums, int64_t n_chunks) {
  __shared__ uint32_t wave_tot[kWavesPerBlock];
  uint32_t carry = 0;
  for (int64_t base = 0; base < n_chunks; base += kBlock) {
    const int64_t i = base + threadIdx.x;
    const uint32_t v = i < n_chunks ? sums[i] : 0u;
    uint32_t total;
    const uint32_t incl = block_inclusive_scan_u32(v, wave_tot, total);
    if (i < n_chunks) sums[i] = carry + incl - v;
    carry += total;
  }
  if (threadIdx.x == 0) sums[n_chunks] = carry;
}

// data[chunk b] -> exclusive prefix inside the chunk + sums[b].  A thread owns kScanItems
// CONSECUTIVE elements (blocked arrangement), so the order of the array is the scan order.
__global__ void __launch_bounds__(kBlock)
    scan_chunks_kernel(uint32_t* __restrict__ data, int64_t n,
                       const uint32_t* __restrict__ sums) {
  __shared__ uint32_t wave_tot[kWavesPerBlock];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanChunk +
                       static_cast<int64_t>(threadIdx.x) * kScanItems;
  uint32_t v[kScanItems];
  uint32_t mine = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    v[j] = base + j < n ? data[base + j] : 0u;
    mine += v[j];
  }
  uint32_t total;
  const uint32_t incl = block_inclusive_scan_u32(mine, wave_tot, total);
  uint32_t run = sums[blockIdx.x] + incl - mine;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    if (base + j < n) data[base + j] = run;
    run += v[j];
  }
}

static inline int64_t scan_chunks_of(int64_t n) { return ceil_div(n, kScanChunk); }
// scratch the scan needs next to the data: one uint32 per chunk + the grand total
static inline size_t scan_scratch_bytes(int64_t n) {
  return static_cast<size_t>(scan_chunks_of(n) + 1) * sizeof(uint32_t);
}

// data[0 .. n) -> exclusive prefix sums, in place; sums[scan_chunks_of(n)] = the grand total
// (device memory).  The caller guarantees that the total fits 32 bits.
static inline int exclusive_scan_u32(uint32_t* data, int64_t n, uint32_t* sums, hipStream_t st) {
  if (n <= 0) return PYGAMD_OK;
  const int64_t chunks = scan_chunks_of(n);
  hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3(static_cast<unsigned>(chunks)), dim3(kBlock), 0,
                     st, data, n, sums);
  PYGAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kBlock), 0, st, sums, chunks);
  PYGAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(scan_chunks_kernel, dim3(static_cast<unsigned>(chunks)), dim3(kBlock), 0, st,
                     data, n, sums);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

}  // namespace pygamd
