// sample.hip — uniform neighbour sampling (without or with replacement) on a CSC graph (SURVEY.md
// §8(f)-1):
// the device-side counterpart of torch.ops.pyg.neighbor_sample for one hop
// (torch_geometric/sampler/neighbor_sampler.py:550-577: colptr, row, seed nodes, fan-out k).
//
// One wavefront per frontier node v.  If deg(v) <= k every in-neighbour is taken (lanes copy the
// slot range, coalesced).  Otherwise the wave draws a uniform k-subset of the deg(v) slots with
// Floyd's algorithm (k <= 64: lane c makes draw c from a counter-based hash of (seed, v, c), the
// k insertions are k wave-wide membership tests), so a batch is reproducible from its seed
// regardless of scheduling; lanes 0..k-1 then emit one edge each.  With replacement
// (`replace`, the reference's `replace=True`, loader/neighbor_loader.py:209) every node with at
// least one in-neighbour emits exactly k edges, lane c an independent uniform draw of the same
// hash.  HBM-bound integer work: 3 index reads + 3 index writes per sampled edge.
#include "common.h"

namespace pygamd {

constexpr int kMaxFanout = 64;

__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    sample_neighbors_kernel(const IdxT* __restrict__ colptr, const IdxT* __restrict__ row,
                            const IdxT* __restrict__ frontier, int64_t n_frontier,
                            const IdxT* __restrict__ offsets, uint64_t seed, int flags,
                            const uint64_t* __restrict__ seed_dev,
                            IdxT* __restrict__ src_out, IdxT* __restrict__ dstpos_out,
                            IdxT* __restrict__ slot_out) {
  const int lane = lane_id();
  const int64_t f = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  if (f >= n_frontier) return;
  const int64_t v = frontier[f];
  const int64_t s = colptr[v];
  const int64_t deg = static_cast<int64_t>(colptr[v + 1]) - s;
  const int64_t o = offsets[f];
  const int64_t cnt = static_cast<int64_t>(offsets[f + 1]) - o;
  if (cnt <= 0) return;
  const bool replace = (flags & 1) != 0;
  // flags & 2: the draws depend on the frontier position too (disjoint trees)
  const uint64_t salt = (flags & 2) ? mix64(0xD1B54A32D192ED03ull * static_cast<uint64_t>(f + 1)) : 0;
  // (a captured graph bumps this word between replays.  It is mixed in on its own, not added:
  // the host seed is rng * 1000003 + hop, so `seed + word` would hand hop h + 1 of one replay the
  // stream of hop h of the next — ADVICE r3)
  if (seed_dev) seed = mix64(seed ^ mix64(*seed_dev * 0x9E3779B97F4A7C15ull));
  const uint64_t key = mix64(seed ^ mix64(static_cast<uint64_t>(v)) ^ salt);
  if (replace) {  // cnt = k independent draws from the deg in-neighbours (deg > 0 here)
    if (lane < cnt) {
      const uint64_t r = mix64(key + static_cast<uint64_t>(lane));
      const int64_t t = static_cast<int64_t>(__umul64hi(r, static_cast<uint64_t>(deg)));
      src_out[o + lane] = row[s + t];
      dstpos_out[o + lane] = static_cast<IdxT>(f);
      slot_out[o + lane] = static_cast<IdxT>(s + t);
    }
    return;
  }
  if (deg <= cnt) {  // take every in-neighbour
    for (int64_t t = lane; t < deg; t += kWave) {
      src_out[o + t] = row[s + t];
      dstpos_out[o + t] = static_cast<IdxT>(f);
      slot_out[o + t] = static_cast<IdxT>(s + t);
    }
    return;
  }
  // Floyd: for j = deg-k .. deg-1: t = U{0..j}; insert t, or j if t is already chosen.  Lane c
  // draws t_c; the k dependent insertions then run as k wave-wide steps (one 64-lane membership
  // test each) instead of an O(k^2) loop on lane 0 — same draws, same result, bit for bit.
  const int k = static_cast<int>(cnt);
  const int64_t jl = deg - cnt + lane;  // Floyd's j of this lane's draw
  int64_t t = 0;
  if (lane < k) {
    const uint64_t r = mix64(key + static_cast<uint64_t>(lane));
    t = static_cast<int64_t>(__umul64hi(r, static_cast<uint64_t>(jl + 1)));
  }
  int64_t mine = -1;  // final choice of lane c (only lanes < k end up with one)
  for (int c = 0; c < k; ++c) {
    const int64_t tc = bcast_uniform(t, c);
    const bool dup = __ballot(lane < c && mine == tc) != 0;
    if (lane == c) mine = dup ? jl : tc;
  }
  if (lane < k) {
    src_out[o + lane] = row[s + mine];
    dstpos_out[o + lane] = static_cast<IdxT>(f);
    slot_out[o + lane] = static_cast<IdxT>(s + mine);
  }
}

// cnt[f] = min(deg(frontier[f]), k)   (k < 0: deg; with replacement: k wherever deg > 0)
template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    sample_counts_kernel(const IdxT* __restrict__ colptr, const IdxT* __restrict__ frontier,
                         int64_t n, int64_t k, int replace, const int64_t* __restrict__ n_valid,
                         IdxT* __restrict__ cnt) {
  const int64_t f = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (f >= n) return;
  if (n_valid && f >= *n_valid) {  // padding of a fixed-capacity frontier
    cnt[f] = 0;
    return;
  }
  const int64_t v = frontier[f];
  const int64_t deg = static_cast<int64_t>(colptr[v + 1]) - static_cast<int64_t>(colptr[v]);
  if (replace && k >= 0) {
    cnt[f] = static_cast<IdxT>(deg > 0 ? k : 0);
  } else {
    cnt[f] = static_cast<IdxT>((k >= 0 && deg > k) ? k : deg);
  }
}

// ---- relabelling: global ids of the sampled sources -> local ids, new nodes in order of first
// appearance (what the reference's hash-map insertion produces), deterministic:
//   claim : local[s] = max(local[s], -(e + 2))  -> the smallest e wins among duplicates; entries
//           that already hold a local id (>= 0) are never lowered
//   flag  : flag[e] = (local[src[e]] == -(e + 2))
//   (inclusive scan of flag on the caller's side)
//   assign: the claimant writes local[s] = base + scan[e] - 1 and new_nodes[scan[e] - 1] = s
//   lookup: row[e] = local[src[e]]
template <typename IdxT>
__device__ __forceinline__ void atomic_max_idx(IdxT* p, IdxT v);
template <>
__device__ __forceinline__ void atomic_max_idx<int32_t>(int32_t* p, int32_t v) {
  atomicMax(p, v);
}
template <>
__device__ __forceinline__ void atomic_max_idx<int64_t>(int64_t* p, int64_t v) {
  atomicMax(reinterpret_cast<long long*>(p), static_cast<long long>(v));
}

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    relabel_claim_kernel(const IdxT* __restrict__ src, int64_t m,
                         const int64_t* __restrict__ m_dev, IdxT* __restrict__ local) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (m_dev && *m_dev < m) m = *m_dev;
  if (e < m) atomic_max_idx<IdxT>(local + src[e], static_cast<IdxT>(-(e + 2)));
}

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    relabel_flag_kernel(const IdxT* __restrict__ src, int64_t m,
                        const int64_t* __restrict__ m_dev, const IdxT* __restrict__ local,
                        int64_t* __restrict__ flag) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= m) return;
  const bool valid = !m_dev || e < *m_dev;  // the padding flags 0: the scan stays fixed-size
  flag[e] = (valid && static_cast<int64_t>(local[src[e]]) == -(e + 2)) ? 1 : 0;
}

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    relabel_assign_kernel(const IdxT* __restrict__ src, int64_t m,
                          const int64_t* __restrict__ m_dev, const int64_t* __restrict__ scan,
                          int64_t base, const int64_t* __restrict__ base_dev,
                          IdxT* __restrict__ local, IdxT* __restrict__ new_nodes) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (m_dev && *m_dev < m) m = *m_dev;
  if (e >= m) return;
  if (base_dev) base = *base_dev;
  const int64_t prev = (e == 0) ? 0 : scan[e - 1];
  if (scan[e] != prev) {  // this entry claimed its source
    const IdxT s = src[e];
    local[s] = static_cast<IdxT>(base + prev);
    new_nodes[prev] = s;
  }
}

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    relabel_lookup_kernel(const IdxT* __restrict__ src, int64_t m,
                          const int64_t* __restrict__ m_dev, const IdxT* __restrict__ local,
                          IdxT* __restrict__ out) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= m) return;
  out[e] = (!m_dev || e < *m_dev) ? local[src[e]] : static_cast<IdxT>(0);
}

}  // namespace pygamd

using namespace pygamd;

extern "C" {

int pygamd_sample_max_fanout(void) { return kMaxFanout; }

int pygamd_sample_neighbors(const void* colptr, const void* row, int idx_dtype,
                            const void* frontier, int64_t n_frontier, const void* offsets,
                            int64_t max_per_node, uint64_t seed, int flags,
                            const uint64_t* seed_dev, void* src_out,
                            void* dstpos_out, void* slot_out, void* stream) {
  if (n_frontier < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n_frontier == 0) return PYGAMD_OK;
  if (!colptr || !row || !frontier || !offsets || !src_out || !dstpos_out || !slot_out)
    return PYGAMD_ERR_INVALID_ARG;
  // a bounded fan-out larger than the LDS draw table is not supported (k < 0 = "all" is)
  if (max_per_node > kMaxFanout) return PYGAMD_ERR_UNSUPPORTED;
  if ((flags & 1) && max_per_node <= 0) return PYGAMD_ERR_INVALID_ARG;  // "all": no replacement
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    const unsigned grid = static_cast<unsigned>(ceil_div(n_frontier, kWavesPerBlock));
    hipLaunchKernelGGL((sample_neighbors_kernel<IdxT>), dim3(grid), dim3(kBlock), 0,
                       as_stream(stream), static_cast<const IdxT*>(colptr),
                       static_cast<const IdxT*>(row), static_cast<const IdxT*>(frontier),
                       n_frontier, static_cast<const IdxT*>(offsets), seed, flags, seed_dev,
                       static_cast<IdxT*>(src_out), static_cast<IdxT*>(dstpos_out),
                       static_cast<IdxT*>(slot_out));
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_sample_counts(const void* colptr, int idx_dtype, const void* frontier, int64_t n,
                         int64_t k, int replace, const int64_t* n_valid, void* cnt_out,
                         void* stream) {
  if (n < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n == 0) return PYGAMD_OK;
  if (!colptr || !frontier || !cnt_out) return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    hipLaunchKernelGGL((sample_counts_kernel<IdxT>),
                       dim3(static_cast<unsigned>(ceil_div(n, kBlock))), dim3(kBlock), 0,
                       as_stream(stream), static_cast<const IdxT*>(colptr),
                       static_cast<const IdxT*>(frontier), n, k, replace ? 1 : 0, n_valid,
                       static_cast<IdxT*>(cnt_out));
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_relabel(int phase, const void* src, int idx_dtype, int64_t m, const int64_t* m_dev,
                   void* local_map, int64_t* flag_or_scan, int64_t base,
                   const int64_t* base_dev, void* out, void* stream) {
  if (m < 0 || phase < 0 || phase > 3) return PYGAMD_ERR_INVALID_ARG;
  if (m == 0) return PYGAMD_OK;
  if (!src || !local_map) return PYGAMD_ERR_INVALID_ARG;
  if ((phase == 1 || phase == 2) && !flag_or_scan) return PYGAMD_ERR_INVALID_ARG;
  if ((phase == 2 || phase == 3) && !out) return PYGAMD_ERR_INVALID_ARG;
  const dim3 grid(static_cast<unsigned>(ceil_div(m, kBlock)));
  hipStream_t st = as_stream(stream);
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    const IdxT* s = static_cast<const IdxT*>(src);
    IdxT* local = static_cast<IdxT*>(local_map);
    switch (phase) {
      case 0:
        hipLaunchKernelGGL((relabel_claim_kernel<IdxT>), grid, dim3(kBlock), 0, st, s, m, m_dev,
                           local);
        break;
      case 1:
        hipLaunchKernelGGL((relabel_flag_kernel<IdxT>), grid, dim3(kBlock), 0, st, s, m, m_dev,
                           local, flag_or_scan);
        break;
      case 2:
        hipLaunchKernelGGL((relabel_assign_kernel<IdxT>), grid, dim3(kBlock), 0, st, s, m, m_dev,
                           flag_or_scan, base, base_dev, local, static_cast<IdxT*>(out));
        break;
      default:
        hipLaunchKernelGGL((relabel_lookup_kernel<IdxT>), grid, dim3(kBlock), 0, st, s, m, m_dev,
                           local, static_cast<IdxT*>(out));
    }
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

}  // extern "C"
