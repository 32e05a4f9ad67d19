// minibatch.hip — STATIC-SHAPE ("slot") neighbour-sampled batches (BASELINE
// config 4: NeighborLoader [15, 10, 5] x batch 1024).  The sampling contract is the one of
// csrc/sample.hip (torch_geometric/sampler/neighbor_sampler.py:550-577: one hop = for every frontier
// node a uniform min(deg, k)-subset of its in-neighbours without replacement, Floyd's algorithm on
// a counter-based hash); what differs is the OUTPUT LAYOUT, chosen so that a whole training step
// has the same shapes, the same row ranges and no host read every batch — and needs two launches
// per hop instead of ten (round 3: 181 kernels per captured batch, half of them scans, fills,
// concatenations and index kernels around the sampler):
//
//   * node ids of a batch are BLOCK POSITIONS: block 0 = the B seeds, block h + 1 = the k_h slots
//     of every position of block h (capacity cap[h + 1] = cap[h] * k_h), bases[b] = first id of
//     block b.  Slot e of hop h IS position e of block h + 1: a sampled edge and the row its source
//     node would get are the same index.
//   * every destination row owns a FIXED slot range [begin[r], begin[r] + k) of which the first
//     cnt[r] = min(deg, k) are filled: `row_begin` is static, the sampler writes `row_end` — the
//     aggregation kernels take the pair (pygamd_spmm_args.rowend); no offsets scan, no compaction.
//   * duplicates (the same graph node sampled twice, or already in the batch) resolve WITHOUT a
//     scan and without resetting anything between batches: every occurrence claims its node in a
//     global map with atomicMax of  epoch << 32 | (2^32 - 1 - id)  — the current batch's epoch
//     beats every older entry, inside a batch the SMALLEST id (earliest block, earliest slot) wins
//     — and a second pass reads the winner back: src_id[slot] = the row that holds the source's
//     features; the winning slot keeps the node (node_g[id] = graph node), the others become holes
//     (node_g = -1: no features gathered, nothing sampled from them, zero gradient).  The
//     numbering is NOT the reference's dense first-appearance numbering — it is the padded-id form
//     a captured step needs; `NeighborSampler.sample_from_nodes` (sample.hip) keeps the
//     reference's contract for everything that looks at node ids.
//   * the backward of layer l needs its edges grouped by SOURCE: the resolve pass counts them per
//     source row on the way, one scan launch turns all the counts into CSR pointers and one fill
//     launch writes the destination ids — a transposed CSR per backward layer in three launches per
//     batch, so that the input gradient is an SpMM (one 1 KiB row read per edge) instead of
//     E x F float atomics (round 3: 0.24 ms per launch for 121 k edges x 256 floats).
// HBM-bound integer work throughout.
#include "common.h"

namespace pygamd {

constexpr int kSlotMaxFanout = 64;
constexpr int kSlotMaxHops = 8;

__device__ __forceinline__ uint64_t slot_mix64(uint64_t z) {  // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ int64_t round_up_dev(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

__device__ __forceinline__ long long slot_key(int64_t epoch, int64_t id) {
  return static_cast<long long>((epoch << 32) | (0xFFFFFFFFll - id));
}
__device__ __forceinline__ int64_t slot_key_id(long long key) {
  return 0xFFFFFFFFll - (static_cast<int64_t>(key) & 0xFFFFFFFFll);
}

// ---- seeds: block 0
template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    slots_seed_kernel(const IdxT* __restrict__ seeds, int64_t B, const int64_t* __restrict__ epoch,
                      long long* __restrict__ local, int64_t* __restrict__ node_g) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= B) return;
  const int64_t v = static_cast<int64_t>(seeds[i]);
  node_g[i] = v;
  atomicMax(local + v, slot_key(*epoch, i));
}

// ---- one hop: G lanes per frontier position (G = 16 / 32 / 64 >= the fan-out), 64 / G positions
// per wave.  Draws as in sample_neighbors_kernel (sample.hip), a position's lanes numbered from 0.
// The hop is a chain of four dependent random reads (node id -> colptr -> row -> claim) over the
// whole graph's arrays: latency, hidden only by positions in flight — a wave per position with a
// fan-out of 5 kept 5 of 64 lanes busy and 8 k positions in flight per chip (63 us for the 154 k
// positions of config 4's last hop); four positions per wave are four times as many.
template <typename IdxT, int G>
__global__ void __launch_bounds__(kBlock)
    slots_sample_kernel(const IdxT* __restrict__ colptr, const IdxT* __restrict__ row,
                        const int64_t* __restrict__ node_g, int64_t frontier_base,
                        int64_t n_frontier, int k, int64_t slot_base, int64_t B, uint64_t seed,
                        int hop, const int64_t* __restrict__ epoch,
                        long long* __restrict__ local, int64_t* __restrict__ src_g,
                        int32_t* __restrict__ row_end, float* __restrict__ inv_cnt) {
  constexpr int kPer = kWave / G;            // positions per wave
  const int wl = lane_id();
  const int sub = wl / G, lane = wl % G;     // position inside the wave, lane inside the position
  const int64_t f = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block()) * kPer + sub;
  // (no early exit per position: the draws below talk across the wave; the tail positions of the
  // last wave run on position n_frontier - 1 and store nothing)
  const bool live = f < n_frontier;
  if (f - sub >= n_frontier) return;         // (wave-uniform: the whole wave is past the end)
  const int64_t fc = live ? f : n_frontier - 1;
  const int64_t r = frontier_base + fc;
  const int64_t begin = slot_base + fc * k;
  const int64_t v = node_g[r];
  int64_t s = 0, deg = 0;
  if (v >= 0) {
    s = colptr[v];
    deg = static_cast<int64_t>(colptr[v + 1]) - s;
  }
  const int cnt = static_cast<int>(deg < k ? deg : k);
  const int64_t ep = *epoch;
  // (the replay counter is mixed in on its own, not added to the seed: seed + hop + epoch would
  // give hop h + 1 of one batch the stream of hop h of the next)
  const uint64_t hop_seed =
      slot_mix64(slot_mix64(seed ^ slot_mix64(static_cast<uint64_t>(ep) * 0x9E3779B97F4A7C15ull)) +
                 static_cast<uint64_t>(hop));
  int64_t mine = lane;  // deg <= k: every in-neighbour, in storage order
  // Floyd: for j = deg-k .. deg-1: t = U{0..j}; insert t, or j if t is already chosen.  (Positions
  // with deg <= k walk the loop too — its cross-lane steps need every lane — and keep `mine`.)
  const bool draw = deg > k;
  const uint64_t key = slot_mix64(hop_seed ^ slot_mix64(static_cast<uint64_t>(v)));
  const int64_t jl = deg - k + lane;
  int64_t t = 0;
  if (draw && lane < k) {
    const uint64_t rr = slot_mix64(key + static_cast<uint64_t>(lane));
    t = static_cast<int64_t>(__umul64hi(rr, static_cast<uint64_t>(jl + 1)));
  }
  if (__any(draw)) {  // (wave-uniform)
    int64_t pick = -1;
    for (int c = 0; c < k; ++c) {
      const int64_t tc = kPer == 1 ? bcast_uniform(t, c) : bcast_lane(t, sub * G + c);
      const uint64_t hits = __ballot(lane < c && pick == tc);
      const bool dup = kPer == 1 ? hits != 0
                                 : ((hits >> (sub * G)) & ((G == 64 ? ~0ull : (1ull << G) - 1))) != 0;
      if (lane == c) pick = dup ? jl : tc;
    }
    if (draw) mine = pick;
  }
  if (live && lane < k) {
    int64_t sg = -1;
    if (lane < cnt) {
      sg = static_cast<int64_t>(row[s + mine]);
      atomicMax(local + sg, slot_key(ep, B + begin + lane));
    }
    src_g[begin + lane] = sg;
  }
  if (live && lane == 0) {
    row_end[r] = static_cast<int32_t>(begin + cnt);
    inv_cnt[r] = 1.f / static_cast<float>(cnt > 0 ? cnt : 1);
  }
}

struct SlotCounters {
  int32_t* p[kSlotMaxHops];
};

// ---- resolve one hop's slots: the winner of every sampled source, the hop's new nodes, and the
// per-source edge counts of the transposed CSRs this hop belongs to
__global__ void __launch_bounds__(kBlock)
    slots_resolve_kernel(const int64_t* __restrict__ src_g, int64_t slot_base, int64_t n_slots,
                         int64_t B, const int64_t* __restrict__ epoch,
                         const long long* __restrict__ local,
                         int32_t* __restrict__ src_id, int64_t* __restrict__ node_g,
                         SlotCounters counts, int n_counts) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= n_slots) return;
  const int64_t slot = slot_base + e;
  const int64_t sg = src_g[slot];
  int64_t keep = -1;
  if (sg >= 0) {
    // (a claim of another epoch can only be there if the caller broke the contract — epochs must
    // grow — and would name a row of some other batch: the slot then keeps its own node)
    const long long key = local[sg];
    const int64_t win = (static_cast<int64_t>(key) >> 32) == *epoch ? slot_key_id(key) : B + slot;
    src_id[slot] = static_cast<int32_t>(win);
    if (win == B + slot) keep = sg;  // this slot introduces the node: its row holds the features
    for (int c = 0; c < n_counts; ++c) atomicAdd(counts.p[c] + win, 1);
  } else {
    src_id[slot] = 0;  // never read (past row_end); a valid index all the same
  }
  node_g[B + slot] = keep;
}

// ---- x[node_g] -> out (holes: zero rows); 16-byte pieces, LPR lanes per row
__global__ void __launch_bounds__(kBlock)
    slots_gather_kernel(const float* __restrict__ x, int64_t ldx, int64_t F,
                        const int64_t* __restrict__ node_g, int64_t n_rows,
                        float* __restrict__ out, int64_t ldo) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t units = F / 4;
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t r = t / units;
  if (r >= n_rows) return;
  const int64_t u = t - r * units;
  const int64_t v = node_g[r];
  f4 val = {0.f, 0.f, 0.f, 0.f};
  if (v >= 0) val = __builtin_nontemporal_load(reinterpret_cast<const f4*>(x + v * ldx + 4 * u));
  *reinterpret_cast<f4*>(out + r * ldo + 4 * u) = val;
}

// ---- exclusive scans of up to kSlotMaxHops count arrays in ONE launch without any hand-off between
// workgroups: array c is cut into kScanChunks chunks, workgroup (w, c) first SUMS everything in
// front of its chunk (a coalesced read of L2-resident counts: 340 KB on average for the 170 k rows
// of config 4 — cheaper than a second launch or a look-back chain) and then scans its own chunk.
// (One 1024-thread workgroup per array took 0.18 ms; sixteen entries per thread 0.09 ms.)
struct SlotScans {
  const int32_t* in[kSlotMaxHops];
  int32_t* out[kSlotMaxHops];  // [n + 1]
  int64_t n[kSlotMaxHops];
};
constexpr int kScanBlock = 1024;
constexpr int kScanChunks = 64;
constexpr int kScanPer = 4;  // consecutive entries per thread and tile (one 16-byte load)

__device__ __forceinline__ int scan_block_sum(int v, int* wave_tot, int lane, int wave) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  if (lane == 0) wave_tot[wave] = v;
  __syncthreads();
  int s = 0;
  for (int w = 0; w < kScanBlock / kWave; ++w) s += wave_tot[w];
  __syncthreads();
  return s;
}

__global__ void __launch_bounds__(kScanBlock) slots_scan_kernel(SlotScans a) {
  typedef int i4 __attribute__((ext_vector_type(4)));
  __shared__ int wave_tot[kScanBlock / kWave];
  const int32_t* __restrict__ in = a.in[blockIdx.y];
  int32_t* __restrict__ out = a.out[blockIdx.y];
  const int64_t n = a.n[blockIdx.y];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int64_t chunk = round_up_dev((n + kScanChunks - 1) / kScanChunks, 4);
  const int64_t start = static_cast<int64_t>(blockIdx.x) * chunk;
  const bool last = blockIdx.x == kScanChunks - 1;
  if (start >= n && !last) return;
  const int64_t s0 = start < n ? start : n;
  const int64_t end = start + chunk < n ? start + chunk : n;
  const bool vec = (reinterpret_cast<uintptr_t>(in) & 15u) == 0;
  // ---- everything in front of this chunk
  int mine = 0;
  if (vec) {
    for (int64_t i = 4 * static_cast<int64_t>(threadIdx.x); i + 4 <= s0; i += 4 * kScanBlock) {
      const i4 t = *reinterpret_cast<const i4*>(in + i);
      mine += t[0] + t[1] + t[2] + t[3];
    }
    // (s0 is a multiple of 4 or equals n: a tail shorter than 4 exists only when s0 == n)
    for (int64_t i = (s0 / 4) * 4 + threadIdx.x; i < s0; i += kScanBlock) mine += in[i];
  } else {
    for (int64_t i = threadIdx.x; i < s0; i += kScanBlock) mine += in[i];
  }
  int base = scan_block_sum(mine, wave_tot, lane, wave);
  // ---- this chunk, tile by tile (one tile of 4096 entries at the config-4 sizes)
  for (int64_t t0 = s0; t0 < end; t0 += static_cast<int64_t>(kScanBlock) * kScanPer) {
    const int64_t i0 = t0 + static_cast<int64_t>(threadIdx.x) * kScanPer;
    int v[kScanPer];
    if (vec && i0 + kScanPer <= end) {
      const i4 t = *reinterpret_cast<const i4*>(in + i0);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = t[e];
    } else {
#pragma unroll
      for (int q = 0; q < kScanPer; ++q) v[q] = i0 + q < end ? in[i0 + q] : 0;
    }
    const int sum = v[0] + v[1] + v[2] + v[3];
    int inc = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int t = __shfl_up(inc, off, kWave);
      if (lane >= off) inc += t;
    }
    if (lane == kWave - 1) wave_tot[wave] = inc;
    __syncthreads();
    int before = base;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    int tile_total = 0;
    for (int w = 0; w < kScanBlock / kWave; ++w) tile_total += wave_tot[w];
    int run = before + inc - sum;
#pragma unroll
    for (int q = 0; q < kScanPer; ++q) {
      if (i0 + q < end) out[i0 + q] = run;
      run += v[q];
    }
    __syncthreads();
    base += tile_total;
  }
  if (last && threadIdx.x == 0) out[n] = base;  // (the last chunk ends at n: base = the total)
}

// ---- fill the transposed CSRs: slot -> (source row, destination row); positions inside a source
// row are handed out by an atomic cursor (the order of a row's few entries is not fixed: sums of
// two or three gradient rows may differ in their last bit between runs, as with the atomics this
// replaces)
struct SlotFill {
  const int32_t* ptr[kSlotMaxHops];   // transposed row pointers
  int32_t* cursor[kSlotMaxHops];      // zeroed, one per source row
  int32_t* col[kSlotMaxHops];         // out: destination row of every transposed entry
  int64_t n_slots[kSlotMaxHops];      // CSR c covers the slots [0, n_slots[c])
  int64_t ebase[kSlotMaxHops + 1];    // first slot of hop h (ebase[hops] = all slots)
  int64_t nbase[kSlotMaxHops + 1];    // bases[h]: first row of block h
  int32_t fanout[kSlotMaxHops];
  int hops, n_csr;
};

__global__ void __launch_bounds__(kBlock)
    slots_fill_kernel(const int64_t* __restrict__ src_g, const int32_t* __restrict__ src_id,
                      SlotFill a) {
  const int64_t slot = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (slot >= a.n_slots[0]) return;  // CSR 0 covers the most hops
  if (src_g[slot] < 0) return;
  int h = 0;
  while (h + 1 < a.hops && slot >= a.ebase[h + 1]) ++h;
  const int32_t dst = static_cast<int32_t>(a.nbase[h] + (slot - a.ebase[h]) / a.fanout[h]);
  const int32_t s = src_id[slot];
  for (int c = 0; c < a.n_csr; ++c) {
    if (slot < a.n_slots[c]) {
      const int32_t pos = a.ptr[c][s] + atomicAdd(a.cursor[c] + s, 1);
      a.col[c][pos] = dst;
    }
  }
}

}  // namespace pygamd

using namespace pygamd;

extern "C" {

int pygamd_slots_max_fanout(void) { return kSlotMaxFanout; }
int pygamd_slots_max_hops(void) { return kSlotMaxHops; }

int pygamd_slots_seed(const void* seeds, int idx_dtype, int64_t B, const int64_t* epoch_dev,
                      int64_t* local_map, int64_t* node_g, void* stream) {
  if (B < 0) return PYGAMD_ERR_INVALID_ARG;
  if (B == 0) return PYGAMD_OK;
  if (!seeds || !epoch_dev || !local_map || !node_g) return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    hipLaunchKernelGGL((slots_seed_kernel<IdxT>), dim3(static_cast<unsigned>(ceil_div(B, kBlock))),
                       dim3(kBlock), 0, as_stream(stream), static_cast<const IdxT*>(seeds), B,
                       epoch_dev, reinterpret_cast<long long*>(local_map), node_g);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_slots_sample(const void* colptr, const void* row, int idx_dtype, const int64_t* node_g,
                        int64_t frontier_base, int64_t n_frontier, int fanout, int64_t slot_base,
                        int64_t B, uint64_t seed, int hop, const int64_t* epoch_dev,
                        int64_t* local_map, int64_t* src_g, int32_t* row_end, float* inv_cnt,
                        void* stream) {
  if (n_frontier < 0 || fanout < 1 || fanout > kSlotMaxFanout || frontier_base < 0 ||
      slot_base < 0 || B < 0)
    return PYGAMD_ERR_INVALID_ARG;
  if ((slot_base + n_frontier * fanout + B) >= (static_cast<int64_t>(1) << 31))
    return PYGAMD_ERR_UNSUPPORTED;  // batch-local ids are 32-bit
  if (n_frontier == 0) return PYGAMD_OK;
  if (!colptr || !row || !node_g || !epoch_dev || !local_map || !src_g || !row_end || !inv_cnt)
    return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    auto launch = [&](auto kernel, int per_wave) {
      hipLaunchKernelGGL(kernel,
                         dim3(static_cast<unsigned>(
                             ceil_div(n_frontier, static_cast<int64_t>(kWavesPerBlock) * per_wave))),
                         dim3(kBlock), 0, as_stream(stream), static_cast<const IdxT*>(colptr),
                         static_cast<const IdxT*>(row), node_g, frontier_base, n_frontier, fanout,
                         slot_base, B, seed, hop, epoch_dev,
                         reinterpret_cast<long long*>(local_map), src_g, row_end, inv_cnt);
    };
    if (fanout <= 16) launch(slots_sample_kernel<IdxT, 16>, 4);
    else if (fanout <= 32) launch(slots_sample_kernel<IdxT, 32>, 2);
    else launch(slots_sample_kernel<IdxT, 64>, 1);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_slots_resolve(const int64_t* src_g, int64_t slot_base, int64_t n_slots, int64_t B,
                         const int64_t* epoch_dev, const int64_t* local_map, int32_t* src_id,
                         int64_t* node_g, int32_t* const* counts, int n_counts, void* stream) {
  if (n_slots < 0 || slot_base < 0 || n_counts < 0 || n_counts > kSlotMaxHops)
    return PYGAMD_ERR_INVALID_ARG;
  if (n_slots == 0) return PYGAMD_OK;
  if (!src_g || !epoch_dev || !local_map || !src_id || !node_g || (n_counts > 0 && !counts))
    return PYGAMD_ERR_INVALID_ARG;
  SlotCounters c = {};
  for (int i = 0; i < n_counts; ++i) {
    if (!counts[i]) return PYGAMD_ERR_INVALID_ARG;
    c.p[i] = counts[i];
  }
  hipLaunchKernelGGL(slots_resolve_kernel, dim3(static_cast<unsigned>(ceil_div(n_slots, kBlock))),
                     dim3(kBlock), 0, as_stream(stream), src_g, slot_base, n_slots, B, epoch_dev,
                     reinterpret_cast<const long long*>(local_map), src_id, node_g, c, n_counts);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

int pygamd_slots_gather(const float* x, int64_t ldx, int64_t F, const int64_t* node_g,
                        int64_t n_rows, float* out, int64_t ldo, void* stream) {
  if (n_rows < 0 || F < 0 || ldx < F || ldo < F) return PYGAMD_ERR_INVALID_ARG;
  if (n_rows == 0 || F == 0) return PYGAMD_OK;
  if (!x || !node_g || !out) return PYGAMD_ERR_INVALID_ARG;
  if ((F % 4) || (ldx % 4) || (ldo % 4) || (reinterpret_cast<uintptr_t>(x) & 15u) ||
      (reinterpret_cast<uintptr_t>(out) & 15u))
    return PYGAMD_ERR_UNSUPPORTED;
  const int64_t threads = n_rows * (F / 4);
  hipLaunchKernelGGL(slots_gather_kernel, dim3(static_cast<unsigned>(ceil_div(threads, kBlock))),
                     dim3(kBlock), 0, as_stream(stream), x, ldx, F, node_g, n_rows, out, ldo);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

/* Transposed CSRs of the first n_slots[c] slots, c < n_csr (n_slots descending: CSR 0 covers the
 * most hops): counts[c] (from pygamd_slots_resolve) -> ptr[c] by one scan launch, col[c] by one
 * fill launch; cursor[c] must be zero (the caller clears counts and cursors with one memset). */
int pygamd_slots_transpose(const int64_t* src_g, const int32_t* src_id, int hops,
                           const int32_t* fanout, const int64_t* bases /*[hops + 2]*/, int n_csr,
                           const int64_t* n_slots, const int64_t* n_src_rows,
                           int32_t* const* counts, int32_t* const* cursor, int32_t* const* ptr,
                           int32_t* const* col, void* stream) {
  if (hops < 1 || hops > kSlotMaxHops || n_csr < 0 || n_csr > kSlotMaxHops)
    return PYGAMD_ERR_INVALID_ARG;
  if (n_csr == 0) return PYGAMD_OK;
  if (!src_g || !src_id || !fanout || !bases || !n_slots || !n_src_rows || !counts || !cursor ||
      !ptr || !col)
    return PYGAMD_ERR_INVALID_ARG;
  SlotScans sc = {};
  SlotFill fl = {};
  for (int c = 0; c < n_csr; ++c) {
    if (!counts[c] || !cursor[c] || !ptr[c] || !col[c] || n_src_rows[c] < 0 || n_slots[c] < 0 ||
        (c > 0 && n_slots[c] > n_slots[c - 1]))
      return PYGAMD_ERR_INVALID_ARG;
    sc.in[c] = counts[c];
    sc.out[c] = ptr[c];
    sc.n[c] = n_src_rows[c];
    fl.ptr[c] = ptr[c];
    fl.cursor[c] = cursor[c];
    fl.col[c] = col[c];
    fl.n_slots[c] = n_slots[c];
  }
  const int64_t B = bases[1];
  for (int h = 0; h <= hops; ++h) {
    fl.nbase[h] = bases[h];
    fl.ebase[h] = bases[h + 1] - B;
    if (h < hops) {
      if (fanout[h] < 1) return PYGAMD_ERR_INVALID_ARG;
      fl.fanout[h] = fanout[h];
    }
  }
  fl.hops = hops;
  fl.n_csr = n_csr;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(slots_scan_kernel, dim3(kScanChunks, static_cast<unsigned>(n_csr)),
                     dim3(kScanBlock), 0, st, sc);
  PYGAMD_LAUNCH_CHECK();
  if (n_slots[0] > 0) {
    hipLaunchKernelGGL(slots_fill_kernel,
                       dim3(static_cast<unsigned>(ceil_div(n_slots[0], kBlock))), dim3(kBlock), 0,
                       st, src_g, src_id, fl);
    PYGAMD_LAUNCH_CHECK();
  }
  return PYGAMD_OK;
}

}  // extern "C"
