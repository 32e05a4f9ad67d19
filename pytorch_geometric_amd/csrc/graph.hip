// graph.hip — integer side of the path: index_sort, index2ptr/ptr2index, range checks and the hub
// plan of a CSR handle.  Everything here is bit-exact by construction (stable LSD radix sort,
// boundary-detection for the pointer array) and hand-written: no library primitive.  See include/pyg_amd.h for the reference call sites.
#include <cstring>
#include <limits>
#include <type_traits>

#include "common.h"
#include "scan_device.h"

namespace pygamd {

static int bits_for(int64_t max_value, int key_bits) {
  if (max_value < 0) return key_bits;
  int b = 1;
  while (b < key_bits - 1 && (static_cast<int64_t>(1) << b) <= max_value) ++b;
  return b;
}

// ---- index_sort: stable LSD radix sort of (key, position) pairs, 8 bits per pass ---------------
// Per pass three steps: (1) every workgroup counts the digits of its tile of 4096 keys into
// counts[digit][workgroup]; (2) an exclusive scan over that table (digit-major, workgroup-minor)
// IS the destination of the first key of each (digit, workgroup) run; (3) every workgroup re-reads
// its tile in order and places each key at run start + its rank among the tile's earlier keys of
// the same digit.  The rank is taken 256 keys at a time: inside a wave by eight ballots (the lanes
// that agree with me on every digit bit, restricted to the lanes before me), across the four waves
// through an LDS table.  Keys of one digit keep their order in every pass, so the sort is stable
// and the permutation is the one torch.sort(stable=True) returns (tests/test_gpu_graph.py).
// Passes only cover the bits max_value can set: 22 bits of node ids = 3 passes.
constexpr int kSortItems = 16;                   // keys per thread
constexpr int kSortTile = kBlock * kSortItems;   // 4096 keys per workgroup
constexpr int kRadix = 256;
static_assert(kBlock == kRadix, "one thread per digit in the table updates");

template <typename KeyT>
__global__ void __launch_bounds__(kBlock)
    radix_count_kernel(const KeyT* __restrict__ keys, int64_t n, int shift,
                       uint32_t* __restrict__ counts, int64_t n_tiles) {
  __shared__ uint32_t hist[kRadix];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kSortTile;
  const int lane = lane_id();
#pragma unroll 4
  for (int j = 0; j < kSortItems; ++j) {
    const int64_t i = base + static_cast<int64_t>(j) * kBlock + threadIdx.x;
    const bool valid = i < n;
    const unsigned d = valid ? static_cast<unsigned>((keys[i] >> shift) & (kRadix - 1)) : 0u;
    // one LDS atomic per distinct digit of the wave (hub-heavy graphs repeat a digit often)
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const uint64_t m = __ballot((d >> bit) & 1u);
      peers &= ((d >> bit) & 1u) ? m : ~m;
    }
    if (valid && (peers & ((1ull << lane) - 1)) == 0)
      atomicAdd(&hist[d], static_cast<uint32_t>(__popcll(peers)));
  }
  __syncthreads();
  counts[static_cast<int64_t>(threadIdx.x) * n_tiles + blockIdx.x] = hist[threadIdx.x];
}

// IOTA: the values are the positions themselves (first pass)
template <typename KeyT, bool IOTA>
__global__ void __launch_bounds__(kBlock)
    radix_place_kernel(const KeyT* __restrict__ kin, const int64_t* __restrict__ vin, int64_t n,
                       int shift, const uint32_t* __restrict__ starts, int64_t n_tiles,
                       KeyT* __restrict__ kout, int64_t* __restrict__ vout) {
  __shared__ uint32_t next[kRadix];                    // destination of a digit's next key
  __shared__ uint32_t wave_cnt[kWavesPerBlock][kRadix];
  next[threadIdx.x] = starts[static_cast<int64_t>(threadIdx.x) * n_tiles + blockIdx.x];
#pragma unroll
  for (int w = 0; w < kWavesPerBlock; ++w) wave_cnt[w][threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kSortTile;
  const int lane = lane_id(), wave = wave_in_block();
  for (int j = 0; j < kSortItems; ++j) {
    const int64_t i = base + static_cast<int64_t>(j) * kBlock + threadIdx.x;
    const bool valid = i < n;
    const KeyT k = valid ? kin[i] : static_cast<KeyT>(0);
    const unsigned d = static_cast<unsigned>((k >> shift) & (kRadix - 1));
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const uint64_t m = __ballot((d >> bit) & 1u);
      peers &= ((d >> bit) & 1u) ? m : ~m;
    }
    const uint32_t rank = static_cast<uint32_t>(__popcll(peers & ((1ull << lane) - 1)));
    if (valid && rank == 0) wave_cnt[wave][d] = static_cast<uint32_t>(__popcll(peers));
    __syncthreads();
    if (valid) {
      uint32_t pos = next[d] + rank;
      for (int w = 0; w < wave; ++w) pos += wave_cnt[w][d];
      kout[pos] = k;
      vout[pos] = IOTA ? i : vin[i];
    }
    __syncthreads();
    uint32_t add = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) {
      add += wave_cnt[w][threadIdx.x];
      wave_cnt[w][threadIdx.x] = 0;
    }
    next[threadIdx.x] += add;
    __syncthreads();
  }
}

static int64_t sort_tiles_of(int64_t n) { return ceil_div(n, kSortTile); }
static size_t align256(size_t b) { return (b + 255) & ~static_cast<size_t>(255); }

// workspace: [second key buffer][second value buffer][digit table][scan scratch]
static size_t index_sort_ws_bytes(int64_t n, size_t key_size) {
  const int64_t table = kRadix * sort_tiles_of(n);
  return align256(static_cast<size_t>(n) * key_size) + align256(static_cast<size_t>(n) * 8) +
         align256(static_cast<size_t>(table) * 4) + align256(scan_scratch_bytes(table));
}

template <typename IdxT>
static int index_sort_impl(const void* keys_in, int64_t n, int64_t max_value, void* keys_out,
                           int64_t* perm_out, void* ws, size_t* ws_bytes, hipStream_t st) {
  // Keys are non-negative, so they sort identically as unsigned; radix passes only cover the
  // bits max_value can set.
  using KeyT = typename std::make_unsigned<IdxT>::type;
  (void)ws_bytes;
  if (n >= (static_cast<int64_t>(1) << 32)) return PYGAMD_ERR_INVALID_ARG;  // 32-bit positions
  const int end_bit = bits_for(max_value, sizeof(IdxT) * 8);
  const int passes = (end_bit + 7) / 8;
  const int64_t tiles = sort_tiles_of(n), table = kRadix * tiles;
  char* wp = static_cast<char*>(ws);
  KeyT* k2 = reinterpret_cast<KeyT*>(wp);
  wp += align256(static_cast<size_t>(n) * sizeof(KeyT));
  int64_t* v2 = reinterpret_cast<int64_t*>(wp);
  wp += align256(static_cast<size_t>(n) * 8);
  uint32_t* counts = reinterpret_cast<uint32_t*>(wp);
  wp += align256(static_cast<size_t>(table) * 4);
  uint32_t* sums = reinterpret_cast<uint32_t*>(wp);
  KeyT* kout = static_cast<KeyT*>(keys_out);
  const KeyT* ksrc = static_cast<const KeyT*>(keys_in);
  const int64_t* vsrc = nullptr;
  const dim3 grid(static_cast<unsigned>(tiles)), block(kBlock);
  for (int p = 0; p < passes; ++p) {
    // the last pass lands in the caller's buffers; the passes before it alternate
    const bool to_out = ((passes - 1 - p) & 1) == 0;
    KeyT* kdst = to_out ? kout : k2;
    int64_t* vdst = to_out ? perm_out : v2;
    const int shift = 8 * p;
    hipLaunchKernelGGL((radix_count_kernel<KeyT>), grid, block, 0, st, ksrc, n, shift, counts,
                       tiles);
    PYGAMD_LAUNCH_CHECK();
    const int rc = exclusive_scan_u32(counts, table, sums, st);
    if (rc != PYGAMD_OK) return rc;
    if (p == 0) {
      hipLaunchKernelGGL((radix_place_kernel<KeyT, true>), grid, block, 0, st, ksrc, vsrc, n,
                         shift, counts, tiles, kdst, vdst);
    } else {
      hipLaunchKernelGGL((radix_place_kernel<KeyT, false>), grid, block, 0, st, ksrc, vsrc, n,
                         shift, counts, tiles, kdst, vdst);
    }
    PYGAMD_LAUNCH_CHECK();
    ksrc = kdst;
    vsrc = vdst;
  }
  return PYGAMD_OK;
}

// ptr[v] = number of entries < v  (index sorted ascending) = lower_bound(index, v).  One thread
// per OUTPUT row and a binary search over the sorted index: perfectly balanced, also when most
// rows are empty (a sampled mini-batch: the last hop's nodes have no in-edges, so a "fill the
// gap" formulation would leave half a million sequential stores to one thread).
template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    index2ptr_kernel(const IdxT* __restrict__ index, int64_t n, int64_t size,
                     IdxT* __restrict__ ptr) {
  const int64_t v = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (v > size) return;
  int64_t lo = 0, hi = n;  // first position whose value is >= v
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (static_cast<int64_t>(index[mid]) < v) {
      lo = mid + 1;
    } else {
      hi = mid;
    }
  }
  ptr[v] = static_cast<IdxT>(lo);
}

// index[k] = the row r with ptr[r] <= k < ptr[r+1]  (upper_bound - 1 per output element).
template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    ptr2index_kernel(const IdxT* __restrict__ ptr, int64_t size, int64_t n,
                     IdxT* __restrict__ index) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (k >= n) return;
  int64_t lo = 0, hi = size;  // ptr[lo] <= k < ptr[hi]
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (static_cast<int64_t>(ptr[mid]) <= k) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  index[k] = static_cast<IdxT>(lo);
}

__global__ void minmax_init_kernel(int64_t* mm) {
  mm[0] = std::numeric_limits<int64_t>::max();
  mm[1] = -1;
}

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    index_minmax_kernel(const IdxT* __restrict__ index, int64_t n, int64_t* __restrict__ mm) {
  int64_t lo = std::numeric_limits<int64_t>::max();
  int64_t hi = std::numeric_limits<int64_t>::min();
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t v = index[i];
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    const int64_t olo = bcast_lane(lo, lane_id() ^ off);
    const int64_t ohi = bcast_lane(hi, lane_id() ^ off);
    lo = olo < lo ? olo : lo;
    hi = ohi > hi ? ohi : hi;
  }
  if (lane_id() == 0 && hi >= lo) {
    atomicMin(reinterpret_cast<long long*>(&mm[0]), static_cast<long long>(lo));
    atomicMax(reinterpret_cast<long long*>(&mm[1]), static_cast<long long>(hi));
  }
}

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    permute_index_kernel(const IdxT* __restrict__ src, const int64_t* __restrict__ perm,
                         int64_t n, IdxT* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i < n) out[i] = src[perm[i]];
}

// key[e] = major[e] * n + minor[e]; major = row when by_row, else col
template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    edge_key_kernel(const IdxT* __restrict__ row, const IdxT* __restrict__ col, int64_t E,
                    int64_t n, int by_row, int64_t* __restrict__ key) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= E) return;
  const int64_t r = row[e], c = col[e];
  key[e] = by_row ? r * n + c : c * n + r;
}

__global__ void __launch_bounds__(kBlock)
    run_flags_kernel(const int64_t* __restrict__ key, int64_t E, int64_t* __restrict__ flag) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i < E) flag[i] = (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
}

// Decodes sorted keys back into (row, col).  Without `scan` entry i goes to slot i; with the
// inclusive scan of the run flags only the first entry of every run is written, to slot
// scan[i] - 1 (coalesce), and every original edge perm[i] learns its slot through `gid_orig`.
template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    edge_unkey_kernel(const int64_t* __restrict__ key, const int64_t* __restrict__ scan,
                      const int64_t* __restrict__ perm, int64_t E, int64_t n, int by_row,
                      IdxT* __restrict__ out_row, IdxT* __restrict__ out_col,
                      int64_t* __restrict__ gid_orig) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= E) return;
  int64_t slot = i;
  bool write = true;
  if (scan != nullptr) {
    slot = scan[i] - 1;
    write = (i == 0) || (scan[i - 1] != scan[i]);
    if (gid_orig != nullptr) gid_orig[perm != nullptr ? perm[i] : i] = slot;
  }
  if (!write) return;
  const int64_t k = key[i];
  const int64_t major = k / n, minor = k - major * n;
  out_row[slot] = static_cast<IdxT>(by_row ? major : minor);
  out_col[slot] = static_cast<IdxT>(by_row ? minor : major);
}

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    cast_index_kernel(const int64_t* __restrict__ src, int64_t n, IdxT* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i < n) out[i] = static_cast<IdxT>(src[i]);
}

// out[i] = index[i] when it lies in [0, size), else the sentinel `size` (one group past the last:
// a sort by these keys puts such entries behind every real group); *err = 1 if any entry was
// replaced.  One plain store per wave that saw one — nobody reads the flag before the stream does.
template <typename IdxT, typename OutT>
__global__ void __launch_bounds__(kBlock)
    index_guard_kernel(const IdxT* __restrict__ index, int64_t n, int64_t size,
                       OutT* __restrict__ out, int32_t* __restrict__ err) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  bool bad = false;
  if (i < n) {
    const int64_t v = static_cast<int64_t>(index[i]);
    bad = v < 0 || v >= size;
    out[i] = static_cast<OutT>(bad ? size : v);
  }
  if (err != nullptr && __any(bad) && (threadIdx.x & 63) == 0) *err = 1;
}

// ---- hub plan -------------------------------------------------------------------------------
template <typename IdxT>
struct IsHub {
  const IdxT* rowptr;
  int64_t threshold;
  __device__ bool operator()(int64_t r) const {
    return static_cast<int64_t>(rowptr[r + 1]) - static_cast<int64_t>(rowptr[r]) > threshold;
  }
};

// single-block exclusive scan of the chunk counts of the (few) hub rows
template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    hub_chunk_scan_kernel(const IdxT* __restrict__ rowptr, const IdxT* __restrict__ hub_rows,
                          const int64_t* __restrict__ n_hub_dev, int64_t cap, int64_t chunk,
                          IdxT* __restrict__ hub_chunk_ptr, int64_t* __restrict__ n_chunks_dev) {
  __shared__ int64_t carry;
  __shared__ int64_t wave_tot[kWavesPerBlock];
  int64_t n_hub = *n_hub_dev;
  if (n_hub > cap) n_hub = cap;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n_hub; base += kBlock) {
    const int64_t h = base + threadIdx.x;
    int64_t c = 0;
    if (h < n_hub) {
      const int64_t r = hub_rows[h];
      const int64_t deg = static_cast<int64_t>(rowptr[r + 1]) - static_cast<int64_t>(rowptr[r]);
      c = (deg + chunk - 1) / chunk;
    }
    // inclusive scan inside the wave
    int64_t s = c;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int64_t o = bcast_lane(s, (lane_id() - off) & (kWave - 1));
      if (lane_id() >= off) s += o;
    }
    const int w = threadIdx.x >> 6;
    if (lane_id() == kWave - 1) wave_tot[w] = s;
    __syncthreads();
    int64_t wave_off = 0;
    for (int i = 0; i < w; ++i) wave_off += wave_tot[i];
    const int64_t excl = carry + wave_off + s - c;
    if (h < n_hub) hub_chunk_ptr[h] = static_cast<IdxT>(excl);
    __syncthreads();
    if (threadIdx.x == kBlock - 1) carry = excl + c;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    hub_chunk_ptr[n_hub] = static_cast<IdxT>(carry);
    *n_chunks_dev = carry;
  }
}

// hub rows in ascending order: per-workgroup counts (256 rows each), an exclusive scan, and a
// second pass that writes every hub row at its workgroup's start + its rank (ballots)
template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    hub_count_kernel(IsHub<IdxT> pred, int64_t n_rows, uint32_t* __restrict__ counts) {
  __shared__ uint32_t wave_cnt[kWavesPerBlock];
  const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const uint64_t m = __ballot(r < n_rows && pred(r));
  if (lane_id() == 0) wave_cnt[threadIdx.x >> 6] = static_cast<uint32_t>(__popcll(m));
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t c = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) c += wave_cnt[w];
    counts[blockIdx.x] = c;
  }
}

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    hub_write_kernel(IsHub<IdxT> pred, int64_t n_rows, const uint32_t* __restrict__ starts,
                     IdxT* __restrict__ hub_rows) {
  __shared__ uint32_t wave_cnt[kWavesPerBlock];
  const int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const bool hub = r < n_rows && pred(r);
  const uint64_t m = __ballot(hub);
  const int w = threadIdx.x >> 6;
  if (lane_id() == 0) wave_cnt[w] = static_cast<uint32_t>(__popcll(m));
  __syncthreads();
  if (hub) {
    uint32_t pos = starts[blockIdx.x] +
                   static_cast<uint32_t>(__popcll(m & ((1ull << lane_id()) - 1)));
    for (int i = 0; i < w; ++i) pos += wave_cnt[i];
    hub_rows[pos] = static_cast<IdxT>(r);
  }
}

__global__ void hub_total_kernel(const uint32_t* __restrict__ total, int64_t* __restrict__ out) {
  *out = static_cast<int64_t>(*total);
}

static size_t hub_plan_ws_bytes(int64_t n_rows) {
  const int64_t blocks = ceil_div(n_rows, kBlock);
  return 16 + align256(static_cast<size_t>(blocks) * 4) + align256(scan_scratch_bytes(blocks));
}

template <typename IdxT>
static int hub_plan_impl(const void* rowptr_v, int64_t n_rows, int64_t threshold, int64_t chunk,
                         void* hub_rows_v, void* hub_chunk_ptr_v, int64_t cap,
                         int64_t* n_hub_out, int64_t* n_chunks_out, void* ws, size_t ws_bytes,
                         hipStream_t st) {
  const IdxT* rowptr = static_cast<const IdxT*>(rowptr_v);
  IdxT* hub_rows = static_cast<IdxT*>(hub_rows_v);
  IdxT* hub_chunk_ptr = static_cast<IdxT*>(hub_chunk_ptr_v);
  if (n_rows >= (static_cast<int64_t>(1) << 32)) return PYGAMD_ERR_INVALID_ARG;
  if (ws_bytes < hub_plan_ws_bytes(n_rows)) return PYGAMD_ERR_WORKSPACE;
  // workspace layout: [n_hub (int64)] [n_chunks (int64)] [per-workgroup counts] [scan scratch]
  int64_t* counters = static_cast<int64_t*>(ws);
  const int64_t blocks = ceil_div(n_rows, kBlock);
  uint32_t* counts = reinterpret_cast<uint32_t*>(static_cast<char*>(ws) + 16);
  uint32_t* sums = reinterpret_cast<uint32_t*>(static_cast<char*>(ws) + 16 +
                                               align256(static_cast<size_t>(blocks) * 4));
  IsHub<IdxT> pred{rowptr, threshold};
  const dim3 grid(static_cast<unsigned>(blocks)), block(kBlock);
  hipLaunchKernelGGL((hub_count_kernel<IdxT>), grid, block, 0, st, pred, n_rows, counts);
  PYGAMD_LAUNCH_CHECK();
  const int rc = exclusive_scan_u32(counts, blocks, sums, st);
  if (rc != PYGAMD_OK) return rc;
  hipLaunchKernelGGL((hub_write_kernel<IdxT>), grid, block, 0, st, pred, n_rows, counts,
                     hub_rows);
  PYGAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(hub_total_kernel, dim3(1), dim3(1), 0, st, sums + scan_chunks_of(blocks),
                     counters);
  PYGAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL((hub_chunk_scan_kernel<IdxT>), dim3(1), dim3(kBlock), 0, st, rowptr,
                     hub_rows, counters, cap, chunk, hub_chunk_ptr, counters + 1);
  PYGAMD_LAUNCH_CHECK();
  int64_t host[2] = {0, 0};
  PYGAMD_HIP_CHECK(hipMemcpyAsync(host, counters, sizeof(host), hipMemcpyDeviceToHost, st));
  PYGAMD_HIP_CHECK(hipStreamSynchronize(st));
  *n_hub_out = host[0];
  *n_chunks_out = host[1];
  return PYGAMD_OK;
}

}  // namespace pygamd

using namespace pygamd;

extern "C" {

int pygamd_index_sort_workspace_bytes(int idx_dtype, int64_t n, size_t* bytes) {
  if (!bytes || n < 0) return PYGAMD_ERR_INVALID_ARG;
  if (idx_dtype != PYGAMD_IDX_I64 && idx_dtype != PYGAMD_IDX_I32) return PYGAMD_ERR_INVALID_ARG;
  const size_t need = index_sort_ws_bytes(n, idx_dtype == PYGAMD_IDX_I64 ? 8 : 4);
  *bytes = need < 16 ? 16 : need;
  return PYGAMD_OK;
}

int pygamd_index_sort(const void* keys_in, int idx_dtype, int64_t n, int64_t max_value,
                      void* keys_out, int64_t* perm_out, void* workspace, size_t workspace_bytes,
                      void* stream) {
  if (n < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n == 0) return PYGAMD_OK;
  if (!keys_in || !keys_out || !perm_out || !workspace || keys_in == keys_out)
    return PYGAMD_ERR_INVALID_ARG;  // (out of place: the first pass reads keys_in while writing)
  size_t need = 0;
  int rc = pygamd_index_sort_workspace_bytes(idx_dtype, n, &need);
  if (rc != PYGAMD_OK) return rc;
  if (workspace_bytes < need) return PYGAMD_ERR_WORKSPACE;
  size_t ws_bytes = workspace_bytes;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    return index_sort_impl<IdxT>(keys_in, n, max_value, keys_out, perm_out, workspace, &ws_bytes,
                                 as_stream(stream));
  });
}

int pygamd_index2ptr(const void* index, int idx_dtype, int64_t n, int64_t size, void* ptr_out,
                     void* stream) {
  if (n < 0 || size < 0 || !ptr_out) return PYGAMD_ERR_INVALID_ARG;
  if (n > 0 && !index) return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    const unsigned grid = static_cast<unsigned>(ceil_div(size + 1, kBlock));
    hipLaunchKernelGGL((index2ptr_kernel<IdxT>), dim3(grid), dim3(kBlock), 0, as_stream(stream),
                       static_cast<const IdxT*>(index), n, size, static_cast<IdxT*>(ptr_out));
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_ptr2index(const void* ptr, int idx_dtype, int64_t size, int64_t n, void* index_out,
                     void* stream) {
  if (n < 0 || size < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n == 0) return PYGAMD_OK;
  if (!ptr || !index_out || size == 0) return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    const unsigned grid = static_cast<unsigned>(ceil_div(n, kBlock));
    hipLaunchKernelGGL((ptr2index_kernel<IdxT>), dim3(grid), dim3(kBlock), 0, as_stream(stream),
                       static_cast<const IdxT*>(ptr), size, n, static_cast<IdxT*>(index_out));
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_index_minmax(const void* index, int idx_dtype, int64_t n, int64_t* minmax_out,
                        void* stream) {
  if (n < 0 || !minmax_out) return PYGAMD_ERR_INVALID_ARG;
  if (n > 0 && !index) return PYGAMD_ERR_INVALID_ARG;
  hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, as_stream(stream), minmax_out);
  PYGAMD_LAUNCH_CHECK();
  if (n == 0) return PYGAMD_OK;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    int64_t blocks = ceil_div(n, kBlock);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((index_minmax_kernel<IdxT>), dim3(static_cast<unsigned>(blocks)),
                       dim3(kBlock), 0, as_stream(stream), static_cast<const IdxT*>(index), n,
                       minmax_out);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_permute_index(const void* src, int idx_dtype, const int64_t* perm, int64_t n,
                         void* out, void* stream) {
  if (n < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n == 0) return PYGAMD_OK;
  if (!src || !perm || !out) return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    const unsigned grid = static_cast<unsigned>(ceil_div(n, kBlock));
    hipLaunchKernelGGL((permute_index_kernel<IdxT>), dim3(grid), dim3(kBlock), 0,
                       as_stream(stream), static_cast<const IdxT*>(src), perm, n,
                       static_cast<IdxT*>(out));
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_edge_key(const void* row, const void* col, int idx_dtype, int64_t E, int64_t num_nodes,
                    int by_row, int64_t* key_out, void* stream) {
  if (E < 0 || num_nodes < 0) return PYGAMD_ERR_INVALID_ARG;
  if (E == 0) return PYGAMD_OK;
  if (!row || !col || !key_out || num_nodes == 0) return PYGAMD_ERR_INVALID_ARG;
  if (num_nodes > 3037000499LL) return PYGAMD_ERR_INVALID_ARG;  // n * n would overflow int64
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    hipLaunchKernelGGL((edge_key_kernel<IdxT>), dim3(static_cast<unsigned>(ceil_div(E, kBlock))),
                       dim3(kBlock), 0, as_stream(stream), static_cast<const IdxT*>(row),
                       static_cast<const IdxT*>(col), E, num_nodes, by_row, key_out);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_run_flags(const int64_t* key_sorted, int64_t E, int64_t* flag_out, void* stream) {
  if (E < 0) return PYGAMD_ERR_INVALID_ARG;
  if (E == 0) return PYGAMD_OK;
  if (!key_sorted || !flag_out) return PYGAMD_ERR_INVALID_ARG;
  hipLaunchKernelGGL(run_flags_kernel, dim3(static_cast<unsigned>(ceil_div(E, kBlock))),
                     dim3(kBlock), 0, as_stream(stream), key_sorted, E, flag_out);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

int pygamd_edge_unkey(const int64_t* key_sorted, const int64_t* scan, const int64_t* perm,
                      int64_t E, int64_t num_nodes, int by_row, int idx_dtype, void* out_row,
                      void* out_col, int64_t* gid_orig, void* stream) {
  if (E < 0 || num_nodes < 0) return PYGAMD_ERR_INVALID_ARG;
  if (E == 0) return PYGAMD_OK;
  if (!key_sorted || !out_row || !out_col || num_nodes == 0) return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    hipLaunchKernelGGL((edge_unkey_kernel<IdxT>),
                       dim3(static_cast<unsigned>(ceil_div(E, kBlock))), dim3(kBlock), 0,
                       as_stream(stream), key_sorted, scan, perm, E, num_nodes, by_row,
                       static_cast<IdxT*>(out_row), static_cast<IdxT*>(out_col), gid_orig);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_cast_index(const int64_t* src, int64_t n, int idx_dtype, void* out, void* stream) {
  if (n < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n == 0) return PYGAMD_OK;
  if (!src || !out) return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    const unsigned grid = static_cast<unsigned>(ceil_div(n, kBlock));
    hipLaunchKernelGGL((cast_index_kernel<IdxT>), dim3(grid), dim3(kBlock), 0, as_stream(stream),
                       src, n, static_cast<IdxT*>(out));
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_index_guard(const void* index, int idx_dtype, int64_t n, int64_t size, void* out,
                       int out_dtype, int32_t* err, void* stream) {
  if (n < 0 || size < 0) return PYGAMD_ERR_INVALID_ARG;
  if (out_dtype != PYGAMD_IDX_I64 && out_dtype != PYGAMD_IDX_I32) return PYGAMD_ERR_INVALID_ARG;
  if (out_dtype == PYGAMD_IDX_I32 && size > 0x7fffffffLL) return PYGAMD_ERR_INVALID_ARG;
  if (idx_dtype != PYGAMD_IDX_I64 && idx_dtype != PYGAMD_IDX_I32) return PYGAMD_ERR_INVALID_ARG;
  if (n == 0) return PYGAMD_OK;
  if (!index || !out || index == out) return PYGAMD_ERR_INVALID_ARG;
  const unsigned grid = static_cast<unsigned>(ceil_div(n, kBlock));
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    if (out_dtype == PYGAMD_IDX_I64)
      hipLaunchKernelGGL((index_guard_kernel<IdxT, int64_t>), dim3(grid), dim3(kBlock), 0,
                         as_stream(stream), static_cast<const IdxT*>(index), n, size,
                         static_cast<int64_t*>(out), err);
    else
      hipLaunchKernelGGL((index_guard_kernel<IdxT, int32_t>), dim3(grid), dim3(kBlock), 0,
                         as_stream(stream), static_cast<const IdxT*>(index), n, size,
                         static_cast<int32_t*>(out), err);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_cumsum_workspace_bytes(int idx_dtype, int64_t n, size_t* bytes) {
  if (!bytes || n < 0) return PYGAMD_ERR_INVALID_ARG;
  if (idx_dtype != PYGAMD_IDX_I64 && idx_dtype != PYGAMD_IDX_I32) return PYGAMD_ERR_INVALID_ARG;
  const size_t need = scan_scratch_bytes(n, idx_dtype == PYGAMD_IDX_I64 ? 8 : 4);
  *bytes = need < 16 ? 16 : need;
  return PYGAMD_OK;
}

int pygamd_cumsum(const void* in, int idx_dtype, int64_t n, void* out, void* workspace,
                  size_t workspace_bytes, void* stream) {
  if (n < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n == 0) return PYGAMD_OK;
  if (!in || !out || !workspace) return PYGAMD_ERR_INVALID_ARG;
  size_t need = 0;
  const int rc = pygamd_cumsum_workspace_bytes(idx_dtype, n, &need);
  if (rc != PYGAMD_OK) return rc;
  if (workspace_bytes < need) return PYGAMD_ERR_WORKSPACE;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    return cumsum_device<IdxT>(static_cast<const IdxT*>(in), n, static_cast<IdxT*>(out),
                               static_cast<IdxT*>(workspace), as_stream(stream));
  });
}

int pygamd_hub_plan_workspace_bytes(int idx_dtype, int64_t n_rows, size_t* bytes) {
  if (!bytes || n_rows < 0) return PYGAMD_ERR_INVALID_ARG;
  if (idx_dtype != PYGAMD_IDX_I64 && idx_dtype != PYGAMD_IDX_I32) return PYGAMD_ERR_INVALID_ARG;
  *bytes = hub_plan_ws_bytes(n_rows);
  return PYGAMD_OK;
}

int pygamd_hub_plan(const void* rowptr, int idx_dtype, int64_t n_rows, int64_t threshold,
                    int64_t chunk, void* hub_rows, void* hub_chunk_ptr, int64_t cap,
                    int64_t* n_hub_out, int64_t* n_chunks_out, void* workspace,
                    size_t workspace_bytes, void* stream) {
  if (n_rows < 0 || threshold < 1 || chunk < 1 || !n_hub_out || !n_chunks_out)
    return PYGAMD_ERR_INVALID_ARG;
  *n_hub_out = 0;
  *n_chunks_out = 0;
  if (n_rows == 0) return PYGAMD_OK;
  if (!rowptr || !hub_rows || !hub_chunk_ptr || !workspace || workspace_bytes < 32 ||
      cap < n_rows)
    return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    return hub_plan_impl<IdxT>(rowptr, n_rows, threshold, chunk, hub_rows, hub_chunk_ptr, cap,
                               n_hub_out, n_chunks_out, workspace, workspace_bytes,
                               as_stream(stream));
  });
}

}  // extern "C"
