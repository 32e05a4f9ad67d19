// sage_fused.hip — one SAGEConv layer forward in ONE kernel:
//
//   out[i, :] = act( [ aggr_{j -> i} x[j]  |  x_root[i] ] @ [W_l | W_r]^T + b )
//
// i.e. `propagate` (gather -> mean/sum) and `lin_l(agg) + lin_r(x)` + bias (+ ReLU) of
// torch_geometric/nn/conv/sage_conv.py:134-139.  The aggregated tile never makes the HBM round
// trip between an SpMM launch and a GEMM launch: it is produced into LDS by the gather phase and
// consumed from there by the MFMA loop (it is additionally stored once, write-only, when the
// caller needs it for the weight gradient).  And because the gather phase is HBM-bound while the
// transform is MFMA-bound, two workgroups per CU in different phases overlap the two.
//
// Workgroup = 512 threads (8 waves), one tile of 32 destination rows, two workgroups per CU:
//   phase 1  the tile's own (root) rows are copied to LDS; every wave then takes the next row of
//            the tile from an LDS counter and aggregates it with the SpMM's row loop (slot indices
//            staged 64 at a time, LPR lanes x 16 bytes per source row, 8 row loads in flight),
//            leaving the (mean-scaled) row in the LDS tile `agg[32][F_pad + 4]`.  Rows longer than
//            the hub threshold are NOT gathered here: the host runs the two-stage hub kernels
//            first and this kernel copies their result from the global `agg` buffer.
//   phase 2  out tile [32 x Fo] = A [32 x 2F] @ B^T with A = [agg | root rows] (both in LDS), B =
//            the concatenated weight [Fo x 2F]; wave w owns output columns [32 w, 32 w + 32)
//            (Fo <= 256) and one 32 x 32 accumulator; K is walked in chunks of 32 with the
//            fragment layout "16 consecutive floats per lane" (k = s + 16 h, see gemm.hip), which
//            doubles as a coalesced global access: the weight fragments go global -> registers
//            one chunk ahead (every weight byte once per workgroup, L2-resident); no staging and
//            no barrier after the one that closes phase 1; v_mfma_f32_32x32x2_f32.
//   epilogue bias, optional ReLU, 128-byte row segments to `out` (leading dimension given).
// LDS: 2 x 33.3 KB (aggregated + root tile, F = 256) -> 2 workgroups per CU: while one gathers
// (HBM-bound) the other transforms (MFMA-bound).
// Measured at the products shape (scripts/fused_probe.py; SpMM + GEMM as two launches: 17.3 ms at
// F = 256, 7.4 ms at F = 100): this kernel 14.7 / 7.2 ms; weight chunks staged through LDS with
// two barriers per chunk 14.3 / 7.7 ms; root-half fragments fetched from global memory by every
// wave 15.7 ms (L1-bound); ONE persistent 1024-thread workgroup per CU with 12 gather waves
// feeding 4 MFMA waves through two LDS buffers 16.0 / 9.6 ms (a barrier per tile drains the memory
// pipeline; lowering the hub threshold to 128 changed nothing); ONE LDS tile used twice (root half
// of the transform first, then the gather into the same tile, then the aggregated half), which
// fits three workgroups per CU: 14.3 / 7.9 ms — not adopted.
#include "spmm_device.h"

namespace pygamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFTile = 32;    // destination rows per workgroup
constexpr int kFBlock = 512;  // threads per workgroup
constexpr int kFWaves = kFBlock / kWave;
constexpr int kFK = 32;       // k chunk
constexpr int kFLD = kFK + 4; // row stride of the staged chunks
constexpr int kFMaxFo = 256;  // one 32-column block per wave

template <typename IdxT>
struct SageFusedArgs {
  SpmmDev<IdxT> g;              // graph + gather source (x, ldx) + global agg buffer (out, ldo)
  const float* __restrict__ x_root;  // [n_rows, F]
  int64_t ld_root;
  const float* __restrict__ w;       // [Fo, 2F]
  int64_t ldw;
  const float* __restrict__ bias;    // [Fo] or null
  float* __restrict__ y;             // [n_rows, Fo]
  int64_t ldy;
  int Fo, relu, save_agg;
  int f_pad;                         // F rounded up to a multiple of 32
  uint32_t* __restrict__ bits;       // null or [y > 0], one bit per element, 32 x 32 tiles
  int64_t ld_bits;
  const uint32_t* __restrict__ mask_bits;  // null or: y = bit ? y : 0 (same tiled layout) — the
  int64_t ld_mask;                         // ReLU backward of the layer below, when this kernel
                                           // runs a layer's input gradient
  const float* __restrict__ row_scale;     // with y2: y2[i, :] = y[i, :] * row_scale[i]
  float* __restrict__ y2;                  // null or a second, row-scaled copy of the output
  int64_t ldy2;
  uint32_t* __restrict__ zout;             // null or the output once more as compressed rows
  int64_t ldz;                             // (spmm_device.h), the next layer's gather source
  int nbuf;   // specialised kernel: aggregated-tile buffers in LDS
  int probe;  // timing probes only (scripts/fused_probe.py): bit 0 = skip the gather loop (the
              // aggregated tile stays undefined), bit 1 = skip the MFMA loop, bits 2-3 = weight
              // prefetch depth, bit 4 = no weight loads after the first chunks, bit 5 = no LDS
              // fragment reads after the first chunk.  0 in production.
};

// ---- epilogue of one 32 x 32 accumulator: bias, optional ReLU / mask bits, 128-byte row segments
// to `y` (+ the row-scaled copy, + the [y > 0] bits).  Reg e of lane l is
// C[(e & 3) + 8 (e >> 2) + 4 (l >> 5)][l & 31].
template <typename IdxT>
__device__ __forceinline__ void fused_epilogue(const SageFusedArgs<IdxT>& a, const f32x16& acc,
                                               int64_t row0, int wave_col0, int lane,
                                               f32x16& vout) {
  const int li = lane & 31, lh = lane >> 5;
  const int col = wave_col0 + li;
  const bool col_ok = col < a.Fo;
  const float bv = (a.bias && col_ok) ? a.bias[col] : 0.f;
  const int64_t rbase = row0 + 4 * lh;
  float* yp = a.y + rbase * a.ldy + col;
  float* yp2 = a.y2 ? a.y2 + rbase * a.ldy2 + col : nullptr;
  // mask word of row (row0 + li) for this 32-column block: one 128-byte line per wave, fetched
  // before the stores and handed out by ds_bpermute
  uint32_t mword = 0xffffffffu;
  if (a.mask_bits && row0 + li < a.g.n_rows)
    mword = a.mask_bits[((row0 >> 5) * a.ld_mask + (wave_col0 >> 5)) * 32 + li];
  uint32_t my_word = 0;  // lane e < 16: row (e & 3) + 8 (e >> 2); lane 16 + e: that row + 4
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int roff = (e & 3) + 8 * (e >> 2);
    float v = acc[e] + bv;
    if (a.relu) v = (v > 0.f || v != v) ? v : 0.f;  // NaN propagates like torch.relu
    if (a.mask_bits) {  // uniform
      const uint32_t mw = __shfl(mword, roff + 4 * lh, kWave);
      v = ((mw >> li) & 1u) ? v : 0.f;
    }
    const bool ok = col_ok && rbase + roff < a.g.n_rows;
    vout[e] = v;
    if (ok) yp[roff * a.ldy] = v;
    if (yp2 && ok) yp2[roff * a.ldy2] = v * a.row_scale[rbase + roff];
    if (a.bits) {  // uniform.  One ballot = this 32-column block of two rows (lane halves)
      const uint64_t m = __ballot(col_ok && v > 0.f);
      if (lane == e) my_word = static_cast<uint32_t>(m);
      if (lane == 16 + e) my_word = static_cast<uint32_t>(m >> 32);
    }
  }
  if (a.bits && lane < 32) {  // the tile's 32 words of this column block: one 128-byte line
    const int e = lane & 15;
    const int r = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 4);
    if (row0 + r < a.g.n_rows)
      a.bits[((row0 >> 5) * a.ld_bits + (wave_col0 >> 5)) * 32 + r] = my_word;
  }
}

// ---- the output tile once more as compressed rows (spmm_device.h): [8 mask words | kept values].
// Wave w holds the 32 x 32 block of columns [32 w, 32 w + 32) in the accumulator layout above; the
// offset of its values inside a row is the number of kept values in the blocks before it, which the
// waves exchange through `zw` ([8][32] mask words in LDS).  EVERY wave of the workgroup calls this
// (one barrier inside); waves without a column block pass active = false.
template <typename IdxT>
__device__ __forceinline__ void fused_compress_tile(const SageFusedArgs<IdxT>& a, const f32x16& v,
                                                    uint32_t* __restrict__ zw, int64_t row0,
                                                    int wave, int lane, bool active) {
  const int li = lane & 31, lh = lane >> 5;
  const bool col_ok = active && wave * 32 + li < a.Fo;
  uint32_t my_word = 0;  // lane e < 16: row (e & 3) + 8 (e >> 2); lane 16 + e: that row + 4
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const uint64_t m = __ballot(col_ok && __float_as_uint(v[e]) != 0u);
    if (lane == e) my_word = static_cast<uint32_t>(m);
    if (lane == 16 + e) my_word = static_cast<uint32_t>(m >> 32);
  }
  if (lane < 32) {
    const int e = lane & 15;
    zw[wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 4)] = my_word;
  }
  __syncthreads();
  // kept values of row li in the column blocks before this wave's
  int before = 0;
  for (int w = 0; w < wave; ++w) before += __popc(zw[w * 32 + li]);
  // the mask words of rows 4 wave .. 4 wave + 3: 32 contiguous bytes per row
  if (lane < 32) {
    const int r = 4 * wave + (lane >> 3);
    if (row0 + r < a.g.n_rows) a.zout[(row0 + r) * a.ldz + (lane & 7)] = zw[(lane & 7) * 32 + r];
  }
  if (!active) return;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
    const bool keep = col_ok && __float_as_uint(v[e]) != 0u;
    const uint64_t m = __ballot(keep);
    const uint32_t half = lh ? static_cast<uint32_t>(m >> 32) : static_cast<uint32_t>(m);
    const int rank = __popc(half & ((1u << li) - 1u));
    const int pre = __shfl(before, r, kWave);
    if (keep && row0 + r < a.g.n_rows)
      a.zout[(row0 + r) * a.ldz + kZrowHdr + pre + rank] = __float_as_uint(v[e]);
  }
}

// ---- phase 2 + epilogue, shared by both kernels: [32 x Fo] = [agg | x_root] @ w^T from the two
// LDS tiles.  The caller has closed phase 1 with a barrier (both tiles visible to every wave).
template <typename IdxT, int PF>
__device__ __forceinline__ void fused_transform(const SageFusedArgs<IdxT>& a,
                                                const float* __restrict__ agg,
                                                const float* __restrict__ xr, int agg_ld,
                                                int64_t row0, int wave, int lane,
                                                uint32_t* __restrict__ zw = nullptr) {
  const int F = static_cast<int>(a.g.F);
  // ---- phase 2: [32 x Fo] = [agg | x_root] @ w^T.  No staging and no barrier: wave w owns the
  // output columns [32 w, 32 w + 32), so of every weight chunk it needs exactly its own 32 rows x
  // 32 k — and the MFMA operand layout (lane (j, h): 16 consecutive k of row j) IS a coalesced
  // global access pattern (a wave reads 32 full 128-byte lines).  The weight fragments therefore
  // go global -> registers directly, one chunk ahead of the MFMAs (every weight byte once per
  // workgroup); both halves of A come from the LDS tiles.
  const int wave_col0 = wave * 32;
  if (wave_col0 >= a.Fo) {
    if (zw) {  // (only the row-at-a-time kernel passes zw: all of its waves come through here)
      f32x16 none;
#pragma unroll
      for (int e = 0; e < 16; ++e) none[e] = 0.f;
      fused_compress_tile<IdxT>(a, none, zw, row0, wave, lane, false);
    }
    return;
  }
  const int li = lane & 31, lh = lane >> 5;
  const int n_half = a.f_pad / kFK;  // chunks per half (aggregated / root)
  const int n_chunks = 2 * n_half;
  const int col = wave_col0 + li;
  const bool col_ok = col < a.Fo;
  const float* __restrict__ wrow = a.w + static_cast<int64_t>(col_ok ? col : a.Fo - 1) * a.ldw;
  const float* agg_row = agg + li * agg_ld + 16 * lh;
  const float* xr_row = xr + li * agg_ld + 16 * lh;
  f32x4 fb[4], fa[4];
  // weight fragments of chunk c: k = (chunk base) + 16 lh + 4 v + e; columns past F are clamped
  // to a valid address here and zeroed right before use
  auto load_b = [&](int c, f32x4 (&dst)[4]) {
    const bool root = c >= n_half;
    const int kl = (root ? c - n_half : c) * kFK + 16 * lh;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int k = kl + 4 * v;
      dst[v] = *reinterpret_cast<const f32x4*>(wrow + (root ? F : 0) + (k < F ? k : 0));
    }
  };
  f32x16 acc, acc2;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = acc2[e] = 0.f;
  const bool dual = (a.probe & 128) != 0;  // probe: two independent accumulation chains
  auto chunk = [&](int c) {  // MFMAs of chunk c with the weight fragments in `fb`
    const bool root = c >= n_half;
    const int base = (root ? c - n_half : c) * kFK;
    const float* ap = (root ? xr_row : agg_row) + base;
    if (!(a.probe & 32) || c == 0) {
#pragma unroll
      for (int v = 0; v < 4; ++v) fa[v] = *reinterpret_cast<const f32x4*>(ap + 4 * v);
    }
    const int rem = F - base;  // > 0: valid k of this chunk (multiple of 4)
    if (rem < kFK || !col_ok) {  // boundary chunk / padding column: zero B past F (the LDS
      const int kl = base + 16 * lh;  // tiles are zero there already)
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          fb[v][e] = (col_ok && (kl + 4 * v + e < F)) ? fb[v][e] : 0.f;
    }
    // a tail shorter than 16 leaves the upper lane half all zero: only `rem` steps carry data
    const int groups = rem >= 16 ? 4 : rem / 4;  // wave-uniform
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (v < groups) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (dual && (v & 1)) {
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[v][e], fb[v][e], acc2, 0, 0, 0);
          } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[v][e], fb[v][e], acc, 0, 0, 0);
          }
        }
      }
    }
  };
  // weight fragments run PF chunks ahead of the MFMAs through a register ring (the loads share the
  // CU's vector-memory path with the other workgroup's gather: one chunk of lead is not enough)
  const int n_run = (a.probe & 2) ? 0 : n_chunks;
  f32x4 ring[PF][4];
#pragma unroll
  for (int q = 0; q < PF; ++q)
    if (q < n_chunks) load_b(q, ring[q]);
  for (int c0 = 0; c0 < n_run; c0 += PF) {
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int c = c0 + q;
      if (c < n_run) {
#pragma unroll
        for (int v = 0; v < 4; ++v) fb[v] = ring[q][v];
        if (c + PF < n_chunks && !(a.probe & 16)) load_b(c + PF, ring[q]);
        chunk(c);
      }
    }
  }

  if (dual) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] += acc2[e];
  }
  f32x16 vout;
  fused_epilogue<IdxT>(a, acc, row0, wave_col0, lane, vout);
  if (zw) fused_compress_tile<IdxT>(a, vout, zw, row0, wave, lane, true);
}

// aggregated row -> LDS tile (+ global agg buffer); lanes < LPR hold VW features per CH
template <typename IdxT, int VW, int LPR, bool ZSRC = false>
__device__ __forceinline__ void fused_gather_row(const SageFusedArgs<IdxT>& a, int64_t row,
                                                 float* __restrict__ agg_row, int lane) {
  constexpr int CH = 1;
  int fo[CH], head[CH];
  bool fv[CH];
  const int lir = lane % LPR;
  fo[0] = lir * VW;
  fv[0] = fo[0] < a.g.F;
  head[0] = 0;
  float acc[CH][VW];
#pragma unroll
  for (int i = 0; i < VW; ++i) acc[0][i] = 0.f;
  IdxT start = 0, end = 0;
  if (row < a.g.n_rows) {
    start = a.g.rowptr[row];
    end = a.g.rowptr[row + 1];
  }
  const IdxT deg = end - start;
  const bool hub = a.g.hub_threshold > 0 && deg > a.g.hub_threshold;
  if (hub) {  // aggregated by the two-stage hub kernels before this launch
    if (lane < LPR && fv[0]) {
      const Vec<VW> v = load_vec<VW>(a.g.out + row * a.g.ldo + fo[0]);
      store_vec<VW>(agg_row + fo[0], v);
    }
    return;
  }
  // (16 instead of 8 row loads in flight per lane was measured slower here: 14.4 / 7.5 ms)
  spmm_accumulate<IdxT, VW, LPR, CH, ZSRC ? 4 : 0, false>(a.g, start, end, lane, fo, fv, head,
                                                          acc);
  combine_subgroups<VW, LPR, CH>(acc);
  if (lane < LPR && fv[0]) {
    const float cntf = static_cast<float>(deg > 0 ? deg : 1);
    Vec<VW> o;
#pragma unroll
    for (int i = 0; i < VW; ++i) o.v[i] = a.g.mean ? acc[0][i] / cntf : acc[0][i];
    store_vec<VW>(agg_row + fo[0], o);
    if (a.save_agg && row < a.g.n_rows) {
#pragma unroll
      for (int i = 0; i < VW; ++i)
        __builtin_nontemporal_store(o.v[i], a.g.out + row * a.g.ldo + fo[0] + i);
    }
  }
}

// ZSRC: the gather source is a block of compressed rows (64 lanes x 4 columns per row, two
// dependent loads per source row: more registers, two workgroups per CU — what the LDS tiles of
// F = 256 allow anyway)
template <typename IdxT, int LPR, bool ZSRC = false>
__global__ void __launch_bounds__(kFBlock, 4)
    sage_fused_fwd_kernel(SageFusedArgs<IdxT> a) {
  extern __shared__ __align__(16) float smem[];
  __shared__ int next_row;
  __shared__ uint32_t zw[kFWaves * 32];
  const int agg_ld = a.f_pad + 4;
  float* agg = smem;                    // [32][f_pad + 4]  aggregated rows
  float* xr = smem + kFTile * agg_ld;   // [32][f_pad + 4]  root rows of the tile
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t tile = xcd_logical_block();
  const int64_t row0 = tile * kFTile;
  if (row0 >= a.g.n_rows) return;
  const int F = static_cast<int>(a.g.F);

  // ---- phase 1: the aggregated tile and the tile's own (root) rows -> LDS.  Padding columns
  // [F, f_pad) are zeroed once.
  if (threadIdx.x == 0) next_row = 0;
  if (a.f_pad > F) {
    const int padw = a.f_pad - F;
    for (int t = threadIdx.x; t < 2 * kFTile * padw; t += kFBlock) {
      const int r = t / padw;
      smem[r * agg_ld + F + (t - r * padw)] = 0.f;
    }
  }
  {
    const int units = F / 4;  // 16-byte pieces per row
    for (int t = threadIdx.x; t < kFTile * units; t += kFBlock) {
      const int r = t / units;
      const int u = t - r * units;
      int64_t rr = row0 + r;
      rr = rr < a.g.n_rows ? rr : a.g.n_rows - 1;
      *reinterpret_cast<f32x4*>(xr + r * agg_ld + 4 * u) =
          *reinterpret_cast<const f32x4*>(a.x_root + rr * a.ld_root + 4 * u);
    }
  }
  __syncthreads();  // next_row armed
  // (probe bit 11: the gather phase at raised issue priority, the transform phase at 0 — VALU /
  // VMEM issue on a SIMD is arbitrated by priority, then age, and a wave issuing dependent MFMAs
  // back to back otherwise wins every slot it asks for)
  if (a.probe & 2048) __builtin_amdgcn_s_setprio(2);
  for (; !(a.probe & 1);) {  // rows are handed out one by one: long and short rows balance over
    int r = 0;               // the 8 waves
    if (lane == 0) r = atomicAdd(&next_row, 1);
    r = __builtin_amdgcn_readfirstlane(r);
    if (r >= kFTile) break;
    fused_gather_row<IdxT, 4, LPR, ZSRC>(a, row0 + r, agg + r * agg_ld, lane);
  }

  if (a.probe & 2048) __builtin_amdgcn_s_setprio(0);
  __syncthreads();  // phase 1 complete: both tiles visible to every wave
  fused_transform<IdxT, 1>(a, agg, xr, agg_ld, row0, wave, lane, a.zout ? zw : nullptr);
}

// ---- v2: the gather phase as a software-pipelined stream ------------------------------------------
// The row-at-a-time phase 1 above pays three dependent memory latencies per destination row
// (rowptr -> slot indices -> source rows) and drains its loads at every batch, with only 16 waves
// per CU to hide them (an SpMM launch has 32): at the products shape the two phases of the kernel
// ran back to back on a CU (48 us per tile = 35 us gather + 13.7 us MFMA) although two workgroups
// share it.  Here the dependent chain is paid ONCE PER TILE and the row loads never drain:
//   1. every wave reads the tile's 33 row pointers into its lanes and scans the non-hub degrees
//      (compacted slot offsets cp[0..32]); the tile's column indices — one contiguous run of the
//      CSR array — go to LDS as int32 with one coalesced pass of the whole workgroup;
//   2. the 32 rows are split into 8 contiguous runs of about equal slot count, one per wave;
//   3. a wave walks its run as a sequence of UNITS (one row, STEP = U * 64/LPR consecutive slots,
//      U row loads per lane) through two register buffers: the loads of unit i+1 are issued before
//      unit i is added up, across row boundaries — 2 x U x 1 KiB in flight per wave at all times,
//      issued unconditionally (clamped slot, select at the add) so that the compiler's vmcnt
//      bookkeeping stays exact.  A finished row is scaled (mean) and written to the LDS tile.
// The order of the additions is the SpMM's (slot order per lane group, groups combined by the same
// butterfly): the result is bitwise that of pygamd_spmm_csr + pygamd_linear_forward.
// Tiles with more than kFCap non-hub slots read their indices from global memory instead (same
// pipeline, rare).  The aggregated tile is written to global memory (save_agg) from LDS after the
// barrier, coalesced, so that phase 1 contains no global store.
constexpr int kFCap = 3072;  // column indices staged per tile (12 KiB)

template <typename IdxT, int LPR, bool LDS_IDX>
__device__ __forceinline__ void stream_gather(const SageFusedArgs<IdxT>& a, float* __restrict__ agg,
                                              int agg_ld, const int32_t* __restrict__ cidx,
                                              IdxT rp_l, int cp_l, int rb, int re, int lane) {
  constexpr int VW = 4, U = 8;
  constexpr int EPI = kWave / LPR, STEP = U * EPI;
  const int sub = lane / LPR;
  const int fo = (lane % LPR) * VW;
  const bool fv = fo < static_cast<int>(a.g.F);
  const float* __restrict__ xb = a.g.x + (fv ? fo : 0);  // loads are unconditional
  // unit iterator (all wave-uniform): row it_r, slots [it_j, it_j + STEP) of its it_deg, first
  // compacted slot it_base
  int it_r = rb - 1, it_j = 0, it_deg = 0, it_base = 0;
  bool done = false;
  auto advance = [&]() -> bool {
    if (done) return false;
    int j = it_j + STEP, r = it_r, deg = it_deg, base = it_base;
    while (j >= deg) {
      ++r;
      if (r >= re) {
        done = true;
        return false;
      }
      base = bcast_uniform(cp_l, r);
      deg = bcast_uniform(cp_l, r + 1) - base;
      j = 0;
    }
    it_r = r;
    it_j = j;
    it_deg = deg;
    it_base = base;
    return true;
  };
  struct Unit {
    int row, j0, deg;
    bool live;
  };
  // (after the last unit the iterator keeps its coordinates: a dead unit re-issues the loads of
  // the last live one — cache hits — so that every pass of the loop issues exactly U loads)
  auto issue = [&](Unit& un, Vec<VW> (&b)[U], bool live) {
    un.row = it_r;
    un.j0 = it_j;
    un.deg = it_deg;
    un.live = live;
    IdxT g0 = 0;
    if constexpr (!LDS_IDX) g0 = bcast_uniform(rp_l, it_r);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int k = it_j + u * EPI + sub;
      k = k < it_deg ? k : it_deg - 1;
      int64_t c;
      if constexpr (LDS_IDX) {
        c = cidx[it_base + k];
      } else {
        c = static_cast<int64_t>(a.g.col[g0 + k]);
      }
      b[u] = load_vec<VW>(xb + c * a.g.ldx);
    }
    // every load of the unit is issued before the first add of the previous one (hipcc otherwise
    // starts the adds between the loads and parks the wave on the oldest load in flight)
    __builtin_amdgcn_sched_barrier(0);
  };
  float acc[1][VW];
#pragma unroll
  for (int i = 0; i < VW; ++i) acc[0][i] = 0.f;
  auto consume = [&](const Unit& un, const Vec<VW> (&b)[U]) {
    const int lim = un.live ? un.deg : 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool valid = un.j0 + u * EPI + sub < lim;
#pragma unroll
      for (int i = 0; i < VW; ++i) acc[0][i] += valid ? b[u].v[i] : 0.f;
    }
    if (un.live && un.j0 + STEP >= un.deg) {  // the row is complete
      combine_subgroups<VW, LPR, 1>(acc);
      if (lane < LPR && fv) {
        const float cntf = static_cast<float>(un.deg);
        Vec<VW> o;
#pragma unroll
        for (int i = 0; i < VW; ++i) o.v[i] = a.g.mean ? acc[0][i] / cntf : acc[0][i];
        store_vec<VW>(agg + un.row * agg_ld + fo, o);
      }
#pragma unroll
      for (int i = 0; i < VW; ++i) acc[0][i] = 0.f;
    }
  };
  Unit ua, ub;
  Vec<VW> va[U], vb[U];
  if (!advance()) return;
  issue(ua, va, true);
  for (;;) {  // invariant: `ua` is live and its loads are in flight
    issue(ub, vb, advance());
    consume(ua, va);
    const bool more = advance();
    issue(ua, va, more);
    consume(ub, vb);
    if (!more) break;
  }
}

template <typename IdxT, int LPR, int PF>
__global__ void __launch_bounds__(kFBlock, 4) sage_fused_stream_kernel(SageFusedArgs<IdxT> a) {
  extern __shared__ __align__(16) float smem[];
  const int agg_ld = a.f_pad + 4;
  float* agg = smem;                    // [32][f_pad + 4]  aggregated rows
  float* xr = smem + kFTile * agg_ld;   // [32][f_pad + 4]  root rows of the tile
  int32_t* cidx = reinterpret_cast<int32_t*>(smem + 2 * kFTile * agg_ld);  // [kFCap]
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t tile = xcd_logical_block();
  const int64_t row0 = tile * kFTile;
  if (row0 >= a.g.n_rows) return;
  const int F = static_cast<int>(a.g.F);

  // ---- the tile's row pointers, one per lane (rows past n_rows repeat the last pointer: degree 0)
  IdxT rp_l = 0;
  if (lane <= kFTile) {
    int64_t rr = row0 + lane;
    rr = rr < a.g.n_rows ? rr : a.g.n_rows;
    rp_l = a.g.rowptr[rr];
  }
  // root rows -> LDS (issued before anything waits on the row pointers)
  {
    const int units = F / 4;  // 16-byte pieces per row
    for (int t = threadIdx.x; t < kFTile * units; t += kFBlock) {
      const int r = t / units;
      const int u = t - r * units;
      int64_t rr = row0 + r;
      rr = rr < a.g.n_rows ? rr : a.g.n_rows - 1;
      *reinterpret_cast<f32x4*>(xr + r * agg_ld + 4 * u) =
          *reinterpret_cast<const f32x4*>(a.x_root + rr * a.ld_root + 4 * u);
    }
  }
  if (a.f_pad > F) {  // padding columns [F, f_pad) of both tiles are zeroed once
    const int padw = a.f_pad - F;
    for (int t = threadIdx.x; t < 2 * kFTile * padw; t += kFBlock) {
      const int r = t / padw;
      smem[r * agg_ld + F + (t - r * padw)] = 0.f;
    }
  }
  const IdxT rp_n = bcast_lane(rp_l, lane + 1 < kWave ? lane + 1 : lane);
  const int64_t deg_l = lane < kFTile ? static_cast<int64_t>(rp_n - rp_l) : 0;
  const bool hub_l = a.g.hub_threshold > 0 && deg_l > a.g.hub_threshold;
  const int act_l = hub_l ? 0 : static_cast<int>(deg_l);
  int inc = act_l;  // inclusive scan over the lanes
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int t = __shfl_up(inc, off, kWave);
    if (lane >= off) inc += t;
  }
  const int cp_l = inc - act_l;  // compacted first slot of row `lane`; lane 32: the tile's total
  const int total = bcast_uniform(cp_l, kFTile);
  const bool any_hub = __ballot(hub_l) != 0;
  const bool staged = total <= kFCap;

  // ---- rows that are not gathered here (wave w looks after rows w, w + 8, ...): hub rows come
  // from the global agg buffer (two-stage hub kernels, before this launch), empty rows are zero;
  // with hub rows in the tile the indices are staged row by row, otherwise in one flat pass
  const IdxT g_first = bcast_uniform(rp_l, 0);
  if (staged && !any_hub) {
    for (int k = threadIdx.x; k < total; k += kFBlock)
      cidx[k] = static_cast<int32_t>(__builtin_nontemporal_load(&a.g.col[g_first + k]));
  }
  for (int r = wave; r < kFTile; r += kFWaves) {
    const IdxT g0 = bcast_uniform(rp_l, r);
    const int64_t deg = static_cast<int64_t>(bcast_uniform(rp_l, r + 1) - g0);
    const bool hub = a.g.hub_threshold > 0 && deg > a.g.hub_threshold;
    float* arow = agg + r * agg_ld;
    if (hub) {
      const float* __restrict__ src = a.g.out + (row0 + r) * a.g.ldo;
      for (int f = 4 * lane; f < F; f += 4 * kWave)
        *reinterpret_cast<f32x4*>(arow + f) = *reinterpret_cast<const f32x4*>(src + f);
    } else if (deg == 0) {
      for (int f = 4 * lane; f < F; f += 4 * kWave)
        *reinterpret_cast<f32x4*>(arow + f) = f32x4{0.f, 0.f, 0.f, 0.f};
    } else if (staged && any_hub) {
      const int s0 = bcast_uniform(cp_l, r);
      for (int i = lane; i < deg; i += kWave)
        cidx[s0 + i] = static_cast<int32_t>(__builtin_nontemporal_load(&a.g.col[g0 + i]));
    }
  }
  // ---- this wave's run of rows: [rb, re) = the rows whose first slot lies in its share
  const int t_lo = static_cast<int>(static_cast<int64_t>(total) * wave / kFWaves);
  const int t_hi = static_cast<int>(static_cast<int64_t>(total) * (wave + 1) / kFWaves);
  const int rb = __popcll(__ballot(lane < kFTile && cp_l < t_lo));
  const int re = __popcll(__ballot(lane < kFTile && cp_l < t_hi));
  __syncthreads();  // indices staged
  if (a.probe & 1) {
  } else if (staged) {
    stream_gather<IdxT, LPR, true>(a, agg, agg_ld, cidx, rp_l, cp_l, rb, re, lane);
  } else {
    stream_gather<IdxT, LPR, false>(a, agg, agg_ld, cidx, rp_l, cp_l, rb, re, lane);
  }
  __syncthreads();  // phase 1 complete: both tiles visible to every wave
  if (a.save_agg) {  // the aggregated rows, once, for the weight gradient (write-only)
    const int units = F / 4;
    for (int t = threadIdx.x; t < kFTile * units; t += kFBlock) {
      const int r = t / units;
      const int u = t - r * units;
      if (row0 + r < a.g.n_rows) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(agg + r * agg_ld + 4 * u);
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(a.g.out + (row0 + r) * a.g.ldo +
                                                                4 * u));
      }
    }
  }
  fused_transform<IdxT, PF>(a, agg, xr, agg_ld, row0, wave, lane);
}

// ---- v3: producer / consumer waves, persistent workgroup ------------------------------------------
// In the two kernels above a workgroup alternates between its HBM-bound phase and its MFMA-bound
// phase, and whether the two phases of DIFFERENT workgroups overlap on a CU is left to chance: at
// the products shape the layer runs 14.1 ms against 11.8 ms with the MFMA loop skipped and 6.4 ms
// with the gather skipped (scripts/fused_probe.py, profiles/r03_fused_phase_probe.txt).  Here the
// overlap holds by construction.  ONE workgroup of 16 waves per CU walks tiles b, b + G, b + 2G...:
//   waves [0, NG)        GATHER.  They draw destination rows one by one from an LDS ticket counter
//                        that runs ACROSS tile boundaries (row ticket t = tile t / 32 of this
//                        workgroup, row t % 32), aggregate each with the SpMM's row loop into one of
//                        `nbuf` LDS tiles, and count the finished row on that tile's `done` counter.
//                        No barrier, nothing drains between tiles; a wave only waits (s_sleep poll)
//                        when the tile `nbuf` tiles back has not been consumed yet.
//   waves [NG, 16)       TRANSFORM (one per SIMD with NM = 4).  Per tile: the root half of K first —
//                        A fragments straight from global memory (the tile's own rows, one 128-byte
//                        line per row and chunk), no dependence on the gather — then wait for
//                        done == 32 * (use + 1), the aggregated half with A from the LDS tile,
//                        release the tile (`free` counter), epilogue.  Fragments run through a
//                        register ring D half-chunks ahead of the MFMAs.
// The MFMA waves need about a third of a tile's gather time, so the gather waves set the pace:
// the layer costs what its aggregation costs.  Sums run root half first, so results equal the
// two-launch path to rounding (not bitwise like v1 / v2).
constexpr int kSBlock = 1024;
constexpr int kSWaves = kSBlock / kWave;

__device__ __forceinline__ int lds_counter_load(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// every earlier LDS access of this wave is complete before the counter moves (the LDS queue is in
// order per CU; the wait is on lgkmcnt only — a workgroup release fence would also wait for the
// gather's global loads and stores).
// The add is executed by ALL lanes without a branch: lane 0 targets the counter, lane l > 0 its own
// word of `sink`.  With `if (lane == 0)` around the atomics, hipcc threads the branch at the end of
// one loop iteration into the identical branch at the head of the next (ticket draw), and the
// readfirstlane between them ends up evaluated by lanes 1..63 alone on their constant 0: those
// lanes then spin on ticket 0 forever (seen on the device: the kernel never ended).
__device__ __forceinline__ int lds_counter_add(int* p, int* sink, int lane) {
  int* q = lane == 0 ? p : sink + lane;
  return __hip_atomic_fetch_add(q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_counter_signal(int* p, int* sink, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  lds_counter_add(p, sink, lane);
}
// A poll that never succeeds would hang the GPU: after ~2 s of polling the workgroup gives up (every
// later wait of the workgroup returns at once; its results are then wrong, which the callers' tests
// catch — the protocol has no cycle, see the kernel's comment, so this is a guard, not a path).
constexpr int kSpinLimit = 1 << 24;
__device__ __forceinline__ void lds_counter_wait(const int* p, int target, int* abort_flag) {
  int spins = 0;
  while (lds_counter_load(p) < target) {
    if (lds_counter_load(abort_flag) != 0) break;
    if (++spins > kSpinLimit) {
      __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  asm volatile("" ::: "memory");
}

// one half of K for NB 32-column blocks: ROOT = A from global rows (row clamped by the caller),
// otherwise from the LDS tile.  `wait_p` (aggregated half): polled after the weight prefetch and
// before the first LDS read.  A step = one 16-byte fragment group v of a chunk (k = 32 c + 16 lh +
// 4 v + e: 4 MFMAs per column block); the fragments of the F / 32 full chunks run through a
// register ring four steps ahead of the MFMAs in a loop WITHOUT branches (hipcc's s_waitcnt
// bookkeeping is exact only then: with a conditional load anywhere in the loop it waits for
// vmcnt(0) before every MFMA group); a partial last chunk (F % 32) is done after the loop.
template <typename IdxT, int NB, bool ROOT>
__device__ __forceinline__ void spec_half(const SageFusedArgs<IdxT>& a, const float* a_row,
                                          const float* const (&wrow)[NB],
                                          const bool (&col_ok)[NB], int lh,
                                          const int* wait_p, int wait_target, int* abort_flag,
                                          f32x16 (&acc)[NB]) {
  constexpr int D = 4;
  const int F = static_cast<int>(a.g.F);
  const int n_steps = 4 * (F / kFK);  // steps of the full chunks
  const float* wp[NB];
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) wp[blk] = wrow[blk] + (ROOT ? F : 0) + 16 * lh;
  const float* ap = a_row + 16 * lh;
  auto off_of = [&](int s) { return (s >> 2) * kFK + 4 * (s & 3); };
  auto mfmas = [&](const f32x4& av, const f32x4 (&bv)[NB]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
        acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], col_ok[blk] ? bv[blk][e] : 0.f,
                                                        acc[blk], 0, 0, 0);
    }
  };
  if (n_steps > 0) {
    f32x4 ra[D], rb[D][NB];
    const int last = n_steps - 1;
    // (prologue in the loop's issue order — weights, then A, step by step: with any other order
    // the s_waitcnt pass merges the two histories at the loop head into vmcnt(0))
#pragma unroll
    for (int q = 0; q < D; ++q) {
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
        rb[q][blk] = *reinterpret_cast<const f32x4*>(wp[blk] + off_of(q));
      if constexpr (ROOT) {
        ra[q] = *reinterpret_cast<const f32x4*>(ap + off_of(q));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (!ROOT) {
      __builtin_amdgcn_sched_barrier(0);
      if (wait_p) lds_counter_wait(wait_p, wait_target, abort_flag);
#pragma unroll
      for (int q = 0; q < D; ++q) ra[q] = *reinterpret_cast<const f32x4*>(ap + off_of(q));
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int s0 = 0; s0 < n_steps; s0 += D) {  // n_steps is a multiple of D = 4
#pragma unroll
      for (int q = 0; q < D; ++q) {
        mfmas(ra[q], rb[q]);
        int sn = s0 + q + D;  // past the end: the last step again (never used)
        sn = sn < last ? sn : last;
        const int off = off_of(sn);
        // (timing probes: bit 5 = every weight fragment from one hot address, bit 6 = every root
        // fragment from one hot address — same instruction stream, no memory latency)
        const int off_b = (a.probe & 32) ? 0 : off;
        const int off_a = (ROOT && (a.probe & 64)) ? 0 : off;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
          rb[q][blk] = *reinterpret_cast<const f32x4*>(wp[blk] + off_b);
        ra[q] = *reinterpret_cast<const f32x4*>(ap + off_a);
        // the loads of a step stay behind its MFMAs and ahead of the next step's (the scheduler
        // otherwise sinks every load down to its use and waits for it there)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if (wait_p) {
    lds_counter_wait(wait_p, wait_target, abort_flag);
  }
  const int rem = F % kFK;  // partial chunk: k = base + 16 lh + 4 v + e < F (rem is a multiple of 4)
  if (rem > 0) {
    const int base = F - rem;
    const int groups = rem >= 16 ? 4 : rem / 4;  // groups whose lower lane half carries data
    for (int v = 0; v < groups; ++v) {
      const int kk = base + 16 * lh + 4 * v;
      const bool ok = kk < F;  // whole 16-byte group valid or not
      const int kc = ok ? kk : 0;
      f32x4 av, bv[NB];
      if constexpr (ROOT) {
        av = *reinterpret_cast<const f32x4*>(a_row + kc);
      } else {
        av = *reinterpret_cast<const f32x4*>(a_row + kk);  // LDS tile: zero past F
      }
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        bv[blk] = *reinterpret_cast<const f32x4*>(wrow[blk] + (ROOT ? F : 0) + kc);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[blk][e] = ok ? bv[blk][e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) av[e] = ok ? av[e] : 0.f;
      mfmas(av, bv);
    }
  }
}

template <typename IdxT, int LPR, int NM>
__global__ void __launch_bounds__(kSBlock) sage_fused_spec_kernel(SageFusedArgs<IdxT> a) {
  constexpr int NG = kSWaves - NM;        // gather waves
  constexpr int NB = (kFMaxFo / 32) / NM;  // 32-column blocks per transform wave
  constexpr int kMaxBuf = 8;
  extern __shared__ __align__(16) float smem[];
  __shared__ int ticket;
  __shared__ int done_cnt[kMaxBuf];  // rows finished, summed over the uses of the buffer
  __shared__ int free_cnt[kMaxBuf];  // transform waves finished with it, summed over the uses
  __shared__ int abort_flag;
  __shared__ int sink[kWave];        // where the lanes > 0 of a counter update add
  __shared__ int simd_cnt[4];        // waves of this workgroup per SIMD
  const int agg_ld = a.f_pad + 4;
  const int tile_floats = kFTile * agg_ld;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int F = static_cast<int>(a.g.F);
  const int nbuf = a.nbuf;
  const int64_t tiles = (a.g.n_rows + kFTile - 1) / kFTile;
  const int64_t G = gridDim.x;

  if (threadIdx.x == 0) ticket = abort_flag = 0;
  if (threadIdx.x < kMaxBuf) done_cnt[threadIdx.x] = free_cnt[threadIdx.x] = 0;
  if (threadIdx.x < 4) simd_cnt[threadIdx.x] = 0;
  if (a.f_pad > F) {  // padding columns [F, f_pad) of every buffer are zeroed once
    const int padw = a.f_pad - F;
    for (int t = threadIdx.x; t < nbuf * kFTile * padw; t += kSBlock) {
      const int r = t / padw;
      smem[r * agg_ld + F + (t - r * padw)] = 0.f;
    }
  }
  __syncthreads();
  // ---- roles.  The transform waves must sit on DIFFERENT SIMDs (each SIMD has its own MFMA pipe):
  // which SIMD a wave of the workgroup lands on is the dispatcher's choice, so every wave reads its
  // SIMD id (HW_REG_HW_ID bits [5:4]) and the first NM / 4 waves to register on each SIMD become
  // the transform waves; if a SIMD holds fewer waves of this workgroup than that, the last NM
  // waves do (correct either way).
  constexpr int MPS = NM / 4;
  const int simd = static_cast<int>(__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4));
  const int rank = __builtin_amdgcn_readfirstlane(lds_counter_add(&simd_cnt[simd], sink, lane));
  __syncthreads();  // the last barrier of the kernel
  const bool spread = simd_cnt[0] >= MPS && simd_cnt[1] >= MPS && simd_cnt[2] >= MPS &&
                      simd_cnt[3] >= MPS && !(a.probe & 16);  // (probe: static roles)
  int m = -1;  // transform wave index, -1 = gather wave
  if (spread) {
    if (rank < MPS) m = simd * MPS + rank;
  } else if (wave >= NG) {
    m = wave - NG;
  }
  m = __builtin_amdgcn_readfirstlane(m);

  if (m < 0) {
    // ---- gather waves
    if (a.probe & 2048) __builtin_amdgcn_s_setprio(2);
    for (;;) {
      const int t = __builtin_amdgcn_readfirstlane(lds_counter_add(&ticket, sink, lane));
      const int lt = t >> 5, r = t & 31;
      const int64_t tile = blockIdx.x + lt * G;
      if (tile >= tiles) break;
      const int b = lt % nbuf, use = lt / nbuf;
      if (use > 0) lds_counter_wait(&free_cnt[b], NM * use, &abort_flag);
      if (!(a.probe & 1))
        fused_gather_row<IdxT, 4, LPR>(a, tile * kFTile + r, smem + b * tile_floats + r * agg_ld,
                                       lane);
      lds_counter_signal(&done_cnt[b], sink, lane);
    }
    return;
  }

  // ---- transform waves
  const int li = lane & 31, lh = lane >> 5;
  const float* wrow[NB];
  bool col_ok[NB];
  bool any_col = false;
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    const int col = (m * NB + blk) * 32 + li;
    col_ok[blk] = col < a.Fo;
    any_col = any_col || (m * NB + blk) * 32 < a.Fo;
    wrow[blk] = a.w + static_cast<int64_t>(col_ok[blk] ? col : a.Fo - 1) * a.ldw;
  }
  for (int lt = 0;; ++lt) {
    const int64_t tile = blockIdx.x + lt * G;
    if (tile >= tiles) break;
    const int b = lt % nbuf, use = lt / nbuf;
    const int64_t row0 = tile * kFTile;
    f32x16 acc[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[blk][e] = 0.f;
    if (any_col && !(a.probe & 2)) {
      int64_t rr = row0 + li;
      rr = rr < a.g.n_rows ? rr : a.g.n_rows - 1;
      spec_half<IdxT, NB, true>(a, a.x_root + rr * a.ld_root, wrow, col_ok, lh, nullptr, 0,
                                &abort_flag, acc);
      spec_half<IdxT, NB, false>(a, smem + b * tile_floats + li * agg_ld, wrow, col_ok, lh,
                                 &done_cnt[b], kFTile * (use + 1), &abort_flag, acc);
    } else {
      lds_counter_wait(&done_cnt[b], kFTile * (use + 1), &abort_flag);
    }
    lds_counter_signal(&free_cnt[b], sink, lane);
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      const int col0 = (m * NB + blk) * 32;
      if (col0 < a.Fo && !(a.probe & 128)) {
        f32x16 unused;
        fused_epilogue<IdxT>(a, acc[blk], row0, col0, lane, unused);
      }
    }
  }
}

template <typename IdxT, int LPR>
static int launch_spec(SageFusedArgs<IdxT> a, int nm, hipStream_t st) {
  int dev = 0, cus = 0;
  PYGAMD_HIP_CHECK(hipGetDevice(&dev));
  PYGAMD_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const size_t tile_bytes = sizeof(float) * kFTile * (a.f_pad + 4);
  int nbuf = static_cast<int>((150 * 1024) / tile_bytes);
  nbuf = nbuf > 6 ? 6 : nbuf;
  const int forced = (a.probe >> 8) & 7;  // timing probe: buffer count
  if (forced >= 2 && forced <= nbuf) nbuf = forced;
  a.nbuf = nbuf;
  const size_t lds = tile_bytes * nbuf;
  auto k = nm == 8 ? sage_fused_spec_kernel<IdxT, LPR, 8> : sage_fused_spec_kernel<IdxT, LPR, 4>;
  PYGAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds)));
  const int64_t tiles = ceil_div(a.g.n_rows, kFTile);
  const unsigned grid = static_cast<unsigned>(tiles < cus ? tiles : cus);
  hipLaunchKernelGGL(k, dim3(grid), dim3(kSBlock), lds, st, a);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

static bool aligned16f(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename IdxT, int LPR>
static int launch_fused(const SageFusedArgs<IdxT>& a, bool streamed, hipStream_t st,
                        bool zsrc = false) {
  size_t lds = sizeof(float) * 2 * kFTile * (a.f_pad + 4);
  if (streamed) lds += sizeof(int32_t) * kFCap;
  if (a.probe & 64) lds = 100 * 1024;  // timing probe: one workgroup per CU
  // (probe bits 2-3: weight-prefetch depth of the transform phase, for A/B timing)
  const int pf = (a.probe >> 2) & 3;
  void (*k)(SageFusedArgs<IdxT>) = nullptr;
  if constexpr (LPR == kWave) {
    if (zsrc) k = sage_fused_fwd_kernel<IdxT, LPR, true>;
  }
  if (!k)
    k = !streamed ? sage_fused_fwd_kernel<IdxT, LPR, false>
           : pf == 1 ? sage_fused_stream_kernel<IdxT, LPR, 1>
           : pf == 3 ? sage_fused_stream_kernel<IdxT, LPR, 3>
                     : sage_fused_stream_kernel<IdxT, LPR, 2>;
  PYGAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds)));
  const int64_t tiles = ceil_div(a.g.n_rows, kFTile);
  const unsigned grid = static_cast<unsigned>(round_up(tiles, 8));
  hipLaunchKernelGGL(k, dim3(grid), dim3(kFBlock), lds, st, a);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

}  // namespace pygamd

using namespace pygamd;

extern "C" {

int pygamd_sage_layer_forward_supported(int64_t F, int64_t Fo, int reduce) {
  return (F > 0 && F % 4 == 0 && F <= 256 && Fo > 0 && Fo <= kFMaxFo &&
          (reduce == PYGAMD_SUM || reduce == PYGAMD_MEAN))
             ? 1
             : 0;
}

int pygamd_sage_layer_fused(const pygamd_spmm_args* graph, const pygamd_sage_fused_args* f,
                            void* workspace, size_t workspace_bytes, void* stream) {
  if (!graph || !f) return PYGAMD_ERR_INVALID_ARG;
  const int64_t F = graph->F, Fo = f->Fo;
  if (graph->n_rows < 0 || F < 0 || Fo < 0 || graph->ldx < F || graph->ldo < F ||
      f->ld_root < F || f->ldw < 2 * F || f->ldy < Fo)
    return PYGAMD_ERR_INVALID_ARG;
  // (col may be NULL only for a graph without edges: it is never dereferenced then)
  if (!pygamd_sage_layer_forward_supported(F, Fo, graph->reduce) || graph->w ||
      graph->src_scale || graph->eid || graph->accumulate || graph->relu_mask ||
      graph->relu_bits)
    return PYGAMD_ERR_UNSUPPORTED;
  const int64_t words = (Fo + 31) / 32;
  if (f->relu_bits_out && (!f->relu || f->ld_bits_out < words)) return PYGAMD_ERR_INVALID_ARG;
  if (f->mask_bits && f->ld_mask_bits < words) return PYGAMD_ERR_INVALID_ARG;
  if (f->y_scaled && (!f->row_scale || f->ldy_scaled < Fo)) return PYGAMD_ERR_INVALID_ARG;
  if (f->variant < 0 || f->variant > 4) return PYGAMD_ERR_INVALID_ARG;
  const bool zsrc = graph->x_format == PYGAMD_X_COMPRESSED;
  if (graph->x_format != PYGAMD_X_DENSE && !zsrc) return PYGAMD_ERR_INVALID_ARG;
  // compressed rows in / out: the row-at-a-time kernel only
  if ((zsrc || f->compressed_out) && f->variant > 1) return PYGAMD_ERR_UNSUPPORTED;
  if (zsrc && (graph->ldx < F + 12 || graph->src_bits)) return PYGAMD_ERR_INVALID_ARG;
  if (f->compressed_out && (Fo % 32 != 0 || f->ld_compressed < Fo + 12))
    return PYGAMD_ERR_INVALID_ARG;
  if (graph->n_rows == 0) return PYGAMD_OK;
  if (!graph->rowptr || !graph->x || !graph->out || !f->x_root || !f->w || !f->y)
    return PYGAMD_ERR_INVALID_ARG;
  if (graph->idx_dtype != PYGAMD_IDX_I32 && graph->idx_dtype != PYGAMD_IDX_I64)
    return PYGAMD_ERR_INVALID_ARG;
  // 16-byte accesses everywhere
  if ((graph->ldx % 4) || (graph->ldo % 4) || (f->ld_root % 4) || (f->ldw % 4) ||
      !aligned16f(graph->x) || !aligned16f(graph->out) || !aligned16f(f->x_root) ||
      !aligned16f(f->w))
    return PYGAMD_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  // hub rows first (two-stage, deterministic) into the global agg buffer; the fused kernel copies
  // them from there
  if (graph->n_hub > 0) {
    pygamd_spmm_args hubs = *graph;
    hubs.hub_phase = 2;
    const int rc = pygamd_spmm_csr(&hubs, workspace, workspace_bytes, stream);
    if (rc != PYGAMD_OK) return rc;
  }
  // default = row-at-a-time (measured fastest at the products shape: 53.8 ms/step against 55.1
  // streamed and 61.6 / 66.8 with the producer / consumer kernels); the streamed gather keeps
  // column indices as int32 in LDS
  const bool streamed = f->variant == 2 && graph->n_src < (static_cast<int64_t>(1) << 31);
  // producer / consumer waves: 3 = four transform waves of 64 columns, 4 = eight of 32
  const int spec_nm = f->variant == 3 ? 4 : f->variant == 4 ? 8 : 0;
  int lpr = 4;
  while (lpr < 64 && lpr * 4 < F) lpr <<= 1;
  if (zsrc) lpr = 64;  // a compressed row is decoded by a whole wave
  return PYGAMD_DISPATCH_IDX(graph->idx_dtype, [&]() -> int {
    SageFusedArgs<IdxT> a;
    a.g.rowptr = static_cast<const IdxT*>(graph->rowptr);
    a.g.col = static_cast<const IdxT*>(graph->col);
    a.g.eid = nullptr;
    a.g.w = nullptr;
    a.g.src_scale = nullptr;
    a.g.x = graph->x;
    a.g.out = graph->out;
    a.g.arg_out = nullptr;
    a.g.arg32_out = nullptr;
    a.g.relu_mask = nullptr;
    a.g.ldm = 0;
    a.g.relu_bits = nullptr;
    a.g.ldb = 0;
    a.g.src_bits = nullptr;
    a.g.src_bits_set = nullptr;
    a.g.n_src = graph->n_src;
    a.g.n_rows = graph->n_rows;
    a.g.F = F;
    a.g.ldx = graph->ldx;
    a.g.ldo = graph->ldo;
    a.g.w_heads = 1;
    a.g.head_dim = static_cast<int>(F);
    a.g.mean = (graph->reduce == PYGAMD_MEAN);
    a.g.accumulate = 0;
    a.g.hub_threshold = graph->n_hub > 0 ? graph->hub_threshold : 0;
    a.x_root = f->x_root;
    a.ld_root = f->ld_root;
    a.w = f->w;
    a.ldw = f->ldw;
    a.bias = f->bias;
    a.y = f->y;
    a.ldy = f->ldy;
    a.Fo = static_cast<int>(Fo);
    a.relu = f->relu ? 1 : 0;
    a.save_agg = f->save_agg ? 1 : 0;
    a.f_pad = static_cast<int>(round_up(F, kFK));
    a.bits = f->relu_bits_out;
    a.ld_bits = f->ld_bits_out;
    a.mask_bits = f->mask_bits;
    a.ld_mask = f->ld_mask_bits;
    a.row_scale = f->row_scale;
    a.y2 = f->y_scaled;
    a.ldy2 = f->ldy_scaled;
    a.zout = f->compressed_out;
    a.ldz = f->ld_compressed;
    a.probe = f->reserved;
    a.nbuf = 0;
    if (spec_nm) {
      switch (lpr) {
        case 4: return launch_spec<IdxT, 4>(a, spec_nm, st);
        case 8: return launch_spec<IdxT, 8>(a, spec_nm, st);
        case 16: return launch_spec<IdxT, 16>(a, spec_nm, st);
        case 32: return launch_spec<IdxT, 32>(a, spec_nm, st);
        default: return launch_spec<IdxT, 64>(a, spec_nm, st);
      }
    }
    switch (lpr) {
      case 4: return launch_fused<IdxT, 4>(a, streamed, st);
      case 8: return launch_fused<IdxT, 8>(a, streamed, st);
      case 16: return launch_fused<IdxT, 16>(a, streamed, st);
      case 32: return launch_fused<IdxT, 32>(a, streamed, st);
      default: return launch_fused<IdxT, 64>(a, streamed, st, zsrc);
    }
  });
}

int pygamd_sage_layer_forward(const pygamd_spmm_args* graph, const float* x_root,
                              int64_t ld_root, const float* w, int64_t ldw, const float* bias,
                              int64_t Fo, int relu, int save_agg, float* y, int64_t ldy,
                              uint32_t* relu_bits_out, int64_t ld_bits, void* workspace,
                              size_t workspace_bytes, void* stream) {
  pygamd_sage_fused_args f = {};
  f.x_root = x_root;
  f.ld_root = ld_root;
  f.w = w;
  f.ldw = ldw;
  f.bias = bias;
  f.Fo = Fo;
  f.relu = relu;
  f.save_agg = save_agg;
  f.y = y;
  f.ldy = ldy;
  f.relu_bits_out = relu_bits_out;
  f.ld_bits_out = ld_bits;
  return pygamd_sage_layer_fused(graph, &f, workspace, workspace_bytes, stream);
}

}  // extern "C"
