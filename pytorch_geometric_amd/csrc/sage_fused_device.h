// sage_fused_device.h — device-side pieces of the one-kernel SAGEConv layer shared by the production
// kernels (sage_fused.hip: the fp32-instruction and the split-arithmetic schedules) and the
// laboratory schedules (sage_fused_lab.hip: streamed gather, producer / consumer waves): the
// argument block, the epilogue of a 32 x 32 accumulator, the fp32 transform phase, the row gather
// into an LDS tile, and the host-side validation / argument marshalling of the C entry points.
#pragma once
#include "spmm_device.h"
#include "split_bf16.h"

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
  const u32x4* __restrict__ wp;  // split arithmetic: the weight as bf16 term planes in fragment
  int f_half;                    // order (sage_fused.hip), and the padded width of one half of K
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
    end = spmm_row_end(a.g, row);
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
// padding columns [F, f_pad) of both LDS tiles zeroed once, and the tile's own (root) rows copied
// next to the aggregated tile
template <typename IdxT>
__device__ __forceinline__ void fused_stage_root(const SageFusedArgs<IdxT>& a,
                                                 float* __restrict__ smem, float* __restrict__ xr,
                                                 int agg_ld, int64_t row0) {
  const int F = static_cast<int>(a.g.F);
  if (a.f_pad > F) {
    const int padw = a.f_pad - F;
    for (int t = threadIdx.x; t < 2 * kFTile * padw; t += kFBlock) {
      const int r = t / padw;
      smem[r * agg_ld + F + (t - r * padw)] = 0.f;
    }
  }
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

static inline bool aligned16f(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int sage_fused_lpr(int64_t F) {
  int lpr = 4;
  while (lpr < 64 && lpr * 4 < F) lpr <<= 1;
  return lpr;
}

// Validation shared by pygamd_sage_layer_fused and its laboratory twin.  Returns PYGAMD_OK with
// *run = false when there is nothing to do (no rows).
inline int sage_fused_validate(const pygamd_spmm_args* graph, const pygamd_sage_fused_args* f,
                               bool* run) {
  *run = false;
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
  const bool zsrc = graph->x_format == PYGAMD_X_COMPRESSED;
  if (graph->x_format != PYGAMD_X_DENSE && !zsrc) return PYGAMD_ERR_INVALID_ARG;
  if (graph->rowend && (graph->n_hub > 0 || zsrc)) return PYGAMD_ERR_UNSUPPORTED;
  if (graph->accumulate_rows != 0) return PYGAMD_ERR_INVALID_ARG;
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
  *run = true;
  return PYGAMD_OK;
}

// hub rows first (two-stage, deterministic) into the global agg buffer; the fused kernels copy
// them from there
inline int sage_fused_hub_pass(const pygamd_spmm_args* graph, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (graph->n_hub <= 0) return PYGAMD_OK;
  pygamd_spmm_args hubs = *graph;
  hubs.hub_phase = 2;
  return pygamd_spmm_csr(&hubs, workspace, workspace_bytes, stream);
}

template <typename IdxT>
inline SageFusedArgs<IdxT> sage_fused_fill(const pygamd_spmm_args* graph,
                                           const pygamd_sage_fused_args* f) {
  SageFusedArgs<IdxT> a;
  const int64_t F = graph->F;
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
  a.g.rowend = static_cast<const IdxT*>(graph->rowend);
  a.g.accumulate_rows = 0;
  a.x_root = f->x_root;
  a.ld_root = f->ld_root;
  a.w = f->w;
  a.ldw = f->ldw;
  a.bias = f->bias;
  a.y = f->y;
  a.ldy = f->ldy;
  a.Fo = static_cast<int>(f->Fo);
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
  a.probe = 0;
  a.nbuf = 0;
  a.wp = nullptr;
  a.f_half = 0;
  return a;
}

// sage_fused.hip: validation + hub pass + (split: weight pre-pass) + the production launch
int sage_layer_fused_run(const pygamd_spmm_args* graph, const pygamd_sage_fused_args* f,
                         bool split, int probe, void* workspace, size_t workspace_bytes,
                         void* stream);

}  // namespace pygamd
