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
// Two production schedules, chosen by the process-wide arithmetic (pygamd_set_gemm_mode):
//
// (A) PYGAMD_GEMM_FP32 — `sage_fused_fwd_kernel`, the exact fp32 matrix instruction.
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
// LDS: 2 x 33.3 KB (aggregated + root tile, F = 256) -> 2 workgroups per CU.
//
// (B) PYGAMD_GEMM_SPLIT_BF16 — `sage_fused_split_kernel` (round 4).  At the products shape the fp32
// transform phase costs 6.0-6.4 ms of a 13.4 ms launch whose gather phase alone takes 10.1-11.5 ms,
// and a CU that runs the fp32 matrix pipe next to a bandwidth-bound gather slows both (CHANGELOG.md
// §5a).  The split arithmetic (split_bf16.h: every fp32 operand as the exact sum of three bf16
// terms, the six leading cross products on v_mfma_f32_32x32x16_bf16, fp32 accumulation) needs 6 x 32
// instead of 8 x 64 matrix-pipe cycles per 16 k — IF the conversion does not eat the gain: done per
// fragment in registers (gemm.hip) it costs 44 VALU instructions per 6 matrix instructions and every
// element of the tile is converted by all eight waves.  Here every operand element is converted
// exactly ONCE, where it is produced:
//   * the aggregated row, by the wave that finishes it (registers -> three bf16 term planes in
//     LDS), the root rows on their way from global memory into the same planes;
//   * the weight, by a pre-pass (`sage_split_weights_kernel`, 16 k threads) into a workspace in
//     FRAGMENT ORDER: for column block cb and k-step s the three 1 KiB term planes are contiguous
//     and lane l's eight bf16 sit at 16 l — a wave's weight load is one fully coalesced 1 KiB
//     request per plane.
// The MFMA loop therefore contains no VALU work: per 16-k step 3 ds_read_b128 (A), 3 global
// 16-byte loads (B, a register ring D steps ahead), 6 matrix instructions alternating between two
// accumulators.  LDS: the planes hold 6 bytes per element, so ONE half of K is resident at a time
// for F > 128 (50.7 KB at F = 256): pass 0 = aggregated half; the root rows wait in registers
// (loaded at kernel start) and are converted into the same planes between the passes.  For
// F <= 128 both halves fit side by side and there is one pass.
// Numerics: error against fp64 at or below the exact instruction's (tests/test_gpu_split_accept.py
// holds the comparison at the headline shapes); not bitwise an fmaf chain; Inf operands give NaN.
#include "sage_fused_device.h"

namespace pygamd {

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

  // ---- phase 1: the aggregated tile and the tile's own (root) rows -> LDS
  if (threadIdx.x == 0) next_row = 0;
  fused_stage_root<IdxT>(a, smem, xr, agg_ld, row0);
  __syncthreads();  // next_row armed
  for (;;) {  // rows are handed out one by one: long and short rows balance over the 8 waves
    int r = 0;
    if (lane == 0) r = atomicAdd(&next_row, 1);
    r = __builtin_amdgcn_readfirstlane(r);
    if (r >= kFTile) break;
    fused_gather_row<IdxT, 4, LPR, ZSRC>(a, row0 + r, agg + r * agg_ld, lane);
  }
  __syncthreads();  // phase 1 complete: both tiles visible to every wave
  fused_transform<IdxT, 1>(a, agg, xr, agg_ld, row0, wave, lane, a.zout ? zw : nullptr);
}

// ---- (B) split arithmetic ----------------------------------------------------------------------------
constexpr int kPK = 16;  // k per bf16 matrix instruction

// Geometry shared by the pre-pass, the kernel and the host: one half of K is padded to `f_half`
// columns (zero weights / zero tile columns past F), the "A column space" is [aggregated half |
// root half] = [0, 2 f_half).  F <= 128: f_half = F rounded up to 16, one pass, ring depth 2 (the
// total step count 2 f_half / 16 is even).  F > 128: f_half = F rounded up to 64, one half per
// pass, ring depth 4.
inline int split_f_half(int64_t F) {
  return static_cast<int>(F <= 128 ? round_up(F, kPK) : round_up(F, 4 * kPK));
}
inline size_t split_planes_bytes(int64_t F, int64_t Fo) {
  const size_t steps = static_cast<size_t>(2 * split_f_half(F) / kPK);
  return static_cast<size_t>(ceil_div(Fo, 32)) * steps * 3 * kWave * sizeof(u32x4);
}

// weight [Fo, 2F] -> wp[((cb * steps + s) * 3 + term) * 64 + lane] = the eight bf16 of term `term`
// that lane (j = lane & 31, h = lane >> 5) feeds to the matrix instruction of step s: column
// 32 cb + j, A-space columns 16 s + 8 h .. + 7
__global__ void __launch_bounds__(kBlock)
    sage_split_weights_kernel(const float* __restrict__ w, int64_t ldw, int F, int Fo, int f_half,
                              u32x4* __restrict__ wp) {
  const int steps = 2 * f_half / kPK;
  const int cbs = (Fo + 31) / 32;
  const int gid = blockIdx.x * kBlock + threadIdx.x;
  if (gid >= cbs * steps * kWave) return;
  const int lane = gid & 63;
  const int s = (gid >> 6) % steps;
  const int cb = (gid >> 6) / steps;
  const int col = cb * 32 + (lane & 31);
  const int c0 = s * kPK + 8 * (lane >> 5);
  u32x4 t3[3];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = c0 + 2 * q + e;
      const bool root = c >= f_half;
      const int k = root ? c - f_half : c;
      v[e] = (col < Fo && k < F) ? w[static_cast<int64_t>(col) * ldw + (root ? F : 0) + k] : 0.f;
    }
    uint32_t t[3];
    split_pair(v[0], v[1], t);
    t3[0][q] = t[0];
    t3[1][q] = t[1];
    t3[2][q] = t[2];
  }
  u32x4* dst = wp + (static_cast<int64_t>(cb) * steps + s) * 3 * kWave + lane;
#pragma unroll
  for (int t = 0; t < 3; ++t) dst[t * kWave] = t3[t];
}

// four consecutive columns of one tile row -> the three term planes (8 bytes each)
__device__ __forceinline__ void split_store4(uint32_t* __restrict__ p, int pstride, float x0,
                                             float x1, float x2, float x3) {
  uint32_t ta[3], tb[3];
  split_pair(x0, x1, ta);
  split_pair(x2, x3, tb);
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const u32x2 v = {ta[t], tb[t]};
    *reinterpret_cast<u32x2*>(p + t * pstride) = v;
  }
}

// fused_gather_row with the finished row going to the term planes (`prow` = the row's first dword
// in plane 0)
template <typename IdxT, int LPR>
__device__ __forceinline__ void split_gather_row(const SageFusedArgs<IdxT>& a, int64_t row,
                                                 uint32_t* __restrict__ prow, int pstride,
                                                 int lane) {
  constexpr int VW = 4, CH = 1;
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
      split_store4(prow + (fo[0] >> 1), pstride, v.v[0], v.v[1], v.v[2], v.v[3]);
    }
    return;
  }
  spmm_accumulate<IdxT, VW, LPR, CH, 0, false>(a.g, start, end, lane, fo, fv, head, acc);
  combine_subgroups<VW, LPR, CH>(acc);
  if (lane < LPR && fv[0]) {
    const float cntf = static_cast<float>(deg > 0 ? deg : 1);
    float o[VW];
#pragma unroll
    for (int i = 0; i < VW; ++i) o[i] = a.g.mean ? acc[0][i] / cntf : acc[0][i];
    split_store4(prow + (fo[0] >> 1), pstride, o[0], o[1], o[2], o[3]);
    if (a.save_agg && row < a.g.n_rows) {
#pragma unroll
      for (int i = 0; i < VW; ++i)
        __builtin_nontemporal_store(o[i], a.g.out + row * a.g.ldo + fo[0] + i);
    }
  }
}

// `ns` steps (a multiple of D) of this wave's 32 x 32 block: A fragments from the planes (`aq` =
// lane's first dword of step 0 in plane 0; a step advances 8 dwords), weight fragments from `wq`
// (= lane's u32x4 of step 0, term 0; a step advances 3 x 64) through a register ring D steps
// ahead.  Loads are unconditional (past the end: the last step again) so that hipcc's s_waitcnt
// bookkeeping stays exact.
template <int D>
__device__ __forceinline__ void split_pass(const u32x4* __restrict__ wq,
                                           const uint32_t* __restrict__ aq, int pstride, int ns,
                                           f32x16& acc0, f32x16& acc1) {
  u32x4 rb[D][3];
  // (prologue in the loop's issue order, slot by slot: with any other order the s_waitcnt pass
  // merges the two histories at the loop head into a wait for nearly every load in flight.
  // Instruction selection clusters loads off one base by their 4 KiB offset window and would issue
  // slot 0 last — the loads are of read-only memory, so neither a scheduling barrier nor a
  // compiler fence orders them; an opaque offset per slot does)
#pragma unroll
  for (int q = 0; q < D; ++q) {
    int off = q * 3 * kWave;
    asm volatile("" : "+v"(off));
#pragma unroll
    for (int t = 0; t < 3; ++t) rb[q][t] = wq[off + t * kWave];
    __builtin_amdgcn_sched_barrier(0);
  }
  u32x4 fa[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) fa[t] = *reinterpret_cast<const u32x4*>(aq + t * pstride);
  __builtin_amdgcn_sched_barrier(0);
  const int last = ns - 1;
  for (int s = 0; s < ns; s += D) {
#pragma unroll
    for (int q = 0; q < D; ++q) {
      u32x4 fan[3];
      int sa = s + q + 1;
      sa = sa < last ? sa : last;
#pragma unroll
      for (int t = 0; t < 3; ++t)
        fan[t] = *reinterpret_cast<const u32x4*>(aq + t * pstride + sa * (kPK / 2));
#pragma unroll
      for (int t = 0; t < kSplitTerms; ++t) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, fa[kSplitTa[t]]);
        const bf16x8 bv = __builtin_bit_cast(bf16x8, rb[q][kSplitTb[t]]);
        if (t & 1) {
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc1, 0, 0, 0);
        } else {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc0, 0, 0, 0);
        }
      }
      // the ring slot is refilled AFTER the matrix instructions that read it (same registers: no
      // rotation copies at the loop end, which would wait for the data of every load in flight);
      // the sched_barriers keep the refill here instead of down at its use D steps later
      __builtin_amdgcn_sched_barrier(0);
      int sn = s + q + D;
      sn = sn < last ? sn : last;
#pragma unroll
      for (int t = 0; t < 3; ++t) rb[q][t] = wq[(sn * 3 + t) * kWave];
#pragma unroll
      for (int t = 0; t < 3; ++t) fa[t] = fan[t];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// NPASS = 1: both halves of K side by side in the planes (F <= 128); 2: one half per pass.
// PROBE: honour a.probe bits 0 / 1 (skip the gather / the matrix loop; timing only).
// One pass: 44.5 KB of planes at F = 100 -> three workgroups per CU if the kernel stays within 80
// registers; two passes: 50.7 KB at F = 256, two workgroups per CU (the ring of depth 4 and two
// accumulators take 124 registers).
template <typename IdxT, int LPR, int NPASS, bool PROBE, bool OCC3 = (NPASS == 1)>
__global__ void __launch_bounds__(kFBlock, OCC3 ? 6 : 4)
    sage_fused_split_kernel(SageFusedArgs<IdxT> a) {
  constexpr int D = OCC3 ? 2 : 4;
  constexpr int NU = 4;  // 16-byte root pieces per thread: 32 rows x (F <= 256) / 4 / 512
  extern __shared__ __align__(16) uint32_t pl[];  // [3][32][row_dw]
  __shared__ int next_row;
  const int F = static_cast<int>(a.g.F);
  const int fh = a.f_half;
  const int KT = NPASS == 1 ? 2 * fh : fh;  // tile columns resident in the planes
  const int row_dw = KT / 2 + 4;            // odd multiple of 4 dwords: conflict-free b128 reads
  const int pstride = kFTile * row_dw;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t tile = xcd_logical_block();
  const int64_t row0 = tile * kFTile;
  if (row0 >= a.g.n_rows) return;
  const int units = F / 4;  // 16-byte pieces per row

  // ---- the tile's own rows: loads issued first.  One pass: converted into the root half of the
  // planes right away; two passes: they wait in registers until the aggregated half is consumed.
  f32x4 rr[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    rr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int t = threadIdx.x + i * kFBlock;
    const int r = t / units;
    const int u = t - r * units;
    int64_t row = row0 + r;
    row = row < a.g.n_rows ? row : a.g.n_rows - 1;
    if (t < kFTile * units)
      rr[i] = *reinterpret_cast<const f32x4*>(a.x_root + row * a.ld_root + 4 * u);
  }
  auto root_to_planes = [&](int col0) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int t = threadIdx.x + i * kFBlock;
      const int r = t / units;
      const int u = t - r * units;
      if (t < kFTile * units)
        split_store4(pl + r * row_dw + ((col0 + 4 * u) >> 1), pstride, rr[i][0], rr[i][1],
                     rr[i][2], rr[i][3]);
    }
  };
  if (threadIdx.x == 0) next_row = 0;
  if (fh > F) {  // padding columns [F, f_half) of each resident half: zero in all three planes
    const int pu = (fh - F) / 4;
    const int halves = NPASS == 1 ? 2 : 1;
    for (int t = threadIdx.x; t < kFTile * pu * halves; t += kFBlock) {
      const int r = t / (pu * halves);
      const int q = t - r * (pu * halves);
      const int h = q / pu;
      const int col = h * fh + F + 4 * (q - h * pu);
      const u32x2 z = {0u, 0u};
#pragma unroll
      for (int tt = 0; tt < 3; ++tt)
        *reinterpret_cast<u32x2*>(pl + tt * pstride + r * row_dw + (col >> 1)) = z;
    }
  }
  if constexpr (NPASS == 1) root_to_planes(fh);
  __syncthreads();  // next_row armed

  // ---- gather phase: rows handed out one by one, finished rows -> term planes
  for (; !(PROBE && (a.probe & 1));) {
    int r = 0;
    if (lane == 0) r = atomicAdd(&next_row, 1);
    r = __builtin_amdgcn_readfirstlane(r);
    if (r >= kFTile) break;
    split_gather_row<IdxT, LPR>(a, row0 + r, pl + r * row_dw, pstride, lane);
  }
  __syncthreads();  // aggregated half (and, one pass: the root half) visible to every wave

  // ---- transform: wave w owns output columns [32 w, 32 w + 32)
  const int wave_col0 = wave * 32;
  const bool active = wave_col0 < a.Fo && !(PROBE && (a.probe & 2));
  const int li = lane & 31, lh = lane >> 5;
  const int steps = 2 * fh / kPK;  // both halves
  const u32x4* __restrict__ wq = a.wp + static_cast<int64_t>(wave) * steps * 3 * kWave + lane;
  const uint32_t* aq = pl + li * row_dw + 4 * lh;
  f32x16 acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
  if constexpr (NPASS == 1) {
    if (active) split_pass<D>(wq, aq, pstride, steps, acc0, acc1);
  } else {
    const int ns = steps / 2;
    if (active) split_pass<D>(wq, aq, pstride, ns, acc0, acc1);
    __syncthreads();  // every wave is done with the aggregated half
    root_to_planes(0);
    __syncthreads();
    if (active) split_pass<D>(wq + static_cast<int64_t>(ns) * 3 * kWave, aq, pstride, ns, acc0, acc1);
  }
  if (wave_col0 >= a.Fo) return;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc0[e] += acc1[e];
  f32x16 vout;
  fused_epilogue<IdxT>(a, acc0, row0, wave_col0, lane, vout);
}

template <typename IdxT, int LPR>
static int launch_fused(const SageFusedArgs<IdxT>& a, hipStream_t st, bool zsrc) {
  const size_t lds = sizeof(float) * 2 * kFTile * (a.f_pad + 4);
  void (*k)(SageFusedArgs<IdxT>) = sage_fused_fwd_kernel<IdxT, LPR, false>;
  if constexpr (LPR == kWave) {
    if (zsrc) k = sage_fused_fwd_kernel<IdxT, LPR, true>;
  }
  PYGAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds)));
  const int64_t tiles = ceil_div(a.g.n_rows, kFTile);
  const unsigned grid = static_cast<unsigned>(round_up(tiles, 8));
  hipLaunchKernelGGL(k, dim3(grid), dim3(kFBlock), lds, st, a);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

template <typename IdxT, int LPR, int NPASS>
static int launch_split(const SageFusedArgs<IdxT>& a, hipStream_t st) {
  const int KT = NPASS == 1 ? 2 * a.f_half : a.f_half;
  const size_t lds = sizeof(uint32_t) * 3 * kFTile * (KT / 2 + 4);
  void (*k)(SageFusedArgs<IdxT>) = a.probe ? sage_fused_split_kernel<IdxT, LPR, NPASS, true>
                                           : sage_fused_split_kernel<IdxT, LPR, NPASS, false>;
  if constexpr (NPASS == 2) {  // (probe bit 4: the two-pass kernel at three workgroups per CU)
    if (a.probe & 16) k = sage_fused_split_kernel<IdxT, LPR, NPASS, true, true>;
  }
  PYGAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds)));
  const int64_t tiles = ceil_div(a.g.n_rows, kFTile);
  const unsigned grid = static_cast<unsigned>(round_up(tiles, 8));
  hipLaunchKernelGGL(k, dim3(grid), dim3(kFBlock), lds, st, a);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

// the split schedule takes dense rows in and writes dense rows out
static bool split_eligible(const pygamd_spmm_args* graph, const pygamd_sage_fused_args* f) {
  return pygamd_get_gemm_mode() == PYGAMD_GEMM_SPLIT_BF16 && graph->x_format == PYGAMD_X_DENSE &&
         !f->compressed_out;
}

static size_t hub_bytes_aligned(const pygamd_spmm_args* graph) {
  size_t hub = 0;
  pygamd_spmm_csr_workspace_bytes(graph, &hub);
  return static_cast<size_t>(round_up(static_cast<int64_t>(hub), 256));
}

// `probe`: 0 in production (pygamd_sage_layer_fused); the laboratory entry point passes its bits
// through for the split schedule (variant 5 there)
int sage_layer_fused_run(const pygamd_spmm_args* graph, const pygamd_sage_fused_args* f,
                         bool split, int probe, void* workspace, size_t workspace_bytes,
                         void* stream) {
  bool run = false;
  int rc = sage_fused_validate(graph, f, &run);
  if (rc != PYGAMD_OK || !run) return rc;
  const int64_t F = graph->F, Fo = f->Fo;
  hipStream_t st = as_stream(stream);
  const bool zsrc = graph->x_format == PYGAMD_X_COMPRESSED;
  u32x4* wp = nullptr;
  if (split) {
    if (zsrc || f->compressed_out) return PYGAMD_ERR_UNSUPPORTED;
    const size_t off = hub_bytes_aligned(graph);
    if (!workspace || workspace_bytes < off + split_planes_bytes(F, Fo))
      return PYGAMD_ERR_WORKSPACE;
    wp = reinterpret_cast<u32x4*>(static_cast<char*>(workspace) + off);
  }
  rc = sage_fused_hub_pass(graph, workspace, workspace_bytes, stream);
  if (rc != PYGAMD_OK) return rc;
  const int fh = split_f_half(F);
  if (split) {
    const int threads = static_cast<int>(ceil_div(Fo, 32)) * (2 * fh / kPK) * kWave;
    hipLaunchKernelGGL(sage_split_weights_kernel, dim3(static_cast<unsigned>(ceil_div(threads, kBlock))),
                       dim3(kBlock), 0, st, f->w, f->ldw, static_cast<int>(F),
                       static_cast<int>(Fo), fh, wp);
    PYGAMD_LAUNCH_CHECK();
  }
  const int lpr = zsrc ? 64 : sage_fused_lpr(F);  // a compressed row is decoded by a whole wave
  return PYGAMD_DISPATCH_IDX(graph->idx_dtype, [&]() -> int {
    SageFusedArgs<IdxT> a = sage_fused_fill<IdxT>(graph, f);
    a.probe = probe;
    a.wp = wp;
    a.f_half = fh;
    if (split) {
      switch (lpr) {
        case 4: return launch_split<IdxT, 4, 1>(a, st);
        case 8: return launch_split<IdxT, 8, 1>(a, st);
        case 16: return launch_split<IdxT, 16, 1>(a, st);
        case 32: return launch_split<IdxT, 32, 1>(a, st);
        default: return launch_split<IdxT, 64, 2>(a, st);
      }
    }
    switch (lpr) {
      case 4: return launch_fused<IdxT, 4>(a, st, false);
      case 8: return launch_fused<IdxT, 8>(a, st, false);
      case 16: return launch_fused<IdxT, 16>(a, st, false);
      case 32: return launch_fused<IdxT, 32>(a, st, false);
      default: return launch_fused<IdxT, 64>(a, st, zsrc);
    }
  });
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

int pygamd_sage_layer_fused_workspace_bytes(const pygamd_spmm_args* graph,
                                            const pygamd_sage_fused_args* f, size_t* bytes) {
  if (!graph || !f || !bytes || graph->F < 0 || f->Fo < 0) return PYGAMD_ERR_INVALID_ARG;
  size_t hub = 0;
  pygamd_spmm_csr_workspace_bytes(graph, &hub);
  // (sized for the split schedule whatever the current mode: the mode may change between the
  // query and the launch)
  const bool planes = graph->x_format == PYGAMD_X_DENSE && !f->compressed_out &&
                      pygamd_sage_layer_forward_supported(graph->F, f->Fo, graph->reduce);
  *bytes = planes ? hub_bytes_aligned(graph) + split_planes_bytes(graph->F, f->Fo) : hub;
  return PYGAMD_OK;
}

int pygamd_sage_layer_fused(const pygamd_spmm_args* graph, const pygamd_sage_fused_args* f,
                            void* workspace, size_t workspace_bytes, void* stream) {
  if (!graph || !f) return PYGAMD_ERR_INVALID_ARG;
  return sage_layer_fused_run(graph, f, split_eligible(graph, f), 0, workspace, workspace_bytes,
                              stream);
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
