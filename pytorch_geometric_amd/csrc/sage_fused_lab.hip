// sage_fused_lab.hip — LABORATORY schedules of the one-kernel SAGEConv layer (sage_fused.hip holds the
// production ones).  Built into libpyg_amd_lab.so ONLY (not into the product library libpyg_amd.so)
// so that scripts/fused_probe.py and the parity tests can run them on the device; the entry point
// below is declared in include/pyg_amd_lab.h, not in include/pyg_amd.h.
//   variant 2      the gather phase as a software-pipelined stream (bitwise the production kernel);
//   variant 1      the production fp32 schedule with its probe bits honoured;
//   variant 5 / 6  the production kernels themselves (split arithmetic / fp32 instruction),
//                  independent of pygamd_set_gemm_mode; 5 honours probe bits 0 and 1;
//   variant 3 / 4  one persistent 1024-thread workgroup per CU whose 12 / 8 gather waves feed 4 / 8
//                  transform waves through LDS tiles with LDS counters instead of barriers.
// Both were measured slower than the production schedule at the products shape (CHANGELOG.md §5a).
// `probe` bits (timing only): see SageFusedArgs::probe.
#include "sage_fused_device.h"
#include "../../include/pyg_amd_lab.h"

namespace pygamd {

// ---- variant 1 with the probe bits honoured: the production schedule (sage_fused.hip) as a timing
// subject (gather loop / MFMA loop skipped, issue priorities, one workgroup per CU)
template <typename IdxT, int LPR>
__global__ void __launch_bounds__(kFBlock, 4) sage_fused_probe_kernel(SageFusedArgs<IdxT> a) {
  extern __shared__ __align__(16) float smem[];
  __shared__ int next_row;
  const int agg_ld = a.f_pad + 4;
  float* agg = smem;                    // [32][f_pad + 4]  aggregated rows
  float* xr = smem + kFTile * agg_ld;   // [32][f_pad + 4]  root rows of the tile
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t tile = xcd_logical_block();
  const int64_t row0 = tile * kFTile;
  if (row0 >= a.g.n_rows) return;
  fused_stage_root<IdxT>(a, smem, xr, agg_ld, row0);
  if (threadIdx.x == 0) next_row = 0;
  __syncthreads();  // next_row armed
  // (probe bit 11: the gather phase at raised issue priority, the transform phase at 0 — VALU /
  // VMEM issue on a SIMD is arbitrated by priority, then age, and a wave issuing dependent MFMAs
  // back to back otherwise wins every slot it asks for)
  if (a.probe & 2048) __builtin_amdgcn_s_setprio(2);
  for (; !(a.probe & 1);) {
    int r = 0;
    if (lane == 0) r = atomicAdd(&next_row, 1);
    r = __builtin_amdgcn_readfirstlane(r);
    if (r >= kFTile) break;
    fused_gather_row<IdxT, 4, LPR, false>(a, row0 + r, agg + r * agg_ld, lane);
  }
  if (a.probe & 2048) __builtin_amdgcn_s_setprio(0);
  __syncthreads();  // phase 1 complete: both tiles visible to every wave
  fused_transform<IdxT, 1>(a, agg, xr, agg_ld, row0, wave, lane, nullptr);
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
  // (ring depth: 4 steps with one column block per wave; 2 with two — round 3's depth 4 spilled
  // 16-17 registers there under the 128-register cap of a 1024-thread workgroup)
  constexpr int D = NB >= 2 ? 2 : 4;
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

template <typename IdxT, int LPR>
static int launch_lab(const SageFusedArgs<IdxT>& a, bool streamed, hipStream_t st) {
  size_t lds = sizeof(float) * 2 * kFTile * (a.f_pad + 4);
  if (streamed) lds += sizeof(int32_t) * kFCap;
  if (a.probe & 64) lds = 100 * 1024;  // timing probe: one workgroup per CU
  // (probe bits 2-3: weight-prefetch depth of the transform phase, for A/B timing)
  const int pf = (a.probe >> 2) & 3;
  void (*k)(SageFusedArgs<IdxT>) =
      !streamed ? sage_fused_probe_kernel<IdxT, LPR>
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

// Streaming copy probe (pygamd_lab_copy): every lane moves 16 bytes per step, `unroll` loads in
// flight before the first store, grid-stride so a fixed number of workgroups per CU covers any size.
typedef float lab_f4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void lab_copy_kernel(const lab_f4* __restrict__ src,
                                                       lab_f4* __restrict__ dst, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    lab_f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
      else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

// the read half alone: the same loads summed per lane; the store never happens for finite data
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void lab_read_kernel(const lab_f4* __restrict__ src,
                                                       lab_f4* __restrict__ dst, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  lab_f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    lab_f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u];
  }
  for (; i < n; i += stride) acc += src[i];
  if (acc.x + acc.y + acc.z + acc.w == __builtin_inff()) dst[threadIdx.x] = acc;
}

}  // namespace pygamd

using namespace pygamd;

extern "C" int pygamd_lab_copy(const void* src, void* dst, int64_t n_bytes, int variant,
                               int blocks_per_cu, void* stream) {
  if (n_bytes < 0 || (n_bytes & 15) || blocks_per_cu < 1) return PYGAMD_ERR_INVALID_ARG;
  if ((n_bytes && (!src || !dst)) || variant < 0 || variant > 7) return PYGAMD_ERR_INVALID_ARG;
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) != 0)
    return PYGAMD_ERR_INVALID_ARG;
  if (n_bytes == 0) return PYGAMD_OK;
  hipStream_t st = as_stream(stream);
  int dev = 0, cus = 256;
  PYGAMD_HIP_CHECK(hipGetDevice(&dev));
  PYGAMD_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int64_t n = n_bytes / 16;
  const int64_t want = ceil_div(n, 256);
  const unsigned grid = static_cast<unsigned>(
      want < static_cast<int64_t>(cus) * blocks_per_cu ? want
                                                        : static_cast<int64_t>(cus) * blocks_per_cu);
  const lab_f4* s = static_cast<const lab_f4*>(src);
  lab_f4* d = static_cast<lab_f4*>(dst);
#define PYGAMD_LAB_COPY(K, U, NT) \
  hipLaunchKernelGGL((K<U, NT>), dim3(grid), dim3(256), 0, st, s, d, n)
  switch (variant) {
    case 0: PYGAMD_LAB_COPY(lab_copy_kernel, 4, false); break;
    case 1: PYGAMD_LAB_COPY(lab_copy_kernel, 4, true); break;
    case 2: PYGAMD_LAB_COPY(lab_copy_kernel, 8, false); break;
    case 3: PYGAMD_LAB_COPY(lab_copy_kernel, 8, true); break;
    case 4: PYGAMD_LAB_COPY(lab_read_kernel, 4, false); break;
    case 5: PYGAMD_LAB_COPY(lab_read_kernel, 4, true); break;
    case 6: PYGAMD_LAB_COPY(lab_read_kernel, 8, false); break;
    default: PYGAMD_LAB_COPY(lab_read_kernel, 8, true); break;
  }
#undef PYGAMD_LAB_COPY
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

extern "C" int pygamd_lab_sage_layer_fused(const pygamd_spmm_args* graph,
                                           const pygamd_sage_fused_args* f, int variant,
                                           int probe, void* workspace, size_t workspace_bytes,
                                           void* stream) {
  bool run = false;
  int rc = sage_fused_validate(graph, f, &run);
  if (rc != PYGAMD_OK || !run) return rc;
  if (variant < 1 || variant > 6) return PYGAMD_ERR_INVALID_ARG;
  // 5 / 6: the production kernels (split arithmetic / fp32 instruction) whatever the process-wide
  // mode, with the probe bits they honour (bit 0 = skip the gather, bit 1 = skip the matrix loop)
  if (variant == 5) return sage_layer_fused_run(graph, f, true, probe, workspace, workspace_bytes, stream);
  if (variant == 6) return sage_layer_fused_run(graph, f, false, 0, workspace, workspace_bytes, stream);
  // compressed rows in / out: the production kernel only
  if (graph->x_format != PYGAMD_X_DENSE || f->compressed_out) return PYGAMD_ERR_UNSUPPORTED;
  // (the streamed / producer-consumer schedules read rowptr[row + 1] themselves)
  if (graph->rowend && variant != 5 && variant != 6 && variant != 1) return PYGAMD_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  rc = sage_fused_hub_pass(graph, workspace, workspace_bytes, stream);
  if (rc != PYGAMD_OK) return rc;
  // the streamed gather keeps column indices as int32 in LDS
  const bool streamed = variant == 2 && graph->n_src < (static_cast<int64_t>(1) << 31);
  // producer / consumer waves: 3 = four transform waves of 64 columns, 4 = eight of 32
  const int spec_nm = variant == 3 ? 4 : variant == 4 ? 8 : 0;
  const int lpr = sage_fused_lpr(graph->F);
  return PYGAMD_DISPATCH_IDX(graph->idx_dtype, [&]() -> int {
    SageFusedArgs<IdxT> a = sage_fused_fill<IdxT>(graph, f);
    a.probe = probe;
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
      case 4: return launch_lab<IdxT, 4>(a, streamed, st);
      case 8: return launch_lab<IdxT, 8>(a, streamed, st);
      case 16: return launch_lab<IdxT, 16>(a, streamed, st);
      case 32: return launch_lab<IdxT, 32>(a, streamed, st);
      default: return launch_lab<IdxT, 64>(a, streamed, st);
    }
  });
}
