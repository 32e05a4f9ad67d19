// spmm.hip — CSR segmented-reduce SpMM for gfx950 (MI355X).
//
// The fused form of MessagePassing.propagate (gather on edge_index[j] -> message -> reduce on
// edge_index[i]); see include/pyg_amd.h for the reference call sites.  HBM-bound: every stored
// entry reads one full source row (4F bytes) exactly once; nothing of size [E, F] is ever
// materialised.
//
// Mapping (wave64):
//   * one wavefront owns one destination row (4 rows per 256-thread workgroup, workgroups
//     XCD-remapped so each XCD walks a contiguous eighth of the rows);
//   * a row's column indices are fetched 64 at a time with one coalesced load (lane l holds
//     slot base+l) and handed out through v_readlane (when a whole wave serves one source row)
//     or ds_bpermute (when a wave serves 64/LPR source rows at once);
//   * LPR lanes cover one source row with VW-wide (16-byte when F % 4 == 0) loads, so a
//     256-float row is ONE global_load_dwordx4 per wave = 1 KiB contiguous;
//   * U independent row loads are issued before the first add (8 x 1 KiB in flight per wave,
//     <= 64 VGPRs -> 8 waves/SIMD) — latency is hidden by memory-level parallelism, not LDS;
//   * accumulation is in registers in slot order, one non-atomic store per output row;
//   * rows longer than hub_threshold are skipped here and processed as fixed-size chunks by
//     spmm_hub_chunks + spmm_hub_combine (deterministic two-stage sum).
#include "common.h"

#include "spmm_device.h"

namespace pygamd {

template <typename IdxT, int VW, int LPR, int CH, int WMODE, bool IDENT>
__global__ void __launch_bounds__(kBlock) spmm_sum_rows(SpmmDev<IdxT> a) {
  const int lane = lane_id();
  const int64_t row = xcd_logical_block() * kWavesPerBlock + wave_in_block();
  if (row >= a.n_rows) return;
  const IdxT start = a.rowptr[row];
  const IdxT end = spmm_row_end(a, row);
  const IdxT deg = end - start;
  if (a.hub_threshold > 0 && deg > a.hub_threshold) return;  // owned by the hub path
  const bool has_old = a.accumulate && (a.accumulate_rows <= 0 || row < a.accumulate_rows);
  int fo[CH], head[CH];
  bool fv[CH];
  feature_slots<VW, LPR, CH>(lane, a.F, a.head_dim, fo, fv, head);
  float acc[CH][VW];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
#pragma unroll
    for (int i = 0; i < VW; ++i) acc[c][i] = 0.f;
  }
  // The row's epilogue operands (the old output row of `accumulate`, the activation row of the
  // fused ReLU backward) do not depend on the sum: their loads are issued BEFORE the gather loop
  // and complete under it, instead of adding one or two exposed HBM latencies to the end of every
  // wave's life.
  float* __restrict__ orow = a.out + row * a.ldo;
  Vec<VW> oldv[CH], maskv[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      oldv[c].v[i] = 0.f;
      maskv[c].v[i] = 1.f;
    }
    if (lane < LPR && fv[c]) {
      if (has_old) oldv[c] = load_vec_streamed<VW>(orow + fo[c]);
      if (a.relu_mask) maskv[c] = load_vec_streamed<VW>(a.relu_mask + row * a.ldm + fo[c]);
      if (a.relu_bits) {  // VW divides 32 and fo[c] is a multiple of VW: one word holds the bits
        const uint32_t w =
            a.relu_bits[((row >> 5) * a.ldb + (fo[c] >> 5)) * 32 + (row & 31)] >> (fo[c] & 31);
#pragma unroll
        for (int i = 0; i < VW; ++i) maskv[c].v[i] = ((w >> i) & 1u) ? 1.f : 0.f;
      }
    }
  }
  spmm_accumulate<IdxT, VW, LPR, CH, WMODE, IDENT>(a, start, end, lane, fo, fv, head, acc);
  combine_subgroups<VW, LPR, CH>(acc);
  if (lane < LPR) {
    const float cntf = static_cast<float>(deg > 0 ? deg : 1);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (fv[c]) {
#pragma unroll
        for (int i = 0; i < VW; ++i) {
          float o = a.mean ? acc[c][i] / cntf : acc[c][i];
          if (has_old) o += oldv[c].v[i];  // (not "+ 0": -0.0 sums stay -0.0)
          o = maskv[c].v[i] > 0.f ? o : 0.f;
          __builtin_nontemporal_store(o, orow + fo[c] + i);
        }
      }
    }
  }
}

// Sparse-source variant of spmm_sum_rows (a.src_bits given, plain sum / mean, no epilogue
// operands): one wave owns R consecutive rows.  With 8 % of the source rows live a row's work is
// three dependent round trips (its column ids, their bits, the two or three live rows) and almost
// no bytes; the one-row kernel then runs at the latency of 3 trips per 8 waves per SIMD (1.0 ms for
// the products graph).  Here the R index loads, then the R bit lookups, are in flight together.
template <typename IdxT, int VW, int LPR, int CH, int R>
__global__ void __launch_bounds__(kBlock) spmm_sum_rows_sparse(SpmmDev<IdxT> a) {
  const int lane = lane_id();
  const int64_t row0 = (xcd_logical_block() * kWavesPerBlock + wave_in_block()) * R;
  if (row0 >= a.n_rows) return;
  const bool use_bits = a.src_bits_set == nullptr || 2 * *a.src_bits_set < a.n_src;
  int fo[CH], head[CH];
  bool fv[CH];
  feature_slots<VW, LPR, CH>(lane, a.F, a.head_dim, fo, fv, head);
  // rowptr[row0 .. row0 + R], clamped to the last entry: rows past the end come out empty
  IdxT myptr = 0;
  if (lane <= R) {
    const int64_t r = row0 + lane;
    myptr = a.rowptr[r < a.n_rows ? r : a.n_rows];
  }
  IdxT start[R], end[R], myc[R];
  int cnt[R];
  bool hub[R];
  // (loads are unconditional on clamped addresses and selected afterwards: a load inside a
  // divergent `if` is waited for at the end of its block, one round trip per row again)
  IdxT nnz = a.rowptr[a.n_rows];
  const IdxT* __restrict__ colp = a.col;
  const uint32_t* __restrict__ bitp = a.src_bits;
  if (nnz == 0) {  // nothing stored: every load below reads slot 0 of a one-word stand-in
    nnz = 1;
    colp = reinterpret_cast<const IdxT*>(a.rowptr);  // rowptr[0] == 0: a valid column id
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    start[r] = bcast_uniform(myptr, r);
    end[r] = bcast_uniform(myptr, r + 1);
    const IdxT deg = end[r] - start[r];
    hub[r] = a.hub_threshold > 0 && deg > a.hub_threshold;  // owned by the hub path
    cnt[r] = hub[r] ? 0 : (deg < kWave ? static_cast<int>(deg) : kWave);
    IdxT slot = start[r] + lane;
    slot = slot < nnz ? slot : nnz - 1;
    myc[r] = __builtin_nontemporal_load(&colp[slot]);
  }
  bool keep[R];
  if (use_bits) {
    uint32_t wbits[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
      wbits[r] = bitp[static_cast<int64_t>(myc[r]) >> 5];
#pragma unroll
    for (int r = 0; r < R; ++r) keep[r] = lane < cnt[r] && ((wbits[r] >> (myc[r] & 31)) & 1u) != 0;
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) keep[r] = lane < cnt[r];
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint64_t m = __ballot(keep[r]);
    const int before = __builtin_amdgcn_mbcnt_hi(
        static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
    const int live = __popcll(m);
    myc[r] = push_lane(myc[r], keep[r] ? before : live + lane - before);
    cnt[r] = live;
  }
  constexpr int EPI = kWave / LPR;
  constexpr int STEP = EPI * spmm_unroll<LPR, CH>();
  const int sub = lane / LPR;
  // the first EPI live slots of every row (with 8 % of the sources live: usually all of them),
  // R loads in flight; the address is always that of a stored column (slot 0 when there is none)
  Vec<VW> v0[R][CH];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    IdxT c;
    if constexpr (EPI == 1) {
      c = bcast_uniform(myc[r], 0);
    } else {
      c = bcast_lane(myc[r], sub < cnt[r] ? sub : 0);
    }
    const float* __restrict__ xr = a.x + static_cast<int64_t>(c) * a.ldx;
#pragma unroll
    for (int c2 = 0; c2 < CH; ++c2) v0[r][c2] = load_vec<VW>(xr + (fv[c2] ? fo[c2] : 0));
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t row = row0 + r;
    if (row >= a.n_rows || hub[r]) continue;
    float acc[CH][VW];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int i = 0; i < VW; ++i) acc[c][i] = (sub < cnt[r] && fv[c]) ? v0[r][c].v[i] : 0.f;
    }
    int j = EPI;
    for (; j + STEP <= cnt[r]; j += STEP) {
      spmm_batch<IdxT, VW, LPR, CH, 0, false, true>(a, j, cnt[r], sub, myc[r], IdxT(0), 1.f, fo,
                                                    fv, head, acc);
    }
    if (j < cnt[r]) {
      spmm_batch<IdxT, VW, LPR, CH, 0, false, false>(a, j, cnt[r], sub, myc[r], IdxT(0), 1.f, fo,
                                                     fv, head, acc);
    }
    if (end[r] - start[r] > kWave) {  // the slots past the first 64, one chunk at a time
      spmm_accumulate<IdxT, VW, LPR, CH, 3, false>(a, start[r] + kWave, end[r], lane, fo, fv,
                                                    head, acc);
    }
    combine_subgroups<VW, LPR, CH>(acc);
    if (lane < LPR) {
      const IdxT deg = end[r] - start[r];
      const float cntf = static_cast<float>(deg > 0 ? deg : 1);
      float* __restrict__ orow = a.out + row * a.ldo;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (fv[c]) {
#pragma unroll
          for (int i = 0; i < VW; ++i) {
            const float o = a.mean ? acc[c][i] / cntf : acc[c][i];
            __builtin_nontemporal_store(o, orow + fo[c] + i);
          }
        }
      }
    }
  }
}

// One wave per hub chunk: partial[ck, :] = sum over the chunk's slots (no post-scale).
template <typename IdxT, int VW, int LPR, int CH, int WMODE, bool IDENT>
__global__ void __launch_bounds__(kBlock)
    spmm_hub_chunks(SpmmDev<IdxT> a, const IdxT* __restrict__ hub_rows,
                    const IdxT* __restrict__ hub_chunk_ptr, int64_t n_hub, int64_t n_chunks,
                    int64_t chunk, float* __restrict__ partial) {
  const int lane = lane_id();
  const int64_t ck = xcd_logical_block() * kWavesPerBlock + wave_in_block();
  if (ck >= n_chunks) return;
  int64_t lo = 0, hi = n_hub;  // hub_chunk_ptr[lo] <= ck < hub_chunk_ptr[hi]
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (static_cast<int64_t>(hub_chunk_ptr[mid]) <= ck) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  const int64_t row = hub_rows[lo];
  const int64_t ci = ck - static_cast<int64_t>(hub_chunk_ptr[lo]);
  const IdxT rs = a.rowptr[row];
  const IdxT re = a.rowptr[row + 1];
  const IdxT start = rs + static_cast<IdxT>(ci * chunk);
  IdxT end = start + static_cast<IdxT>(chunk);
  if (end > re) end = re;
  int fo[CH], head[CH];
  bool fv[CH];
  feature_slots<VW, LPR, CH>(lane, a.F, a.head_dim, fo, fv, head);
  float acc[CH][VW];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
#pragma unroll
    for (int i = 0; i < VW; ++i) acc[c][i] = 0.f;
  }
  spmm_accumulate<IdxT, VW, LPR, CH, WMODE, IDENT>(a, start, end, lane, fo, fv, head, acc);
  combine_subgroups<VW, LPR, CH>(acc);
  if (lane < LPR) {
    float* __restrict__ prow = partial + ck * a.F;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (fv[c]) {
        Vec<VW> o;
#pragma unroll
        for (int i = 0; i < VW; ++i) o.v[i] = acc[c][i];
        store_vec<VW>(prow + fo[c], o);
      }
    }
  }
}

// One wave per hub row: out[row, :] = post * (partial chunks summed in chunk order).
template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    spmm_hub_combine(const IdxT* __restrict__ rowptr, const IdxT* __restrict__ hub_rows,
                     const IdxT* __restrict__ hub_chunk_ptr, int64_t n_hub,
                     const float* __restrict__ partial, float* __restrict__ out, int64_t F,
                     int64_t ldo, int mean, int accumulate, const float* __restrict__ relu_mask,
                     int64_t ldm, const uint32_t* __restrict__ relu_bits, int64_t ldb) {
  const int lane = lane_id();
  const int64_t h = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  if (h >= n_hub) return;
  const int64_t row = hub_rows[h];
  const int64_t c0 = hub_chunk_ptr[h];
  const int64_t c1 = hub_chunk_ptr[h + 1];
  const IdxT deg = rowptr[row + 1] - rowptr[row];
  const float cntf = static_cast<float>(deg > 0 ? deg : 1);
  for (int64_t f = lane; f < F; f += kWave) {
    // (eight chunks' loads in flight, added in chunk order: one at a time a hub row of 16 k slots
    // in 256-slot chunks was 64 serial trips to memory per 64 columns)
    float s = 0.f;
    int64_t c = c0;
    for (; c + 8 <= c1; c += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(c + u) * F + f];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < c1; ++c) s += partial[c * F + f];
    s = mean ? s / cntf : s;
    s = accumulate ? out[row * ldo + f] + s : s;
    if (relu_mask) s = relu_mask[row * ldm + f] > 0.f ? s : 0.f;
    if (relu_bits)
      s = ((relu_bits[((row >> 5) * ldb + (f >> 5)) * 32 + (row & 31)] >> (f & 31)) & 1u) ? s : 0.f;
    out[row * ldo + f] = s;
  }
}

// ---- min / max with first-on-tie arg ------------------------------------------------------
template <bool IS_MAX>
__device__ __forceinline__ bool better_val(float v, float best) {
  // NaN propagates (torch amax/amin semantics): a NaN beats any non-NaN.
  const bool vn = (v != v), bn = (best != best);
  const bool cmp = IS_MAX ? (v > best) : (v < best);
  return (!bn) & (vn | cmp);
}

// Hub rows (longer than hub_threshold) are not split here — an extremum needs no two-stage sum —
// but they go FIRST: the lowest wave ids take the rows of the hub list, so the 16 k-slot rows
// start at time 0 and run under the rest of the launch instead of forming its tail.
template <typename IdxT, int VW, int LPR, int CH, bool IS_MAX, bool IDENT>
__global__ void __launch_bounds__(kBlock)
    spmm_minmax_rows(SpmmDev<IdxT> a, const IdxT* __restrict__ hub_rows, int64_t n_hub) {
  constexpr int EPI = kWave / LPR;
  constexpr int U = spmm_unroll<LPR, CH>();
  constexpr int STEP = EPI * U;
  const int lane = lane_id();
  const int64_t wid = xcd_logical_block() * kWavesPerBlock + wave_in_block();
  int64_t row;
  if (wid < n_hub) {
    row = hub_rows[wid];
  } else {
    row = wid - n_hub;
    if (row >= a.n_rows) return;
  }
  const IdxT start = a.rowptr[row];
  const IdxT end = a.rowptr[row + 1];
  if (wid >= n_hub && n_hub > 0 && end - start > a.hub_threshold) return;  // done by a hub wave
  int fo[CH], head[CH];
  bool fv[CH];
  feature_slots<VW, LPR, CH>(lane, a.F, 1, fo, fv, head);
  const int sub = lane / LPR;
  const float init = IS_MAX ? -INFINITY : INFINITY;
  // per feature: the extremum, the slot OFFSET inside the row of its first occurrence (32 bits: a
  // row holds < 2^31 slots) and whether it was met more than once
  // ... from a DIFFERENT source row: parallel edges (the same neighbour twice) attain the value
  // together but send the whole gradient to one row, which is what the one-winner backward does
  // (round 3: on a multigraph like the products-shaped one they were the bulk of the marked
  // outputs and kept the tie kernel at 10 ms)
  // (r5) the bookkeeping costs VALU issue slots on every gathered element, and this kernel is
  // bound by them (5.4 ms on top of the 11.9 ms the same gather takes as a sum, F = 256): the
  // first source is kept in 32 bits (node ids of a graph differ in their low 32 bits), "empty" and
  // "the extremum is a NaN" are lane masks carried along instead of compares per element, and the
  // two orderings share their compares — 7 VALU instructions per element (3 compares of the value,
  // 1 of the source, 3 selects) + scalar mask logic, instead of ~14.
  float best[CH][VW];
  int32_t barg[CH][VW];
  uint32_t bsrc[CH][VW];
  bool tie[CH][VW], has[CH][VW], bnan[CH][VW];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      best[c][i] = init;
      barg[c][i] = -1;
      bsrc[c][i] = 0u;
      tie[c][i] = false;
      has[c][i] = false;
      bnan[c][i] = false;
    }
  }
  for (IdxT base = start; base < end; base += kWave) {
    const IdxT rem = end - base;
    const int cnt = rem < kWave ? static_cast<int>(rem) : kWave;
    IdxT myc = 0;
    if (lane < cnt) {
      if constexpr (IDENT) {
        myc = base + lane;
      } else {
        myc = __builtin_nontemporal_load(&a.col[base + lane]);  // streamed once
      }
    }
    const int32_t off0 = static_cast<int32_t>(base - start);
    for (int j = 0; j < cnt; j += STEP) {
      Vec<VW> v[U][CH];
      bool ok[U];
      uint32_t cs[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = j + u * EPI + sub;
        ok[u] = k < cnt;
        const int kk = ok[u] ? k : cnt - 1;
        IdxT c;
        if constexpr (EPI == 1) {
          c = bcast_uniform(myc, kk);
        } else {
          c = bcast_lane(myc, kk);
        }
        cs[u] = static_cast<uint32_t>(c);
        const float* __restrict__ xr = a.x + static_cast<int64_t>(c) * a.ldx;
#pragma unroll
        for (int c2 = 0; c2 < CH; ++c2) {
          if (fv[c2] && ok[u]) {
            v[u][c2] = load_vec<VW>(xr + fo[c2]);
          } else {
#pragma unroll
            for (int i = 0; i < VW; ++i) v[u][c2].v[i] = init;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int32_t slot = off0 + (j + u * EPI + sub);
#pragma unroll
        for (int c2 = 0; c2 < CH; ++c2) {
#pragma unroll
          for (int i = 0; i < VW; ++i) {
            const float val = v[u][c2].v[i];
            const bool vn = val != val;
            const bool c_gt = IS_MAX ? (val > best[c2][i]) : (val < best[c2][i]);
            const bool c_lt = IS_MAX ? (val < best[c2][i]) : (val > best[c2][i]);
            // better_val(val, best) and better_val(best, val): a NaN beats any non-NaN
            const bool gt = !bnan[c2][i] & (vn | c_gt);
            const bool lt = !vn & (bnan[c2][i] | c_lt);
            // the first valid slot always wins over the initial state
            const bool take = ok[u] & (!has[c2][i] | gt);
            const bool other_src = cs[u] != bsrc[c2][i];
            tie[c2][i] = take ? false : (tie[c2][i] | (ok[u] & has[c2][i] & !lt & other_src));
            best[c2][i] = take ? val : best[c2][i];
            barg[c2][i] = take ? slot : barg[c2][i];
            bsrc[c2][i] = take ? cs[u] : bsrc[c2][i];
            bnan[c2][i] = take ? vn : bnan[c2][i];
            has[c2][i] = has[c2][i] | take;
          }
        }
      }
    }
  }
  // merge the EPI sub-groups: better value wins, equal values keep the smaller slot (and tie)
#pragma unroll
  for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int i = 0; i < VW; ++i) {
        const float ov = bcast_lane(best[c][i], lane ^ off);
        const int32_t oa = bcast_lane(barg[c][i], lane ^ off);
        const bool ot = bcast_lane(static_cast<int32_t>(tie[c][i]), lane ^ off) != 0;
        const uint32_t os = static_cast<uint32_t>(
            bcast_lane(static_cast<int32_t>(bsrc[c][i]), lane ^ off));
        // Branch-free on purpose: hipcc 7.2 drops the guarded assignment of the nested-if form
        // of this update for VW = 4 (found by tests/test_gpu_ops.py::test_spmm_minmax_vs_oracle).
        const bool other_valid = oa >= 0;
        const bool mine_empty = barg[c][i] < 0;
        const bool other_better = better_val<IS_MAX>(ov, best[c][i]);
        const bool mine_better = better_val<IS_MAX>(best[c][i], ov);
        const bool equal = other_valid & !mine_empty & !other_better & !mine_better;
        const bool tie_earlier = equal & (oa < barg[c][i]);
        const bool take = other_valid & (mine_empty | other_better | tie_earlier);
        const bool take_tie = other_valid & (mine_empty | other_better);
        tie[c][i] = equal ? (tie[c][i] | ot | (os != bsrc[c][i])) : (take_tie ? ot : tie[c][i]);
        best[c][i] = take ? ov : best[c][i];
        barg[c][i] = take ? oa : barg[c][i];
        bsrc[c][i] = take ? os : bsrc[c][i];
      }
    }
  }
  if (lane < LPR) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (fv[c]) {
        Vec<VW> o;
#pragma unroll
        for (int i = 0; i < VW; ++i) o.v[i] = barg[c][i] < 0 ? 0.f : best[c][i];
        store_vec<VW>(a.out + row * a.ldo + fo[c], o);
        if (a.arg_out) {
#pragma unroll
          for (int i = 0; i < VW; ++i)
            a.arg_out[row * a.ldo + fo[c] + i] =
                barg[c][i] < 0 ? static_cast<IdxT>(-1) : start + static_cast<IdxT>(barg[c][i]);
        }
        if (a.arg32_out) {
          // -2: the gradient is split (several attaining neighbours, or an extremum of exactly 0,
          // which ties with the zero-initialised output of scatter_reduce_(include_self=False))
#pragma unroll
          for (int i = 0; i < VW; ++i) {
            const bool split = tie[c][i] | (best[c][i] == 0.f);
            a.arg32_out[row * a.F + fo[c] + i] = barg[c][i] < 0 ? -1 : (split ? -2 : barg[c][i]);
          }
        }
      }
    }
  }
}

// The extremum ALONE (no argument, no tie flag requested: `scatter(..., 'max')` of the unfused
// path, whose backward compares the saved output instead; inference): the kernel above spends
// ~14 VALU instructions per gathered element on its argument / tie / first-source bookkeeping —
// 5.4 ms on top of the 11.9 ms the same gather takes as a sum at the products shape, F = 256
// (profiles/r04_unfused_propagate.md: 0.35-0.48 of the HBM peak) — against 4 here:
// take = (v > best) | isnan(v): a NaN replaces anything and is then only replaced by a NaN.
template <typename IdxT, int VW, int LPR, int CH, bool IS_MAX, bool IDENT>
__global__ void __launch_bounds__(kBlock)
    spmm_minmax_rows_plain(SpmmDev<IdxT> a, const IdxT* __restrict__ hub_rows, int64_t n_hub) {
  constexpr int EPI = kWave / LPR;
  constexpr int U = spmm_unroll<LPR, CH>();
  constexpr int STEP = EPI * U;
  const int lane = lane_id();
  const int64_t wid = xcd_logical_block() * kWavesPerBlock + wave_in_block();
  int64_t row;
  if (wid < n_hub) {
    row = hub_rows[wid];
  } else {
    row = wid - n_hub;
    if (row >= a.n_rows) return;
  }
  const IdxT start = a.rowptr[row];
  const IdxT end = a.rowptr[row + 1];
  if (wid >= n_hub && n_hub > 0 && end - start > a.hub_threshold) return;  // done by a hub wave
  int fo[CH], head[CH];
  bool fv[CH];
  feature_slots<VW, LPR, CH>(lane, a.F, 1, fo, fv, head);
  const int sub = lane / LPR;
  const float init = IS_MAX ? -INFINITY : INFINITY;
  float best[CH][VW];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int i = 0; i < VW; ++i) best[c][i] = init;
  auto merge = [](float val, float cur) {
    const bool take = (IS_MAX ? (val > cur) : (val < cur)) | (val != val);
    return take ? val : cur;
  };
  for (IdxT base = start; base < end; base += kWave) {
    const IdxT rem = end - base;
    const int cnt = rem < kWave ? static_cast<int>(rem) : kWave;
    IdxT myc = 0;
    if (lane < cnt) {
      if constexpr (IDENT) {
        myc = base + lane;
      } else {
        myc = __builtin_nontemporal_load(&a.col[base + lane]);  // streamed once
      }
    }
    for (int j = 0; j < cnt; j += STEP) {
      Vec<VW> v[U][CH];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = j + u * EPI + sub;
        const bool ok = k < cnt;
        const int kk = ok ? k : cnt - 1;
        IdxT c;
        if constexpr (EPI == 1) {
          c = bcast_uniform(myc, kk);
        } else {
          c = bcast_lane(myc, kk);
        }
        const float* __restrict__ xr = a.x + static_cast<int64_t>(c) * a.ldx;
#pragma unroll
        for (int c2 = 0; c2 < CH; ++c2) {
          if (fv[c2] && ok) {
            v[u][c2] = load_vec<VW>(xr + fo[c2]);
          } else {
#pragma unroll
            for (int i = 0; i < VW; ++i) v[u][c2].v[i] = init;  // (neutral: never taken)
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c2 = 0; c2 < CH; ++c2)
#pragma unroll
          for (int i = 0; i < VW; ++i) best[c2][i] = merge(v[u][c2].v[i], best[c2][i]);
    }
  }
#pragma unroll
  for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int i = 0; i < VW; ++i) best[c][i] = merge(bcast_lane(best[c][i], lane ^ off), best[c][i]);
  }
  if (lane < LPR) {
    const bool empty = end == start;  // (scatter_reduce_ with include_self=False: empty groups -> 0)
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (fv[c]) {
        Vec<VW> o;
#pragma unroll
        for (int i = 0; i < VW; ++i) o.v[i] = empty ? 0.f : best[c][i];
        store_vec<VW>(a.out + row * a.ldo + fo[c], o);
      }
    }
  }
}

// Fast half of the min/max backward: outputs with a unique extremum.  Thread = (row, VW features):
// one read of arg32 and grad_out, a col lookup inside the row's own slot range, one fp32 atomic
// into the single attaining source row.  No edge pass.
template <typename IdxT, int VW>
__global__ void __launch_bounds__(kBlock)
    spmm_minmax_bwd_arg_kernel(const IdxT* __restrict__ rowptr, const IdxT* __restrict__ col,
                               const int32_t* __restrict__ arg32,
                               const float* __restrict__ grad_out, int64_t ldgo, int64_t n_rows,
                               int64_t units, int64_t F, float* __restrict__ grad_x,
                               int64_t ldg) {
  const int64_t total = n_rows * units;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t row = t / units;
    const int64_t f = (t - row * units) * VW;
    const IdxT start = rowptr[row];
    int32_t a4[VW];
    if constexpr (VW == 4) {
      const int4 v = *reinterpret_cast<const int4*>(arg32 + row * F + f);
      a4[0] = v.x; a4[1] = v.y; a4[2] = v.z; a4[3] = v.w;
    } else {
      a4[0] = arg32[row * F + f];
    }
    const Vec<VW> g = load_vec<VW>(grad_out + row * ldgo + f);
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      if (a4[i] >= 0) {
        const IdxT k = start + static_cast<IdxT>(a4[i]);
        const int64_t src = col ? static_cast<int64_t>(col[k]) : static_cast<int64_t>(k);
        atomicAdd(grad_x + src * ldg + f + i, g.v[i]);
      }
    }
  }
}

// ---- atomic-free fast half of the min/max backward (round 3) ------------------------------------
// The one-atomic-per-output kernel above moves 4 bytes per atomic but a whole cache line per touched
// source row segment: N x F scattered fp32 atomics = 30 ms at the products shape, F = 256 (the
// forward takes 12).  Every output (i, f) with a unique extremum has exactly ONE winning edge, so
// the N x F gradient values can be regrouped BY EDGE in one streaming pass and then read once:
//   pack        destination-driven, a wave per row i: lane l holds arg32[i, 4l..4l+3] and
//               grad_out[i, 4l..4l+3]; for every slot s of the row, ballots (arg == s) give each
//               winner its rank, and the winners' (feature, value) pairs go to the row's own
//               F-entry region of `entries` in slot order; seg[k] = (offset in the region << 16 |
//               count) per edge.  No global scan: a row owns entries [i F, (i + 1) F).
//   accumulate  source-driven over the transposed CSR, a wave per source j with an F-float row in
//               LDS: for each out-edge (8 in flight) the lanes read the edge's packed pairs —
//               one or two cache lines instead of the scattered grad_out row — and add them with
//               ds_add_f32 (the features of one edge are distinct, edges are processed in slot
//               order: deterministic); the row is written to grad_x once.  No atomics on global
//               memory, no memset.
// Per edge: 16 bytes of indices, 4 of seg, ~8 bytes x (F / deg) of pairs.  Two earlier forms were
// measured at the products shape, F = 256: one bit per (edge, feature) + predicated 16-byte loads
// from grad_out[i]: 3.5 ms (masks) + 16.3 ms (every edge still touches ~6 of the row's 8 lines).
// Outputs marked -2 (ties / extremum 0) go to spmm_minmax_bwd_dst afterwards, as before.
struct MinmaxEntry {
  uint32_t f;
  float v;
};
constexpr int kPackCap = 1024;  // slots of a row counted in LDS (longer rows take the ballot loop)

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    minmax_pack_kernel(const IdxT* __restrict__ rowptr, const int32_t* __restrict__ arg32,
                       const float* __restrict__ grad_out, int64_t ldgo, int64_t n_rows, int64_t F,
                       MinmaxEntry* __restrict__ entries, uint32_t* __restrict__ seg) {
  __shared__ int pack_cnt[kWavesPerBlock * kPackCap];
  const int lane = lane_id();
  const int64_t row = xcd_logical_block() * kWavesPerBlock + wave_in_block();
  if (row >= n_rows) return;
  const int64_t start = rowptr[row];
  const int64_t deg = static_cast<int64_t>(rowptr[row + 1]) - start;
  if (deg == 0) return;
  const int Q = static_cast<int>((F + 255) / 256);
  const uint64_t lt = (1ull << lane) - 1;  // lanes below this one
  MinmaxEntry* __restrict__ reg = entries + row * F;
  if (Q == 1 && deg <= kPackCap) {
    // counting sort of the row's <= 256 winners by slot, in LDS: one ds_add_rtn per entry gives
    // its rank inside its edge (the order inside an edge is free: its features are distinct), a
    // wave scan over the deg counters gives the edge offsets; 4 global stores per lane and ROW.
    // (A loop over the slots with four ballots + four conditional stores per slot — 250 M store
    // instructions at the products shape — measured 7.5 ms for this kernel.)
    int* __restrict__ cnt = pack_cnt + wave_in_block() * kPackCap;
    for (int s = lane; s < deg; s += kWave) cnt[s] = 0;
    const int64_t f0 = 4 * lane;
    int4 a = {-1, -1, -1, -1};
    Vec<4> g = {{0.f, 0.f, 0.f, 0.f}};
    if (f0 < F) {
      a = *reinterpret_cast<const int4*>(arg32 + row * F + f0);
      g = load_vec<4>(grad_out + row * ldgo + f0);
    }
    int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    if (a.x >= 0) r0 = atomicAdd(cnt + a.x, 1);
    if (a.y >= 0) r1 = atomicAdd(cnt + a.y, 1);
    if (a.z >= 0) r2 = atomicAdd(cnt + a.z, 1);
    if (a.w >= 0) r3 = atomicAdd(cnt + a.w, 1);
    // exclusive scan of cnt[0 .. deg) in place -> edge offsets; seg words written on the way
    int carry = 0;
    for (int s0 = 0; s0 < deg; s0 += kWave) {
      const int s = s0 + lane;
      const int c = s < deg ? cnt[s] : 0;
      int inc = c;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const int t = __shfl_up(inc, off, kWave);
        if (lane >= off) inc += t;
      }
      const int excl = carry + inc - c;
      if (s < deg) {
        cnt[s] = excl;
        seg[start + s] = (static_cast<uint32_t>(excl) << 16) | static_cast<uint32_t>(c);
      }
      carry += bcast_uniform(inc, kWave - 1);
    }
    if (a.x >= 0) reg[cnt[a.x] + r0] = {static_cast<uint32_t>(f0), g.v[0]};
    if (a.y >= 0) reg[cnt[a.y] + r1] = {static_cast<uint32_t>(f0 + 1), g.v[1]};
    if (a.z >= 0) reg[cnt[a.z] + r2] = {static_cast<uint32_t>(f0 + 2), g.v[2]};
    if (a.w >= 0) reg[cnt[a.w] + r3] = {static_cast<uint32_t>(f0 + 3), g.v[3]};
    return;
  }
  if (Q == 1) {  // a longer row: one pass over the slots, ballots give the ranks
    const int64_t f0 = 4 * lane;
    int4 a = {-1, -1, -1, -1};
    Vec<4> g = {{0.f, 0.f, 0.f, 0.f}};
    if (f0 < F) {
      a = *reinterpret_cast<const int4*>(arg32 + row * F + f0);
      g = load_vec<4>(grad_out + row * ldgo + f0);
    }
    uint32_t run = 0;
    for (int64_t s = 0; s < deg; ++s) {
      const int32_t ss = static_cast<int32_t>(s);
      const uint64_t b0 = __ballot(a.x == ss), b1 = __ballot(a.y == ss);
      const uint64_t b2 = __ballot(a.z == ss), b3 = __ballot(a.w == ss);
      const uint32_t n0 = __popcll(b0), n1 = __popcll(b1), n2 = __popcll(b2), n3 = __popcll(b3);
      if (a.x == ss) reg[run + __popcll(b0 & lt)] = {static_cast<uint32_t>(f0), g.v[0]};
      if (a.y == ss) reg[run + n0 + __popcll(b1 & lt)] = {static_cast<uint32_t>(f0 + 1), g.v[1]};
      if (a.z == ss)
        reg[run + n0 + n1 + __popcll(b2 & lt)] = {static_cast<uint32_t>(f0 + 2), g.v[2]};
      if (a.w == ss)
        reg[run + n0 + n1 + n2 + __popcll(b3 & lt)] = {static_cast<uint32_t>(f0 + 3), g.v[3]};
      const uint32_t n = n0 + n1 + n2 + n3;
      if (lane == 0) seg[start + s] = (run << 16) | n;
      run += n;
    }
    return;
  }
  // wide rows: slot-major too, the feature blocks of a slot one after the other
  uint32_t run = 0;
  for (int64_t s = 0; s < deg; ++s) {
    const int32_t ss = static_cast<int32_t>(s);
    const uint32_t run0 = run;
    for (int q = 0; q < Q; ++q) {
      const int64_t f0 = static_cast<int64_t>(q) * 256 + 4 * lane;
      int4 a = {-1, -1, -1, -1};
      if (f0 < F) a = *reinterpret_cast<const int4*>(arg32 + row * F + f0);
      const uint64_t b0 = __ballot(a.x == ss), b1 = __ballot(a.y == ss);
      const uint64_t b2 = __ballot(a.z == ss), b3 = __ballot(a.w == ss);
      const uint32_t n0 = __popcll(b0), n1 = __popcll(b1), n2 = __popcll(b2), n3 = __popcll(b3);
      const float* __restrict__ gp = grad_out + row * ldgo + f0;
      if (a.x == ss) reg[run + __popcll(b0 & lt)] = {static_cast<uint32_t>(f0), gp[0]};
      if (a.y == ss) reg[run + n0 + __popcll(b1 & lt)] = {static_cast<uint32_t>(f0 + 1), gp[1]};
      if (a.z == ss) reg[run + n0 + n1 + __popcll(b2 & lt)] = {static_cast<uint32_t>(f0 + 2), gp[2]};
      if (a.w == ss)
        reg[run + n0 + n1 + n2 + __popcll(b3 & lt)] = {static_cast<uint32_t>(f0 + 3), gp[3]};
      run += n0 + n1 + n2 + n3;
    }
    if (lane == 0) seg[start + s] = (run0 << 16) | (run - run0);
  }
}

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    minmax_bwd_src_kernel(const IdxT* __restrict__ rowptr_t, const IdxT* __restrict__ col_t,
                          const IdxT* __restrict__ slot_map,
                          const MinmaxEntry* __restrict__ entries, const uint32_t* __restrict__ seg,
                          int64_t n_src, int64_t F, float* __restrict__ grad_x, int64_t ldg) {
  extern __shared__ __align__(16) float rows_lds[];  // [waves per block][F]
  const int lane = lane_id();
  const int64_t j = xcd_logical_block() * kWavesPerBlock + wave_in_block();
  if (j >= n_src) return;
  float* __restrict__ acc = rows_lds + static_cast<int64_t>(wave_in_block()) * F;
  for (int64_t f = 4 * lane; f < F; f += 4 * kWave)
    *reinterpret_cast<Vec<4>*>(acc + f) = Vec<4>{{0.f, 0.f, 0.f, 0.f}};
  const IdxT start = rowptr_t[j];
  const IdxT end = rowptr_t[j + 1];
  constexpr int U = 8;
  for (IdxT base = start; base < end; base += kWave) {
    const IdxT rem = end - base;
    const int cnt = rem < kWave ? static_cast<int>(rem) : kWave;
    int64_t my_dst = 0;
    uint32_t my_seg = 0;
    if (lane < cnt) {
      my_dst = static_cast<int64_t>(__builtin_nontemporal_load(&col_t[base + lane]));
      my_seg = seg[__builtin_nontemporal_load(&slot_map[base + lane])];
    }
    for (int e = 0; e < cnt; e += U) {
      // the pairs of U edges (first 64 of each) are requested before the first add
      MinmaxEntry en[U];
      int n[U];
      const MinmaxEntry* ep[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ee = e + u < cnt ? e + u : cnt - 1;
        const int64_t i = bcast_uniform(my_dst, ee);
        const uint32_t sg = static_cast<uint32_t>(bcast_uniform(static_cast<int32_t>(my_seg), ee));
        n[u] = e + u < cnt ? static_cast<int>(sg & 0xffffu) : 0;
        ep[u] = entries + i * F + (sg >> 16);
        en[u] = {0u, 0.f};
        if (lane < n[u]) en[u] = ep[u][lane];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (lane < n[u]) atomicAdd(acc + en[u].f, en[u].v);  // ds_add_f32: features of an edge differ
        for (int t = kWave; t < n[u]; t += kWave) {  // an edge with more than 64 winners (rare)
          if (t + lane < n[u]) {
            const MinmaxEntry x = ep[u][t + lane];
            atomicAdd(acc + x.f, x.v);
          }
        }
      }
    }
  }
  for (int64_t f = 4 * lane; f < F; f += 4 * kWave)
    store_vec<4>(grad_x + j * ldg + f, *reinterpret_cast<const Vec<4>*>(acc + f));
}

// ---- one-pass multi-reduce (FusedAggregation, nn/aggr/fused.py:191-336) ----------------------
// sum, sum of squares, min and max of the rows of a group in ONE read of the rows: the SpMM's lane
// mapping with four accumulators per feature.  Outputs that are not requested are null; empty
// groups give 0 everywhere (utils/_scatter.py semantics).
template <typename IdxT, int VW, int LPR, int CH, bool IDENT>
__global__ void __launch_bounds__(kBlock)
    spmm_multi_rows(SpmmDev<IdxT> a, float* __restrict__ out_sum, float* __restrict__ out_sq,
                    float* __restrict__ out_min, float* __restrict__ out_max) {
  constexpr int EPI = kWave / LPR;
  constexpr int U = CH == 1 ? (LPR < 4 ? LPR : 4) : 2;
  constexpr int STEP = EPI * U;
  const int lane = lane_id();
  const int64_t row = xcd_logical_block() * kWavesPerBlock + wave_in_block();
  if (row >= a.n_rows) return;
  const IdxT start = a.rowptr[row];
  const IdxT end = a.rowptr[row + 1];
  int fo[CH], head[CH];
  bool fv[CH];
  feature_slots<VW, LPR, CH>(lane, a.F, 1, fo, fv, head);
  const int sub = lane / LPR;
  float s1[CH][VW], s2[CH][VW], mn[CH][VW], mx[CH][VW];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      s1[c][i] = 0.f;
      s2[c][i] = 0.f;
      mn[c][i] = INFINITY;
      mx[c][i] = -INFINITY;
    }
  }
  for (IdxT base = start; base < end; base += kWave) {
    const IdxT rem = end - base;
    const int cnt = rem < kWave ? static_cast<int>(rem) : kWave;
    IdxT myc = 0;
    if (lane < cnt) {
      if constexpr (IDENT) {
        myc = base + lane;
      } else {
        myc = __builtin_nontemporal_load(&a.col[base + lane]);
      }
    }
    for (int j = 0; j < cnt; j += STEP) {
      Vec<VW> v[U][CH];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = j + u * EPI + sub;
        ok[u] = k < cnt;
        const int kk = ok[u] ? k : cnt - 1;
        IdxT c;
        if constexpr (EPI == 1) {
          c = bcast_uniform(myc, kk);
        } else {
          c = bcast_lane(myc, kk);
        }
        const float* __restrict__ xr = a.x + static_cast<int64_t>(c) * a.ldx;
#pragma unroll
        for (int c2 = 0; c2 < CH; ++c2) {
          if (fv[c2] && ok[u]) v[u][c2] = load_vec<VW>(xr + fo[c2]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int c2 = 0; c2 < CH; ++c2) {
          if (fv[c2] && ok[u]) {
#pragma unroll
            for (int i = 0; i < VW; ++i) {
              const float val = v[u][c2].v[i];
              s1[c2][i] += val;
              s2[c2][i] = fmaf(val, val, s2[c2][i]);
              // NaN propagates like torch's amin / amax
              mn[c2][i] = better_val<false>(val, mn[c2][i]) ? val : mn[c2][i];
              mx[c2][i] = better_val<true>(val, mx[c2][i]) ? val : mx[c2][i];
            }
          }
        }
      }
    }
  }
  combine_subgroups<VW, LPR, CH>(s1);
  combine_subgroups<VW, LPR, CH>(s2);
#pragma unroll
  for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int i = 0; i < VW; ++i) {
        const float on = __shfl_xor(mn[c][i], off, kWave);
        const float ox = __shfl_xor(mx[c][i], off, kWave);
        mn[c][i] = better_val<false>(on, mn[c][i]) ? on : mn[c][i];
        mx[c][i] = better_val<true>(ox, mx[c][i]) ? ox : mx[c][i];
      }
    }
  }
  if (lane < LPR) {
    const bool empty = end <= start;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (!fv[c]) continue;
      Vec<VW> o;
      const int64_t off = row * a.ldo + fo[c];
      if (out_sum) {
#pragma unroll
        for (int i = 0; i < VW; ++i) o.v[i] = s1[c][i];
        store_vec<VW>(out_sum + off, o);
      }
      if (out_sq) {
#pragma unroll
        for (int i = 0; i < VW; ++i) o.v[i] = s2[c][i];
        store_vec<VW>(out_sq + off, o);
      }
      if (out_min) {
#pragma unroll
        for (int i = 0; i < VW; ++i) o.v[i] = empty ? 0.f : mn[c][i];
        store_vec<VW>(out_min + off, o);
      }
      if (out_max) {
#pragma unroll
        for (int i = 0; i < VW; ++i) o.v[i] = empty ? 0.f : mx[c][i];
        store_vec<VW>(out_max + off, o);
      }
    }
  }
}

// ---- min/max backward helpers (reference tie rule, see pyg_amd.h) --------------------------
template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    spmm_tie_count_rows(const IdxT* __restrict__ rowptr, const IdxT* __restrict__ col,
                        const float* __restrict__ x, int64_t ldx, const float* __restrict__ out,
                        int64_t ldo, int64_t n_rows, int64_t F, int count_self,
                        float* __restrict__ ntie) {
  const int lane = lane_id();
  const int64_t row = xcd_logical_block() * kWavesPerBlock + wave_in_block();
  if (row >= n_rows) return;
  const IdxT start = rowptr[row];
  const IdxT end = rowptr[row + 1];
  for (int64_t f = lane; f < F; f += kWave) {
    const float o = out[row * ldo + f];
    float n = (count_self && o == 0.f) ? 1.f : 0.f;
    for (IdxT k = start; k < end; ++k) {
      const int64_t c = col ? static_cast<int64_t>(col[k]) : static_cast<int64_t>(k);
      n += (x[c * ldx + f] == o) ? 1.f : 0.f;
    }
    ntie[row * ldo + f] = n;
  }
}

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    spmm_minmax_bwd_rows(const IdxT* __restrict__ rowptr_t, const IdxT* __restrict__ col_t,
                         const float* __restrict__ x, int64_t ldx, const float* __restrict__ out,
                         const float* __restrict__ grad_out, const float* __restrict__ ntie,
                         int64_t ldo, int64_t n_src, int64_t F, float* __restrict__ grad_x,
                         int64_t ldg) {
  const int lane = lane_id();
  const int64_t j = xcd_logical_block() * kWavesPerBlock + wave_in_block();
  if (j >= n_src) return;
  const IdxT start = rowptr_t[j];
  const IdxT end = rowptr_t[j + 1];
  for (int64_t f = lane; f < F; f += kWave) {
    const float xv = x[j * ldx + f];
    float g = 0.f;
    for (IdxT k = start; k < end; ++k) {
      const int64_t i = col_t[k];
      if (xv == out[i * ldo + f]) g += grad_out[i * ldo + f] / ntie[i * ldo + f];
    }
    grad_x[j * ldg + f] = g;
  }
}

// Destination-driven min/max backward with the reference tie rule, ONE launch: a wave owns
// destination row i with the forward's lane mapping (slot indices staged 64 at a time, 16-byte
// source-row loads, U rows in flight).  Pass 1 counts, per feature, the neighbours that attain
// out[i, f] (+ the zero-initialised self when out == 0, scatter_reduce_(include_self=False)'s
// backward rule); pass 2 walks the row again and sends grad_out[i, f] / ties to every attaining
// source with one fp32 atomic.  Gathers the E source rows (twice) instead of the three
// destination-row gathers per edge of the source-driven form, and needs neither a by-source sort
// of the graph nor a tie-count tensor: 42 ms instead of 144 ms at the products shape, F = 256
// (the forward takes 15 ms; the N x F scattered atomics are the larger half of the difference —
// a single-pass variant that finishes tie-free features inside pass 1 measured the same).
template <typename IdxT, int VW, int LPR, int CH>
__global__ void __launch_bounds__(kBlock)
    spmm_minmax_bwd_dst(const IdxT* __restrict__ rowptr, const IdxT* __restrict__ col,
                        const float* __restrict__ x, int64_t ldx, const float* __restrict__ out,
                        int64_t ldo, const float* __restrict__ grad_out, int64_t ldgo,
                        int64_t n_rows, int64_t F, int count_self,
                        const int32_t* __restrict__ arg32, float* __restrict__ grad_x,
                        int64_t ldg) {
  constexpr int EPI = kWave / LPR;
  constexpr int U = spmm_unroll<LPR, CH>();
  constexpr int STEP = EPI * U;
  const int lane = lane_id();
  const int64_t row = xcd_logical_block() * kWavesPerBlock + wave_in_block();
  if (row >= n_rows) return;
  const IdxT start = rowptr[row];
  const IdxT end = rowptr[row + 1];
  int fo[CH], head[CH];
  bool fv[CH];
  feature_slots<VW, LPR, CH>(lane, F, 1, fo, fv, head);
  const int sub = lane / LPR;
  // with the forward's arg32: only the outputs marked -2 (split gradient) are handled here, and
  // a row without any is left after one read of its arg32 row
  bool mine[CH][VW];
  bool any = arg32 == nullptr;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      mine[c][i] = fv[c] && (arg32 == nullptr || arg32[row * F + fo[c] + i] == -2);
      any |= mine[c][i];
    }
  }
  if (__ballot(any) == 0) return;
  Vec<VW> o[CH];
  float ties[CH][VW];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (fv[c]) {
      o[c] = load_vec<VW>(out + row * ldo + fo[c]);
    } else {
#pragma unroll
      for (int i = 0; i < VW; ++i) o[c].v[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < VW; ++i) ties[c][i] = 0.f;
  }
  Vec<VW> q[CH];
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    for (IdxT base = start; base < end; base += kWave) {
      const IdxT rem = end - base;
      const int cnt = rem < kWave ? static_cast<int>(rem) : kWave;
      IdxT myc = 0;
      if (lane < cnt) myc = col ? col[base + lane] : base + lane;
      for (int j = 0; j < cnt; j += STEP) {
        Vec<VW> v[U][CH];
        bool ok[U];
        int64_t src[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = j + u * EPI + sub;
          ok[u] = k < cnt;
          const int kk = ok[u] ? k : cnt - 1;
          IdxT c;
          if constexpr (EPI == 1) {
            c = bcast_uniform(myc, kk);
          } else {
            c = bcast_lane(myc, kk);
          }
          src[u] = static_cast<int64_t>(c);
          const float* __restrict__ xr = x + src[u] * ldx;
#pragma unroll
          for (int c2 = 0; c2 < CH; ++c2) {
            if (fv[c2] && ok[u]) v[u][c2] = load_vec<VW>(xr + fo[c2]);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int c2 = 0; c2 < CH; ++c2) {
            if (fv[c2] && ok[u]) {
#pragma unroll
              for (int i = 0; i < VW; ++i) {
                const bool hit = v[u][c2].v[i] == o[c2].v[i];
                if (pass == 0) {
                  ties[c2][i] += hit ? 1.f : 0.f;
                } else if (hit && mine[c2][i]) {
                  atomicAdd(grad_x + src[u] * ldg + fo[c2] + i, q[c2].v[i]);
                }
              }
            }
          }
        }
      }
    }
    if (pass == 0) {
      combine_subgroups<VW, LPR, CH>(ties);  // every sub-group now holds the row's tie counts
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        Vec<VW> g;
        if (fv[c]) {
          g = load_vec<VW>(grad_out + row * ldgo + fo[c]);
        } else {
#pragma unroll
          for (int i = 0; i < VW; ++i) g.v[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < VW; ++i) {
          const float n = ties[c][i] + ((count_self && o[c].v[i] == 0.f) ? 1.f : 0.f);
          q[c].v[i] = n > 0.f ? g.v[i] / n : 0.f;
        }
      }
    }
  }
}

// ---- SDDMM ----------------------------------------------------------------------------------
// grad_w[e(k), h] = <grad_out[i, head h], x[col[k], head h]> for every slot k of row i.  Same
// mapping as the SpMM: one wave per destination row, the row of grad_out stays in registers, slot
// indices are staged 64 at a time, U source rows are in flight per lane.  The per-head dot product
// is finished with a SEGMENTED xor-free shuffle reduction over the lanes that share a head
// (p += shfl_down(p, 2^s) while lane + 2^s is still inside the head): exact for any head width
// that is a multiple of VW, e.g. 8 lanes for C = 32, 10 lanes for C = 40.
//
// AGG (pygamd_sddmm_spmm_csr): the same pass also aggregates the gathered rows with per-(slot, head)
// weights, agg[i, f] = sum_k w[e(k), head(f)] * x[col[k], f] — the two halves of a GAT layer's
// backward over the by-source CSR (row held in registers = the source's own features, gathered
// rows = the destinations' gradients): d alpha of every edge AND the aggregation's d x from ONE
// gather of the gradient rows, which is what both cost.
template <typename IdxT, int VW, int LPR, int CH, bool AGG>
__global__ void __launch_bounds__(kBlock)
    sddmm_rows(const IdxT* __restrict__ rowptr, const IdxT* __restrict__ col,
               const IdxT* __restrict__ eid, const float* __restrict__ grad_out, int64_t ldg,
               const float* __restrict__ x, int64_t ldx, int64_t n_rows, int64_t F, int w_heads,
               int head_dim, int use_atomic, float* __restrict__ grad_w,
               const float* __restrict__ w, float* __restrict__ agg, int64_t lda) {
  constexpr int EPI = kWave / LPR;
  constexpr int U = (CH == 1) ? (LPR < 4 ? LPR : 4) : (AGG ? 4 : 2);
  constexpr int STEP = EPI * U;
  const int lane = lane_id();
  const int64_t row = xcd_logical_block() * kWavesPerBlock + wave_in_block();
  if (row >= n_rows) return;
  const IdxT start = rowptr[row];
  const IdxT end = rowptr[row + 1];
  int fo[CH], head[CH];
  bool fv[CH];
  feature_slots<VW, LPR, CH>(lane, F, head_dim, fo, fv, head);
  const int sub = lane / LPR;
  // segment bookkeeping: which shuffle distances stay inside my (sub-group, head) segment, and
  // whether I am the first lane of it
  int okmask[CH];
  bool first[CH];
  Vec<VW> gv[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int hid = fv[c] ? head[c] : -1 - lane;
    okmask[c] = 0;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const int off = 1 << s;
      const int oh = __shfl_down(hid, off, kWave);
      const bool same = (lane + off < kWave) && ((lane + off) / LPR == sub) && (oh == hid);
      okmask[c] |= same ? (1 << s) : 0;
    }
    const int ph = __shfl_up(hid, 1, kWave);
    first[c] = fv[c] && ((lane % LPR) == 0 || ph != hid);
    if (fv[c]) {
      gv[c] = load_vec<VW>(grad_out + row * ldg + fo[c]);
    } else {
#pragma unroll
      for (int i = 0; i < VW; ++i) gv[c].v[i] = 0.f;
    }
  }
  float sum[CH][VW];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int i = 0; i < VW; ++i) sum[c][i] = 0.f;
  for (IdxT base = start; base < end; base += kWave) {
    const IdxT rem = end - base;
    const int cnt = rem < kWave ? static_cast<int>(rem) : kWave;
    IdxT myc = 0, mye = 0;
    if (lane < cnt) {
      const IdxT k = base + lane;
      myc = col ? col[k] : k;
      mye = eid ? eid[k] : k;
    }
    for (int j = 0; j < cnt; j += STEP) {
      float p[U][CH];
      IdxT e[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = j + u * EPI + sub;
        ok[u] = k < cnt;
        const int kk = ok[u] ? k : cnt - 1;
        IdxT c;
        if constexpr (EPI == 1) {
          c = bcast_uniform(myc, kk);
          e[u] = bcast_uniform(mye, kk);
        } else {
          c = bcast_lane(myc, kk);
          e[u] = bcast_lane(mye, kk);
        }
        const float* __restrict__ xr = x + static_cast<int64_t>(c) * ldx;
#pragma unroll
        for (int c2 = 0; c2 < CH; ++c2) {
          float acc = 0.f;
          if (fv[c2] && ok[u]) {
            const Vec<VW> v = load_vec<VW>(xr + fo[c2]);
#pragma unroll
            for (int i = 0; i < VW; ++i) acc = fmaf(v.v[i], gv[c2].v[i], acc);
            if constexpr (AGG) {
              const float wv = w[static_cast<int64_t>(e[u]) * w_heads + head[c2]];
#pragma unroll
              for (int i = 0; i < VW; ++i) sum[c2][i] = fmaf(v.v[i], wv, sum[c2][i]);
            }
          }
          p[u][c2] = acc;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int c2 = 0; c2 < CH; ++c2) {
          float v = p[u][c2];
#pragma unroll
          for (int s = 0; s < 6; ++s) {
            const float o = __shfl_down(v, 1 << s, kWave);
            v += ((okmask[c2] >> s) & 1) ? o : 0.f;
          }
          if (first[c2] && ok[u]) {
            float* dst = grad_w + static_cast<int64_t>(e[u]) * w_heads + head[c2];
            if (use_atomic) {
              atomicAdd(dst, v);
            } else {
              *dst = v;
            }
          }
        }
      }
    }
  }
  if constexpr (AGG) {
    combine_subgroups<VW, LPR, CH>(sum);
    if (sub == 0) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (fv[c]) {
          Vec<VW> o;
#pragma unroll
          for (int i = 0; i < VW; ++i) o.v[i] = sum[c][i];
          store_vec<VW>(agg + row * lda + fo[c], o);
        }
      }
    }
  }
}

// ---- host dispatch --------------------------------------------------------------------------
template <typename IdxT>
static SpmmDev<IdxT> make_dev(const pygamd_spmm_args* p) {
  SpmmDev<IdxT> a;
  a.rowptr = static_cast<const IdxT*>(p->rowptr);
  a.col = static_cast<const IdxT*>(p->col);
  a.eid = static_cast<const IdxT*>(p->eid);
  a.w = p->w;
  a.src_scale = p->src_scale;
  a.x = p->x;
  a.out = p->out;
  a.arg_out = static_cast<IdxT*>(p->arg_out);
  a.arg32_out = p->arg32_out;
  a.relu_mask = p->relu_mask;
  a.ldm = p->ld_mask;
  a.relu_bits = p->relu_bits;
  a.ldb = p->ld_bits;
  a.src_bits = p->src_bits;
  a.src_bits_set = p->src_bits_set;
  a.n_src = p->n_src;
  a.n_rows = p->n_rows;
  a.F = p->F;
  a.ldx = p->ldx;
  a.ldo = p->ldo;
  a.w_heads = p->w_heads < 1 ? 1 : p->w_heads;
  a.head_dim = (p->w_heads > 1) ? p->head_dim : static_cast<int>(p->F > 0 ? p->F : 1);
  a.mean = (p->reduce == PYGAMD_MEAN);
  a.accumulate = p->accumulate;
  a.hub_threshold = (p->n_hub > 0) ? p->hub_threshold : 0;
  a.rowend = static_cast<const IdxT*>(p->rowend);
  a.accumulate_rows = p->accumulate_rows;
  return a;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

struct Shape {
  int vw, lpr, ch;
  unsigned tiles;
};

static Shape pick_shape(const pygamd_spmm_args* p) {
  Shape s;
  const bool v4 = (p->F % 4 == 0) && (p->ldx % 4 == 0) && (p->ldo % 4 == 0) &&
                  aligned16(p->x) && aligned16(p->out) &&
                  (!p->relu_mask || (p->ld_mask % 4 == 0 && aligned16(p->relu_mask))) &&
                  (p->w_heads <= 1 || p->head_dim % 4 == 0);
  s.vw = v4 ? 4 : 1;
  const int64_t units = ceil_div(p->F, s.vw);  // lanes needed to cover a row once
  int lpr = 4;
  while (lpr < 64 && lpr < units) lpr <<= 1;
  s.lpr = lpr;
  s.ch = (lpr == 64 && units > 64) ? 2 : 1;
  s.tiles = static_cast<unsigned>(ceil_div(units, static_cast<int64_t>(s.lpr) * s.ch));
  return s;
}

template <typename IdxT, int VW, int LPR, int CH, int WMODE, bool IDENT>
static int launch_sum(const pygamd_spmm_args* p, const Shape& s, float* partial,
                      hipStream_t st) {
  SpmmDev<IdxT> a = make_dev<IdxT>(p);
  if (p->hub_phase != 2) {
    bool done = false;
    if constexpr (WMODE == 3) {
      if (!p->accumulate && !p->relu_mask && !p->relu_bits && p->n_src > 0) {
        constexpr int R = 4;
        dim3 grid(wave_grid(ceil_div(p->n_rows, static_cast<int64_t>(R))), s.tiles);
        hipLaunchKernelGGL((spmm_sum_rows_sparse<IdxT, VW, LPR, CH, R>), grid, dim3(kBlock), 0,
                           st, a);
        done = true;
      }
    }
    if (!done) {
      dim3 grid(wave_grid(p->n_rows), s.tiles);
      hipLaunchKernelGGL((spmm_sum_rows<IdxT, VW, LPR, CH, WMODE, IDENT>), grid, dim3(kBlock), 0,
                         st, a);
    }
    PYGAMD_LAUNCH_CHECK();
  }
  if (p->n_hub > 0 && p->hub_phase != 1) {
    dim3 hgrid(wave_grid(p->n_chunks), s.tiles);
    hipLaunchKernelGGL((spmm_hub_chunks<IdxT, VW, LPR, CH, WMODE, IDENT>), hgrid, dim3(kBlock),
                       0, st, a, static_cast<const IdxT*>(p->hub_rows),
                       static_cast<const IdxT*>(p->hub_chunk_ptr), p->n_hub, p->n_chunks,
                       p->hub_chunk, partial);
    PYGAMD_LAUNCH_CHECK();
    dim3 cgrid(static_cast<unsigned>(ceil_div(p->n_hub, kWavesPerBlock)));
    hipLaunchKernelGGL((spmm_hub_combine<IdxT>), cgrid, dim3(kBlock), 0, st, a.rowptr,
                       static_cast<const IdxT*>(p->hub_rows),
                       static_cast<const IdxT*>(p->hub_chunk_ptr), p->n_hub, partial, a.out,
                       a.F, a.ldo, a.mean, a.accumulate, a.relu_mask, a.ldm, a.relu_bits, a.ldb);
    PYGAMD_LAUNCH_CHECK();
  }
  return PYGAMD_OK;
}

template <typename IdxT, int VW, int LPR, int CH, bool IDENT>
static int launch_sum_w(const pygamd_spmm_args* p, const Shape& s, float* partial,
                        hipStream_t st) {
  const bool has_w = p->w != nullptr;
  const bool has_scale = p->src_scale != nullptr;
  if (has_w && p->w_heads > 1)
    return launch_sum<IdxT, VW, LPR, CH, 2, IDENT>(p, s, partial, st);
  if (has_w || has_scale) return launch_sum<IdxT, VW, LPR, CH, 1, IDENT>(p, s, partial, st);
  if constexpr (!IDENT) {
    if (p->src_bits) return launch_sum<IdxT, VW, LPR, CH, 3, IDENT>(p, s, partial, st);
  }
  return launch_sum<IdxT, VW, LPR, CH, 0, IDENT>(p, s, partial, st);
}

template <typename IdxT, int VW, int LPR, int CH, bool IDENT>
static int launch_minmax(const pygamd_spmm_args* p, const Shape& s, hipStream_t st) {
  SpmmDev<IdxT> a = make_dev<IdxT>(p);
  const int64_t n_hub = p->n_hub > 0 ? p->n_hub : 0;
  const IdxT* hub_rows = static_cast<const IdxT*>(p->hub_rows);
  dim3 grid(wave_grid(p->n_rows + n_hub), s.tiles);
  if (!a.arg_out && !a.arg32_out) {  // nobody asked which slot attained the extremum
    if (p->reduce == PYGAMD_MAX) {
      hipLaunchKernelGGL((spmm_minmax_rows_plain<IdxT, VW, LPR, CH, true, IDENT>), grid,
                         dim3(kBlock), 0, st, a, hub_rows, n_hub);
    } else {
      hipLaunchKernelGGL((spmm_minmax_rows_plain<IdxT, VW, LPR, CH, false, IDENT>), grid,
                         dim3(kBlock), 0, st, a, hub_rows, n_hub);
    }
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  }
  if (p->reduce == PYGAMD_MAX) {
    hipLaunchKernelGGL((spmm_minmax_rows<IdxT, VW, LPR, CH, true, IDENT>), grid, dim3(kBlock), 0,
                       st, a, hub_rows, n_hub);
  } else {
    hipLaunchKernelGGL((spmm_minmax_rows<IdxT, VW, LPR, CH, false, IDENT>), grid, dim3(kBlock),
                       0, st, a, hub_rows, n_hub);
  }
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

template <typename IdxT, int VW, int LPR, int CH>
static int launch_shape(const pygamd_spmm_args* p, const Shape& s, float* partial,
                        hipStream_t st) {
  const bool ident = (p->col == nullptr);
  const bool mm = (p->reduce == PYGAMD_MIN || p->reduce == PYGAMD_MAX);
  if (mm) {
    return ident ? launch_minmax<IdxT, VW, LPR, CH, true>(p, s, st)
                 : launch_minmax<IdxT, VW, LPR, CH, false>(p, s, st);
  }
  return ident ? launch_sum_w<IdxT, VW, LPR, CH, true>(p, s, partial, st)
               : launch_sum_w<IdxT, VW, LPR, CH, false>(p, s, partial, st);
}

template <typename IdxT, int VW>
static int launch_vw(const pygamd_spmm_args* p, const Shape& s, float* partial, hipStream_t st) {
  switch (s.lpr) {
    case 4:
      return launch_shape<IdxT, VW, 4, 1>(p, s, partial, st);
    case 8:
      return launch_shape<IdxT, VW, 8, 1>(p, s, partial, st);
    case 16:
      return launch_shape<IdxT, VW, 16, 1>(p, s, partial, st);
    case 32:
      return launch_shape<IdxT, VW, 32, 1>(p, s, partial, st);
    default:
      return s.ch == 2 ? launch_shape<IdxT, VW, 64, 2>(p, s, partial, st)
                       : launch_shape<IdxT, VW, 64, 1>(p, s, partial, st);
  }
}

template <typename IdxT, int VW, int LPR, int CH>
static int launch_sddmm_shape(const Shape& s, const void* rowptr, const void* col,
                              const void* eid, const float* grad_out, int64_t ldg, const float* x,
                              int64_t ldx, int64_t n_rows, int64_t F, int w_heads, int head_dim,
                              int use_atomic, float* grad_w, const float* w, float* agg,
                              int64_t lda, hipStream_t st) {
  dim3 grid(wave_grid(n_rows), s.tiles);
  if (agg != nullptr) {
    hipLaunchKernelGGL((sddmm_rows<IdxT, VW, LPR, CH, true>), grid, dim3(kBlock), 0, st,
                       static_cast<const IdxT*>(rowptr), static_cast<const IdxT*>(col),
                       static_cast<const IdxT*>(eid), grad_out, ldg, x, ldx, n_rows, F, w_heads,
                       head_dim, use_atomic, grad_w, w, agg, lda);
  } else {
    hipLaunchKernelGGL((sddmm_rows<IdxT, VW, LPR, CH, false>), grid, dim3(kBlock), 0, st,
                       static_cast<const IdxT*>(rowptr), static_cast<const IdxT*>(col),
                       static_cast<const IdxT*>(eid), grad_out, ldg, x, ldx, n_rows, F, w_heads,
                       head_dim, use_atomic, grad_w, w, agg, lda);
  }
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

template <typename IdxT>
static int launch_sddmm(const Shape& s, const void* rowptr, const void* col, const void* eid,
                        const float* grad_out, int64_t ldg, const float* x, int64_t ldx,
                        int64_t n_rows, int64_t F, int w_heads, int head_dim, int use_atomic,
                        float* grad_w, const float* w, float* agg, int64_t lda, hipStream_t st) {
#define PYGAMD_SDDMM(VW, LPR, CH)                                                               \
  return launch_sddmm_shape<IdxT, VW, LPR, CH>(s, rowptr, col, eid, grad_out, ldg, x, ldx,     \
                                               n_rows, F, w_heads, head_dim, use_atomic, grad_w, \
                                               w, agg, lda, st)
  if (s.vw == 4) {
    switch (s.lpr) {
      case 4: PYGAMD_SDDMM(4, 4, 1);
      case 8: PYGAMD_SDDMM(4, 8, 1);
      case 16: PYGAMD_SDDMM(4, 16, 1);
      case 32: PYGAMD_SDDMM(4, 32, 1);
      default:
        if (s.ch == 2) PYGAMD_SDDMM(4, 64, 2);
        PYGAMD_SDDMM(4, 64, 1);
    }
  }
  switch (s.lpr) {
    case 4: PYGAMD_SDDMM(1, 4, 1);
    case 8: PYGAMD_SDDMM(1, 8, 1);
    case 16: PYGAMD_SDDMM(1, 16, 1);
    case 32: PYGAMD_SDDMM(1, 32, 1);
    default:
      if (s.ch == 2) PYGAMD_SDDMM(1, 64, 2);
      PYGAMD_SDDMM(1, 64, 1);
  }
#undef PYGAMD_SDDMM
}

template <typename IdxT>
static int launch_minmax_bwd_dst(const Shape& s, const void* rowptr, const void* col,
                                 const float* x, int64_t ldx, const float* out, int64_t ldo,
                                 const float* grad_out, int64_t ldgo, int64_t n_rows, int64_t F,
                                 int count_self, const int32_t* arg32, float* grad_x, int64_t ldg,
                                 hipStream_t st) {
  dim3 grid(wave_grid(n_rows), s.tiles);
#define PYGAMD_MMB(VW, LPR, CH)                                                                \
  hipLaunchKernelGGL((spmm_minmax_bwd_dst<IdxT, VW, LPR, CH>), grid, dim3(kBlock), 0, st,      \
                     static_cast<const IdxT*>(rowptr), static_cast<const IdxT*>(col), x, ldx,  \
                     out, ldo, grad_out, ldgo, n_rows, F, count_self, arg32, grad_x, ldg);     \
  break
  if (s.vw == 4) {
    switch (s.lpr) {
      case 4: PYGAMD_MMB(4, 4, 1);
      case 8: PYGAMD_MMB(4, 8, 1);
      case 16: PYGAMD_MMB(4, 16, 1);
      case 32: PYGAMD_MMB(4, 32, 1);
      default:
        if (s.ch == 2) {
          PYGAMD_MMB(4, 64, 2);
        }
        PYGAMD_MMB(4, 64, 1);
    }
  } else {
    switch (s.lpr) {
      case 4: PYGAMD_MMB(1, 4, 1);
      case 8: PYGAMD_MMB(1, 8, 1);
      case 16: PYGAMD_MMB(1, 16, 1);
      case 32: PYGAMD_MMB(1, 32, 1);
      default:
        if (s.ch == 2) {
          PYGAMD_MMB(1, 64, 2);
        }
        PYGAMD_MMB(1, 64, 1);
    }
  }
#undef PYGAMD_MMB
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

template <typename IdxT>
static int launch_multi(const pygamd_spmm_args* p, const Shape& s, bool has_perm, float* out_sum,
                        float* out_sq, float* out_min, float* out_max, hipStream_t st) {
  SpmmDev<IdxT> a = make_dev<IdxT>(p);
  a.hub_threshold = 0;
  dim3 grid(wave_grid(p->n_rows), s.tiles);
#define PYGAMD_MULTI(VW, LPR, CH)                                                                \
  do {                                                                                           \
    if (has_perm) {                                                                              \
      hipLaunchKernelGGL((spmm_multi_rows<IdxT, VW, LPR, CH, false>), grid, dim3(kBlock), 0, st, \
                         a, out_sum, out_sq, out_min, out_max);                                  \
    } else {                                                                                     \
      hipLaunchKernelGGL((spmm_multi_rows<IdxT, VW, LPR, CH, true>), grid, dim3(kBlock), 0, st,  \
                         a, out_sum, out_sq, out_min, out_max);                                  \
    }                                                                                            \
  } while (0)
  if (s.vw == 4) {
    switch (s.lpr) {
      case 4: PYGAMD_MULTI(4, 4, 1); break;
      case 8: PYGAMD_MULTI(4, 8, 1); break;
      case 16: PYGAMD_MULTI(4, 16, 1); break;
      case 32: PYGAMD_MULTI(4, 32, 1); break;
      default:
        if (s.ch == 2) {
          PYGAMD_MULTI(4, 64, 2);
        } else {
          PYGAMD_MULTI(4, 64, 1);
        }
    }
  } else {
    switch (s.lpr) {
      case 4: PYGAMD_MULTI(1, 4, 1); break;
      case 8: PYGAMD_MULTI(1, 8, 1); break;
      case 16: PYGAMD_MULTI(1, 16, 1); break;
      case 32: PYGAMD_MULTI(1, 32, 1); break;
      default:
        if (s.ch == 2) {
          PYGAMD_MULTI(1, 64, 2);
        } else {
          PYGAMD_MULTI(1, 64, 1);
        }
    }
  }
#undef PYGAMD_MULTI
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

static int validate(const pygamd_spmm_args* p) {
  if (!p) return PYGAMD_ERR_INVALID_ARG;
  if (p->n_rows < 0 || p->F < 0 || p->ldx < p->F || p->ldo < p->F) return PYGAMD_ERR_INVALID_ARG;
  if (p->idx_dtype != PYGAMD_IDX_I32 && p->idx_dtype != PYGAMD_IDX_I64)
    return PYGAMD_ERR_INVALID_ARG;
  if (p->reduce != PYGAMD_SUM && p->reduce != PYGAMD_MEAN && p->reduce != PYGAMD_MIN &&
      p->reduce != PYGAMD_MAX)
    return PYGAMD_ERR_UNSUPPORTED;
  if (p->n_rows > 0 && p->F > 0 && (!p->rowptr || !p->x || !p->out)) return PYGAMD_ERR_INVALID_ARG;
  const bool mm = (p->reduce == PYGAMD_MIN || p->reduce == PYGAMD_MAX);
  if (mm && (p->w || p->src_scale || p->accumulate || p->relu_mask || p->relu_bits))
    return PYGAMD_ERR_UNSUPPORTED;
  if (p->relu_mask && p->ld_mask < p->F) return PYGAMD_ERR_INVALID_ARG;
  if (p->relu_bits && (p->relu_mask || p->ld_bits < (p->F + 31) / 32)) return PYGAMD_ERR_INVALID_ARG;
  if (p->w && p->w_heads > 1) {
    if (p->head_dim < 1 || static_cast<int64_t>(p->head_dim) * p->w_heads != p->F)
      return PYGAMD_ERR_INVALID_ARG;
  }
  if (p->n_hub > 0 && (!p->hub_rows || !p->hub_chunk_ptr || p->hub_chunk < 1 ||
                       p->hub_threshold < 1 || p->n_chunks < p->n_hub))
    return PYGAMD_ERR_INVALID_ARG;
  if (p->hub_phase < 0 || p->hub_phase > 2) return PYGAMD_ERR_INVALID_ARG;
  if (p->src_bits && (mm || p->w || p->src_scale)) return PYGAMD_ERR_UNSUPPORTED;
  if (p->src_bits_set && !p->src_bits) return PYGAMD_ERR_INVALID_ARG;
  if (p->x_format != PYGAMD_X_DENSE && p->x_format != PYGAMD_X_COMPRESSED)
    return PYGAMD_ERR_INVALID_ARG;
  // fixed-stride slot blocks / a row limit on `accumulate`: the plain and weighted sum kernels of
  // rows without hubs (what a sampled batch needs)
  if (p->rowend && (mm || p->n_hub > 0 || p->src_bits || p->x_format != PYGAMD_X_DENSE))
    return PYGAMD_ERR_UNSUPPORTED;
  if (p->accumulate_rows < 0 || (p->accumulate_rows > 0 && (!p->accumulate || p->n_hub > 0)))
    return PYGAMD_ERR_INVALID_ARG;
  if (p->x_format == PYGAMD_X_COMPRESSED) {
    if (mm || p->w || p->src_scale || p->src_bits || !p->col) return PYGAMD_ERR_UNSUPPORTED;
    if (p->F > 256 || p->F % 4 != 0 || p->ldx < p->F + 12 || p->ldo % 4 != 0 ||
        (reinterpret_cast<uintptr_t>(p->out) & 15u) != 0)
      return PYGAMD_ERR_UNSUPPORTED;
  }
  return PYGAMD_OK;
}

// One wave per 32 rows (= one word of the row bitmap), 2^lshift lanes per row: copies of a row
// block with a 1/deg-style row factor and zero padding, and [row has a non-zero entry] per row.
// The loads of kPackRows row groups are issued before the first one is consumed (one row group at
// a time leaves a wave with a single 188-byte request in flight: 0.99 ms for the [2.45 M, 47] block
// of the products step, against 0.3 ms of traffic).
constexpr int kPackRows = 8;
__global__ void __launch_bounds__(kBlock)
    rows_pack_kernel(const float* __restrict__ g, int64_t ldg, int64_t n_rows, int F, int lshift,
                     const float* __restrict__ row_scale, float* __restrict__ scaled,
                     int64_t lds, int Fs, float* __restrict__ copy, int64_t ldc, int Fc,
                     uint32_t* __restrict__ bits) {
  const int lane = lane_id();
  const int64_t word = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  if (word * 32 >= n_rows) return;
  const int L = 1 << lshift;
  const int rpi = kWave >> lshift;  // rows per row group
  const int iters = 32 / rpi;
  const int sub = lane >> lshift;
  const int lir = lane & (L - 1);
  const int fmax = Fs > Fc ? (Fs > F ? Fs : F) : (Fc > F ? Fc : F);
  const uint64_t group = L == kWave ? ~0ull : ((1ull << L) - 1ull);
  uint32_t w = 0;
  for (int it0 = 0; it0 < iters; it0 += kPackRows) {
    int64_t row[kPackRows];
    bool live[kPackRows], nz[kPackRows];
    float sc[kPackRows];
#pragma unroll
    for (int u = 0; u < kPackRows; ++u) {
      row[u] = word * 32 + static_cast<int64_t>(it0 + u) * rpi + sub;
      live[u] = it0 + u < iters && row[u] < n_rows;
      row[u] = live[u] ? row[u] : n_rows - 1;  // (clamped: the load below is unconditional)
      nz[u] = false;
      sc[u] = row_scale ? row_scale[row[u]] : 1.f;
    }
    for (int f0 = 0; f0 < fmax; f0 += L) {
      const int f = f0 + lir;
      const int fc = f < F ? f : F - 1;
      float v[kPackRows];
#pragma unroll
      for (int u = 0; u < kPackRows; ++u) v[u] = F > 0 ? g[row[u] * ldg + fc] : 0.f;
#pragma unroll
      for (int u = 0; u < kPackRows; ++u) {
        const float x = (live[u] && f < F) ? v[u] : 0.f;
        nz[u] |= (__float_as_uint(x) << 1) != 0u;  // anything but +-0 (NaN, subnormals count)
        if (live[u]) {
          if (scaled && f < Fs) scaled[row[u] * lds + f] = x * sc[u];
          if (copy && f < Fc) copy[row[u] * ldc + f] = x;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kPackRows; ++u) {
      const uint64_t m = __ballot(nz[u]);
      for (int r = 0; r < rpi; ++r) {
        if ((m >> (r << lshift)) & group) w |= 1u << ((it0 + u) * rpi + r);
      }
    }
  }
  if (lane == 0) bits[word] = w;
}

// *n_set = number of set bits, one workgroup (the bitmap of 2.45 M rows is 300 KiB; one atomic per
// word from the kernel above — 70 k of them on one address — cost 0.6 ms)
__global__ void __launch_bounds__(1024)
    bits_count_kernel(const uint32_t* __restrict__ bits, int64_t words, int64_t* __restrict__ n_set) {
  __shared__ unsigned long long part[16];
  unsigned long long c = 0;
  // (eight loads in flight: the plain loop waits for each word before it asks for the next —
  // 75 serial trips to L2 for the 300 KiB bitmap of the products shape, 31 us)
  int64_t i = threadIdx.x;
  for (; i + 7 * 1024 < words; i += 8 * 1024) {
    uint32_t w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = bits[i + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u) c += __popc(w[u]);
  }
  for (; i < words; i += 1024) c += __popc(bits[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, kWave);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int i = 0; i < 16; ++i) t += part[i];
    *n_set = static_cast<int64_t>(t);
  }
}

// One wave per row: x[row, :F] (F <= 256) -> the compressed row [8 mask words | kept values] of
// spmm_device.h ("compressed rows").  Kept = bit pattern other than +0.0.
__global__ void __launch_bounds__(kBlock)
    rows_compress_kernel(const float* __restrict__ x, int64_t ldx, int64_t n_rows, int F,
                         uint32_t* __restrict__ out, int64_t ldo) {
  const int lane = lane_id();
  const int64_t row = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  if (row >= n_rows) return;
  uint32_t v[4];
  uint32_t nib = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 * lane + i;
    v[i] = c < F ? __float_as_uint(x[row * ldx + c]) : 0u;
    nib |= (v[i] != 0u ? 1u : 0u) << i;
  }
  uint32_t word = nib << (4 * (lane & 7));
  word |= __shfl_xor(word, 1, kWave);
  word |= __shfl_xor(word, 2, kWave);
  word |= __shfl_xor(word, 4, kWave);
  uint32_t nib2;
  int off = zrow_lane_offset(word, lane, nib2);
  uint32_t* __restrict__ orow = out + row * ldo;
  if ((lane & 7) == 0) orow[lane >> 3] = word;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (nib & (1u << i)) orow[kZrowHdr + off++] = v[i];
  }
}

}  // namespace pygamd

using namespace pygamd;

extern "C" {

int pygamd_spmm_csr_workspace_bytes(const pygamd_spmm_args* args, size_t* bytes) {
  if (!args || !bytes) return PYGAMD_ERR_INVALID_ARG;
  const bool mm = (args->reduce == PYGAMD_MIN || args->reduce == PYGAMD_MAX);
  *bytes = (args->n_hub > 0 && !mm)
               ? static_cast<size_t>(args->n_chunks) * static_cast<size_t>(args->F) * sizeof(float)
               : 0;
  return PYGAMD_OK;
}

int pygamd_spmm_csr(const pygamd_spmm_args* args, void* workspace, size_t workspace_bytes,
                    void* stream) {
  int rc = validate(args);
  if (rc != PYGAMD_OK) return rc;
  if (args->n_rows == 0 || args->F == 0) return PYGAMD_OK;
  size_t need = 0;
  pygamd_spmm_csr_workspace_bytes(args, &need);
  if (need > 0 && (!workspace || workspace_bytes < need)) return PYGAMD_ERR_WORKSPACE;
  hipStream_t st = as_stream(stream);
  float* partial = static_cast<float*>(workspace);
  if (args->x_format == PYGAMD_X_COMPRESSED) {  // one wave per row, 64 lanes x 4 columns
    Shape cs;
    cs.vw = 4; cs.lpr = 64; cs.ch = 1; cs.tiles = 1;
    return PYGAMD_DISPATCH_IDX(args->idx_dtype, [&]() -> int {
      return launch_sum<IdxT, 4, 64, 1, 4, false>(args, cs, partial, st);
    });
  }
  const Shape s = pick_shape(args);
  return PYGAMD_DISPATCH_IDX(args->idx_dtype, [&]() -> int {
    return s.vw == 4 ? launch_vw<IdxT, 4>(args, s, partial, st)
                     : launch_vw<IdxT, 1>(args, s, partial, st);
  });
}

int pygamd_rows_compress(const float* x, int64_t ldx, int64_t n_rows, int64_t F, uint32_t* out,
                         int64_t ld_out, void* stream) {
  if (n_rows < 0 || F < 0 || F > 256 || ldx < F || ld_out < F + 12) return PYGAMD_ERR_INVALID_ARG;
  if (n_rows == 0) return PYGAMD_OK;
  if (!out || (F > 0 && !x)) return PYGAMD_ERR_INVALID_ARG;
  dim3 grid(static_cast<unsigned>(ceil_div(n_rows, static_cast<int64_t>(kWavesPerBlock))));
  hipLaunchKernelGGL(rows_compress_kernel, grid, dim3(kBlock), 0, as_stream(stream), x, ldx,
                     n_rows, static_cast<int>(F), out, ld_out);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

int pygamd_rows_pack(const float* g, int64_t ldg, int64_t n_rows, int64_t F,
                     const float* row_scale, float* scaled, int64_t ld_scaled, int64_t F_scaled,
                     float* copy, int64_t ld_copy, int64_t F_copy, uint32_t* row_bits,
                     int64_t* n_set, void* stream) {
  if (n_rows < 0 || F < 0 || F > (1 << 20) || ldg < F || (n_rows > 0 && !row_bits))
    return PYGAMD_ERR_INVALID_ARG;
  if (scaled && (F_scaled < F || ld_scaled < F_scaled)) return PYGAMD_ERR_INVALID_ARG;
  if (copy && (F_copy < F || ld_copy < F_copy)) return PYGAMD_ERR_INVALID_ARG;
  if (n_rows > 0 && F > 0 && !g) return PYGAMD_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  if (n_rows == 0) {
    if (n_set) PYGAMD_HIP_CHECK(hipMemsetAsync(n_set, 0, sizeof(int64_t), st));
    return PYGAMD_OK;
  }
  int64_t fmax = F;
  if (scaled && F_scaled > fmax) fmax = F_scaled;
  if (copy && F_copy > fmax) fmax = F_copy;
  int lshift = 2;
  while (lshift < 6 && (1 << lshift) < fmax) ++lshift;
  const int64_t words = ceil_div(n_rows, static_cast<int64_t>(32));
  dim3 grid(static_cast<unsigned>(ceil_div(words, static_cast<int64_t>(kWavesPerBlock))));
  hipLaunchKernelGGL(rows_pack_kernel, grid, dim3(kBlock), 0, st, g, ldg, n_rows,
                     static_cast<int>(F), lshift, row_scale, scaled, ld_scaled,
                     static_cast<int>(scaled ? F_scaled : 0), copy, ld_copy,
                     static_cast<int>(copy ? F_copy : 0), row_bits);
  PYGAMD_LAUNCH_CHECK();
  if (n_set) {
    hipLaunchKernelGGL(bits_count_kernel, dim3(1), dim3(1024), 0, st, row_bits, words, n_set);
    PYGAMD_LAUNCH_CHECK();
  }
  return PYGAMD_OK;
}

int pygamd_spmm_csr_tie_count(const void* rowptr, const void* col, int idx_dtype, const float* x,
                              int64_t ldx, const float* out, int64_t ldo, int64_t n_rows,
                              int64_t F, int count_self, float* ntie_out, void* stream) {
  if (n_rows < 0 || F < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n_rows == 0 || F == 0) return PYGAMD_OK;
  if (!rowptr || !x || !out || !ntie_out) return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    hipLaunchKernelGGL((spmm_tie_count_rows<IdxT>), dim3(wave_grid(n_rows)), dim3(kBlock), 0,
                       as_stream(stream), static_cast<const IdxT*>(rowptr),
                       static_cast<const IdxT*>(col), x, ldx, out, ldo, n_rows, F, count_self,
                       ntie_out);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_spmm_csr_minmax_backward(const void* rowptr_t, const void* col_t, int idx_dtype,
                                    const float* x, int64_t ldx, const float* out,
                                    const float* grad_out, const float* ntie, int64_t ldo,
                                    int64_t n_src, int64_t F, float* grad_x, int64_t ldg,
                                    void* stream) {
  if (n_src < 0 || F < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n_src == 0 || F == 0) return PYGAMD_OK;
  if (!rowptr_t || !col_t || !x || !out || !grad_out || !ntie || !grad_x)
    return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    hipLaunchKernelGGL((spmm_minmax_bwd_rows<IdxT>), dim3(wave_grid(n_src)), dim3(kBlock), 0,
                       as_stream(stream), static_cast<const IdxT*>(rowptr_t),
                       static_cast<const IdxT*>(col_t), x, ldx, out, grad_out, ntie, ldo, n_src,
                       F, grad_x, ldg);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

static int minmax_backward_common(const void* rowptr, const void* col, int idx_dtype,
                                  const int32_t* arg32, const float* x, int64_t ldx,
                                  const float* out, int64_t ldo, const float* grad_out,
                                  int64_t ldgo, int64_t n_rows, int64_t n_src, int64_t F,
                                  int count_self, float* grad_x, int64_t ldg, void* stream) {
  if (n_rows < 0 || n_src < 0 || F < 0 || ldx < F || ldo < F || ldgo < F || ldg < F)
    return PYGAMD_ERR_INVALID_ARG;
  if (n_src == 0 || F == 0) return PYGAMD_OK;
  if (!grad_x) return PYGAMD_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  PYGAMD_HIP_CHECK(hipMemset2DAsync(grad_x, sizeof(float) * ldg, 0, sizeof(float) * F, n_src, st));
  if (n_rows == 0) return PYGAMD_OK;
  if (!rowptr || !x || !out || !grad_out) return PYGAMD_ERR_INVALID_ARG;
  if (arg32) {  // unique extrema: one atomic per output, no edge pass
    const bool v4 = (F % 4 == 0) && (ldgo % 4 == 0) && aligned16(grad_out) && aligned16(arg32);
    const int rc = PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
      const IdxT* rp = static_cast<const IdxT*>(rowptr);
      const IdxT* cp = static_cast<const IdxT*>(col);
      if (v4) {
        const int64_t units = F / 4;
        int64_t blocks = ceil_div(n_rows * units, kBlock);
        blocks = blocks > 256 * 64 ? 256 * 64 : blocks;
        hipLaunchKernelGGL((spmm_minmax_bwd_arg_kernel<IdxT, 4>),
                           dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, st, rp, cp, arg32,
                           grad_out, ldgo, n_rows, units, F, grad_x, ldg);
      } else {
        int64_t blocks = ceil_div(n_rows * F, kBlock);
        blocks = blocks > 256 * 64 ? 256 * 64 : blocks;
        hipLaunchKernelGGL((spmm_minmax_bwd_arg_kernel<IdxT, 1>),
                           dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, st, rp, cp, arg32,
                           grad_out, ldgo, n_rows, F, F, grad_x, ldg);
      }
      PYGAMD_LAUNCH_CHECK();
      return PYGAMD_OK;
    });
    if (rc != PYGAMD_OK) return rc;
  }
  pygamd_spmm_args probe = {};
  probe.F = F;
  probe.ldx = (ldx % 4 == 0 && ldgo % 4 == 0 && ldg % 4 == 0 && aligned16(grad_out)) ? ldx : 1;
  probe.ldo = ldo;
  probe.x = x;
  probe.out = const_cast<float*>(out);
  probe.w_heads = 1;
  probe.head_dim = static_cast<int>(F);
  const Shape s = pick_shape(&probe);
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    return launch_minmax_bwd_dst<IdxT>(s, rowptr, col, x, ldx, out, ldo, grad_out, ldgo, n_rows,
                                       F, count_self, arg32, grad_x, ldg, st);
  });
}

int pygamd_spmm_csr_minmax_backward_dst(const void* rowptr, const void* col, int idx_dtype,
                                        const float* x, int64_t ldx, const float* out,
                                        int64_t ldo, const float* grad_out, int64_t ldgo,
                                        int64_t n_rows, int64_t n_src, int64_t F, int count_self,
                                        float* grad_x, int64_t ldg, void* stream) {
  return minmax_backward_common(rowptr, col, idx_dtype, nullptr, x, ldx, out, ldo, grad_out, ldgo,
                                n_rows, n_src, F, count_self, grad_x, ldg, stream);
}

int pygamd_spmm_csr_minmax_backward_arg(const void* rowptr, const void* col, int idx_dtype,
                                        const int32_t* arg32, const float* x, int64_t ldx,
                                        const float* out, int64_t ldo, const float* grad_out,
                                        int64_t ldgo, int64_t n_rows, int64_t n_src, int64_t F,
                                        int count_self, float* grad_x, int64_t ldg,
                                        void* stream) {
  if (!arg32 && n_rows > 0 && F > 0) return PYGAMD_ERR_INVALID_ARG;
  return minmax_backward_common(rowptr, col, idx_dtype, arg32, x, ldx, out, ldo, grad_out, ldgo,
                                n_rows, n_src, F, count_self, grad_x, ldg, stream);
}

static size_t minmax_src_entries_bytes(int64_t n_rows, int64_t F) {
  return (static_cast<size_t>(n_rows) * static_cast<size_t>(F) * sizeof(MinmaxEntry) + 255) &
         ~static_cast<size_t>(255);
}

size_t pygamd_minmax_backward_src_workspace_bytes(int64_t n_rows, int64_t nnz, int64_t F) {
  if (n_rows <= 0 || nnz <= 0 || F <= 0) return 0;
  return minmax_src_entries_bytes(n_rows, F) + static_cast<size_t>(nnz) * sizeof(uint32_t);
}

int pygamd_spmm_csr_minmax_backward_src(const void* rowptr, const void* col, const void* rowptr_t,
                                        const void* col_t, const void* slot_map, int idx_dtype,
                                        const int32_t* arg32, const float* x, int64_t ldx,
                                        const float* out, int64_t ldo, const float* grad_out,
                                        int64_t ldgo, int64_t n_rows, int64_t n_src, int64_t nnz,
                                        int64_t F, int count_self, void* workspace,
                                        size_t workspace_bytes, float* grad_x, int64_t ldg,
                                        void* stream) {
  if (n_rows < 0 || n_src < 0 || nnz < 0 || F < 0 || ldx < F || ldo < F || ldgo < F || ldg < F)
    return PYGAMD_ERR_INVALID_ARG;
  if (n_src == 0 || F == 0) return PYGAMD_OK;
  if (!grad_x || !rowptr_t) return PYGAMD_ERR_INVALID_ARG;
  if (n_rows > 0 && (!rowptr || !x || !out || !grad_out || !arg32)) return PYGAMD_ERR_INVALID_ARG;
  if (nnz > 0 && (!col_t || !slot_map)) return PYGAMD_ERR_INVALID_ARG;
  // 16-byte accesses on arg32 / grad_out / grad_x rows; seg packs (offset, count) in 16 + 16 bits;
  // one LDS row per wave
  if ((F % 4) || (ldgo % 4) || (ldg % 4) || !aligned16(grad_out) || !aligned16(grad_x) ||
      !aligned16(arg32) || F > 8192)
    return PYGAMD_ERR_UNSUPPORTED;
  if (workspace_bytes < pygamd_minmax_backward_src_workspace_bytes(n_rows, nnz, F) ||
      (nnz > 0 && n_rows > 0 && !workspace))
    return PYGAMD_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  MinmaxEntry* entries = static_cast<MinmaxEntry*>(workspace);
  uint32_t* seg = reinterpret_cast<uint32_t*>(static_cast<char*>(workspace) +
                                              minmax_src_entries_bytes(n_rows, F));
  int rc = PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    if (n_rows > 0 && nnz > 0) {
      hipLaunchKernelGGL((minmax_pack_kernel<IdxT>), dim3(wave_grid(n_rows)), dim3(kBlock), 0, st,
                         static_cast<const IdxT*>(rowptr), arg32, grad_out, ldgo, n_rows, F,
                         entries, seg);
      PYGAMD_LAUNCH_CHECK();
    }
    const size_t lds = sizeof(float) * kWavesPerBlock * static_cast<size_t>(F);
    auto k = minmax_bwd_src_kernel<IdxT>;
    if (lds > 48 * 1024)
      PYGAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds)));
    hipLaunchKernelGGL(k, dim3(wave_grid(n_src)), dim3(kBlock), lds, st,
                       static_cast<const IdxT*>(rowptr_t), static_cast<const IdxT*>(col_t),
                       static_cast<const IdxT*>(slot_map), entries, seg, n_src, F, grad_x, ldg);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
  if (rc != PYGAMD_OK || n_rows == 0) return rc;
  // outputs marked -2: the two-pass tie kernel adds their shares (fp32 atomics, rows without a
  // mark leave after one read of their arg32 row)
  pygamd_spmm_args probe = {};
  probe.F = F;
  probe.ldx = (ldx % 4 == 0) ? ldx : 1;
  probe.ldo = ldo;
  probe.x = x;
  probe.out = const_cast<float*>(out);
  probe.w_heads = 1;
  probe.head_dim = static_cast<int>(F);
  const Shape s = pick_shape(&probe);
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    return launch_minmax_bwd_dst<IdxT>(s, rowptr, col, x, ldx, out, ldo, grad_out, ldgo, n_rows,
                                       F, count_self, arg32, grad_x, ldg, st);
  });
}

int pygamd_multi_reduce_csr(const void* rowptr, const void* perm, int idx_dtype, const float* x,
                            int64_t ldx, int64_t n_rows, int64_t F, float* out_sum,
                            float* out_sq, float* out_min, float* out_max, int64_t ldo,
                            void* stream) {
  if (n_rows < 0 || F < 0 || ldx < F || ldo < F) return PYGAMD_ERR_INVALID_ARG;
  if (n_rows == 0 || F == 0) return PYGAMD_OK;
  // (x may be NULL: a zero-row source with only empty groups is legal and never dereferenced)
  if (!rowptr || !(out_sum || out_sq || out_min || out_max)) return PYGAMD_ERR_INVALID_ARG;
  pygamd_spmm_args args = {};
  args.rowptr = rowptr;
  args.col = perm;
  args.x = x;
  // every requested output must allow the vector width the probe picks
  float* outs[4] = {out_sum, out_sq, out_min, out_max};
  args.out = nullptr;
  bool all_aligned = true;
  for (float* o : outs) {
    if (o) {
      if (!args.out) args.out = o;
      all_aligned = all_aligned && aligned16(o);
    }
  }
  args.n_rows = n_rows;
  args.F = F;
  args.ldx = all_aligned ? ldx : 1;  // ldx % 4 != 0 forces the scalar shape in pick_shape
  args.ldo = ldo;
  args.idx_dtype = idx_dtype;
  args.w_heads = 1;
  args.head_dim = static_cast<int>(F);
  const Shape s = pick_shape(&args);
  args.ldx = ldx;
  hipStream_t st = as_stream(stream);
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    return launch_multi<IdxT>(&args, s, perm != nullptr, out_sum, out_sq, out_min, out_max, st);
  });
}

int pygamd_sddmm_csr(const void* rowptr, const void* col, const void* eid, int idx_dtype,
                     const float* grad_out, int64_t ldg, const float* x, int64_t ldx,
                     int64_t n_rows, int64_t F, int32_t w_heads, int32_t head_dim, float* grad_w,
                     void* stream) {
  if (n_rows < 0 || F < 0 || w_heads < 1 || ldg < F || ldx < F) return PYGAMD_ERR_INVALID_ARG;
  if (n_rows == 0 || F == 0) return PYGAMD_OK;
  if (!rowptr || !grad_out || !x || !grad_w) return PYGAMD_ERR_INVALID_ARG;
  const int hd = (w_heads > 1) ? head_dim : static_cast<int>(F);
  if (static_cast<int64_t>(hd) * w_heads != F) return PYGAMD_ERR_INVALID_ARG;
  // shape selection mirrors the SpMM (vector width must divide the head width)
  pygamd_spmm_args probe = {};
  probe.F = F;
  probe.ldx = ldx;
  probe.ldo = ldg;
  probe.x = x;
  probe.out = const_cast<float*>(grad_out);
  probe.w_heads = w_heads;
  probe.head_dim = hd;
  const Shape s = pick_shape(&probe);
  const int use_atomic = (s.ch > 1 || s.tiles > 1) ? 1 : 0;  // a head may span two lane groups
  hipStream_t st = as_stream(stream);
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    return launch_sddmm<IdxT>(s, rowptr, col, eid, grad_out, ldg, x, ldx, n_rows, F, w_heads, hd,
                              use_atomic, grad_w, nullptr, nullptr, 0, st);
  });
}

int pygamd_sddmm_spmm_csr(const void* rowptr, const void* col, const void* eid, int idx_dtype,
                          const float* rows, int64_t ldr, const float* x, int64_t ldx,
                          int64_t n_rows, int64_t F, int32_t w_heads, int32_t head_dim,
                          const float* w, float* grad_w, float* agg, int64_t lda, void* stream) {
  if (n_rows < 0 || F < 0 || w_heads < 1 || ldr < F || ldx < F || lda < F)
    return PYGAMD_ERR_INVALID_ARG;
  if (n_rows == 0 || F == 0) return PYGAMD_OK;
  if (!rowptr || !rows || !x || !grad_w || !w || !agg) return PYGAMD_ERR_INVALID_ARG;
  const int hd = (w_heads > 1) ? head_dim : static_cast<int>(F);
  if (static_cast<int64_t>(hd) * w_heads != F) return PYGAMD_ERR_INVALID_ARG;
  pygamd_spmm_args probe = {};
  probe.F = F;
  probe.ldx = ldx;
  // 16-byte lanes only if the held rows AND the output take them (the probe has one `out`)
  const bool both16 = ldr % 4 == 0 && lda % 4 == 0 &&
                      ((reinterpret_cast<uintptr_t>(rows) | reinterpret_cast<uintptr_t>(agg)) &
                       15u) == 0;
  probe.ldo = both16 ? 4 : 1;
  probe.x = x;
  probe.out = agg;
  probe.w_heads = w_heads;
  probe.head_dim = hd;
  const Shape s = pick_shape(&probe);
  const int use_atomic = (s.ch > 1 || s.tiles > 1) ? 1 : 0;
  hipStream_t st = as_stream(stream);
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    return launch_sddmm<IdxT>(s, rowptr, col, eid, rows, ldr, x, ldx, n_rows, F, w_heads, hd,
                              use_atomic, grad_w, w, agg, lda, st);
  });
}

}  // extern "C"
