// spmm_device.h — device-side building blocks of the CSR aggregation shared by spmm.hip (the
// stand-alone SpMM family) and sage_fused.hip (aggregation fused with the feature transform):
// the argument block, the per-lane feature slots, and the slot-staged, U-loads-in-flight row
// accumulation loop.  See spmm.hip for the mapping.
#pragma once
#include "common.h"

namespace pygamd {

template <typename IdxT>
struct SpmmDev {
  const IdxT* __restrict__ rowptr;
  const IdxT* __restrict__ col;
  const IdxT* __restrict__ eid;
  const float* __restrict__ w;
  const float* __restrict__ src_scale;
  const float* __restrict__ x;
  float* __restrict__ out;
  IdxT* __restrict__ arg_out;
  int32_t* __restrict__ arg32_out;  // MIN/MAX: saved for the backward (see pyg_amd.h)
  const float* __restrict__ relu_mask;  // SUM/MEAN: out = relu_mask <= 0 ? 0 : out (or null)
  int64_t ldm;
  const uint32_t* __restrict__ relu_bits;  // the same mask, one bit per element (or null)
  int64_t ldb;
  // null, or one bit per SOURCE row (bit j & 31 of src_bits[j >> 5]): rows whose bit is clear are
  // all-zero and are not read (plain sums only).  src_bits_set (or null): how many bits are set,
  // on the device — with more than half of the n_src rows live the lookup costs more than it
  // saves and the kernel ignores the bits.
  const uint32_t* __restrict__ src_bits;
  const int64_t* __restrict__ src_bits_set;
  int64_t n_src;
  int64_t n_rows, F, ldx, ldo;
  int w_heads, head_dim;
  int mean;
  int accumulate;  // out[i] += result instead of out[i] = result
  int64_t hub_threshold;
  // null, or [n_rows]: the slots of row r are [rowptr[r], rowend[r]) instead of [rowptr[r],
  // rowptr[r + 1]) — rows own fixed-stride slot blocks that are only partly filled (sampled batches
  // at their static fan-out capacity, csrc/minibatch.hip)
  const IdxT* __restrict__ rowend;
  int64_t accumulate_rows;  // with accumulate: rows >= this start from 0 (0 = every row has an old value)
};

template <typename IdxT>
__device__ __forceinline__ IdxT spmm_row_end(const SpmmDev<IdxT>& a, int64_t row) {
  return a.rowend ? a.rowend[row] : a.rowptr[row + 1];
}

// Independent row loads issued per lane before the first add.  A staged index chunk holds 64
// slots, so EPI * U never needs to exceed 64 (U <= LPR).
template <int LPR, int CH>
constexpr int spmm_unroll() {
  return CH == 1 ? (LPR < 8 ? LPR : 8) : 4;
}

// WMODE: 0 = plain sum; 1 = one staged multiplier per slot (w with one head and/or src_scale);
//        2 = per-head weights fetched per slot (+ optional staged src_scale);
//        3 = plain sum over the source rows whose bit in a.src_bits is set (the others are known
//            to be all-zero and are not read);
//        4 = plain sum over COMPRESSED source rows (a.x = the compressed block, a.ldx its pitch in words;
//            see "compressed rows" below).
template <typename IdxT, int VW, int LPR, int CH, int WMODE, bool IDENT, bool FULL, int UF = 0>
__device__ __forceinline__ void spmm_batch(const SpmmDev<IdxT>& a, int j, int cnt, int sub,
                                           IdxT myc, IdxT mye, float mym, const int (&fo)[CH],
                                           const bool (&fv)[CH], const int (&head)[CH],
                                           float (&acc)[CH][VW]) {
  constexpr int EPI = kWave / LPR;
  constexpr int U = UF > 0 ? UF : spmm_unroll<LPR, CH>();
  Vec<VW> v[U][CH];
  float m[U];
  float wv[U][CH];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int k = j + u * EPI + sub;
    const bool valid = FULL || (k < cnt);
    const int kk = FULL ? k : (k < cnt ? k : cnt - 1);
    IdxT c;
    if constexpr (EPI == 1) {
      c = bcast_uniform(myc, kk);
    } else {
      c = bcast_lane(myc, kk);
    }
    if constexpr (WMODE == 1 || WMODE == 2) {
      float mm;
      if constexpr (EPI == 1) {
        mm = bcast_uniform(mym, kk);
      } else {
        mm = bcast_lane(mym, kk);
      }
      m[u] = valid ? mm : 0.f;
    } else {
      m[u] = 1.f;
    }
    const float* __restrict__ xr = a.x + static_cast<int64_t>(c) * a.ldx;
    if constexpr (WMODE == 2) {
      IdxT e;
      if constexpr (EPI == 1) {
        e = bcast_uniform(mye, kk);
      } else {
        e = bcast_lane(mye, kk);
      }
      const float* __restrict__ wr = a.w + static_cast<int64_t>(e) * a.w_heads;
#pragma unroll
      for (int c2 = 0; c2 < CH; ++c2) wv[u][c2] = fv[c2] ? wr[head[c2]] : 0.f;
    }
#pragma unroll
    for (int c2 = 0; c2 < CH; ++c2) {
      if (fv[c2] && valid) {
#ifdef PYGAMD_GATHER_NT  // lab: gathered rows with the non-temporal hint (scripts/README.md)
        v[u][c2] = load_vec_streamed<VW>(xr + fo[c2]);
#else
        v[u][c2] = load_vec<VW>(xr + fo[c2]);
#endif
      } else {
#pragma unroll
        for (int i = 0; i < VW; ++i) v[u][c2].v[i] = 0.f;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
#pragma unroll
    for (int c2 = 0; c2 < CH; ++c2) {
#pragma unroll
      for (int i = 0; i < VW; ++i) {
        if constexpr (WMODE == 0 || WMODE == 3) {
          acc[c2][i] += v[u][c2].v[i];
        } else if constexpr (WMODE == 1) {
          acc[c2][i] = fmaf(v[u][c2].v[i], m[u], acc[c2][i]);
        } else {
          acc[c2][i] = fmaf(v[u][c2].v[i], m[u] * wv[u][c2], acc[c2][i]);
        }
      }
    }
  }
}

// ---- compressed rows ------------------------------------------------------------------------------------
// A [n, F <= 256] activation block with many exact zeros (a ReLU output: half of it) stored row by
// row as  [8 mask words | the kept values in column order]  at a fixed row pitch (ld words, a
// multiple of 32 = 128 bytes, >= F + 12): bit (c & 31) of word (c >> 5) is set where element c is
// kept (its bit pattern is not +0.0).  A gather then touches 32 + 4 nnz bytes of a row instead of
// 4 F — 4.4 instead of 8 lines of 128 bytes for a half-empty 256-float row — and the cost of a
// row gather on this chip is proportional to the lines it touches (scripts/gather_lines_probe.py:
// 10.1 ms for 8 lines per row, 6.0 ms for 5, 4.6 ms for 4 at the products shape).  Lossless: the
// sums below add the same values in the same order as the dense gather (a dropped +0.0 changes no
// sum).  Lane l of the wave that reads a row owns columns 4 l .. 4 l + 3: its mask nibble is bits
// 4 (l & 7) .. + 3 of word l >> 3, its values start at (kept in the words below) + (kept in the
// lower nibbles of its word).
constexpr int kZrowHdr = 8;
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// exclusive prefix over the eight 8-lane groups of a wave of a value `t` that is equal inside each
// group (three DPP adds: the other half of the 16-lane row, then row 0 -> 1 / 2 -> 3, then the
// lower half -> rows 2, 3)
__device__ __forceinline__ int group8_exclusive_prefix(int t) {
  const int a1 = t + __builtin_amdgcn_update_dpp(0, t, 0x118 /* row_shr:8 */, 0xF, 0xF, false);
  const int a2 = a1 + __builtin_amdgcn_update_dpp(0, a1, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);
  const int a3 = a2 + __builtin_amdgcn_update_dpp(0, a2, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
  return a3 - t;
}

// (offset of the lane's first kept value, its mask nibble) from the lane's mask word
__device__ __forceinline__ int zrow_lane_offset(uint32_t hw, int lane, uint32_t& nib) {
  const int sh = (lane & 7) * 4;
  nib = (hw >> sh) & 15u;
  const int below = __popc(hw & ((1u << sh) - 1u));
  return group8_exclusive_prefix(__popc(hw)) + below;
}

// the lane's four columns from its (up to four) kept values p[0..]
__device__ __forceinline__ void zrow_expand(uint32_t nib, const f32x4u& p, float (&o)[4]) {
  const bool b0 = nib & 1u, b1 = nib & 2u, b2 = nib & 4u, b3 = nib & 8u;
  const int r2 = (b0 ? 1 : 0) + (b1 ? 1 : 0);
  const int r3 = r2 + (b2 ? 1 : 0);
  o[0] = b0 ? p[0] : 0.f;
  o[1] = b1 ? (b0 ? p[1] : p[0]) : 0.f;
  o[2] = b2 ? (r2 == 0 ? p[0] : (r2 == 1 ? p[1] : p[2])) : 0.f;
  o[3] = b3 ? (r3 == 0 ? p[0] : (r3 == 1 ? p[1] : (r3 == 2 ? p[2] : p[3]))) : 0.f;
}

// U source rows of a compressed block into acc (one wave per destination row, 64 lanes x 4 columns):
// the U mask words are in flight together, then the U value loads.
template <typename IdxT, bool FULL, int U>
__device__ __forceinline__ void spmm_batch_zrows(const SpmmDev<IdxT>& a, int j, int cnt, IdxT myc,
                                                  int lane, float (&acc)[1][4]) {
  const uint32_t* __restrict__ rows = reinterpret_cast<const uint32_t*>(a.x);
  const uint32_t* base[U];
  uint32_t hw[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int k = j + u;
    const int kk = FULL ? k : (k < cnt ? k : cnt - 1);
    const IdxT c = bcast_uniform(myc, kk);
    base[u] = rows + static_cast<int64_t>(c) * a.ldx;
    hw[u] = base[u][lane >> 3];
  }
  f32x4u v[U];
  uint32_t nib[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int off = zrow_lane_offset(hw[u], lane, nib[u]);
    if (!FULL && j + u >= cnt) nib[u] = 0u;
    v[u] = *reinterpret_cast<const f32x4u*>(base[u] + kZrowHdr + off);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float o[4];
    zrow_expand(nib[u], v[u], o);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[0][i] += o[i];
  }
}

// UF > 0 overrides the number of row loads in flight per lane (a kernel with few resident waves
// needs more memory-level parallelism per wave)
template <typename IdxT, int VW, int LPR, int CH, int WMODE, bool IDENT, int UF = 0>
__device__ __forceinline__ void spmm_accumulate(const SpmmDev<IdxT>& a, IdxT start, IdxT end,
                                                int lane, const int (&fo)[CH],
                                                const bool (&fv)[CH], const int (&head)[CH],
                                                float (&acc)[CH][VW]) {
  constexpr int EPI = kWave / LPR;
  constexpr int U = UF > 0 ? UF : spmm_unroll<LPR, CH>();
  constexpr int STEP = EPI * U;
  const int sub = lane / LPR;
  bool sparse_src = false;
  if constexpr (WMODE == 3) {
    sparse_src = a.src_bits_set == nullptr || 2 * *a.src_bits_set < a.n_src;
  }
  for (IdxT base = start; base < end; base += kWave) {
    const IdxT rem = end - base;
    int cnt = rem < kWave ? static_cast<int>(rem) : kWave;
    IdxT myc = 0, mye = 0;
    float mym = 1.f;
    if (lane < cnt) {
      const IdxT k = base + lane;
      if constexpr (IDENT) {
        myc = k;
      } else {
        myc = __builtin_nontemporal_load(&a.col[k]);  // streamed once: keep L2 for feature rows
      }
      if constexpr (WMODE == 1 || WMODE == 2) {
        mye = a.eid ? a.eid[k] : k;
        if (a.src_scale) mym = a.src_scale[myc];
        if constexpr (WMODE == 1) {
          if (a.w) mym *= a.w[mye];
        }
      }
    }
    if constexpr (WMODE == 3) {
      if (sparse_src) {
        // keep the slots whose source row is live, packed into the low lanes in slot order (an
        // all-zero row adds +0.0 to every sum: dropping it changes no result)
        const bool keep =
            lane < cnt && ((a.src_bits[static_cast<int64_t>(myc) >> 5] >> (myc & 31)) & 1u) != 0;
        const uint64_t m = __ballot(keep);
        const int before = __builtin_amdgcn_mbcnt_hi(
            static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
        cnt = __popcll(m);
        myc = push_lane(myc, keep ? before : cnt + lane - before);
      }
    }
    int j = 0;
    if constexpr (WMODE == 4) {
      static_assert(LPR == kWave && VW == 4 && CH == 1 && !IDENT, "compressed rows: 64 lanes x 4");
      for (; j + U <= cnt; j += U) spmm_batch_zrows<IdxT, true, U>(a, j, cnt, myc, lane, acc);
      if (j < cnt) spmm_batch_zrows<IdxT, false, U>(a, j, cnt, myc, lane, acc);
    } else {
      for (; j + STEP <= cnt; j += STEP) {
        spmm_batch<IdxT, VW, LPR, CH, WMODE, IDENT, true, UF>(a, j, cnt, sub, myc, mye, mym, fo,
                                                              fv, head, acc);
      }
      if (j < cnt) {
        spmm_batch<IdxT, VW, LPR, CH, WMODE, IDENT, false, UF>(a, j, cnt, sub, myc, mye, mym, fo,
                                                               fv, head, acc);
      }
    }
  }
}

template <int VW, int LPR, int CH>
__device__ __forceinline__ void feature_slots(int lane, int64_t F, int head_dim, int (&fo)[CH],
                                              bool (&fv)[CH], int (&head)[CH]) {
  const int lir = lane % LPR;
  const int f0 = blockIdx.y * (LPR * CH * VW);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    fo[c] = f0 + (lir + LPR * c) * VW;
    fv[c] = fo[c] < F;
    head[c] = fv[c] ? fo[c] / head_dim : 0;
  }
}

template <int VW, int LPR, int CH>
__device__ __forceinline__ void combine_subgroups(float (&acc)[CH][VW]) {
#pragma unroll
  for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int i = 0; i < VW; ++i) acc[c][i] += __shfl_xor(acc[c][i], off, kWave);
    }
  }
}

}  // namespace pygamd
