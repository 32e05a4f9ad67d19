// gemm.hip — the dense feature transform of the layers (nn/dense/linear.py:121-127 `F.linear`,
// sage_conv.py:134-139 `lin_l(agg) + lin_r(x)`) on the fp32 matrix cores of gfx950.
//
// MFMA-bound: v_mfma_f32_32x32x2_f32 (exact fp32 — bitwise an fmaf chain — 64 cycles per
// instruction per SIMD, 157 TFLOP/s chip peak).  Three entry points, all row-major with explicit
// leading dimensions so they read / write halves of the `[agg | x]` buffers in place:
//
//   linear_forward : out[M, N]  = act(x[M, K] @ w[N, K]^T + bias)          ("NT")
//   linear_dgrad   : out[M, K]  = g[M, N] @ w[N, K]   (w passed transposed -> the same NT kernel),
//                    optionally out[:, :n_scaled] *= row_scale[row]  (the 1/deg of the mean that
//                    the transposed SpMM would otherwise gather once per edge)
//   linear_wgrad   : out[N, K]  = g[M, N]^T @ x[M, K]                      ("TN", split over M)
//
// NT kernel.  A workgroup (4 waves) owns a (WM*TM*32) x (WN*TN*32) output tile, a wave TM x TN
// accumulator blocks of 32 x 32.  K is walked in chunks of 32 through a double-buffered LDS ring
// (one barrier per chunk): global -> registers (16-byte loads, issued one chunk ahead) ->
// ds_write_b128 -> fragments.  The k index consumed by MFMA step s of a chunk is (s + 16 h) for
// lane half h = lane >> 5 — any assignment of the 32 k values to (step, half) is legal as long as
// A and B agree — so a lane's sixteen A (or B) operands of a chunk are 16 CONSECUTIVE floats of
// one row: four ds_read_b128 per 32 x 32 block per chunk instead of sixteen ds_read_b32, from a
// row stride of 36 floats that is conflict-free for both the 16-byte writes and reads.
// Workgroups are numbered so that the column tiles of one row block run back to back on one XCD
// (x rows are fetched from HBM once and served to the sibling tiles by that XCD's L2).
//
// TN kernel.  out tile 128 x 128 (2 x 2 waves, 2 x 2 blocks each); the reduction runs over the
// rows of a split of M in chunks of 32 rows staged row-major in LDS ([32][132]); the two operands
// of an MFMA step are rows (2 s + h) of the staged g and x blocks (conflict-free ds_read_b32).
// Splits write their partial tiles to a slab; a second kernel adds the slabs in split order
// (deterministic, no atomics).  Workgroups of one split are adjacent on one XCD: each row of g and
// x leaves HBM once.
#include "split_bf16.h"
#include "../../include/pyg_amd_lab.h"

namespace pygamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- "split" mode: fp32 operands as sums of three bf16 terms (split_bf16.h).  Opt-in
// (pygamd_set_gemm_mode); the default is the exact fp32 instruction.
struct SplitFrag {
  bf16x8 p[3];
};

// eight consecutive k values of one row (two 16-byte LDS reads) -> three bf16x8 operands
__device__ __forceinline__ SplitFrag split_frag(const f32x4& v0, const f32x4& v1) {
  u32x4 w[3];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = q < 2 ? v0[2 * q] : v1[2 * q - 4];
    const float x1 = q < 2 ? v0[2 * q + 1] : v1[2 * q - 3];
    uint32_t t[3];
    split_pair(x0, x1, t);
    w[0][q] = t[0];
    w[1][q] = t[1];
    w[2][q] = t[2];
  }
  SplitFrag f;
#pragma unroll
  for (int t = 0; t < 3; ++t) f.p[t] = __builtin_bit_cast(bf16x8, w[t]);
  return f;
}

// Callers walk the six terms in the OUTER loop and their 32 x 32 blocks in the inner one:
// consecutive instructions then write different accumulators (six back-to-back instructions on
// one accumulator, with the conversion VALU work scheduled in between, leave the matrix pipe idle
// waiting on the dependency — measured 31 % busy).
__device__ __forceinline__ void split_term(int t, const SplitFrag& a, const SplitFrag& b,
                                           f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[kSplitTa[t]], b.p[kSplitTb[t]], acc, 0, 0, 0);
}

constexpr int kGK = 32;        // k chunk
constexpr int kGLD = kGK + 4;  // LDS row stride (floats) of the NT kernel's tiles

struct GemmNT {
  const float* __restrict__ a;  // [M, K]
  const float* __restrict__ b;  // [N, K]
  const float* __restrict__ bias;       // [N] or null
  const float* __restrict__ row_scale;  // [M] or null
  const float* __restrict__ mask;       // [M, ldm] or null: c = mask <= 0 ? 0 : c (ReLU backward)
  const uint32_t* __restrict__ mask_bits;  // the same mask as bits (32 x 32 tiles, see pyg_amd.h)
  int64_t ldmb;
  float* __restrict__ c;                // [M, N]
  float* __restrict__ c2;               // null or [M, N]: c * row_scale[row], every column (a
  int64_t ldc2;                         // second, row-scaled copy of the result)
  int64_t M, lda, ldb, ldc, ldm;
  int N, K;
  int relu;
  int n_scaled;   // columns [0, n_scaled) are multiplied by row_scale[row]
  int accumulate; // c += result
  int tiles_m, tiles_n;
  int split;      // 3 x bf16 operand split instead of the fp32 instruction
  // split-K (few row tiles: sampled blocks, Cora-sized inputs): grid.y = ksplits slices of
  // k_per_split (a multiple of the chunk) reduction columns each; a slice writes its raw partial
  // tile to partial[slice][M][N] and gemm_nt_splitk_epilogue sums the slices in order and applies
  // every epilogue of this struct (deterministic, no atomics)
  int ksplits, k_per_split;
  float* __restrict__ partial;
};

// Tile numbering: hardware block b runs on XCD b % 8.  Logical order inside an XCD: for each row
// block, all of its column tiles consecutively.
__device__ __forceinline__ void nt_tile_of_block(const GemmNT& p, int& tm, int& tn) {
  const int64_t b = blockIdx.x;
  const int64_t per_xcd = gridDim.x >> 3;
  const int64_t q = (b & 7) * per_xcd + (b >> 3);  // logical id, contiguous per XCD
  tm = static_cast<int>(q / p.tiles_n);
  tn = static_cast<int>(q - static_cast<int64_t>(tm) * p.tiles_n);
}

template <int WM, int WN, int TM, int TN, bool VEC, bool SPLIT>
__global__ void __launch_bounds__(kBlock, 2) gemm_nt_kernel(GemmNT p) {
  static_assert(WM * WN == 4, "four waves per workgroup");
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int A_LD4 = BM * (kGK / 4) / kBlock;  // 16-byte loads per thread per chunk
  constexpr int B_LD4 = (BN * (kGK / 4) + kBlock - 1) / kBlock;
  extern __shared__ __align__(16) float smem[];
  float* As = smem;                       // [2][BM][kGLD]
  float* Bs = smem + 2 * BM * kGLD;       // [2][BN][kGLD]
  int tile_m, tile_n;
  nt_tile_of_block(p, tile_m, tile_n);
  if (tile_m >= p.tiles_m) return;
  const int64_t m0 = static_cast<int64_t>(tile_m) * BM;
  const int n0 = tile_n * BN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  // this workgroup's slice of the reduction: [kb, ke) (the whole of K without split-K)
  const int kb = p.ksplits > 1 ? static_cast<int>(blockIdx.y) * p.k_per_split : 0;
  const int ke = p.ksplits > 1 ? (kb + p.k_per_split < p.K ? kb + p.k_per_split : p.K) : p.K;

  // ---- global -> register staging map: thread -> (row r0 + 32 q, 16-byte column kq).  Row
  // pointers are fixed for the whole K walk; rows past M / N are clamped (never stored / zeroed).
  const int kq = threadIdx.x & 7;
  const int r0 = threadIdx.x >> 3;
  const float* pa[A_LD4];
  const float* pb[B_LD4];
  bool okb[B_LD4];
#pragma unroll
  for (int q = 0; q < A_LD4; ++q) {
    int64_t row = m0 + r0 + 32 * q;
    row = row < p.M ? row : p.M - 1;
    pa[q] = p.a + row * p.lda;
  }
#pragma unroll
  for (int q = 0; q < B_LD4; ++q) {
    int row = n0 + r0 + 32 * q;
    okb[q] = row < p.N && (BN % 32 == 0 || r0 + 32 * q < BN);
    row = row < p.N ? row : p.N - 1;
    pb[q] = p.b + static_cast<int64_t>(row) * p.ldb;
  }
  f32x4 ra[A_LD4], rb[B_LD4];
  // Loads only ISSUE here (clamped to a valid, aligned column); the zeroing of the K tail and of
  // the rows past N happens when the registers move to LDS one iteration later — a select next to
  // the load would make the wave wait for the data right away.
  auto load_chunk = [&](int k0) {
    const int k = k0 + 4 * kq;
    if (VEC) {
      const int kc = k < ke ? k : 0;  // column 0 exists whenever a chunk is loaded (K > 0)
#pragma unroll
      for (int q = 0; q < A_LD4; ++q) ra[q] = *reinterpret_cast<const f32x4*>(pa[q] + kc);
#pragma unroll
      for (int q = 0; q < B_LD4; ++q) rb[q] = *reinterpret_cast<const f32x4*>(pb[q] + kc);
    } else {
#pragma unroll
      for (int q = 0; q < A_LD4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[q][e] = pa[q][k + e < ke ? k + e : 0];
#pragma unroll
      for (int q = 0; q < B_LD4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) rb[q][e] = pb[q][k + e < ke ? k + e : 0];
    }
  };
  auto store_chunk = [&](int buf, int k0) {
    float* as = As + buf * BM * kGLD + r0 * kGLD + 4 * kq;
    float* bs = Bs + buf * BN * kGLD + r0 * kGLD + 4 * kq;
    const int k = k0 + 4 * kq;
    // both operands are zeroed past K (a clamped A value could be Inf/NaN: Inf * 0 = NaN)
#pragma unroll
    for (int q = 0; q < A_LD4; ++q) {
      f32x4 v = ra[q];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (k + e < ke) ? v[e] : 0.f;
      *reinterpret_cast<f32x4*>(as + 32 * q * kGLD) = v;
    }
#pragma unroll
    for (int q = 0; q < B_LD4; ++q) {
      if (BN % 32 == 0 || r0 + 32 * q < BN) {
        f32x4 v = rb[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (okb[q] && k + e < ke) ? v[e] : 0.f;
        *reinterpret_cast<f32x4*>(bs + 32 * q * kGLD) = v;
      }
    }
  };
  // fragments of half a chunk (8 MFMA steps): 8 consecutive floats per 32 x 32 block
  struct Frag {
    f32x4 a[TM][2], b[TN][2];
  };
  const int a_off = (wm * TM * 32 + li) * kGLD + 16 * lh;
  const int b_off = (wn * TN * 32 + li) * kGLD + 16 * lh;
  auto read_frag = [&](Frag& f, int buf, int hf) {
    const float* as = As + buf * BM * kGLD + a_off + 8 * hf;
    const float* bs = Bs + buf * BN * kGLD + b_off + 8 * hf;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int v = 0; v < 2; ++v)
        f.a[i][v] = *reinterpret_cast<const f32x4*>(as + i * 32 * kGLD + 4 * v);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 2; ++v)
        f.b[j][v] = *reinterpret_cast<const f32x4*>(bs + j * 32 * kGLD + 4 * v);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  auto mma_steps = [&](const Frag& f, int s_begin, int s_end) {
#pragma unroll
    for (int s = s_begin; s < s_end; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][s >> 2][s & 3],
                                                            f.b[j][s >> 2][s & 3], acc[i][j], 0, 0,
                                                            0);
  };
  // Split mode: a half-chunk fragment is ONE bf16 step over the same 16 k values (lane half h holds
  // its 8 consecutive ones), six instructions per 32 x 32 block.
  auto split_all = [&](const Frag& f, SplitFrag (&sa)[TM], SplitFrag (&sb)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) sa[i] = split_frag(f.a[i][0], f.a[i][1]);
#pragma unroll
    for (int j = 0; j < TN; ++j) sb[j] = split_frag(f.b[j][0], f.b[j][1]);
  };
  auto mma_terms = [&](const SplitFrag (&sa)[TM], const SplitFrag (&sb)[TN], int t_begin,
                       int t_end) {
#pragma unroll
    for (int t = t_begin; t < t_end; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) split_term(t, sa[i], sb[j], acc[i][j]);
  };

  // Pipeline (ONE barrier per chunk, every LDS read issued at least 16 MFMAs before its use):
  //   iteration c:  LDS[c+1] <- registers;  registers <- global chunk c+2;
  //                 f1 <- LDS[c] upper half;  MFMA(f0 = lower half of chunk c, steps 0-3);
  //                 barrier;  MFMA(f0, steps 4-7);  f0 <- LDS[c+1] lower half;  MFMA(f1).
  // The barrier orders (a) the stores of chunk c+1 before anybody's reads of it and (b) every
  // wave's reads of LDS[c] before the stores of chunk c+2 into the same buffer next iteration.
  const int n_chunks = (ke - kb + kGK - 1) / kGK;
  Frag f0, f1;
  if (n_chunks > 0) {
    load_chunk(kb);
    store_chunk(0, kb);
  }
  if (n_chunks > 1) load_chunk(kb + kGK);
  __syncthreads();
  if (n_chunks > 0) read_frag(f0, 0, 0);
  if constexpr (SPLIT) {
    // Software pipeline over half-chunks: while the 6 * TM * TN matrix instructions of one
    // half-chunk run, the VALU converts the NEXT half-chunk's fp32 fragments into bf16 terms
    // (a wave issues in order: conversion and multiplication of the same fragment back to back
    // would leave each pipe idle while the other one works).  sched_group_barrier pins the
    // interleave: one matrix instruction, then its share of the conversion work.
    constexpr int kMfma = kSplitTerms * TM * TN;       // per half-chunk
    constexpr int kValuPer = (TM + TN) * 44 / kMfma + 1;  // ~44 VALU per converted fragment
    SplitFrag ca[TM], cb[TN], na[TM], nb[TN];
    if (n_chunks > 0) split_all(f0, ca, cb);
    for (int c = 0; c < n_chunks; ++c) {
      const int buf = c & 1;
      if (c + 1 < n_chunks) store_chunk(buf ^ 1, kb + (c + 1) * kGK);
      if (c + 2 < n_chunks) load_chunk(kb + (c + 2) * kGK);
      __builtin_amdgcn_sched_barrier(0);
      read_frag(f1, buf, 1);
      split_all(f1, na, nb);
      mma_terms(ca, cb, 0, kSplitTerms);
#pragma unroll
      for (int q = 0; q < kMfma; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, kValuPer, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      // (the last iteration reads a buffer nobody filled: converted, never multiplied)
      read_frag(f0, buf ^ 1, 0);
      split_all(f0, ca, cb);
      // rows of the K tail are staged as zeros, so the upper half-chunk is always multiplied
      mma_terms(na, nb, 0, kSplitTerms);
#pragma unroll
      for (int q = 0; q < kMfma; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, kValuPer, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    for (int c = 0; c < n_chunks; ++c) {
      const int buf = c & 1;
      if (c + 1 < n_chunks) store_chunk(buf ^ 1, kb + (c + 1) * kGK);
      if (c + 2 < n_chunks) load_chunk(kb + (c + 2) * kGK);
      read_frag(f1, buf, 1);
      mma_steps(f0, 0, 4);
      // keep the barrier in the MIDDLE of the MFMA stream (hipcc hoists it to the top otherwise:
      // the wave would then sit in lgkmcnt(0) + s_barrier with one MFMA in flight)
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      mma_steps(f0, 4, 8);
      if (c + 1 < n_chunks) read_frag(f0, buf ^ 1, 0);
      // a K tail of <= 8 leaves the upper 8 steps all zero in both lane halves: skip them
      if (ke - kb - c * kGK > 8) mma_steps(f1, 0, 8);
    }
  }

  // ---- epilogue: reg e of lane l is C[(e & 3) + 8 (e >> 2) + 4 (l >> 5)][l & 31]
  if (p.ksplits > 1) {  // raw partial tile of this K slice; the epilogue runs in the combine pass
    float* __restrict__ slab = p.partial + static_cast<int64_t>(blockIdx.y) * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int64_t rbase = m0 + (wm * TM + i) * 32 + 4 * lh;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + li;
        if (col >= p.N) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t row = rbase + (e & 3) + 8 * (e >> 2);
          if (row < p.M) slab[row * p.N + col] = acc[i][j][e];
        }
      }
    }
    return;
  }
  const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.N);
  const bool wave_scaled = p.n_scaled > n0 + wn * TN * 32;  // wave-uniform
  if (full && !p.accumulate) {
    // interior tile (all but the last row block): no bounds checks, no read-modify-write
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int64_t rbase = m0 + (wm * TM + i) * 32 + 4 * lh;
      f32x4 sc[4];
      if (wave_scaled || p.c2) {  // rows rbase + 8 g + {0..3}: four aligned 16-byte loads
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
          for (int e = 0; e < 4; ++e) sc[g4][e] = p.row_scale[rbase + 8 * g4 + e];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + li;
        const float bv = p.bias ? p.bias[col] : 0.f;
        const bool scaled = wave_scaled && col < p.n_scaled;
        float* cp = p.c + rbase * p.ldc + col;
        float* cp2 = p.c2 ? p.c2 + rbase * p.ldc2 + col : nullptr;
        auto emit = [&](auto keep) {  // keep(e): the result of register e survives the mask
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float v = acc[i][j][e] + bv;
            if (wave_scaled) v = scaled ? v * sc[e >> 2][e & 3] : v;
            if (p.relu) v = (v > 0.f || v != v) ? v : 0.f;  // NaN propagates like torch.relu
            v = keep(e) ? v : 0.f;
            cp[((e & 3) + 8 * (e >> 2)) * p.ldc] = v;
            if (cp2) cp2[((e & 3) + 8 * (e >> 2)) * p.ldc2] = v * sc[e >> 2][e & 3];
          }
        };
        if (p.mask) {  // uniform: the ReLU-backward epilogue of a dgrad launch
          const float* mp = p.mask + rbase * p.ldm + col;
          float mv[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) mv[e] = mp[((e & 3) + 8 * (e >> 2)) * p.ldm];
          emit([&](int e) { return mv[e] > 0.f; });
        } else if (p.mask_bits) {  // uniform.  This 32 x 32 block is one bit tile = one 128-byte line
          const int64_t tile_row = (rbase - 4 * lh) >> 5;  // (m0 and the block offsets are % 32)
          const uint32_t word = p.mask_bits[(tile_row * p.ldmb + (col >> 5)) * 32 + li];
          uint32_t bw[16];
#pragma unroll
          for (int e = 0; e < 16; ++e)
            bw[e] = __shfl(word, (e & 3) + 8 * (e >> 2) + 4 * lh, kWave);
          emit([&](int e) { return ((bw[e] >> li) & 1u) != 0; });
        } else {
          emit([](int) { return true; });
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int64_t rbase = m0 + (wm * TM + i) * 32 + 4 * lh;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + li;
      if (col >= p.N) continue;
      const float bv = p.bias ? p.bias[col] : 0.f;
      const bool scaled = col < p.n_scaled;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = rbase + (e & 3) + 8 * (e >> 2);
        if (row < p.M) {
          float v = acc[i][j][e] + bv;
          if (scaled) v *= p.row_scale[row];
          if (p.relu) v = (v > 0.f || v != v) ? v : 0.f;
          float* dst = p.c + row * p.ldc + col;
          if (p.accumulate) v += *dst;
          if (p.mask) v = p.mask[row * p.ldm + col] > 0.f ? v : 0.f;
          if (p.mask_bits)
            v = ((p.mask_bits[((row >> 5) * p.ldmb + (col >> 5)) * 32 + (row & 31)] >> (col & 31)) &
                 1u)
                    ? v
                    : 0.f;
          *dst = v;
          if (p.c2) p.c2[row * p.ldc2 + col] = v * p.row_scale[row];
        }
      }
    }
  }
}

// split-K combine: c = epilogue(sum over slices, in slice order) — one thread per output element
__global__ void __launch_bounds__(kBlock) gemm_nt_splitk_epilogue(GemmNT p) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t MN = p.M * p.N;
  if (t >= MN) return;
  const int64_t row = t / p.N;
  const int col = static_cast<int>(t - row * p.N);
  float v = 0.f;
  int sp = 0;
  for (; sp + 4 <= p.ksplits; sp += 4) {  // (four slices' loads in flight, added in slice order)
    float u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = p.partial[(sp + i) * MN + t];
#pragma unroll
    for (int i = 0; i < 4; ++i) v += u[i];
  }
  for (; sp < p.ksplits; ++sp) v += p.partial[sp * MN + t];
  if (p.bias) v += p.bias[col];
  if (col < p.n_scaled) v *= p.row_scale[row];
  if (p.relu) v = (v > 0.f || v != v) ? v : 0.f;
  float* dst = p.c + row * p.ldc + col;
  if (p.accumulate) v += *dst;
  if (p.mask) v = p.mask[row * p.ldm + col] > 0.f ? v : 0.f;
  if (p.mask_bits)
    v = ((p.mask_bits[((row >> 5) * p.ldmb + (col >> 5)) * 32 + (row & 31)] >> (col & 31)) & 1u)
            ? v
            : 0.f;
  *dst = v;
  if (p.c2) p.c2[row * p.ldc2 + col] = v * p.row_scale[row];
}

// ---- TN: out[N, K] = g[M, N]^T @ x[M, K], split over M -------------------------------------------
constexpr int kWRows = 32;          // rows of M per staged block
constexpr int kWTile = 128;         // output tile edge
constexpr int kWLD = kWTile + 4;    // LDS row stride

struct GemmTN {
  const float* __restrict__ g;  // [M, N]
  const float* __restrict__ x;  // [M, K]  (x2 == null)  or  [M, K1]: output columns [0, K1)
  const float* __restrict__ x2; // null or [M, K - K1]: output columns [K1, K) — the weight
                                // gradient against [x | x2] without the concatenated copy
  int64_t ldx2;
  int K1, tiles_k1;             // (x2 != null) column tiles never straddle the two operands
  float* __restrict__ partial;  // [splits][N][K]
  float* __restrict__ colsum;   // [splits][N] or null: per-split column sums of g (bias gradient)
  int64_t M, ldg, ldx;
  int N, K;
  int tiles_n, tiles_k, splits;
  int64_t rows_per_split;       // multiple of kWRows
  int vec_g, vec_x, vec_x2;     // (split kernel) 16-byte loads are legal for this operand
};

// eight k values gathered one by one (the TN kernel's operands run DOWN the staged rows)
__device__ __forceinline__ SplitFrag split_frag8(const float (&v)[8]) {
  const f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
  return split_frag(lo, hi);
}

template <bool VEC, bool SPLIT>
__global__ void __launch_bounds__(kBlock, 2) gemm_tn_kernel(GemmTN p) {
  extern __shared__ __align__(16) float smem[];
  float (*Gs)[kWRows][kWLD] = reinterpret_cast<float (*)[kWRows][kWLD]>(smem);
  float (*Xs)[kWRows][kWLD] =
      reinterpret_cast<float (*)[kWRows][kWLD]>(smem + 2 * kWRows * kWLD);
  // logical id: per XCD, all tiles of one split consecutively
  const int64_t b = blockIdx.x;
  const int64_t per_xcd = gridDim.x >> 3;
  const int64_t q = (b & 7) * per_xcd + (b >> 3);
  const int tiles = p.tiles_n * p.tiles_k;
  const int64_t split = q / tiles;
  if (split >= p.splits) return;
  const int t = static_cast<int>(q - split * tiles);
  const int tn = t / p.tiles_k, tk = t - tn * p.tiles_k;
  const int n0 = tn * kWTile;
  // the tile's operand, its first column inside that operand (k0), the operand's width (Kop) and
  // the tile's first OUTPUT column (kout0)
  const bool two = p.K1 < p.K;  // a second operand supplies the output columns [K1, K)
  const bool second = two && tk >= p.tiles_k1;
  const float* __restrict__ xsrc = second ? p.x2 : p.x;
  const int64_t xld = second ? p.ldx2 : p.ldx;
  const int k0 = (second ? tk - p.tiles_k1 : tk) * kWTile;
  const int Kop = second ? p.K - p.K1 : p.K1;
  const int kout0 = (second ? p.K1 : 0) + k0;
  const int64_t ra = split * p.rows_per_split;
  int64_t rb = ra + p.rows_per_split;
  rb = rb < p.M ? rb : p.M;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wn = wave >> 1, wk = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  // staging: thread -> (row sr + 8 j, 16-byte column sc): a wave covers two 512-byte rows
  const int sc = threadIdx.x & 31;
  const int sr = threadIdx.x >> 5;
  const int gcol = n0 + 4 * sc, xcol = k0 + 4 * sc;
  f32x4 rg[4], rx[4];
  auto load_rows = [&](int64_t r) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t row = r + sr + 8 * j;
      const bool ok = row < rb;
      const int64_t rs = ok ? row : ra;
      const float* gp = p.g + rs * p.ldg + gcol;
      const float* xp = xsrc + rs * xld + xcol;
      if (VEC) {
        rg[j] = (ok && gcol < p.N) ? *reinterpret_cast<const f32x4*>(gp)
                                   : f32x4{0.f, 0.f, 0.f, 0.f};
        rx[j] = (ok && xcol < Kop) ? *reinterpret_cast<const f32x4*>(xp)
                                   : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          rg[j][e] = (ok && gcol + e < p.N) ? gp[e] : 0.f;
          rx[j][e] = (ok && xcol + e < Kop) ? xp[e] : 0.f;
        }
      }
    }
  };
  // bias gradient for free: the workgroups of the first k tile add up the g rows they stage
  // (rows past the split are staged as zeros); fixed order: rows of a thread, the 8 staging rows,
  // then the splits in the reduce kernel
  const bool do_colsum = p.colsum != nullptr && tk == 0;  // workgroup-uniform
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  auto store_rows = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<f32x4*>(&Gs[buf][sr + 8 * j][4 * sc]) = rg[j];
      *reinterpret_cast<f32x4*>(&Xs[buf][sr + 8 * j][4 * sc]) = rx[j];
    }
    if (do_colsum) {
#pragma unroll
      for (int j = 0; j < 4; ++j) csum += rg[j];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // wave-uniform: which 32-wide blocks of this wave's range exist
  const bool nb0 = n0 + wn * 64 < p.N, nb1 = n0 + wn * 64 + 32 < p.N;
  const bool kb0 = k0 + wk * 64 < Kop, kb1 = k0 + wk * 64 + 32 < Kop;

  const int64_t n_blocks = (rb - ra + kWRows - 1) / kWRows;
  if (n_blocks > 0) {
    load_rows(ra);
    store_rows(0);
    if (n_blocks > 1) load_rows(ra + kWRows);
  }
  __syncthreads();
  // One barrier per 32-row block, in the MIDDLE of the block's MFMAs (cf. the NT kernel): the
  // operands of the block's second half (8 steps) are read into registers before the barrier, so
  // that after it nobody touches buffer `buf` any more and the next iteration may overwrite it.
  const bool whole = nb0 && nb1 && kb0 && kb1;  // wave-uniform: all four 32 x 32 blocks exist
  auto mma_step = [&](float a0, float a1, float b0, float b1) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
    if (kb1) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    if (nb1) {
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      if (kb1) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  };
  // split mode: eight values down a staged column -> three bf16x8 operands
  auto read_split = [&](const float* col) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = col[e * kWLD];
    return split_frag8(v);
  };
  auto mma4 = [&](const SplitFrag& a0, const SplitFrag& a1, const SplitFrag& b0,
                  const SplitFrag& b1) {
#pragma unroll
    for (int t = 0; t < kSplitTerms; ++t) {
      split_term(t, a0, b0, acc[0][0]);
      split_term(t, a0, b1, acc[0][1]);
      split_term(t, a1, b0, acc[1][0]);
      split_term(t, a1, b1, acc[1][1]);
    }
  };
  const bool active = nb0 && kb0;  // wave-uniform: this wave's column ranges are not empty
  SplitFrag ca0, ca1, cb0, cb1;
  if (SPLIT && active && n_blocks > 0) {
    ca0 = read_split(&Gs[0][8 * lh][wn * 64 + li]);
    ca1 = read_split(&Gs[0][8 * lh][wn * 64 + 32 + li]);
    cb0 = read_split(&Xs[0][8 * lh][wk * 64 + li]);
    cb1 = read_split(&Xs[0][8 * lh][wk * 64 + 32 + li]);
  }
  for (int64_t c = 0; c < n_blocks; ++c) {
    const int buf = static_cast<int>(c & 1);
    if (c + 1 < n_blocks) store_rows(buf ^ 1);
    if (c + 2 < n_blocks) load_rows(ra + (c + 2) * kWRows);
    if constexpr (SPLIT) {
      // bf16 step t of a block covers rows 16 t + 8 h + e (e = 0..7 down the staged rows) in lane
      // half h — the same rows for g and x, which is all the reduction needs.  Columns past N / K
      // are staged as zeros, so all four 32 x 32 blocks are always computed.  Software pipeline
      // as in the NT kernel: the matrix instructions of one step run while the VALU converts the
      // operands of the next one.
      constexpr int kMfma = kSplitTerms * 4;
      constexpr int kValuPer = 4 * 44 / kMfma + 1;
      if (active) {
        __builtin_amdgcn_sched_barrier(0);
        const SplitFrag na0 = read_split(&Gs[buf][16 + 8 * lh][wn * 64 + li]);
        const SplitFrag na1 = read_split(&Gs[buf][16 + 8 * lh][wn * 64 + 32 + li]);
        const SplitFrag nb0 = read_split(&Xs[buf][16 + 8 * lh][wk * 64 + li]);
        const SplitFrag nb1 = read_split(&Xs[buf][16 + 8 * lh][wk * 64 + 32 + li]);
        mma4(ca0, ca1, cb0, cb1);
#pragma unroll
        for (int q = 0; q < kMfma; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, kValuPer, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        // (after the last block this reads a buffer nobody filled: converted, never multiplied)
        ca0 = read_split(&Gs[buf ^ 1][8 * lh][wn * 64 + li]);
        ca1 = read_split(&Gs[buf ^ 1][8 * lh][wn * 64 + 32 + li]);
        cb0 = read_split(&Xs[buf ^ 1][8 * lh][wk * 64 + li]);
        cb1 = read_split(&Xs[buf ^ 1][8 * lh][wk * 64 + 32 + li]);
        mma4(na0, na1, nb0, nb1);
#pragma unroll
        for (int q = 0; q < kMfma; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, kValuPer, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      } else {
        __syncthreads();
      }
      continue;
    }
    const float* gs = &Gs[buf][lh][wn * 64 + li];
    const float* xs = &Xs[buf][lh][wk * 64 + li];
    float ha0[8], ha1[8], hb0[8], hb1[8];  // second half of the block, steps 8..15
    if (nb0 && kb0) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        ha0[s] = gs[2 * (s + 8) * kWLD];
        ha1[s] = gs[2 * (s + 8) * kWLD + 32];
        hb0[s] = xs[2 * (s + 8) * kWLD];
        hb1[s] = xs[2 * (s + 8) * kWLD + 32];
      }
      if (whole) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const float a0 = gs[2 * s * kWLD], a1 = gs[2 * s * kWLD + 32];
          const float b0 = xs[2 * s * kWLD], b1 = xs[2 * s * kWLD + 32];
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int s = 0; s < 8; ++s)
          mma_step(gs[2 * s * kWLD], gs[2 * s * kWLD + 32], xs[2 * s * kWLD],
                   xs[2 * s * kWLD + 32]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (nb0 && kb0) {
      if (whole) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ha0[s], hb0[s], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ha0[s], hb1[s], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ha1[s], hb0[s], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ha1[s], hb1[s], acc[1][1], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) mma_step(ha0[s], ha1[s], hb0[s], hb1[s]);
      }
    }
  }
  if (do_colsum) {  // (uniform branch: the barriers are legal)
    __syncthreads();  // every wave is done with the ring
    float* red = smem;  // [8][128]
    *reinterpret_cast<f32x4*>(&red[sr * kWTile + 4 * sc]) = csum;
    __syncthreads();
    if (threadIdx.x < kWTile && n0 + static_cast<int>(threadIdx.x) < p.N) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += red[r * kWTile + threadIdx.x];
      p.colsum[split * p.N + n0 + threadIdx.x] = s;
    }
  }
  float* __restrict__ slab = p.partial + split * static_cast<int64_t>(p.N) * p.K;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (k0 + wk * 64 + j * 32 + li >= Kop) continue;
      const int col = kout0 + wk * 64 + j * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = n0 + wn * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row < p.N) slab[static_cast<int64_t>(row) * p.K + col] = acc[i][j][e];
      }
    }
  }
}

// ---- TN, split arithmetic, operands converted ONCE ------------------------------------------------
// The kernel above splits its operands in registers next to the matrix instructions: a lane
// gathers eight values DOWN a staged column (eight ds_read_b32) and converts them, and the two
// waves that share a column range convert the same values twice — the VALU, not the matrix pipe,
// sets its pace (123 TFLOP/s of the 417 the six-term product allows, profiles/r04_bench_kernel_
// stats.md).  Here the staging thread converts: it owns eight consecutive rows of four columns
// (eight 16-byte global loads), splits them into the three bf16 terms as (row, row + 1) pairs and
// writes, per column and term, ONE 16-byte vector of eight k-consecutive bf16 values — the layout
// the operand read wants, so a fragment is ONE ds_read_b128 per term.  Per 32-row block and
// thread: 16 split_pair (176 VALU instructions, half of before), 12 ds_write_b128, and per wave
// 24 ds_read_b128 for 48 matrix instructions.
//
// LDS image of a wave group: [operand g | x][term 0..2][column 0..127][20 dwords]: 32 rows of a
// column are 16 packed dwords, padded to 20 so that (i) the 16 lanes of a ds_read_b128 group,
// which read 16 different columns at the same row group, start at 16 different 4-bank slots
// (5 * column mod 16 is a bijection) and (ii) the eight lanes of a ds_write_b128 group — two
// neighbouring column quads x four row groups, see `c4` / `rg` below — cover all 32 banks.
//
// Schedule: ONE 512-thread workgroup per CU = two groups of four waves (one wave of each group on
// every SIMD), each with its own image (2 x 61,440 bytes), accumulators and half of the split's
// blocks (group A the even ones, group B the odd ones; B's sums are added to A's through LDS at
// the end).  The groups run in ANTI-PHASE, held there by the workgroup barrier: while A converts
// block 2 i (VALU, LDS stores), B multiplies block 2 i - 1 (matrix pipe), then the roles swap.  Two
// independent 256-thread workgroups per CU do not do this by themselves — they start together
// and stay IN phase (both convert, then both multiply: measured, phases switched off one by one,
// profiles/r04_wgrad_probe.txt: products alone 1.86 ms, conversion alone 1.46 ms, loads alone
// 1.77 ms, all three 4.17 ms on the 2.45 M x (256 | 256) -> 256 gradient).
constexpr int kCLD = 20;                 // dwords per staged column
constexpr int kCPlane = kWTile * kCLD;   // dwords per (operand, term) plane
constexpr int kTnGroup = 6 * kCPlane;    // dwords per wave group
constexpr size_t kTnSplitLds = sizeof(uint32_t) * 2 * kTnGroup;
constexpr int kTnThreads = 512;

// (single v_add_f32 for the bias-gradient sums: no packed fp32 next to the partner's products)
__device__ __forceinline__ float add_f32_asm(float a, float b) {
  float r;
  asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// PROBE (lab, timing only — results undefined for bits 1..3): bit 1 = no products, bit 2 = no
// conversion / LDS stores, bit 3 = no global loads, bit 5 = packed residual subtractions
template <bool VG, bool VX, int PROBE = 0>
__global__ void __launch_bounds__(kTnThreads, 1) gemm_tn_split_kernel(GemmTN p) {
  extern __shared__ __align__(16) uint32_t tn_lds[];
  const int64_t b = blockIdx.x;
  const int64_t per_xcd = gridDim.x >> 3;
  const int64_t q = (b & 7) * per_xcd + (b >> 3);
  const int tiles = p.tiles_n * p.tiles_k;
  const int64_t split = q / tiles;
  if (split >= p.splits) return;
  const int t = static_cast<int>(q - split * tiles);
  const int tn = t / p.tiles_k, tk = t - tn * p.tiles_k;
  const int n0 = tn * kWTile;
  const bool two = p.K1 < p.K;
  const bool second = two && tk >= p.tiles_k1;
  const float* __restrict__ xsrc = second ? p.x2 : p.x;
  const int64_t xld = second ? p.ldx2 : p.ldx;
  const int k0 = (second ? tk - p.tiles_k1 : tk) * kWTile;
  const int Kop = second ? p.K - p.K1 : p.K1;
  const int kout0 = (second ? p.K1 : 0) + k0;
  const int64_t ra = split * p.rows_per_split;
  int64_t rb = ra + p.rows_per_split;
  rb = rb < p.M ? rb : p.M;
  // (readfirstlane: the roles and everything derived from them live in scalar registers, so the
  // role branches below are scalar branches)
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int grp = wave8 >> 2, wave = wave8 & 3;
  uint32_t* const planes = tn_lds + grp * kTnGroup;
  const int wn = wave >> 1, wk = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  // staging (per group): waves 0, 1 stage g (columns 0..63, 64..127 of the tile), waves 2, 3
  // stage x.  Lane bits: 0 = low bit of the column quad, 1..2 = row group (8 rows), 3..5 = high
  // bits of the quad: a load instruction of a wave fetches 4 rows x 256 contiguous bytes.
  const bool isx = wave >= 2;
  const int c4 = (lane & 1) + 2 * (lane >> 3);
  const int rg = (lane >> 1) & 3;
  const int tc = 64 * (wave & 1) + 4 * c4;              // first of the thread's 4 tile columns
  const float* __restrict__ src = isx ? xsrc : p.g;
  const int64_t sld = isx ? xld : p.ldg;
  const int col_lim = isx ? Kop - k0 : p.N - n0;        // valid columns of this tile (> 0)
  const int col0 = isx ? k0 : n0;
  // 16-byte loads per operand (VG: g; VX: x and x2): a 47-column g next to 256-column x's keeps
  // the wide loads for the x's (scalar branch; folded when VG == VX)
  const bool vec = isx ? VX : VG;
  // Loads are UNCONDITIONAL (columns clamped to the tile's last valid one, the rows of a split's
  // ragged last block to its last row) and the masks are applied to the values when they are
  // converted: a predicated load makes the compiler wait for the whole batch right behind it
  // (s_waitcnt vmcnt(0) after the last load of the block: no prefetch at all).
  bool cok[4];
  int64_t coff[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    cok[e] = tc + e < col_lim;
    coff[e] = col0 + (cok[e] ? tc + e : col_lim - 1);
  }
  const int64_t voff = col0 + (cok[0] ? tc : 0);  // (vector rows: the whole quad is valid or not)
  const bool ragged_cols = !(cok[0] && cok[3]);  // some column of the thread is staged as zero
  const float* const tp = src + (ra + 8 * rg) * sld;  // the thread's first row
  // full blocks only (splits are multiples of 32 rows: only the LAST split has a ragged block,
  // handled after the loops)
  auto load_rows = [&](f32x4 (&st)[8], int64_t blk) {
    if constexpr ((PROBE & 8) != 0) return;
    const float* bp = tp + blk * kWRows * sld;
    if (vec) {
#pragma unroll
      for (int r = 0; r < 8; ++r) st[r] = *reinterpret_cast<const f32x4*>(bp + r * sld + voff);
    } else {
      // four 4-byte loads off ONE row address (rows that are not 16-byte aligned: a 47-column
      // g).  A quad that straddles the operand's last column reads up to three elements of the
      // NEXT row (masked when converted) — which is why the block holding the operand's very
      // last row always takes load_tail's exact addresses (`exact_tail` below).
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float* rp = bp + r * sld + voff;
#pragma unroll
        for (int e = 0; e < 4; ++e) st[r][e] = rp[e];
      }
    }
  };
  auto load_tail = [&](f32x4 (&st)[8], int64_t r0) {  // rows clamped to the last one
    const int64_t last = rb - 1;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      int64_t row = r0 + 8 * rg + r;
      row = row < last ? row : last;
      const float* sp = src + row * sld;
#pragma unroll
      for (int e = 0; e < 4; ++e) st[r][e] = sp[coff[e]];
    }
  };
  const bool do_colsum = p.colsum != nullptr && tk == 0;  // workgroup-uniform
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  uint32_t* const wbase = planes + (isx ? 3 * kCPlane : 0) + tc * kCLD + 4 * rg;
  // registers -> the three term planes (no masks) + the bias-gradient sums
  auto convert_store = [&](f32x4 (&st)[8]) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      u32x4 w[3];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        uint32_t tt[3];
        // (unpacked subtractions, split_bf16.h; PROBE bit 5 = the compiler's packed ones)
        if constexpr ((PROBE & 32) != 0) split_pair(st[2 * qq][cc], st[2 * qq + 1][cc], tt);
        else split_pair_single(st[2 * qq][cc], st[2 * qq + 1][cc], tt);
        w[0][qq] = tt[0];
        w[1][qq] = tt[1];
        w[2][qq] = tt[2];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k)
        *reinterpret_cast<u32x4*>(wbase + k * kCPlane + cc * kCLD) = w[k];
    }
    if (do_colsum && !isx) {  // (single v_add_f32: no packed fp32 next to the partner's products)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[e] = add_f32_asm(csum[e], st[r][e]);
      }
    }
  };
  // A full block inside the loop.  16-byte operands: a column quad is valid or not as a whole and
  // the invalid ones were zeroed in the planes ONCE (below), so there is no mask and no merge of a
  // masked with an unmasked copy of the registers (45 moves per block before).  Element-wise
  // operands: the boundary quad is masked per element.
  auto store_rows = [&](f32x4 (&st)[8]) {
    if constexpr ((PROBE & 4) != 0) return;
    if (!vec && ragged_cols) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int e = 0; e < 4; ++e) st[r][e] = cok[e] ? st[r][e] : 0.f;
      }
    }
    if (!vec || cok[0]) convert_store(st);  // (ONE inlined copy: the loop body is large already)
  };
  // A block after the loop: rows >= r_end and invalid columns are staged as zero
  auto store_tail = [&](f32x4 (&st)[8], int64_t r_end) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const bool rok = 8 * rg + r < r_end;
#pragma unroll
      for (int e = 0; e < 4; ++e) st[r][e] = (rok && cok[e]) ? st[r][e] : 0.f;
    }
    convert_store(st);
  };
  if (vec && !cok[0]) {  // this thread's slots stay zero for the whole loop
    const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<u32x4*>(wbase + k * kCPlane + cc * kCLD) = zero;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // wave-uniform: this wave's column ranges are not empty (columns past N / K inside a range are
  // staged as zeros, so all four 32 x 32 blocks are computed)
  const bool active = n0 + wn * 64 < p.N && k0 + wk * 64 < Kop;
  // fragment of step s (rows 16 s + 8 lh .. + 7), term k, 32-column block i: one 16-byte read
  const uint32_t* const ga = planes + (wn * 64 + li) * kCLD + 4 * lh;
  const uint32_t* const xb = planes + 3 * kCPlane + (wk * 64 + li) * kCLD + 4 * lh;
  auto frag = [&](const uint32_t* base, int i, int s) {
    SplitFrag f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      f.p[k] = __builtin_bit_cast(
          bf16x8, *reinterpret_cast<const u32x4*>(base + k * kCPlane + i * 32 * kCLD + 8 * s));
    return f;
  };
  auto products = [&]() {
    if constexpr ((PROBE & 2) != 0) return;
    if (active) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const SplitFrag a0 = frag(ga, 0, s), a1 = frag(ga, 1, s);
        const SplitFrag b0 = frag(xb, 0, s), b1 = frag(xb, 1, s);
#pragma unroll
        for (int k = 0; k < kSplitTerms; ++k) {
          split_term(k, a0, b0, acc[0][0]);
          split_term(k, a0, b1, acc[0][1]);
          split_term(k, a1, b0, acc[1][0]);
          split_term(k, a1, b1, acc[1][1]);
        }
      }
    }
  };

  const int64_t n_rows = rb > ra ? rb - ra : 0;
  // blocks of the pipelined loops: the full ones — minus the one that holds the operands' last
  // row when some operand is read element-wise (see load_rows)
  constexpr bool exact_tail = !(VG && VX);
  const int64_t n_blocks = (exact_tail && rb == p.M && n_rows > 0) ? (n_rows - 1) / kWRows
                                                                    : n_rows / kWRows;
  // the pipelined loop takes an EVEN number of blocks (group A: block 2 i, group B: block 2 i + 1:
  // both always have one); what is left — at most three blocks' worth of rows — follows it
  const int64_t n_pipe = n_blocks & ~static_cast<int64_t>(1);
  const int64_t n_iter = n_pipe / 2;
  const int64_t last_blk = n_pipe - 1;
  auto blk_of = [&](int64_t i) {  // this group's block of iteration i, clamped (re-loads)
    const int64_t bi = 2 * i + grp;
    return bi < last_blk ? bi : last_blk;
  };
  // Two register sets per thread: the loads of iteration i + 1 are issued at the start of
  // iteration i, by BOTH groups at the same point of the code — before the role branches — and
  // unconditionally (prefetches past the end re-load the last block): s_waitcnt vmcnt counts
  // loads in issue order, so every path into a wait must have issued the same loads in the same
  // order, or the compiler has to wait for the NEWEST batch (measured: with a predicated load, or
  // with the two roles as two separate loops, the prefetch collapses to none).
  //   interval:   2 i                      2 i + 1
  //   group A     convert block 2 i        multiply block 2 i
  //   group B     multiply block 2 i - 1   convert block 2 i + 1
  f32x4 sta[8], stb[8];
  if constexpr ((PROBE & 8) != 0) {
#pragma unroll
    for (int r = 0; r < 8; ++r) sta[r] = stb[r] = f32x4{1.f, 2.f, 3.f, 4.f};
  }
  const bool is_a = grp == 0;  // scalar
  if (n_iter > 0) load_rows(sta, blk_of(0));
  for (int64_t i = 0; i + 1 < n_iter; i += 2) {
    load_rows(stb, blk_of(i + 1));
    if (is_a) store_rows(sta); else if (i > 0) products();
    __syncthreads();
    if (is_a) products(); else store_rows(sta);
    __syncthreads();
    load_rows(sta, blk_of(i + 2));
    if (is_a) store_rows(stb); else products();
    __syncthreads();
    if (is_a) products(); else store_rows(stb);
    __syncthreads();
  }
  if (n_iter & 1) {
    if (is_a) store_rows(sta); else if (n_iter > 1) products();
    __syncthreads();
    if (is_a) products(); else store_rows(sta);
    __syncthreads();
  }
  if (!is_a && n_iter > 0) products();  // B's last block
  // the rows behind the pipelined blocks (an odd full block, the last split's ragged block, the
  // exactly addressed last block of an element-wise operand): group A, 32 rows at a time
  for (int64_t r0 = n_pipe * kWRows; r0 < n_rows; r0 += kWRows) {  // (workgroup-uniform)
    if (is_a) {
      load_tail(sta, ra + r0);
      const int64_t left = n_rows - r0;
      store_tail(sta, left < kWRows ? left : kWRows);
    }
    __syncthreads();
    if (is_a) products();
    __syncthreads();
  }
  // B's sums and both groups' bias-gradient partials go to A through LDS
  __syncthreads();  // every product is done: the images are free
  float* const cmb = reinterpret_cast<float*>(tn_lds);          // [4 waves][64][64 lanes]
  float* const red = cmb + 4 * 64 * 64;                         // [2 groups][4 row groups][128]
  if (grp == 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          cmb[((wave * 4 + i * 2 + j) * 16 + e) * 64 + lane] = acc[i][j][e];
  }
  if (do_colsum && !isx)
    *reinterpret_cast<f32x4*>(&red[(grp * 4 + rg) * kWTile + tc]) = csum;
  __syncthreads();
  if (grp == 1) return;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        acc[i][j][e] += cmb[((wave * 4 + i * 2 + j) * 16 + e) * 64 + lane];
  if (do_colsum && threadIdx.x < kWTile && n0 + static_cast<int>(threadIdx.x) < p.N) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r * kWTile + threadIdx.x];
    p.colsum[split * p.N + n0 + threadIdx.x] = s;
  }
  float* __restrict__ slab = p.partial + split * static_cast<int64_t>(p.N) * p.K;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (k0 + wk * 64 + j * 32 + li >= Kop) continue;
      const int col = kout0 + wk * 64 + j * 32 + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = n0 + wn * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row < p.N) slab[static_cast<int64_t>(row) * p.K + col] = acc[i][j][e];
      }
    }
  }
}

// ---- TN, split arithmetic, NARROW g (N <= 96): the whole [N, 256-column] tile per workgroup --------
// The last layer of a classifier stack (47 classes, aggregated at the output width: g = [A^T g' |
// g'] is 96 columns wide) multiplies a narrow g against a wide x.  Under gemm_tn_split_kernel that
// launch is bound by its LOADS, not by its products (profiles/r06_wgrad_probe.txt: halving the
// matrix work moved nothing): a 128 x 128 tile re-reads g once per x tile, a quarter of the g
// loads fetch clamped padding columns, and two register sets per thread are all the 64-register
// accumulators leave room for.  Here ONE 512-thread workgroup per CU owns every g column and 256
// x columns: a wave owns one 32-column block of x against all NB <= 3 blocks of g (16 NB
// accumulator registers), every operand row leaves L2 once, and THREE register sets per staging
// thread keep three 32-row blocks (45 KB at N = 96) in flight per CU.  One LDS image (the column
// planes of gemm_tn_split_kernel: [term][column][20 dwords]), two barriers per block; conversion
// and products alternate — both together are shorter than the block's share of HBM time.
constexpr int kSkG = 128;                 // staged g columns (N <= 128)
constexpr int kSkX = 256;                 // x columns per workgroup
constexpr int kSkCols = kSkG + kSkX;
constexpr int kSkPlane = kSkCols * kCLD;  // dwords per term plane
constexpr size_t kTnSkinnyLds = sizeof(uint32_t) * 3 * kSkPlane;
// (N in 97..128 — four blocks of g — and a WIDE form with 256 columns of g + 128 of x per workgroup
// were built too and measured no faster than the tiled kernel, whose two wave groups overlap
// conversion and products: profiles/r06_wgrad_probe.txt)
constexpr int kSkMaxN = 96;
constexpr int64_t kSkinnyMinRows = 32768;  // below: the tiled kernel with its finer splits

// PROBE (lab, timing only — results undefined): bit 1 = no products, bit 2 = no conversion / LDS
// stores, bit 3 = no global loads
template <int NB, int PROBE = 0>
__global__ void __launch_bounds__(kTnThreads, 1) gemm_tn_skinny_kernel(GemmTN p) {
  extern __shared__ __align__(16) uint32_t sk_lds[];
  const int64_t b = blockIdx.x;
  const int64_t per_xcd = gridDim.x >> 3;
  const int64_t q = (b & 7) * per_xcd + (b >> 3);
  const int64_t split = q / p.tiles_k;
  if (split >= p.splits) return;
  const int tk = static_cast<int>(q - split * p.tiles_k);
  const int k0 = tk * kSkX;
  const int64_t ra = split * p.rows_per_split;
  int64_t rb = ra + p.rows_per_split;
  rb = rb < p.M ? rb : p.M;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;

  // staging: waves 0, 1 stage g (tile columns 0..63, 64..127), waves 2..5 stage x (64 columns
  // each); waves 6, 7 stage g ONCE MORE (the same values to the same LDS words: a benign
  // duplicate, their loads hit the CU's L1): every wave runs the same instruction stream — a
  // wave-uniform branch around the staging costs the prefetch (the s_waitcnt pass merges the two
  // paths at every join and loses one batch of distance per join).  Lane bits as in
  // gemm_tn_split_kernel: 0 = low bit of the column quad, 1..2 = row group (8 rows), 3..5 = high
  // bits of the quad.
  const int sw = wave < 6 ? wave : wave - 6;
  const bool isx = sw >= 2;
  const int c4 = (lane & 1) + 2 * (lane >> 3);
  const int rg = (lane >> 1) & 3;
  const int sc = 64 * sw + 4 * c4;       // staged column (0..383)
  const int oc = isx ? sc - kSkG : sc;                     // column inside the operand's tile
  const float* __restrict__ src = isx ? p.x : p.g;
  const int64_t sld = isx ? p.ldx : p.ldg;
  const int col_lim = isx ? p.K - k0 : p.N;                // valid columns (> 0, % 4 == 0)
  const int col0 = isx ? k0 : 0;
  const bool cok = oc < col_lim;                           // the whole quad is valid or not
  const float* const tp = src + (ra + 8 * rg) * sld + col0 + (cok ? oc : 0);
  auto load_rows = [&](f32x4 (&st)[8], int64_t blk) {
    if constexpr ((PROBE & 8) != 0) return;
    const float* bp = tp + blk * kWRows * sld;
#pragma unroll
    for (int r = 0; r < 8; ++r) st[r] = *reinterpret_cast<const f32x4*>(bp + r * sld);
  };
  auto load_tail = [&](f32x4 (&st)[8], int64_t r0) {  // rows clamped to the split's last one
    const int64_t last = rb - 1;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      int64_t row = r0 + 8 * rg + r;
      row = row < last ? row : last;
      st[r] = *reinterpret_cast<const f32x4*>(src + row * sld + col0 + (cok ? oc : 0));
    }
  };
  const bool do_colsum = p.colsum != nullptr && tk == 0;  // workgroup-uniform
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  uint32_t* const wbase = sk_lds + sc * kCLD + 4 * rg;
  auto convert_store = [&](f32x4 (&st)[8]) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      u32x4 w[3];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        uint32_t tt[3];
        split_pair_single(st[2 * qq][cc], st[2 * qq + 1][cc], tt);
        w[0][qq] = tt[0];
        w[1][qq] = tt[1];
        w[2][qq] = tt[2];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k)
        *reinterpret_cast<u32x4*>(wbase + k * kSkPlane + cc * kCLD) = w[k];
    }
    if (do_colsum && wave < 2) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[e] = add_f32_asm(csum[e], st[r][e]);
      }
    }
  };
  // a full block inside the loop: unconditional (a lane-divergent branch around the conversion
  // makes the compiler drain every load at the join); padding quads are staged as zeros
  auto store_rows = [&](f32x4 (&st)[8]) {
    if constexpr ((PROBE & 4) != 0) return;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int e = 0; e < 4; ++e) st[r][e] = cok ? st[r][e] : 0.f;
    }
    convert_store(st);
  };
  auto store_tail = [&](f32x4 (&st)[8], int64_t r_end) {  // rows >= r_end are staged as zero
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const bool rok = 8 * rg + r < r_end;
#pragma unroll
      for (int e = 0; e < 4; ++e) st[r][e] = (rok && cok) ? st[r][e] : 0.f;
    }
    convert_store(st);
  };

  f32x16 acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const bool active = k0 + wave * 32 < p.K;  // scalar: this wave's x block is not all padding
  const uint32_t* const ga = sk_lds + li * kCLD + 4 * lh;
  const uint32_t* const xb = sk_lds + (kSkG + wave * 32 + li) * kCLD + 4 * lh;
  auto frag = [&](const uint32_t* base, int s) {
    SplitFrag f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      f.p[k] = __builtin_bit_cast(bf16x8,
                                  *reinterpret_cast<const u32x4*>(base + k * kSkPlane + 8 * s));
    return f;
  };
  auto products = [&]() {
    if constexpr ((PROBE & 2) != 0) return;
    if (active) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const SplitFrag bx = frag(xb, s);
        SplitFrag a[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) a[i] = frag(ga + i * 32 * kCLD, s);
#pragma unroll
        for (int k = 0; k < kSplitTerms; ++k)
#pragma unroll
          for (int i = 0; i < NB; ++i) split_term(k, a[i], bx, acc[i]);
      }
    }
  };

  const int64_t n_rows = rb > ra ? rb - ra : 0;
  const int64_t n_blocks = n_rows / kWRows;          // full blocks
  const int64_t n_pipe = n_blocks - n_blocks % 3;    // the pipelined loop takes them three at a time
  const int64_t last_blk = n_pipe - 1;
  auto blk_of = [&](int64_t i) { return i < last_blk ? i : last_blk; };  // (re-loads past the end)
  f32x4 st0[8], st1[8], st2[8];
  if constexpr ((PROBE & 8) != 0) {
#pragma unroll
    for (int r = 0; r < 8; ++r) st0[r] = st1[r] = st2[r] = f32x4{1.f, 2.f, 3.f, 4.f};
  }
  // (the three batches in program order, here and in the loop: s_waitcnt counts loads in issue
  // order and merges the paths into the loop header pessimistically)
  if (n_pipe > 0) {
    load_rows(st0, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_rows(st1, 1);
    __builtin_amdgcn_sched_barrier(0);
    load_rows(st2, 2);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int64_t i = 0; i < n_pipe; i += 3) {
    store_rows(st0);
    load_rows(st0, blk_of(i + 3));
    __syncthreads();
    products();
    __syncthreads();
    store_rows(st1);
    load_rows(st1, blk_of(i + 4));
    __syncthreads();
    products();
    __syncthreads();
    store_rows(st2);
    load_rows(st2, blk_of(i + 5));
    __syncthreads();
    products();
    __syncthreads();
  }
  // what the pipelined loop left: up to two full blocks and the split's ragged last block
  for (int64_t r0 = n_pipe * kWRows; r0 < n_rows; r0 += kWRows) {  // (workgroup-uniform)
    load_tail(st0, ra + r0);
    const int64_t left = n_rows - r0;
    store_tail(st0, left < kWRows ? left : kWRows);
    __syncthreads();
    products();
    __syncthreads();
  }
  if (do_colsum) {  // (uniform: the barrier is legal; the image is free after the last products)
    float* const red = reinterpret_cast<float*>(sk_lds);  // [4 row groups][128]
    if (wave < 2) *reinterpret_cast<f32x4*>(&red[rg * kSkG + sc]) = csum;
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < p.N) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) s += red[r * kSkG + threadIdx.x];
      p.colsum[split * p.N + threadIdx.x] = s;
    }
  }
  if (!active) return;
  float* __restrict__ slab = p.partial + split * static_cast<int64_t>(p.N) * p.K;
  const int col = k0 + wave * 32 + li;
  if (col >= p.K) return;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      if (row < p.N) slab[static_cast<int64_t>(row) * p.K + col] = acc[i][e];
    }
  }
}

// out[n, k] (+)= sum over splits, in split order (deterministic)
__global__ void __launch_bounds__(kBlock)
    gemm_tn_reduce_kernel(const float* __restrict__ partial, int splits, int64_t NK, int K,
                          float* __restrict__ out, int64_t ldo, int accumulate,
                          const float* __restrict__ colsum, int N, float* __restrict__ bias_grad) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= NK) {  // the tail threads add up the bias-gradient partials
    const int64_t n = t - NK;
    if (bias_grad && n < N) {
      float s = 0.f;
      int sp = 0;
      for (; sp + 8 <= splits; sp += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = colsum[static_cast<int64_t>(sp + i) * N + n];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
      }
      for (; sp < splits; ++sp) s += colsum[static_cast<int64_t>(sp) * N + n];
      bias_grad[n] = s;
    }
    return;
  }
  // (sixteen slabs' loads in flight, added in split order: the plain loop waits for every load
  // before it issues the next one — 64 splits were 64 trips to L2, 18 us for 16.8 MB)
  float s = 0.f;
  int sp = 0;
  for (; sp + 16 <= splits; sp += 16) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = partial[(sp + i) * NK + t];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
  }
  for (; sp + 4 <= splits; sp += 4) {
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = partial[(sp + i) * NK + t];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += v[i];
  }
  for (; sp < splits; ++sp) s += partial[sp * NK + t];
  const int64_t r = t / K;
  float* dst = out + r * ldo + (t - r * K);
  *dst = accumulate ? *dst + s : s;
}

static bool aligned16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int WM, int WN, int TM, int TN>
static int launch_nt(GemmNT p, bool vec, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  p.tiles_m = static_cast<int>(ceil_div(p.M, BM));
  p.tiles_n = static_cast<int>(ceil_div(p.N, BN));
  const int64_t blocks = round_up(static_cast<int64_t>(p.tiles_m) * p.tiles_n, 8);
  const size_t lds = sizeof(float) * 2 * (BM + BN) * kGLD;
  // (the opt-in to > 64 KiB of dynamic LDS is per kernel and per device: set it on every launch,
  // it is a host-side attribute write)
  void (*k)(GemmNT) = nullptr;
  if (p.split) {
    k = vec ? gemm_nt_kernel<WM, WN, TM, TN, true, true>
            : gemm_nt_kernel<WM, WN, TM, TN, false, true>;
  } else {
    k = vec ? gemm_nt_kernel<WM, WN, TM, TN, true, false>
            : gemm_nt_kernel<WM, WN, TM, TN, false, false>;
  }
  PYGAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds)));
  const unsigned gy = static_cast<unsigned>(p.ksplits > 1 ? p.ksplits : 1);
  hipLaunchKernelGGL(k, dim3(static_cast<unsigned>(blocks), gy), dim3(kBlock), lds, st, p);
  PYGAMD_LAUNCH_CHECK();
  if (p.ksplits > 1) {
    const int64_t MN = p.M * p.N;
    hipLaunchKernelGGL(gemm_nt_splitk_epilogue, dim3(static_cast<unsigned>(ceil_div(MN, kBlock))),
                       dim3(kBlock), 0, st, p);
    PYGAMD_LAUNCH_CHECK();
  }
  return PYGAMD_OK;
}

static int g_gemm_mode = 0;  // pygamd_set_gemm_mode
#ifdef PYGAMD_LAB
static int g_wgrad_variant = 0;  // pygamd_lab_set_wgrad_variant (libpyg_amd_lab.so only)
#else
constexpr int g_wgrad_variant = 0;  // the product library has the production schedule only
#endif

// Shape of an NT launch: the tile and the number of K slices.  Outputs with >= kFillTiles tiles of
// the large shape fill the chip by their rows alone (the full-batch layers); below that — sampled
// blocks of 1 k .. 16 k rows, Cora-sized inputs — smaller tiles and, if the workspace allows,
// slices of the reduction bring the launch back to about one workgroup per CU.
constexpr int kFillTiles = 128;
struct NtShape {
  int tile;     // 0: 128x128, 1: 128x96, 2: 128x64, 3: 128x32, 4: 64x64
  int ksplits;
  int k_per_split;
};
static NtShape nt_shape(int64_t M, int N, int K, bool allow_splitk) {
  NtShape s;
  s.tile = N > 96 ? 0 : N > 64 ? 1 : N > 32 ? 2 : 3;
  s.ksplits = 1;
  s.k_per_split = K;
  const int bn[5] = {128, 96, 64, 32, 64};
  int64_t tiles = ceil_div(M, 128) * ceil_div(N, bn[s.tile]);
  if (tiles >= kFillTiles) return s;
  if (N > 32) {  // 64 x 64: four waves, one 32 x 32 block each
    s.tile = 4;
    tiles = ceil_div(M, 64) * ceil_div(N, 64);
  }
  if (allow_splitk && tiles < kFillTiles && K >= 128) {
    int want = static_cast<int>(ceil_div(2 * kFillTiles, tiles));
    const int most = K / 64;  // a slice walks at least two chunks
    want = want < most ? want : most;
    want = want < 32 ? want : 32;
    if (want > 1) {
      s.k_per_split = static_cast<int>(round_up(ceil_div(K, want), kGK));
      s.ksplits = static_cast<int>(ceil_div(K, s.k_per_split));
    }
  }
  return s;
}

static int run_nt(GemmNT p, void* workspace, size_t workspace_bytes, hipStream_t st) {
  if (p.M == 0 || p.N == 0) return PYGAMD_OK;
  p.split = g_gemm_mode == PYGAMD_GEMM_SPLIT_BF16 ? 1 : 0;
  const bool vec = (p.K % 4 == 0) && (p.lda % 4 == 0) && (p.ldb % 4 == 0) && aligned16p(p.a) &&
                   aligned16p(p.b);
  // tile shape by output width: wide outputs 128 x 128; 65..96 columns one 128 x 96 tile row;
  // narrow outputs 128 x 64 / 128 x 32 tiles (all four waves stacked along M); few tiles: 64 x 64
  // and slices of K (nt_shape).  Without a workspace for the slices the launch runs unsliced.
  NtShape sh = nt_shape(p.M, p.N, p.K, workspace != nullptr);
  if (sh.ksplits > 1 &&
      workspace_bytes < static_cast<size_t>(sh.ksplits) * p.M * p.N * sizeof(float))
    sh = nt_shape(p.M, p.N, p.K, false);
  p.ksplits = sh.ksplits;
  p.k_per_split = sh.k_per_split;
  p.partial = sh.ksplits > 1 ? static_cast<float*>(workspace) : nullptr;
  switch (sh.tile) {
    case 0: return launch_nt<2, 2, 2, 2>(p, vec, st);
    case 1: return launch_nt<4, 1, 1, 3>(p, vec, st);
    case 2: return launch_nt<4, 1, 1, 2>(p, vec, st);
    case 3: return launch_nt<4, 1, 1, 1>(p, vec, st);
    default: return launch_nt<2, 2, 1, 1>(p, vec, st);
  }
}

}  // namespace pygamd

using namespace pygamd;

extern "C" {

int pygamd_set_gemm_mode(int mode) {
  if (mode != PYGAMD_GEMM_FP32 && mode != PYGAMD_GEMM_SPLIT_BF16) return PYGAMD_ERR_INVALID_ARG;
  g_gemm_mode = mode;
  return PYGAMD_OK;
}

int pygamd_get_gemm_mode(void) { return g_gemm_mode; }

#ifdef PYGAMD_LAB
int pygamd_lab_set_wgrad_variant(int variant) {
  if (variant < 0 || variant > 127 || ((variant & 1) && variant != 1))
    return PYGAMD_ERR_INVALID_ARG;
  g_wgrad_variant = variant;
  return PYGAMD_OK;
}
#endif

int pygamd_linear_forward(const float* x, int64_t ldx, const float* w, int64_t ldw,
                          const float* bias, int64_t M, int64_t K, int64_t N, int relu,
                          int accumulate, float* out, int64_t ldo, void* workspace,
                          size_t workspace_bytes, void* stream) {
  if (M < 0 || K < 0 || N < 0 || K > INT32_MAX || N > INT32_MAX || ldx < K || ldw < K ||
      ldo < N)
    return PYGAMD_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return PYGAMD_OK;
  if (!out || (K > 0 && (!x || !w))) return PYGAMD_ERR_INVALID_ARG;
  GemmNT p = {};
  p.a = x; p.b = w; p.bias = bias; p.row_scale = nullptr; p.mask = nullptr; p.c = out;
  p.M = M; p.lda = ldx; p.ldb = ldw; p.ldc = ldo; p.ldm = 0;
  p.N = static_cast<int>(N); p.K = static_cast<int>(K);
  p.relu = relu ? 1 : 0; p.n_scaled = 0; p.accumulate = accumulate ? 1 : 0;
  return run_nt(p, workspace, workspace_bytes, as_stream(stream));
}

int pygamd_linear_dgrad2(const float* g, int64_t ldg, const float* w_t, int64_t ldwt,
                         const float* row_scale, int64_t n_scaled, int64_t M, int64_t N,
                         int64_t K, int accumulate, const float* relu_mask, int64_t ld_mask,
                         const uint32_t* relu_bits, int64_t ld_bits, float* out, int64_t ldo,
                         float* out_scaled, int64_t ld_scaled, void* workspace,
                         size_t workspace_bytes, void* stream) {
  // out[M, K] = g[M, N] @ w[N, K], with w given TRANSPOSED as w_t[K, N]: the same NT kernel
  if (M < 0 || K < 0 || N < 0 || K > INT32_MAX || N > INT32_MAX || ldg < N || ldwt < N ||
      ldo < K || n_scaled < 0 || n_scaled > K || (relu_mask && ld_mask < K) ||
      (relu_bits && (relu_mask || ld_bits < (K + 31) / 32)) ||
      (out_scaled && (ld_scaled < K || accumulate)))
    return PYGAMD_ERR_INVALID_ARG;
  if (M == 0 || K == 0) return PYGAMD_OK;
  if (!out || (N > 0 && (!g || !w_t)) || ((n_scaled > 0 || out_scaled) && !row_scale))
    return PYGAMD_ERR_INVALID_ARG;
  GemmNT p = {};
  p.a = g; p.b = w_t; p.bias = nullptr; p.row_scale = row_scale; p.mask = relu_mask;
  p.c = out;
  p.c2 = out_scaled; p.ldc2 = ld_scaled;
  p.M = M; p.lda = ldg; p.ldb = ldwt; p.ldc = ldo; p.ldm = ld_mask;
  p.mask_bits = relu_bits; p.ldmb = ld_bits;
  p.N = static_cast<int>(K); p.K = static_cast<int>(N);
  p.relu = 0; p.n_scaled = static_cast<int>(n_scaled); p.accumulate = accumulate ? 1 : 0;
  return run_nt(p, workspace, workspace_bytes, as_stream(stream));
}

int pygamd_linear_dgrad(const float* g, int64_t ldg, const float* w_t, int64_t ldwt,
                        const float* row_scale, int64_t n_scaled, int64_t M, int64_t N,
                        int64_t K, int accumulate, const float* relu_mask, int64_t ld_mask,
                        const uint32_t* relu_bits, int64_t ld_bits, float* out, int64_t ldo,
                        void* stream) {
  return pygamd_linear_dgrad2(g, ldg, w_t, ldwt, row_scale, n_scaled, M, N, K, accumulate,
                              relu_mask, ld_mask, relu_bits, ld_bits, out, ldo, nullptr, 0, nullptr,
                              0, stream);
}

int pygamd_linear_nt_workspace_bytes(int64_t M, int64_t N_out, int64_t K_red, size_t* bytes) {
  if (!bytes || M < 0 || N_out < 0 || K_red < 0 || N_out > INT32_MAX || K_red > INT32_MAX)
    return PYGAMD_ERR_INVALID_ARG;
  const NtShape sh = nt_shape(M, static_cast<int>(N_out), static_cast<int>(K_red), true);
  *bytes = sh.ksplits > 1 ? static_cast<size_t>(sh.ksplits) * M * N_out * sizeof(float) : 0;
  return PYGAMD_OK;
}

static int64_t wgrad_splits(int64_t M, int64_t tiles, int wgs_per_cu = 2) {
  // two workgroups per CU (the LDS ring allows no more) = one full wave of workgroups, every
  // split a multiple of 32 rows and at least 1024 rows (128 below 64 k rows).  wgs_per_cu = 1: half the footprint — the
  // launch then shares the chip with a bandwidth-bound kernel on another stream (the matrix cores
  // are idle under an SpMM) instead of taking every wave slot.
  int64_t s = ceil_div(256 * (wgs_per_cu == 1 ? 1 : 2), tiles < 1 ? 1 : tiles);
  // (sampled blocks of a few thousand rows: splits of 256 / 64 rows keep the chip busy; their
  // slabs are a few MB)
  // (1024 rather than 2048 rows from 64 k rows: a sampled batch's first layer — 170 k rows, four
  // tiles — then fills all 256 CUs instead of 168 of them; the full-batch shapes are far from it)
  const int64_t max_s = ceil_div(M, M >= 65536 ? 1024 : M >= 8192 ? 128 : 64);
  s = s > max_s ? max_s : s;
  return s < 1 ? 1 : s;
}

int pygamd_linear_wgrad_workspace_bytes(int64_t M, int64_t N, int64_t K, size_t* bytes) {
  if (!bytes || M < 0 || N < 0 || K < 0) return PYGAMD_ERR_INVALID_ARG;
  // (a two-operand launch has at most one more column tile: fewer splits, never more)
  const int64_t tiles = ceil_div(N, kWTile) * ceil_div(K, kWTile);
  // [splits][N][K] partial tiles + [splits][N] bias-gradient partials
  *bytes = static_cast<size_t>(wgrad_splits(M, tiles)) * static_cast<size_t>(N) *
           (static_cast<size_t>(K) + 1) * sizeof(float);
  return PYGAMD_OK;
}

int pygamd_linear_wgrad(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t M,
                        int64_t N, int64_t K, int accumulate, int wgs_per_cu, float* out,
                        int64_t ldo, float* bias_grad, void* workspace, size_t workspace_bytes,
                        void* stream) {
  return pygamd_linear_wgrad2(g, ldg, x, ldx, K, nullptr, 0, 0, M, N, accumulate, wgs_per_cu, out,
                              ldo, bias_grad, workspace, workspace_bytes, stream);
}

int pygamd_linear_wgrad2(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t K1,
                         const float* x2, int64_t ldx2, int64_t K2, int64_t M, int64_t N,
                         int accumulate, int wgs_per_cu, float* out, int64_t ldo,
                         float* bias_grad, void* workspace, size_t workspace_bytes,
                         void* stream) {
  if (K1 < 0 || K2 < 0 || (K2 > 0 && (K1 == 0 || ldx2 < K2 || (M > 0 && !x2))))
    return PYGAMD_ERR_INVALID_ARG;
  const int64_t K = K1 + K2;
  if (M < 0 || K < 0 || N < 0 || K > INT32_MAX || N > INT32_MAX || ldg < N || ldx < K1 ||
      ldo < K)
    return PYGAMD_ERR_INVALID_ARG;
  if (N == 0 || K == 0) return PYGAMD_OK;
  if (!out || (M > 0 && (!g || !x))) return PYGAMD_ERR_INVALID_ARG;
  size_t need = 0;
  pygamd_linear_wgrad_workspace_bytes(M, N, K, &need);
  if (!workspace || workspace_bytes < need) return PYGAMD_ERR_WORKSPACE;
  hipStream_t st = as_stream(stream);
  GemmTN p = {};
  p.g = g; p.x = x; p.partial = static_cast<float*>(workspace);
  p.M = M; p.ldg = ldg; p.ldx = ldx;
  p.N = static_cast<int>(N); p.K = static_cast<int>(K);
  p.x2 = x2; p.ldx2 = ldx2; p.K1 = static_cast<int>(K1);
  p.tiles_n = static_cast<int>(ceil_div(N, kWTile));
  p.tiles_k1 = static_cast<int>(ceil_div(K1, kWTile));
  p.tiles_k = p.tiles_k1 + static_cast<int>(ceil_div(K2, kWTile));
  const int64_t tiles = static_cast<int64_t>(p.tiles_n) * p.tiles_k;
  // the production split kernel: one 512-thread workgroup per CU (two wave groups share a
  // split): half as many splits for the same number of waves
  const bool split_once = g_gemm_mode == PYGAMD_GEMM_SPLIT_BF16 && !(g_wgrad_variant & 1);
  p.splits = static_cast<int>(wgrad_splits(M, tiles, wgs_per_cu));
  if (split_once) p.splits = (p.splits + 1) / 2;
  p.colsum = bias_grad ? p.partial + static_cast<int64_t>(p.splits) * N * K : nullptr;
  p.rows_per_split = round_up(ceil_div(M > 0 ? M : 1, p.splits), kWRows);
  const bool vec = (N % 4 == 0) && (K1 % 4 == 0) && (K2 % 4 == 0) && (ldg % 4 == 0) &&
                   (ldx % 4 == 0) && (ldx2 % 4 == 0) && aligned16p(g) && aligned16p(x) &&
                   aligned16p(x2);
  // (the split kernel decides per operand: a 47-column g next to 256-column x's keeps the wide
  // loads for the x's)
  p.vec_g = (N % 4 == 0) && (ldg % 4 == 0) && aligned16p(g);
  p.vec_x = (K1 % 4 == 0) && (ldx % 4 == 0) && aligned16p(x);
  p.vec_x2 = (K2 % 4 == 0) && (ldx2 % 4 == 0) && aligned16p(x2);
  // a narrow g against one wide x (the classifier layer aggregated at the output width): every g
  // column and 256 x columns per workgroup, one workgroup per CU (gemm_tn_skinny_kernel)
  // (lab switch 16: the tiled kernel for this shape too — the A/B of scripts/wgrad_probe.py)
  if (split_once && !(g_wgrad_variant & 16) && !x2 && N <= kSkMaxN && p.vec_g && p.vec_x &&
      M >= kSkinnyMinRows) {
    const int64_t ws_splits = wgrad_splits(M, tiles);  // what the workspace was sized for
    p.tiles_k = static_cast<int>(ceil_div(K, kSkX));
    int64_t s = ceil_div(256, p.tiles_k);
    s = s > ws_splits ? ws_splits : s;
    p.splits = static_cast<int>(s);
    p.colsum = bias_grad ? p.partial + static_cast<int64_t>(p.splits) * N * K : nullptr;
    p.rows_per_split = round_up(ceil_div(M, p.splits), kWRows);
    void (*sk)(GemmTN) = N <= 32 ? gemm_tn_skinny_kernel<1>
                         : N <= 64 ? gemm_tn_skinny_kernel<2>
                                   : gemm_tn_skinny_kernel<3>;
#ifdef PYGAMD_LAB
    if (N > 64 && N <= 96) {  // timing probes of the three-block variant (64 + phase bits)
      switch (g_wgrad_variant) {
        case 66: sk = gemm_tn_skinny_kernel<3, 2>; break;
        case 68: sk = gemm_tn_skinny_kernel<3, 4>; break;
        case 70: sk = gemm_tn_skinny_kernel<3, 6>; break;
        case 72: sk = gemm_tn_skinny_kernel<3, 8>; break;
        case 74: sk = gemm_tn_skinny_kernel<3, 10>; break;
        case 76: sk = gemm_tn_skinny_kernel<3, 12>; break;
        case 78: sk = gemm_tn_skinny_kernel<3, 14>; break;
        default: break;
      }
    }
#endif
    PYGAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sk),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(kTnSkinnyLds)));
    const int64_t wgs = round_up(static_cast<int64_t>(p.tiles_k) * p.splits, 8);
    hipLaunchKernelGGL(sk, dim3(static_cast<unsigned>(wgs)), dim3(kTnThreads), kTnSkinnyLds, st, p);
    PYGAMD_LAUNCH_CHECK();
    const int64_t NKs = N * K;
    hipLaunchKernelGGL(gemm_tn_reduce_kernel,
                       dim3(static_cast<unsigned>(ceil_div(NKs + (bias_grad ? N : 0), kBlock))),
                       dim3(kBlock), 0, st, p.partial, p.splits, NKs, p.K, out, ldo,
                       accumulate ? 1 : 0, p.colsum, p.N, bias_grad);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  }
  const int64_t blocks = round_up(tiles * p.splits, 8);
  size_t lds = sizeof(float) * 4 * kWRows * kWLD;
  int threads = kBlock;
  void (*kern)(GemmTN) = nullptr;
  if (split_once) {
    const bool vx = p.vec_x && p.vec_x2;
    kern = p.vec_g ? (vx ? gemm_tn_split_kernel<true, true> : gemm_tn_split_kernel<true, false>)
                   : (vx ? gemm_tn_split_kernel<false, true> : gemm_tn_split_kernel<false, false>);
#ifdef PYGAMD_LAB
    switch (g_wgrad_variant) {  // timing probes of the all-vector variant
      case 2: kern = gemm_tn_split_kernel<true, true, 2>; break;
      case 4: kern = gemm_tn_split_kernel<true, true, 4>; break;
      case 6: kern = gemm_tn_split_kernel<true, true, 6>; break;
      case 8: kern = gemm_tn_split_kernel<true, true, 8>; break;
      case 10: kern = gemm_tn_split_kernel<true, true, 10>; break;
      case 12: kern = gemm_tn_split_kernel<true, true, 12>; break;
      case 14: kern = gemm_tn_split_kernel<true, true, 14>; break;
      case 32: kern = gemm_tn_split_kernel<true, true, 32>; break;
      case 40: kern = gemm_tn_split_kernel<true, true, 40>; break;
      default: break;
    }
#endif
    lds = kTnSplitLds;
    threads = kTnThreads;
#ifdef PYGAMD_LAB
  } else if (g_gemm_mode == PYGAMD_GEMM_SPLIT_BF16) {  // round 3: operands split in registers
    kern = vec ? gemm_tn_kernel<true, true> : gemm_tn_kernel<false, true>;
#endif
  } else {
    kern = vec ? gemm_tn_kernel<true, false> : gemm_tn_kernel<false, false>;
  }
  PYGAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds)));
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(threads), lds, st, p);
  PYGAMD_LAUNCH_CHECK();
  const int64_t NK = N * K;
  hipLaunchKernelGGL(gemm_tn_reduce_kernel,
                     dim3(static_cast<unsigned>(ceil_div(NK + (bias_grad ? N : 0), kBlock))),
                     dim3(kBlock), 0, st, p.partial, p.splits, NK, p.K, out, ldo,
                     accumulate ? 1 : 0, p.colsum, p.N, bias_grad);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

}  // extern "C"
