// segmm.hip — segment_matmul (grouped GEMM over row segments) on the fp32 matrix cores.
//
//   out[ptr[g] : ptr[g+1]] = x[ptr[g] : ptr[g+1]] @ W[g]          (pyg_lib.ops.segment_matmul,
//   torch_geometric/nn/conv/rgcn_conv.py:288, nn/dense/linear.py:255)
//
// MFMA-bound (the one GEMM-shaped op on this path): v_mfma_f32_32x32x2_f32, exact fp32
// (bitwise an fmaf chain), fragment maps per cdna_hip_programming.md §3:
//   A: lane l holds A[i = l & 31][k = l >> 5];  B: lane l holds B[k = l >> 5][j = l & 31];
//   D: reg r of lane l is D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
// ONE launch covers every segment: the host builds a table of (group, first row, rows) tiles of
// 128 rows; a 256-thread workgroup (2 x 2 waves, 64 x 64 outputs = four accumulators each) owns a
// 128 x 128 output tile.  Both operands go through LDS in k chunks of 32 — every element is read
// from L2 once per workgroup (operands streamed straight from global made the kernel L1/L2
// bandwidth bound: ~10 TB/s at the MFMA rate) — and the next chunk is prefetched into registers
// while the matrix cores work on the current one (PMC of the unpipelined kernel: 64 % of the
// wave cycles parked in s_waitcnt / barrier, MFMA pipe 31 % busy).
#include "split_bf16.h"

namespace pygamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTM = 128;   // rows per tile
constexpr int kTN = 128;   // cols per workgroup
constexpr int kTK = 32;    // k chunk staged in LDS

__device__ __forceinline__ void zero_acc(f32x16& a) {
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = 0.f;
}

__global__ void __launch_bounds__(kBlock)
    segmm_kernel(const float* __restrict__ x, int64_t ldx, const int64_t* __restrict__ x_rows,
                 const float* __restrict__ w, int64_t w_seg_stride, int64_t w_sk, int64_t w_sn,
                 const int32_t* __restrict__ tiles, int K, int N, int blocks,
                 float* __restrict__ out, int64_t ldo) {
  __shared__ float As[kTM][kTK + 1];   // stride 33: the MFMA's column reads are conflict-free
  __shared__ float Bs[kTK][kTN + 4];   // a half-wave reads 32 consecutive floats of one row
  const int t = blockIdx.x;
  const int seg = tiles[3 * t];
  const int64_t row0 = tiles[3 * t + 1];
  const int rows = tiles[3 * t + 2];
  const int n0 = blockIdx.y * kTN;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const float* __restrict__ wseg = w + static_cast<int64_t>(seg) * w_seg_stride;
  // block-diagonal weights: group seg = relation * blocks + b works on column block b
  const int blk = blocks > 1 ? seg % blocks : 0;
  x += static_cast<int64_t>(blk) * K;
  out += static_cast<int64_t>(blk) * N;
  f32x16 acc00, acc01, acc10, acc11;   // [row half][col half] of this wave's 64 x 64
  zero_acc(acc00);
  zero_acc(acc01);
  zero_acc(acc10);
  zero_acc(acc11);
  // A tile: thread -> (k = tid % 32, rows tid / 32 + 8 j): every load instruction of a wave covers
  // two full 128-byte runs (a per-thread run of consecutive k would touch 64 lines per load)
  const int ak = threadIdx.x & 31;
  const int ar = threadIdx.x >> 5;
  // (x_rows: row r of the operand is x[x_rows[r]] — the rows of another tensor gathered on the fly)
  const float* __restrict__ xa = x + ak;
  // B tile: thread -> (k = tid / 128 + 2 j, col = tid % 128): 512-byte coalesced weight rows
  const int bc = threadIdx.x & (kTN - 1);
  const int bk = threadIdx.x >> 7;
  const bool bcol_ok = n0 + bc < N;
  const float* __restrict__ wb = wseg + static_cast<int64_t>(bcol_ok ? n0 + bc : 0) * w_sn;
  float a_nx[16], b_nx[16];
  int64_t arow[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int r = ar + 8 * q;
    const int64_t rr = row0 + (r < rows ? r : 0);
    arow[q] = x_rows ? x_rows[rr] : rr;
  }
  auto prefetch = [&](int k0) {
    const bool ak_ok = k0 + ak < K;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = ar + 8 * q;
      const bool ok = ak_ok && r < rows;
      a_nx[q] = ok ? xa[arow[q] * ldx + k0] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k = k0 + bk + 2 * j;
      const bool ok = bcol_ok && k < K;
      b_nx[j] = ok ? wb[static_cast<int64_t>(ok ? k : 0) * w_sk] : 0.f;
    }
  };
  // which of this wave's two 32-row halves hold real rows (wave-uniform)
  const bool rows0 = wm * 64 < rows, rows1 = wm * 64 + 32 < rows;
  const int kh = lane >> 5;
  prefetch(0);
  for (int k0 = 0; k0 < K; k0 += kTK) {
#pragma unroll
    for (int q = 0; q < 16; ++q) As[ar + 8 * q][ak] = a_nx[q];
#pragma unroll
    for (int j = 0; j < 16; ++j) Bs[bk + 2 * j][bc] = b_nx[j];
    __syncthreads();
    if (k0 + kTK < K) prefetch(k0 + kTK);   // in flight during the MFMAs below
    if (rows0) {
#pragma unroll
      for (int kk = 0; kk < kTK; kk += 2) {
        if (k0 + kk < K) {   // scalar guard: no matrix work on the k padding
          const int kl = kk + kh;
          const float b0 = Bs[kl][wn * 64 + (lane & 31)];
          const float b1 = Bs[kl][wn * 64 + 32 + (lane & 31)];
          const float a0 = As[wm * 64 + (lane & 31)][kl];
          acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
          acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
          if (rows1) {
            const float a1 = As[wm * 64 + 32 + (lane & 31)][kl];
            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }
  const int col0 = n0 + wn * 64 + (lane & 31);
  const int col1 = col0 + 32;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r0 = wm * 64 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    const int r1 = r0 + 32;
    if (r0 < rows) {
      float* __restrict__ orow = out + (row0 + r0) * ldo;
      if (col0 < N) orow[col0] = acc00[i];
      if (col1 < N) orow[col1] = acc01[i];
    }
    if (r1 < rows) {
      float* __restrict__ orow = out + (row0 + r1) * ldo;
      if (col0 < N) orow[col0] = acc10[i];
      if (col1 < N) orow[col1] = acc11[i];
    }
  }
}

// grad_W[g] = x[seg]^T @ grad[seg].  A 256-thread workgroup owns a (row chunk of a segment) x
// (128 k-columns) x (128 n-columns) piece: 2 x 2 waves, each 64 x 64 of grad_W in four
// accumulators.  The chunk is walked in blocks of 32 rows: the x and grad blocks are staged in LDS
// (each element read from L2 once per workgroup, coalesced 512-byte rows; next block prefetched
// into registers during the MFMAs), the reduction consumes two rows per MFMA (its k = 2).  Long
// segments are split into chunks by the host table so that no workgroup walks a 100k-row
// relation; chunks of one segment meet in grad_w through fp32 atomics (grad_w is zeroed first).
constexpr int kWR = 32;    // rows per staged block
constexpr int kWT = 128;   // k / n columns per workgroup

__global__ void __launch_bounds__(kBlock)
    segmm_wgrad_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ g,
                       int64_t ldg, const int64_t* __restrict__ g_rows,
                       const int32_t* __restrict__ chunks, int K, int N, int blocks,
                       float* __restrict__ gw) {
  __shared__ float Xs[kWR][kWT + 4];
  __shared__ float Gs[kWR][kWT + 4];
  const int t = blockIdx.x;
  const int seg = chunks[3 * t];
  const int64_t ra = chunks[3 * t + 1];
  const int64_t rb = ra + chunks[3 * t + 2];
  const int k0 = blockIdx.y * kWT;
  const int n0 = blockIdx.z * kWT;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int wk = wave >> 1, wn = wave & 1;
  const int blk = blocks > 1 ? seg % blocks : 0;
  x += static_cast<int64_t>(blk) * K;
  g += static_cast<int64_t>(blk) * N;
  f32x16 acc00, acc01, acc10, acc11;   // [k half][n half] of this wave's 64 x 64
  zero_acc(acc00);
  zero_acc(acc01);
  zero_acc(acc10);
  zero_acc(acc11);
  // staging: thread -> (col = tid % 128, rows tid / 128 + 2 j)
  const int sc = threadIdx.x & (kWT - 1);
  const int sr = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 7));  // (scalar)
  const bool xk_ok = k0 + sc < K, gn_ok = n0 + sc < N;
  const float* __restrict__ xs = x + (xk_ok ? k0 + sc : 0);
  const float* __restrict__ gs = g + (gn_ok ? n0 + sc : 0);
  float x_nx[kWR / 2], g_nx[kWR / 2];
  // g_rows: the gradient rows are read through an index.  A wave stages whole rows (`sr` is
  // wave-uniform), so the index is a scalar load — issued ONE BLOCK AHEAD of the row loads that
  // depend on it (loaded next to them, every block waited for two memory round trips: the launch
  // went from 0.81 to 1.18 ms at the FB15k-237 shape).
  int64_t gi[kWR / 2];
  auto load_index = [&](int64_t r) {
#pragma unroll
    for (int j = 0; j < kWR / 2; ++j) {
      const int64_t rr = r + sr + 2 * j;
      const int64_t rs = rr < rb ? rr : ra;
      gi[j] = g_rows ? g_rows[rs] : rs;
    }
  };
  auto prefetch = [&](int64_t r) {
#pragma unroll
    for (int j = 0; j < kWR / 2; ++j) {
      const int64_t rr = r + sr + 2 * j;
      const bool ok = rr < rb;
      const int64_t rs = ok ? rr : ra;
      x_nx[j] = (ok && xk_ok) ? xs[rs * ldx] : 0.f;
      g_nx[j] = (ok && gn_ok) ? gs[gi[j] * ldg] : 0.f;
    }
    load_index(r + kWR);   // (clamped inside: past the end it re-reads the chunk's first row)
  };
  // wave-uniform: which 32-wide halves of this wave's k / n range exist at all
  const bool kq0 = k0 + wk * 64 < K, kq1 = k0 + wk * 64 + 32 < K;
  const bool nq0 = n0 + wn * 64 < N, nq1 = n0 + wn * 64 + 32 < N;
  const int kh = lane >> 5;
  load_index(ra);
  prefetch(ra);
  for (int64_t r = ra; r < rb; r += kWR) {
#pragma unroll
    for (int j = 0; j < kWR / 2; ++j) {
      Xs[sr + 2 * j][sc] = x_nx[j];
      Gs[sr + 2 * j][sc] = g_nx[j];
    }
    __syncthreads();
    if (r + kWR < rb) prefetch(r + kWR);   // in flight during the MFMAs below
    if (kq0 && nq0) {
#pragma unroll
      for (int rr = 0; rr < kWR; rr += 2) {
        if (r + rr < rb) {   // scalar guard: rows past the chunk end are zero padding
          const float a0 = Xs[rr + kh][wk * 64 + (lane & 31)];
          const float b0 = Gs[rr + kh][wn * 64 + (lane & 31)];
          acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
          float b1 = 0.f;
          if (nq1) {
            b1 = Gs[rr + kh][wn * 64 + 32 + (lane & 31)];
            acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
          }
          if (kq1) {
            const float a1 = Xs[rr + kh][wk * 64 + 32 + (lane & 31)];
            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
            if (nq1) acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }
  float* __restrict__ gseg = gw + static_cast<int64_t>(seg) * K * N;
  const int col0 = n0 + wn * 64 + (lane & 31), col1 = col0 + 32;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int kr0 = k0 + wk * 64 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    const int kr1 = kr0 + 32;
    if (kr0 < K) {
      if (col0 < N) atomicAdd(gseg + static_cast<int64_t>(kr0) * N + col0, acc00[i]);
      if (col1 < N) atomicAdd(gseg + static_cast<int64_t>(kr0) * N + col1, acc01[i]);
    }
    if (kr1 < K) {
      if (col0 < N) atomicAdd(gseg + static_cast<int64_t>(kr1) * N + col0, acc10[i]);
      if (col1 < N) atomicAdd(gseg + static_cast<int64_t>(kr1) * N + col1, acc11[i]);
    }
  }
}

// ---- split arithmetic, K <= 128 (the per-relation weights of an RGCN layer, a HeteroLinear) ------
// The kernel above loses its time, not its bytes or flops: a tile is 128 x 100 x 100 — four k
// chunks, each with a global -> LDS -> barrier -> MFMA round trip, so the matrix pipe is 31 %
// busy (0.80 ms per launch at the FB15k-237 shape against 0.27 ms of HBM bytes and 0.35 ms of
// fp32 matrix time).  This one runs the 3 x bf16 split (split_bf16.h: same error bound as the
// fp32 instruction, 2.7 x fewer matrix cycles) and keeps the VALU — which the split makes the
// scarce pipe — for the one conversion nobody else can do:
//   * W is converted ONCE per call by a pre-pass (segmm_split_weights_kernel) into bf16 term
//     planes in a caller-provided workspace, [group][term][k-group of 8][column][16 bytes]; a
//     workgroup copies its 64-column slice into LDS ([term][column][68 dwords]: a fragment is one
//     ds_read_b128 per term; 17 * column + k-group is conflict-free for reads and writes) with
//     16-byte loads and no arithmetic — ONE barrier per workgroup, none in the k loop.  (Converted
//     inside the workgroup, as the first version did, every weight element was split 9 times per
//     launch: a third of the kernel's 1,192 VALU instructions per wave and tile.)
//   * A: every wave owns 32 rows, loads ALL of their K columns up front (one batch of loads per
//     workgroup: with a chunk-at-a-time prefetch every chunk waited ~1 us for HBM behind 0.3 us of
//     products), stages them 32 k-columns at a time in a WAVE-PRIVATE fp32 buffer ([32][36]:
//     coalesced 16-byte loads in, 8 consecutive floats per lane out), splits its fragment in
//     registers (each element once per column half: the rows are the wave's alone) and multiplies
//     — no barrier; the split of step s + 1 is issued between the matrix instructions of step s.
// A workgroup owns (tile of 128 rows) x (64 output columns): 70.6 KB of LDS, two workgroups per
// CU, so one copies its weights while the other multiplies; the two column halves of a tile are
// neighbours in ONE XCD's queue (the second read of the tile's rows is an L2 hit).
// k index of bf16 step (chunk c, half-chunk f) in lane half h: 32 c + 16 h + 8 f + (0..7) = k-group
// 4 c + 2 h + f — A and B are filed under the same rule, which is all the reduction needs.
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSK = 128;             // largest K
constexpr int kSN = 64;              // output columns per workgroup
constexpr int kSLDB = kSK / 2 + 4;   // dwords per staged B column (128 bf16 + 4 pad)
constexpr int kSLDA = 36;            // floats per staged A row (32 + 4 pad)
constexpr int kSBPlane = kSN * kSLDB;                 // dwords per term plane
constexpr int kSAWave = 32 * kSLDA;                   // floats per wave buffer
constexpr size_t kSegSplitLds = sizeof(uint32_t) * 3 * kSBPlane + sizeof(float) * 4 * kSAWave;

struct SegFrag {
  bf16x8 p[3];
};

__device__ __forceinline__ SegFrag seg_split8(const f32x4& v0, const f32x4& v1) {
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
  SegFrag f;
#pragma unroll
  for (int t = 0; t < 3; ++t) f.p[t] = __builtin_bit_cast(bf16x8, w[t]);
  return f;
}

// planes[((g * 3 + term) * KG + kg) * N + n] (16 bytes) = bf16 term `term` of W[g][8 kg .. + 7][n],
// W[g][k][n] at w[g * seg_stride + k * sk + n * sn]: a transposed view (sk == 1, the input
// gradient's W^T) is read where it lies — neighbouring lanes then read 32-byte pieces of rows whose
// other pieces the kg-neighbours (N lanes further) read, so every line is fetched once — instead of
// being copied into a [G, N, K] tensor first (0.10 ms per layer at the FB15k-237 shape).
__global__ void __launch_bounds__(kBlock)
    segmm_split_weights_kernel(const float* __restrict__ w, int64_t seg_stride, int64_t sk,
                               int64_t sn, int K, int N, int KG, int64_t total,
                               u32x4* __restrict__ planes) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;   // (g, kg, n)
  if (i >= total) return;
  const int n = static_cast<int>(i % N);
  const int64_t gk = i / N;
  const int kg = static_cast<int>(gk % KG);
  const int64_t g = gk / KG;
  const float* __restrict__ wc = w + g * seg_stride + 8 * kg * sk + n * sn;
  float v[8];
  if (sk == 1 && 8 * kg + 8 <= K && ((reinterpret_cast<uintptr_t>(wc) & 15u) == 0)) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(wc);
    const f32x4 b = *reinterpret_cast<const f32x4*>(wc + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 8 * kg + e < K ? wc[static_cast<int64_t>(e) * sk] : 0.f;
  }
  u32x4 pw[3];
#pragma unroll
  for (int p4 = 0; p4 < 4; ++p4) {
    uint32_t tt[3];
    split_pair(v[2 * p4], v[2 * p4 + 1], tt);
    pw[0][p4] = tt[0];
    pw[1][p4] = tt[1];
    pw[2][p4] = tt[2];
  }
#pragma unroll
  for (int tm = 0; tm < 3; ++tm) planes[((g * 3 + tm) * KG + kg) * N + n] = pw[tm];
}

// NC = number of 32-wide k chunks (K <= 32 NC): compile-time, so that every global load of a
// workgroup is issued up front in one batch and the k loop unrolls without branches.
template <int NC>
__global__ void __launch_bounds__(kBlock, 2)
    segmm_split_kernel(const float* __restrict__ x, int64_t ldx,
                       const int64_t* __restrict__ x_rows, const u32x4* __restrict__ planes,
                       const int32_t* __restrict__ tiles, int64_t n_items, int n_halves, int K,
                       int N, int blocks, float* __restrict__ out, int64_t ldo) {
  extern __shared__ __align__(16) uint32_t seg_lds[];
  uint32_t* const Bp = seg_lds;                                        // [3][kSN][kSLDB]
  const int64_t q = xcd_logical_block();
  if (q >= n_items) return;
  const int64_t t = q / n_halves;
  const int n0 = static_cast<int>(q - t * n_halves) * kSN;
  const int seg = tiles[3 * t];
  const int64_t row0 = tiles[3 * t + 1];
  const int rows = tiles[3 * t + 2];
  const int wave = wave_in_block(), lane = lane_id();
  float* const As = reinterpret_cast<float*>(seg_lds + 3 * kSBPlane) + wave * kSAWave;
  const int blk = blocks > 1 ? seg % blocks : 0;
  x += static_cast<int64_t>(blk) * K;
  out += static_cast<int64_t>(blk) * N;
  const int KG = (K + 7) >> 3;

  // ---- weight slice: lane -> column, wave + 4 i -> (term, k-group) slot of the LDS image.  Loads
  // are unconditional (clamped to a stored element); dead slots (k-groups past K, columns past N)
  // become zeros on their way into LDS.
  constexpr int kSlots = 3 * 4 * NC;               // (term, k-group) pairs of the image
  const int bn = lane;
  const bool bn_ok = n0 + bn < N;
  const u32x4* __restrict__ pseg =
      planes + static_cast<int64_t>(seg) * 3 * KG * N + (bn_ok ? n0 + bn : N - 1);
  u32x4 wv[kSlots / 4];
#pragma unroll
  for (int i = 0; i < kSlots / 4; ++i) {
    const int sl = wave + 4 * i;                   // (uniform)
    const int tm = sl / (4 * NC), kg = sl - tm * (4 * NC);
    wv[i] = pseg[static_cast<int64_t>(tm * KG + (kg < KG ? kg : KG - 1)) * N];
  }
  // ---- this wave's rows, every chunk (clamped: rows >= 1, K % 4 == 0)
  const int my_rows = rows - 32 * wave;            // <= 0: nothing to do but the staging below
  const int kq = lane & 7, rr = lane >> 3;
  f32x4 ra[NC][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int r = 32 * wave + rr + 8 * j;
    r = r < rows ? r : rows - 1;
    const float* pa = x + (x_rows ? x_rows[row0 + r] : row0 + r) * ldx;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      int k = 32 * c + 4 * kq;
      k = k < K ? k : K - 4;
      ra[c][j] = *reinterpret_cast<const f32x4*>(pa + k);
    }
  }
  // ---- B planes into LDS
#pragma unroll
  for (int i = 0; i < kSlots / 4; ++i) {
    const int sl = wave + 4 * i;
    const int tm = sl / (4 * NC), kg = sl - tm * (4 * NC);
    const bool ok = bn_ok && kg < KG;
    u32x4 v = wv[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0u;
    *reinterpret_cast<u32x4*>(Bp + tm * kSBPlane + bn * kSLDB + 4 * kg) = v;
  }
  __syncthreads();
  if (my_rows <= 0) return;

  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc[2];
  zero_acc(acc[0]);
  zero_acc(acc[1]);
  float* const a_st = As + rr * kSLDA + 4 * kq;    // store slot of load j: + 8 j rows
  const float* const a_rd = As + li * kSLDA + 16 * lh;
  const uint32_t* const b_rd = Bp + li * kSLDB + 8 * lh;
  const bool full_rows = my_rows >= 32;            // (uniform)
  // registers -> the wave's buffer; columns past K and rows past the tile are staged as zeros (a
  // clamped value could be Inf / NaN: Inf * 0 = NaN).  Interior chunks of full tiles: plain stores.
  auto stage = [&](const f32x4 (&rc)[4], int c) {
    if (full_rows && 32 * c + 32 <= K) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(a_st + 8 * j * kSLDA) = rc[j];
    } else {
      const int k = 32 * c + 4 * kq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool r_ok = rr + 8 * j < my_rows;
        f32x4 v = rc[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (r_ok && k + e < K) ? v[e] : 0.f;
        *reinterpret_cast<f32x4*>(a_st + 8 * j * kSLDA) = v;
      }
    }
  };
  auto read_a = [&](int f, f32x4& v0, f32x4& v1) {
    v0 = *reinterpret_cast<const f32x4*>(a_rd + 8 * f);
    v1 = *reinterpret_cast<const f32x4*>(a_rd + 8 * f + 4);
  };
  auto read_b = [&](int c, int f, SegFrag& b0, SegFrag& b1) {
#pragma unroll
    for (int tm = 0; tm < 3; ++tm) {
      const uint32_t* bp = b_rd + tm * kSBPlane + 16 * c + 4 * f;
      b0.p[tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bp));
      b1.p[tm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bp + 32 * kSLDB));
    }
  };
  // One bf16 step = 12 matrix instructions (terms outside, the two column blocks inside:
  // neighbours write different accumulators — the compiler, left alone, sorts them into two
  // dependent chains of six and the in-order wave then stalls at every second instruction) with
  // the split of the NEXT step's fragment issued between them: four groups of three matrix
  // instructions + one split_pair, fenced by scheduling barriers.  Columns past N are zeros in the
  // planes, so both blocks are always computed.
  auto step = [&](const SegFrag& a, const SegFrag& b0, const SegFrag& b1, const f32x4& v0,
                  const f32x4& v1, SegFrag& nxt) {
    u32x4 w3[3];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
      for (int m = 3 * g4; m < 3 * g4 + 3; ++m) {
        const int tm = m >> 1;
        if (m & 1)
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[kSplitTa[tm]], b1.p[kSplitTb[tm]],
                                                           acc[1], 0, 0, 0);
        else
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[kSplitTa[tm]], b0.p[kSplitTb[tm]],
                                                           acc[0], 0, 0, 0);
      }
      const float x0 = g4 < 2 ? v0[2 * g4] : v1[2 * g4 - 4];
      const float x1 = g4 < 2 ? v0[2 * g4 + 1] : v1[2 * g4 - 3];
      uint32_t tt[3];
      split_pair(x0, x1, tt);
      // (pinned HERE: in the unrolled loop LLVM otherwise sinks the split to its use — behind the
      // step's last matrix instruction, where nothing covers it)
      asm volatile("" : "+v"(tt[0]), "+v"(tt[1]), "+v"(tt[2]));
      w3[0][g4] = tt[0];
      w3[1][g4] = tt[1];
      w3[2][g4] = tt[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int tm = 0; tm < 3; ++tm) nxt.p[tm] = __builtin_bit_cast(bf16x8, w3[tm]);
  };
  f32x4 v0, v1;
  SegFrag a_cur, a_nxt, b0c, b1c, b0n, b1n;
  stage(ra[0], 0);
  read_a(0, v0, v1);
  read_b(0, 0, b0c, b1c);
  a_cur = seg_split8(v0, v1);
  // the K tail is staged as zeros; its second step is skipped when it holds nothing at all
  const bool tail_step = K > 32 * (NC - 1) + 8;    // (uniform)
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    read_a(1, v0, v1);                             // (first: the split waits for these two only)
    __builtin_amdgcn_sched_barrier(0);
    read_b(c, 1, b0n, b1n);
    __builtin_amdgcn_sched_barrier(0);
    step(a_cur, b0c, b1c, v0, v1, a_nxt);
    // (the wave's own LDS operations execute in order: its reads of chunk c above are behind it)
    if (c + 1 < NC) {                              // (compile-time: the loop is unrolled)
      stage(ra[c + 1 < NC ? c + 1 : c], c + 1);
      read_a(0, v0, v1);
      __builtin_amdgcn_sched_barrier(0);
      read_b(c + 1, 0, b0c, b1c);
      __builtin_amdgcn_sched_barrier(0);
      step(a_nxt, b0n, b1n, v0, v1, a_cur);
    } else if (tail_step) {
      __builtin_amdgcn_sched_barrier(0);
      step(a_nxt, b0n, b1n, v0, v1, a_cur);        // (converts a stale fragment: never used)
    }
  }
  // ---- epilogue: reg e of lane l is D[(e & 3) + 8 (e >> 2) + 4 (l >> 5)][l & 31]
  const int col0 = n0 + li, col1 = col0 + 32;
  float* __restrict__ ob = out + (row0 + 32 * wave + 4 * lh) * ldo + col0;
  if (full_rows && n0 + kSN <= N) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float* op = ob + ((e & 3) + 8 * (e >> 2)) * ldo;
      op[0] = acc[0][e];
      op[32] = acc[1][e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
      if (r < my_rows) {
        float* op = ob + ((e & 3) + 8 * (e >> 2)) * ldo;
        if (col0 < N) op[0] = acc[0][e];
        if (col1 < N) op[32] = acc[1][e];
      }
    }
  }
}

// ---- weight gradient, split arithmetic, operands converted once ---------------------------------
// grad_W[g] = x[seg]^T @ grad[seg] reduces over ROWS: both operands reach the matrix instruction
// transposed (eight k-consecutive bf16 values per lane = eight consecutive rows of one column).
// As in gemm_tn_split_kernel (csrc/gemm.hip) the STAGING thread converts: it owns eight consecutive
// rows of one column (eight 4-byte loads, each a 256-byte run across the wave), splits them into
// the three bf16 terms as (row, row + 1) pairs and writes, per term, ONE 16-byte vector — so a
// fragment is one ds_read_b128 per term, every element is converted exactly once per workgroup
// (176 VALU instructions per thread and 32-row block; the fp32 kernel above needs 64 matrix
// instructions of 64 cycles per wave and block, this one 48 of 32).  LDS image:
// [operand x | g][term][column 0..127][20 dwords] (4 row groups x 4 dwords + 4 pad: 5 * column
// mod 16 is a bijection for the 16-lane read groups) = 61,440 bytes, single-buffered with two
// barriers per block, two workgroups per CU (one converts while the other multiplies).  The loads
// of block b + 1 are issued before the products of block b; the row index of the gradient
// operand (g_rows) is a scalar load one block further ahead.
constexpr int kWLD = 20;                    // dwords per staged column
constexpr int kWPlane = kWT * kWLD;         // dwords per (operand, term) plane
constexpr size_t kWgSplitLds = sizeof(uint32_t) * 6 * kWPlane;

__global__ void __launch_bounds__(kBlock, 2)
    segmm_wgrad_split_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ g,
                             int64_t ldg, const int64_t* __restrict__ g_rows,
                             const int32_t* __restrict__ chunks, int K, int N, int blocks,
                             float* __restrict__ gw) {
  extern __shared__ __align__(16) uint32_t wg_lds[];
  const int t = blockIdx.x;
  const int seg = chunks[3 * t];
  const int64_t ra = chunks[3 * t + 1];
  const int64_t rb = ra + chunks[3 * t + 2];
  const int k0 = blockIdx.y * kWT;
  const int n0 = blockIdx.z * kWT;
  const int wave = wave_in_block(), lane = lane_id();
  const int wk = wave >> 1, wn = wave & 1;
  const int blk = blocks > 1 ? seg % blocks : 0;
  x += static_cast<int64_t>(blk) * K;
  g += static_cast<int64_t>(blk) * N;
  // staging: thread -> column tid % 128, row groups (tid / 128) and (tid / 128) + 2 of the block
  const int sc = threadIdx.x & (kWT - 1);
  const int gsel = wave >> 1;                         // (scalar: tid / 128)
  const bool xk_ok = k0 + sc < K, gn_ok = n0 + sc < N;
  const float* __restrict__ xs = x + (xk_ok ? k0 + sc : 0);
  const float* __restrict__ gs = g + (gn_ok ? n0 + sc : 0);
  float xv[2][8], gv[2][8];
  int64_t gi[2][8];
  auto load_index = [&](int64_t r) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int64_t rr = r + 8 * (gsel + 2 * u) + e;
        const int64_t rs = rr < rb ? rr : ra;
        gi[u][e] = g_rows ? g_rows[rs] : rs;
      }
  };
  auto load_rows = [&](int64_t r) {                   // unconditional (clamped), masked at the split
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int64_t rr = r + 8 * (gsel + 2 * u) + e;
        const int64_t rs = rr < rb ? rr : ra;
        xv[u][e] = xs[rs * ldx];
        gv[u][e] = gs[gi[u][e] * ldg];
      }
    load_index(r + kWR);                              // (clamped inside)
  };
  uint32_t* const wx = wg_lds + sc * kWLD;
  uint32_t* const wgp = wg_lds + 3 * kWPlane + sc * kWLD;
  auto convert_store = [&](int64_t r) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int gq = gsel + 2 * u;
      u32x4 px[3], pg[3];
#pragma unroll
      for (int p4 = 0; p4 < 4; ++p4) {
        const bool ok0 = r + 8 * gq + 2 * p4 < rb, ok1 = r + 8 * gq + 2 * p4 + 1 < rb;
        uint32_t tt[3];
        split_pair((ok0 && xk_ok) ? xv[u][2 * p4] : 0.f, (ok1 && xk_ok) ? xv[u][2 * p4 + 1] : 0.f,
                   tt);
        px[0][p4] = tt[0];
        px[1][p4] = tt[1];
        px[2][p4] = tt[2];
        split_pair((ok0 && gn_ok) ? gv[u][2 * p4] : 0.f, (ok1 && gn_ok) ? gv[u][2 * p4 + 1] : 0.f,
                   tt);
        pg[0][p4] = tt[0];
        pg[1][p4] = tt[1];
        pg[2][p4] = tt[2];
      }
#pragma unroll
      for (int tm = 0; tm < 3; ++tm) {
        *reinterpret_cast<u32x4*>(wx + tm * kWPlane + 4 * gq) = px[tm];
        *reinterpret_cast<u32x4*>(wgp + tm * kWPlane + 4 * gq) = pg[tm];
      }
    }
  };
  f32x16 acc[2][2];                                   // [k half][n half] of this wave's 64 x 64
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) zero_acc(acc[i][j]);
  // wave-uniform: this wave's 64-column ranges are not both empty (columns past K / N inside a
  // range are staged as zeros, so all four 32 x 32 blocks are computed)
  const bool active = k0 + wk * 64 < K && n0 + wn * 64 < N;
  const int li = lane & 31, lh = lane >> 5;
  const uint32_t* const fa = wg_lds + (wk * 64 + li) * kWLD + 4 * lh;
  const uint32_t* const fb = wg_lds + 3 * kWPlane + (wn * 64 + li) * kWLD + 4 * lh;
  auto frag = [&](const uint32_t* base, int half, int sstep) {
    SegFrag f;
#pragma unroll
    for (int tm = 0; tm < 3; ++tm)
      f.p[tm] = __builtin_bit_cast(
          bf16x8, *reinterpret_cast<const u32x4*>(base + tm * kWPlane + half * 32 * kWLD +
                                                  8 * sstep));
    return f;
  };
  load_index(ra);
  load_rows(ra);
  for (int64_t r = ra; r < rb; r += kWR) {
    convert_store(r);
    __syncthreads();
    load_rows(r + kWR);                               // in flight during the products below
    if (active) {
#pragma unroll
      for (int sstep = 0; sstep < 2; ++sstep) {
        const SegFrag a0 = frag(fa, 0, sstep), a1 = frag(fa, 1, sstep);
        const SegFrag b0 = frag(fb, 0, sstep), b1 = frag(fb, 1, sstep);
#pragma unroll
        for (int tm = 0; tm < kSplitTerms; ++tm) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.p[kSplitTa[tm]],
                                                              b0.p[kSplitTb[tm]], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.p[kSplitTa[tm]],
                                                              b1.p[kSplitTb[tm]], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.p[kSplitTa[tm]],
                                                              b0.p[kSplitTb[tm]], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.p[kSplitTa[tm]],
                                                              b1.p[kSplitTb[tm]], acc[1][1], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  if (!active) return;
  float* __restrict__ gseg = gw + static_cast<int64_t>(seg) * K * N;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + 32 * j + li;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int kr = k0 + wk * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (kr < K && col < N) atomicAdd(gseg + static_cast<int64_t>(kr) * N + col, acc[i][j][e]);
      }
    }
  }
}

}  // namespace pygamd

using namespace pygamd;

extern "C" {

int pygamd_segment_matmul_tile_rows(void) { return kTM; }

static bool segmm_split_ok(const float* x, int64_t ldx, int64_t K, int64_t N) {
  // 16-byte rows (K % 4, ldx % 4, aligned base) and K <= 128; the weights in any strides (the
  // pre-pass reads them once)
  return pygamd_get_gemm_mode() == PYGAMD_GEMM_SPLIT_BF16 && K >= 4 && K <= kSK && K % 4 == 0 &&
         N >= 1 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
}

int pygamd_segment_matmul_workspace_bytes(int64_t n_groups, int64_t K, int64_t N, size_t* bytes) {
  if (!bytes || n_groups < 0 || K < 0 || N < 0) return PYGAMD_ERR_INVALID_ARG;
  // the bf16 term planes of the convert-once kernel (K <= 128, K % 4 == 0); nothing otherwise
  *bytes = (K >= 4 && K <= kSK && K % 4 == 0)
               ? static_cast<size_t>(n_groups) * 3 * ceil_div(K, 8) * N * sizeof(u32x4)
               : 0;
  return PYGAMD_OK;
}

int pygamd_segment_matmul(const float* x, int64_t ldx, const int64_t* x_rows, const float* w,
                          int64_t w_seg_stride, int64_t w_stride_k, int64_t w_stride_n,
                          int64_t n_groups, const int32_t* tiles, int64_t n_tiles, int64_t K,
                          int64_t N, int64_t blocks, float* out, int64_t ldo, void* workspace,
                          size_t workspace_bytes, void* stream) {
  if (n_tiles < 0 || n_groups < 0 || K < 0 || N < 0 || blocks < 1 || blocks > INT32_MAX ||
      K > INT32_MAX || N > INT32_MAX || ldx < blocks * K || ldo < blocks * N)
    return PYGAMD_ERR_INVALID_ARG;
  if (n_tiles == 0 || N == 0) return PYGAMD_OK;
  if (!x || !w || !tiles || !out) return PYGAMD_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  size_t need = 0;
  pygamd_segment_matmul_workspace_bytes(n_groups, K, N, &need);
  // split arithmetic (the library's default, pygamd_set_gemm_mode), a weight that fits the LDS
  // planes and a workspace for them: the convert-once kernel.  Anything else keeps the general
  // fp32 kernel (exact products: never less accurate).
  if (workspace && need > 0 && workspace_bytes >= need &&
      (reinterpret_cast<uintptr_t>(workspace) & 15u) == 0 &&
      segmm_split_ok(x, ldx, K, N)) {
    u32x4* planes = static_cast<u32x4*>(workspace);
    const int KG = static_cast<int>(ceil_div(K, 8));
    const int64_t total = n_groups * KG * N;
    hipLaunchKernelGGL(segmm_split_weights_kernel,
                       dim3(static_cast<unsigned>(ceil_div(total, kBlock))), dim3(kBlock), 0, st, w,
                       w_seg_stride, w_stride_k, w_stride_n, static_cast<int>(K),
                       static_cast<int>(N), KG, total, planes);
    PYGAMD_LAUNCH_CHECK();
    const int n_halves = static_cast<int>(ceil_div(N, kSN));
    const int64_t n_items = n_tiles * n_halves;
    const unsigned sgrid = static_cast<unsigned>(round_up(n_items, 8));
    const int nc = static_cast<int>(ceil_div(K, 32));
    auto launch = [&](auto kernel) -> int {
      // (the planes need more than 64 KB; set on every launch like gemm.hip / sage_fused.hip: the
      // attribute belongs to the current device's code object, a process may drive several GPUs)
      PYGAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(kSegSplitLds)));
      hipLaunchKernelGGL(kernel, dim3(sgrid), dim3(kBlock), kSegSplitLds, st, x, ldx, x_rows,
                         planes, tiles, n_items, n_halves, static_cast<int>(K), static_cast<int>(N),
                         static_cast<int>(blocks), out, ldo);
      return PYGAMD_OK;
    };
    int rc;
    switch (nc) {
      case 1: rc = launch(segmm_split_kernel<1>); break;
      case 2: rc = launch(segmm_split_kernel<2>); break;
      case 3: rc = launch(segmm_split_kernel<3>); break;
      default: rc = launch(segmm_split_kernel<4>); break;
    }
    if (rc != PYGAMD_OK) return rc;
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  }
  const dim3 grid(static_cast<unsigned>(n_tiles), static_cast<unsigned>(ceil_div(N, kTN)));
  hipLaunchKernelGGL(segmm_kernel, grid, dim3(kBlock), 0, st, x, ldx, x_rows, w,
                     w_seg_stride, w_stride_k, w_stride_n, tiles, static_cast<int>(K),
                     static_cast<int>(N), static_cast<int>(blocks), out, ldo);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

int pygamd_segment_matmul_wgrad(const float* x, int64_t ldx, const float* g, int64_t ldg,
                                const int64_t* g_rows, const int32_t* chunks, int64_t n_chunks, int64_t n_seg,
                                int64_t K, int64_t N, int64_t blocks, float* grad_w,
                                void* stream) {
  if (n_seg < 0 || n_chunks < 0 || K < 0 || N < 0 || blocks < 1 || blocks > INT32_MAX ||
      K > INT32_MAX || N > INT32_MAX || ldx < blocks * K || ldg < blocks * N)
    return PYGAMD_ERR_INVALID_ARG;
  if (n_seg == 0 || K == 0 || N == 0) return PYGAMD_OK;
  if (!grad_w) return PYGAMD_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  PYGAMD_HIP_CHECK(hipMemsetAsync(grad_w, 0, sizeof(float) * n_seg * K * N, st));
  if (n_chunks == 0) return PYGAMD_OK;
  if (!x || !g || !chunks) return PYGAMD_ERR_INVALID_ARG;
  if (ceil_div(K, kWT) > 65535 || ceil_div(N, kWT) > 65535) return PYGAMD_ERR_UNSUPPORTED;
  const dim3 grid(static_cast<unsigned>(n_chunks), static_cast<unsigned>(ceil_div(K, kWT)),
                  static_cast<unsigned>(ceil_div(N, kWT)));
  if (pygamd_get_gemm_mode() == PYGAMD_GEMM_SPLIT_BF16) {
    PYGAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(segmm_wgrad_split_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(kWgSplitLds)));
    hipLaunchKernelGGL(segmm_wgrad_split_kernel, grid, dim3(kBlock), kWgSplitLds, st, x, ldx, g,
                       ldg, g_rows, chunks, static_cast<int>(K), static_cast<int>(N),
                       static_cast<int>(blocks), grad_w);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  }
  hipLaunchKernelGGL(segmm_wgrad_kernel, grid, dim3(kBlock), 0, st, x, ldx, g, ldg, g_rows, chunks,
                     static_cast<int>(K), static_cast<int>(N), static_cast<int>(blocks), grad_w);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

}  // extern "C"
