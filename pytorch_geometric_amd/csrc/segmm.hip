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
#include "common.h"

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
    segmm_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                 int64_t w_seg_stride, int64_t w_sk, int64_t w_sn,
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
  const float* __restrict__ xa = x + row0 * ldx + ak;
  // B tile: thread -> (k = tid / 128 + 2 j, col = tid % 128): 512-byte coalesced weight rows
  const int bc = threadIdx.x & (kTN - 1);
  const int bk = threadIdx.x >> 7;
  const bool bcol_ok = n0 + bc < N;
  const float* __restrict__ wb = wseg + static_cast<int64_t>(bcol_ok ? n0 + bc : 0) * w_sn;
  float a_nx[16], b_nx[16];
  auto prefetch = [&](int k0) {
    const bool ak_ok = k0 + ak < K;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = ar + 8 * q;
      const bool ok = ak_ok && r < rows;
      a_nx[q] = ok ? xa[static_cast<int64_t>(ok ? r : 0) * ldx + k0] : 0.f;
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
                       int64_t ldg, const int32_t* __restrict__ chunks, int K, int N, int blocks,
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
  const int sr = threadIdx.x >> 7;
  const bool xk_ok = k0 + sc < K, gn_ok = n0 + sc < N;
  const float* __restrict__ xs = x + (xk_ok ? k0 + sc : 0);
  const float* __restrict__ gs = g + (gn_ok ? n0 + sc : 0);
  float x_nx[kWR / 2], g_nx[kWR / 2];
  auto prefetch = [&](int64_t r) {
#pragma unroll
    for (int j = 0; j < kWR / 2; ++j) {
      const int64_t rr = r + sr + 2 * j;
      const bool ok = rr < rb;
      const int64_t rs = ok ? rr : ra;
      x_nx[j] = (ok && xk_ok) ? xs[rs * ldx] : 0.f;
      g_nx[j] = (ok && gn_ok) ? gs[rs * ldg] : 0.f;
    }
  };
  // wave-uniform: which 32-wide halves of this wave's k / n range exist at all
  const bool kq0 = k0 + wk * 64 < K, kq1 = k0 + wk * 64 + 32 < K;
  const bool nq0 = n0 + wn * 64 < N, nq1 = n0 + wn * 64 + 32 < N;
  const int kh = lane >> 5;
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

}  // namespace pygamd

using namespace pygamd;

extern "C" {

int pygamd_segment_matmul_tile_rows(void) { return kTM; }

int pygamd_segment_matmul(const float* x, int64_t ldx, const float* w, int64_t w_seg_stride,
                          int64_t w_stride_k, int64_t w_stride_n, const int32_t* tiles,
                          int64_t n_tiles, int64_t K, int64_t N, int64_t blocks, float* out,
                          int64_t ldo, void* stream) {
  if (n_tiles < 0 || K < 0 || N < 0 || blocks < 1 || blocks > INT32_MAX || K > INT32_MAX ||
      N > INT32_MAX || ldx < blocks * K || ldo < blocks * N)
    return PYGAMD_ERR_INVALID_ARG;
  if (n_tiles == 0 || N == 0) return PYGAMD_OK;
  if (!x || !w || !tiles || !out) return PYGAMD_ERR_INVALID_ARG;
  const dim3 grid(static_cast<unsigned>(n_tiles), static_cast<unsigned>(ceil_div(N, kTN)));
  hipLaunchKernelGGL(segmm_kernel, grid, dim3(kBlock), 0, as_stream(stream), x, ldx, w,
                     w_seg_stride, w_stride_k, w_stride_n, tiles, static_cast<int>(K),
                     static_cast<int>(N), static_cast<int>(blocks), out, ldo);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

int pygamd_segment_matmul_wgrad(const float* x, int64_t ldx, const float* g, int64_t ldg,
                                const int32_t* chunks, int64_t n_chunks, int64_t n_seg,
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
  hipLaunchKernelGGL(segmm_wgrad_kernel, grid, dim3(kBlock), 0, st, x, ldx, g, ldg, chunks,
                     static_cast<int>(K), static_cast<int>(N), static_cast<int>(blocks), grad_w);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

}  // extern "C"
