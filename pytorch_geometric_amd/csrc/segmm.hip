// segmm.hip — segment_matmul (grouped GEMM over row segments) on the fp32 matrix cores.
//
//   out[ptr[g] : ptr[g+1]] = x[ptr[g] : ptr[g+1]] @ W[g]          (pyg_lib.ops.segment_matmul,
//   torch_geometric/nn/conv/rgcn_conv.py:288, nn/dense/linear.py:255)
//
// MFMA-bound (the one GEMM-shaped op on this path): v_mfma_f32_32x32x2_f32, exact fp32
// (bitwise an fmaf chain), fragment maps per cdna_hip_programming.md §3:
//   A: lane l holds A[i = l & 31][k = l >> 5];  B: lane l holds B[k = l >> 5][j = l & 31];
//   D: reg r of lane l is D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
// ONE launch covers every segment: the host builds a table of (segment, first row, rows) tiles of
// 64 rows; a 256-thread workgroup (2 x 2 waves) owns a 64 x 128 output tile, stages the A tile
// through LDS (row stride 17 floats: conflict-free column reads) and streams B (the segment's
// weight, L2-resident) straight from global with coalesced 128-byte rows; both are prefetched one
// k chunk ahead into registers (PMC of the unpipelined kernel: 64 % of wave cycles parked in
// s_waitcnt/barrier, MFMA pipe 31 % busy).
#include "common.h"

namespace pygamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTM = 64;    // rows per tile
constexpr int kTN = 128;   // cols per workgroup
constexpr int kTK = 16;    // k chunk staged in LDS

__global__ void __launch_bounds__(kBlock)
    segmm_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                 int64_t w_seg_stride, int64_t w_sk, int64_t w_sn,
                 const int32_t* __restrict__ tiles, int K, int N, int blocks,
                 float* __restrict__ out, int64_t ldo) {
  __shared__ float As[kTM][kTK + 1];
  const int t = blockIdx.x;
  const int seg = tiles[3 * t];
  const int64_t row0 = tiles[3 * t + 1];
  const int rows = tiles[3 * t + 2];
  const int n0 = blockIdx.y * kTN;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int col0 = n0 + wn * 64 + (lane & 31);
  const int col1 = col0 + 32;
  const float* __restrict__ wseg = w + static_cast<int64_t>(seg) * w_seg_stride;
  // block-diagonal weights: group seg = relation * blocks + b works on column block b
  const int blk = blocks > 1 ? seg % blocks : 0;
  x += static_cast<int64_t>(blk) * K;
  out += static_cast<int64_t>(blk) * N;
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc0[i] = 0.f;
    acc1[i] = 0.f;
  }
  const int lr = threadIdx.x >> 2;        // 0..63: row of the A tile this thread loads
  const int lk = (threadIdx.x & 3) * 4;   // 0,4,8,12: first k of its 4 values
  const bool row_ok = lr < rows;
  const float* __restrict__ xa = x + (row0 + (row_ok ? lr : 0)) * ldx + lk;
  const int kh = lane >> 5;               // which of the MFMA's two k values this lane feeds
  const bool c0_ok = col0 < N, c1_ok = col1 < N;
  const float* __restrict__ wb0 = wseg + static_cast<int64_t>(c0_ok ? col0 : 0) * w_sn;
  const float* __restrict__ wb1 = wseg + static_cast<int64_t>(c1_ok ? col1 : 0) * w_sn;
  // Register double buffer: the global loads of chunk i+1 (A values for the LDS tile, B values
  // for this lane's MFMA operands) are issued before the MFMAs of chunk i, so their latency is
  // covered by matrix work of the same wave instead of parking it at s_waitcnt.
  float a_nx[4], b0_nx[kTK / 2], b1_nx[kTK / 2];
  auto prefetch = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      a_nx[j] = (row_ok && k0 + lk + j < K) ? xa[k0 + j] : 0.f;
#pragma unroll
    for (int s2 = 0; s2 < kTK / 2; ++s2) {
      const int k = k0 + 2 * s2 + kh;
      const bool k_ok = k < K;
      const int64_t off = static_cast<int64_t>(k_ok ? k : 0) * w_sk;
      b0_nx[s2] = (k_ok && c0_ok) ? wb0[off] : 0.f;
      b1_nx[s2] = (k_ok && c1_ok) ? wb1[off] : 0.f;
    }
  };
  prefetch(0);
  for (int k0 = 0; k0 < K; k0 += kTK) {
    float b0[kTK / 2], b1[kTK / 2];
#pragma unroll
    for (int j = 0; j < 4; ++j) As[lr][lk + j] = a_nx[j];
#pragma unroll
    for (int s2 = 0; s2 < kTK / 2; ++s2) {
      b0[s2] = b0_nx[s2];
      b1[s2] = b1_nx[s2];
    }
    __syncthreads();
    if (k0 + kTK < K) prefetch(k0 + kTK);
#pragma unroll
    for (int s2 = 0; s2 < kTK / 2; ++s2) {
      const float a = As[wm * 32 + (lane & 31)][2 * s2 + kh];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0[s2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1[s2], acc1, 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    if (r < rows) {
      float* __restrict__ orow = out + (row0 + r) * ldo;
      if (col0 < N) orow[col0] = acc0[i];
      if (col1 < N) orow[col1] = acc1[i];
    }
  }
}

// grad_W[g] = x[seg]^T @ grad[seg]: one wave per (row chunk of a segment, 32 k-columns, 64
// n-columns); the reduction runs over the chunk's rows two at a time (the MFMA's k = 2), operands
// straight from global (both 128-byte coalesced per half-wave).  Long segments are split into
// chunks by the host table so that no single wave walks a 100k-row relation; chunks of one
// segment meet in grad_w through fp32 atomics (grad_w is zeroed first), single-chunk segments
// could store directly but share the same path for simplicity.
__global__ void __launch_bounds__(kWave)
    segmm_wgrad_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ g,
                       int64_t ldg, const int32_t* __restrict__ chunks, int K, int N, int blocks,
                       float* __restrict__ gw) {
  const int t = blockIdx.x;
  const int seg = chunks[3 * t];
  const int64_t ra = chunks[3 * t + 1];
  const int64_t rb = ra + chunks[3 * t + 2];
  const int k0 = blockIdx.y * 32;
  const int n0 = blockIdx.z * 64;
  const int lane = threadIdx.x;
  const int kc = k0 + (lane & 31);
  const int col0 = n0 + (lane & 31), col1 = col0 + 32;
  const int blk = blocks > 1 ? seg % blocks : 0;
  x += static_cast<int64_t>(blk) * K;
  g += static_cast<int64_t>(blk) * N;
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc0[i] = 0.f;
    acc1[i] = 0.f;
  }
  const bool k_ok = kc < K, c0_ok = col0 < N, c1_ok = col1 < N;
  const float* __restrict__ xa = x + (k_ok ? kc : 0);
  const float* __restrict__ g0 = g + (c0_ok ? col0 : 0);
  const float* __restrict__ g1 = g + (c1_ok ? col1 : 0);
  float a_nx[4], b0_nx[4], b1_nx[4];
  auto prefetch = [&](int64_t r) {  // the 8 rows starting at r, two per MFMA
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t rr = r + 2 * u + (lane >> 5);
      const bool ok = rr < rb;
      const int64_t rs = ok ? rr : ra;
      a_nx[u] = (ok && k_ok) ? xa[rs * ldx] : 0.f;
      b0_nx[u] = (ok && c0_ok) ? g0[rs * ldg] : 0.f;
      b1_nx[u] = (ok && c1_ok) ? g1[rs * ldg] : 0.f;
    }
  };
  prefetch(ra);
  for (int64_t r = ra; r < rb; r += 8) {
    float a[4], b0[4], b1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = a_nx[u];
      b0[u] = b0_nx[u];
      b1[u] = b1_nx[u];
    }
    if (r + 8 < rb) prefetch(r + 8);  // in flight during the MFMAs below
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b0[u], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b1[u], acc1, 0, 0, 0);
    }
  }
  float* __restrict__ gseg = gw + static_cast<int64_t>(seg) * K * N;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int kr = k0 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    if (kr < K) {
      if (col0 < N) atomicAdd(gseg + static_cast<int64_t>(kr) * N + col0, acc0[i]);
      if (col1 < N) atomicAdd(gseg + static_cast<int64_t>(kr) * N + col1, acc1[i]);
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
  if (ceil_div(K, 32) > 65535 || ceil_div(N, 64) > 65535) return PYGAMD_ERR_UNSUPPORTED;
  const dim3 grid(static_cast<unsigned>(n_chunks), static_cast<unsigned>(ceil_div(K, 32)),
                  static_cast<unsigned>(ceil_div(N, 64)));
  hipLaunchKernelGGL(segmm_wgrad_kernel, grid, dim3(kWave), 0, st, x, ldx, g, ldg, chunks,
                     static_cast<int>(K), static_cast<int>(N), static_cast<int>(blocks), grad_w);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

}  // extern "C"
