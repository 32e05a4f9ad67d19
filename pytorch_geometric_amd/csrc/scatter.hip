// scatter.hip — the unfused halves of MessagePassing.propagate on an UNSORTED index:
// gather (a2: index_select), scatter-{sum,mean,min,max,mul,any} (a5) and scatter_argmax (a6).
// These exist for callers that hold a raw, unsorted `index` and no graph handle
// (torch_geometric/utils/_scatter.py:14-184).  Work items are flat (row, 16-byte chunk) pairs so
// every global access is a full-width coalesced vector; reductions use device-scope atomics
// (fp32 add is native on CDNA; min/max go through the integer trick in common.h).
#include "common.h"

namespace pygamd {

template <typename IdxT, int VW>
__global__ void __launch_bounds__(kBlock)
    gather_rows_kernel(const float* __restrict__ x, int64_t ldx, int64_t n_src,
                       const IdxT* __restrict__ index, int64_t n, int64_t units, int64_t F,
                       float* __restrict__ out, int64_t ldo, int32_t* __restrict__ err) {
  const int64_t total = n * units;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t e = t / units;
    const int64_t f = (t - e * units) * VW;
    int64_t r = index[e];
    if (r < 0 || r >= n_src) {
      if (err) *err = 1;
      r = 0;
    }
    if (f < F) store_vec<VW>(out + e * ldo + f, load_vec<VW>(x + r * ldx + f));
  }
}

__global__ void __launch_bounds__(kBlock)
    fill_rows_kernel(float* __restrict__ out, int64_t ldo, int64_t rows, int64_t F, float v) {
  const int64_t total = rows * F;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t r = t / F;
    out[r * ldo + (t - r * F)] = v;
  }
}

// Work item = (row e, VW-float chunk): one 16-byte source load and ONE index load per item (the
// scalar version re-read index[e] for every float), then VW device-scope atomics.
template <typename IdxT, int REDUCE, int VW>
__global__ void __launch_bounds__(kBlock)
    scatter_rows_kernel(const float* __restrict__ src, int64_t lds,
                        const IdxT* __restrict__ index, int64_t n, int64_t units, int64_t F,
                        float* __restrict__ out, int64_t ldo, int64_t dim_size,
                        float* __restrict__ count, int32_t* __restrict__ err) {
  const int64_t total = n * units;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t e = t / units;
    const int64_t u = t - e * units;
    const int64_t f = u * VW;
    const int64_t g = index[e];
    if (g < 0 || g >= dim_size) {
      if (err) *err = 1;
      continue;
    }
    const Vec<VW> v = load_vec<VW>(src + e * lds + f);
    float* dst = out + g * ldo + f;
#pragma unroll
    for (int q = 0; q < VW; ++q) {
      if (REDUCE == PYGAMD_SUM || REDUCE == PYGAMD_MEAN) {
        atomicAdd(dst + q, v.v[q]);
      } else if (REDUCE == PYGAMD_MAX) {
        atomic_max_f32(dst + q, v.v[q]);
      } else if (REDUCE == PYGAMD_MIN) {
        atomic_min_f32(dst + q, v.v[q]);
      } else if (REDUCE == PYGAMD_MUL) {
        atomic_mul_f32(dst + q, v.v[q]);
      } else {
        dst[q] = v.v[q];  // 'any': some contributing row wins
      }
    }
    if (count && u == 0) atomicAdd(count + g, 1.f);
  }
}

template <int REDUCE>
__global__ void __launch_bounds__(kBlock)
    scatter_finalize_kernel(float* __restrict__ out, int64_t ldo, int64_t rows, int64_t F,
                            const float* __restrict__ count) {
  const int64_t total = rows * F;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t r = t / F;
    float* p = out + r * ldo + (t - r * F);
    const float c = count[r];
    if (REDUCE == PYGAMD_MEAN) {
      *p = *p / (c < 1.f ? 1.f : c);
    } else {
      if (c == 0.f) *p = 0.f;
    }
  }
}

__global__ void __launch_bounds__(kBlock)
    tie_init_kernel(const float* __restrict__ out, int64_t ldo, int64_t rows, int64_t F,
                    float* __restrict__ ntie) {
  const int64_t total = rows * F;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t r = t / F;
    const int64_t o = r * ldo + (t - r * F);
    ntie[o] = (out[o] == 0.f) ? 1.f : 0.f;
  }
}

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    scatter_tie_count_kernel(const float* __restrict__ src, int64_t lds,
                             const IdxT* __restrict__ index, int64_t n, int64_t F,
                             const float* __restrict__ out, int64_t ldo, int64_t dim_size,
                             float* __restrict__ ntie) {
  const int64_t total = n * F;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t e = t / F;
    const int64_t f = t - e * F;
    const int64_t g = index[e];
    if (g < 0 || g >= dim_size) continue;
    if (src[e * lds + f] == out[g * ldo + f]) atomicAdd(ntie + g * ldo + f, 1.f);
  }
}

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    scatter_minmax_bwd_kernel(const float* __restrict__ src, int64_t lds,
                              const IdxT* __restrict__ index, int64_t n, int64_t F,
                              const float* __restrict__ out, const float* __restrict__ grad_out,
                              const float* __restrict__ ntie, int64_t ldo, int64_t dim_size,
                              float* __restrict__ grad_src, int64_t ldg) {
  const int64_t total = n * F;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t e = t / F;
    const int64_t f = t - e * F;
    const int64_t g = index[e];
    if (g < 0 || g >= dim_size) {  // the forward flagged it; never read out of bounds here
      grad_src[e * ldg + f] = 0.f;
      continue;
    }
    const int64_t o = g * ldo + f;
    grad_src[e * ldg + f] = (src[e * lds + f] == out[o]) ? grad_out[o] / ntie[o] : 0.f;
  }
}

// ---- backward of scatter(reduce='mul') ---------------------------------------------------------
// The reference's CPU path is ones.scatter_reduce_('prod', include_self=True)
// (torch_geometric/utils/_scatter.py:119-133); its gradient (ATen scatter_reduce_backward, 'prod')
// is NOT g * out / src when zeros are present: with z = number of zeros scattered into (group, f)
//   z == 0 : grad_e = g * out / src_e
//   z == 1 : the zero element gets g * prod(other elements of the group), every other element 0
//   z >= 2 : every element 0.
// Three flat passes: count the zeros (atomics only where src == 0), multiply the non-zero
// elements of the groups that hold exactly one zero (atomics only there), then the gradient.
template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    scatter_mul_zero_count_kernel(const float* __restrict__ src, int64_t lds,
                                  const IdxT* __restrict__ index, int64_t n, int64_t F,
                                  int64_t dim_size, int32_t* __restrict__ nzero) {
  const int64_t total = n * F;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t e = t / F;
    const int64_t f = t - e * F;
    const int64_t g = index[e];
    if (g < 0 || g >= dim_size) continue;
    if (src[e * lds + f] == 0.f) atomicAdd(nzero + g * F + f, 1);
  }
}

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    scatter_mul_others_kernel(const float* __restrict__ src, int64_t lds,
                              const IdxT* __restrict__ index, int64_t n, int64_t F,
                              int64_t dim_size, const int32_t* __restrict__ nzero,
                              float* __restrict__ others) {
  const int64_t total = n * F;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t e = t / F;
    const int64_t f = t - e * F;
    const int64_t g = index[e];
    if (g < 0 || g >= dim_size) continue;
    const float v = src[e * lds + f];
    if (v != 0.f && nzero[g * F + f] == 1) atomic_mul_f32(others + g * F + f, v);
  }
}

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    scatter_mul_bwd_kernel(const float* __restrict__ src, int64_t lds,
                           const IdxT* __restrict__ index, int64_t n, int64_t F,
                           const float* __restrict__ out, const float* __restrict__ grad_out,
                           int64_t ldo, int64_t dim_size, const int32_t* __restrict__ nzero,
                           const float* __restrict__ others, float* __restrict__ grad_src,
                           int64_t ldg) {
  const int64_t total = n * F;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t e = t / F;
    const int64_t f = t - e * F;
    const int64_t g = index[e];
    float r = 0.f;
    if (g >= 0 && g < dim_size) {
      const float v = src[e * lds + f];
      const float go = grad_out[g * ldo + f];
      if (v != 0.f) {
        r = (go * out[g * ldo + f]) / v;  // out == 0 as soon as the group holds a zero
      } else if (nzero[g * F + f] == 1) {
        r = go * others[g * F + f];
      }
    }
    grad_src[e * ldg + f] = r;
  }
}

// ---- fused gather -> scale -> scatter-add on an UNSORTED edge list (no [E, F] intermediate) ---
// out[scatter_idx[e], :] += scale[gather_idx[e]] * w[e] * x[gather_idx[e], :]
// The edge-parallel fallback of the CSR SpMM for graphs that are used once (mini-batches): the
// backward of a destination-sorted batch would otherwise need a second sort.  fp32 atomics
// (native on CDNA), so the sum order is not deterministic.
template <typename IdxT, int VW>
__global__ void __launch_bounds__(kBlock)
    gather_scatter_add_kernel(const float* __restrict__ x, int64_t ldx,
                              const IdxT* __restrict__ gather_idx,
                              const IdxT* __restrict__ scatter_idx,
                              const float* __restrict__ scale, const float* __restrict__ w,
                              int64_t n_edges, const int64_t* __restrict__ n_valid, int64_t units,
                              int64_t F, float* __restrict__ out, int64_t ldo) {
  // (n_valid: device-side count of the real entries of a fixed-capacity edge list — a sampled
  // hop inside a captured graph)
  if (n_valid && *n_valid < n_edges) n_edges = *n_valid;
  const int64_t total = n_edges * units;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t e = t / units;
    const int64_t f = (t - e * units) * VW;
    if (f >= F) continue;
    const int64_t j = gather_idx[e];
    const int64_t i = scatter_idx[e];
    float m = 1.f;
    if (scale) m *= scale[j];
    if (w) m *= w[e];
    const Vec<VW> v = load_vec<VW>(x + j * ldx + f);
    float* dst = out + i * ldo + f;
#pragma unroll
    for (int q = 0; q < VW; ++q) atomicAdd(dst + q, v.v[q] * m);
  }
}

// ---- scatter_argmax (1-D) -------------------------------------------------------------------
template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    argmax_init_kernel(float* __restrict__ gmax, IdxT* __restrict__ arg, int64_t dim_size) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (g < dim_size) {
    gmax[g] = -INFINITY;
    arg[g] = -1;
  }
}
template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    argmax_max_kernel(const float* __restrict__ src, const IdxT* __restrict__ index, int64_t n,
                      int64_t dim_size, float* __restrict__ gmax) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= n) return;
  const int64_t g = index[e];
  if (g >= 0 && g < dim_size) atomic_max_f32(gmax + g, src[e]);
}
template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    argmax_pick_kernel(const float* __restrict__ src, const IdxT* __restrict__ index, int64_t n,
                       int64_t dim_size, const float* __restrict__ gmax,
                       IdxT* __restrict__ arg) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= n) return;
  const int64_t g = index[e];
  if (g >= 0 && g < dim_size && src[e] == gmax[g]) {
    if (sizeof(IdxT) == 8) {
      atomicMax(reinterpret_cast<long long*>(arg + g), static_cast<long long>(e));
    } else {
      atomicMax(reinterpret_cast<int*>(arg + g), static_cast<int>(e));
    }
  }
}
template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
    argmax_fix_kernel(IdxT* __restrict__ arg, int64_t dim_size) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (g < dim_size && arg[g] < 0) arg[g] = static_cast<IdxT>(dim_size - 1);
}

// ---- column sum (bias gradient), optionally fused with the ReLU backward mask ------------------
// A 256-thread block covers `groups = 256 / F` rows per pass with thread t on column t % F, so
// consecutive lanes read consecutive addresses of a contiguous [rows, F] block; row-group
// partials meet in LDS and each block issues F atomics.  MASK: v = act <= 0 ? 0 : x (what
// aten::threshold_backward computes), written to y before it is summed — one pass instead of
// mask (read 2, write 1) + column sum (read 1).
template <bool MASK>
__global__ void __launch_bounds__(kBlock)
    colsum_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ act,
                  int64_t lda, float* __restrict__ y, int64_t ldy, int64_t n_rows, int64_t F,
                  float* __restrict__ out) {
  __shared__ float part[kBlock];
  for (int64_t f0 = 0; f0 < F; f0 += kBlock) {
    const int width = static_cast<int>(F - f0 < kBlock ? F - f0 : kBlock);
    const int groups = kBlock / width;
    const int col = threadIdx.x % width;
    const int rg = threadIdx.x / width;
    float acc = 0.f;
    if (rg < groups) {
      for (int64_t r = static_cast<int64_t>(blockIdx.x) * groups + rg; r < n_rows;
           r += static_cast<int64_t>(gridDim.x) * groups) {
        float v = x[r * ldx + f0 + col];
        if constexpr (MASK) {
          v = act[r * lda + f0 + col] <= 0.f ? 0.f : v;
          y[r * ldy + f0 + col] = v;
        }
        acc += v;
      }
    }
    if (out != nullptr) {
      part[threadIdx.x] = acc;
      __syncthreads();
      if (rg == 0) {
        float s = 0.f;
        for (int g = 0; g < groups; ++g) s += part[g * width + col];
        atomicAdd(out + f0 + col, s);
      }
      __syncthreads();
    }
  }
}

// 16-byte variant for F % 4 == 0 (the layer widths): a thread owns 4 consecutive columns, a
// BLOCK-thread workgroup covers 4 BLOCK / F rows per pass with full-width coalesced accesses.
constexpr int kColsumBlock = 1024;

template <bool MASK, int U, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
    colsum_vec4_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ act,
                       int64_t lda, float* __restrict__ y, int64_t ldy, int64_t n_rows, int64_t F,
                       float* __restrict__ out) {
  __shared__ float part[BLOCK][4];
  const int units = static_cast<int>(F / 4);          // 16-byte pieces per row (<= 256 here)
  const int groups = BLOCK / units;                    // rows per pass
  const int u = threadIdx.x % units;
  const int rg = threadIdx.x / units;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (rg < groups) {
    // four row groups in flight per thread (one 16-byte load per operand at a time left the pass
    // latency-bound: 0.42 ms for a [169 k, 256] block, 2.5 x its traffic time)
    const int64_t stride = static_cast<int64_t>(gridDim.x) * groups;
    for (int64_t r0 = static_cast<int64_t>(blockIdx.x) * groups + rg; r0 < n_rows;
         r0 += U * stride) {
      Vec<4> v[U], a[U];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int64_t r = r0 + q * stride;
        const int64_t rc = r < n_rows ? r : r0;
        v[q] = load_vec_streamed<4>(x + rc * ldx + 4 * u);
        if constexpr (MASK) a[q] = load_vec_streamed<4>(act + rc * lda + 4 * u);
      }
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int64_t r = r0 + q * stride;
        if (r < n_rows) {
          if constexpr (MASK) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[q].v[i] = a[q].v[i] <= 0.f ? 0.f : v[q].v[i];
            store_vec<4>(y + r * ldy + 4 * u, v[q]);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += v[q].v[i];
        }
      }
    }
  }
  if (out != nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) part[threadIdx.x][i] = acc[i];
    __syncthreads();
    if (rg == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float s = 0.f;
        for (int g = 0; g < groups; ++g) s += part[g * units + u][i];
        atomicAdd(out + 4 * u + i, s);
      }
    }
  }
}

static unsigned flat_grid(int64_t total) {
  int64_t blocks = ceil_div(total, kBlock);
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 workgroups per CU
  return static_cast<unsigned>(blocks);
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename IdxT, int VW>
static int launch_scatter(const float* src, int64_t lds, const IdxT* idx, int64_t n, int64_t F,
                          float* out, int64_t ldo, int64_t dim_size, int reduce, float* count,
                          int32_t* err_flag, hipStream_t st) {
  const int64_t units = F / VW;
  const dim3 grid(flat_grid(n * units));
#define PYGAMD_SCATTER_CASE(R)                                                                  \
  case R:                                                                                       \
    hipLaunchKernelGGL((scatter_rows_kernel<IdxT, R, VW>), grid, dim3(kBlock), 0, st, src, lds, \
                       idx, n, units, F, out, ldo, dim_size, count, err_flag);                  \
    break;
  switch (reduce) {
    PYGAMD_SCATTER_CASE(PYGAMD_SUM)
    PYGAMD_SCATTER_CASE(PYGAMD_MEAN)
    PYGAMD_SCATTER_CASE(PYGAMD_MIN)
    PYGAMD_SCATTER_CASE(PYGAMD_MAX)
    PYGAMD_SCATTER_CASE(PYGAMD_MUL)
    PYGAMD_SCATTER_CASE(PYGAMD_ANY)
    default:
      return PYGAMD_ERR_INVALID_ARG;
  }
#undef PYGAMD_SCATTER_CASE
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

}  // namespace pygamd

using namespace pygamd;

// out = act(x + bias): the tail of a conv layer (gat_conv.py:378-385 / gcn_conv.py:278-281 `out + bias`
// followed by the model's ReLU, basic_gnn.py:262-263) in ONE pass instead of two ATen passes
template <int VW>
__global__ void __launch_bounds__(kBlock)
    bias_act_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ bias,
                    int64_t n_rows, int64_t units, int relu, float* __restrict__ out, int64_t ldo) {
  const int64_t total = n_rows * units;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;  // (flat_grid caps the grid)
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total;
       t += stride) {
    const int64_t r = t / units;
    const int64_t c = (t - r * units) * VW;
    Vec<VW> v = load_vec<VW>(x + r * ldx + c);
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      float y = v.v[i] + (bias ? bias[c + i] : 0.f);
      if (relu) y = (y > 0.f || y != y) ? y : 0.f;  // NaN propagates like torch.relu
      v.v[i] = y;
    }
    store_vec<VW>(out + r * ldo + c, v);
  }
}

extern "C" {

int pygamd_bias_act(const float* x, int64_t ldx, const float* bias, int64_t n_rows, int64_t F,
                    int relu, float* out, int64_t ldo, void* stream) {
  if (n_rows < 0 || F < 0 || ldx < F || ldo < F) return PYGAMD_ERR_INVALID_ARG;
  if (n_rows == 0 || F == 0) return PYGAMD_OK;
  if (!x || !out) return PYGAMD_ERR_INVALID_ARG;
  const bool v4 = (F % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && aligned16(x) &&
                  aligned16(out);
  if (v4) {
    hipLaunchKernelGGL(bias_act_kernel<4>, dim3(flat_grid(n_rows * (F / 4))), dim3(kBlock), 0,
                       as_stream(stream), x, ldx, bias, n_rows, F / 4, relu ? 1 : 0, out, ldo);
  } else {
    hipLaunchKernelGGL(bias_act_kernel<1>, dim3(flat_grid(n_rows * F)), dim3(kBlock), 0,
                       as_stream(stream), x, ldx, bias, n_rows, F, relu ? 1 : 0, out, ldo);
  }
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

static int launch_colsum(const float* x, int64_t ldx, const float* act, int64_t lda, float* y,
                         int64_t ldy, int64_t n_rows, int64_t F, float* out, hipStream_t st) {
  if (out) PYGAMD_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(float) * F, st));
  if (n_rows == 0) return PYGAMD_OK;
  const bool v4 = (F % 4 == 0) && (F / 4 <= kBlock) && (ldx % 4 == 0) && aligned16(x) &&
                  (!act || ((lda % 4 == 0) && (ldy % 4 == 0) && aligned16(act) && aligned16(y)));
  if (v4) {
    // Every workgroup ends in F atomics on the same F addresses and the workgroups of a resident
    // grid finish together: with 1024 workgroups of 256 lanes that burst was half of the launch
    // ([169 k, 256]: 181 us with the ReLU mask, 128 us without; 118 / 57 us with 256 workgroups).
    // One workgroup of 1024 lanes per CU keeps the loads in flight and a quarter of the atomics:
    // 117 / 55 us, and 26 instead of 56 us for a [14.5 k, 500] block.
    const int groups = kColsumBlock / static_cast<int>(F / 4);
    int64_t blocks = ceil_div(n_rows, static_cast<int64_t>(groups) * 16);
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    const dim3 grid(static_cast<unsigned>(blocks));
    if (act) {
      hipLaunchKernelGGL((colsum_vec4_kernel<true, 4, kColsumBlock>), grid, dim3(kColsumBlock), 0,
                         st, x, ldx, act, lda, y, ldy, n_rows, F, out);
    } else {
      hipLaunchKernelGGL((colsum_vec4_kernel<false, 4, kColsumBlock>), grid, dim3(kColsumBlock), 0,
                         st, x, ldx, act, lda, y, ldy, n_rows, F, out);
    }
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  }
  const int width = static_cast<int>(F < kBlock ? F : kBlock);
  const int groups = kBlock / width;
  int64_t blocks = ceil_div(n_rows, static_cast<int64_t>(groups) * 16);
  if (blocks > 256) blocks = 256;  // (the same burst of closing atomics as above)
  if (blocks < 1) blocks = 1;
  const dim3 grid(static_cast<unsigned>(blocks));
  if (act) {
    hipLaunchKernelGGL(colsum_kernel<true>, grid, dim3(kBlock), 0, st, x, ldx, act, lda, y, ldy,
                       n_rows, F, out);
  } else {
    hipLaunchKernelGGL(colsum_kernel<false>, grid, dim3(kBlock), 0, st, x, ldx, act, lda, y, ldy,
                       n_rows, F, out);
  }
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

int pygamd_colsum(const float* x, int64_t ldx, int64_t n_rows, int64_t F, float* out,
                  void* stream) {
  if (n_rows < 0 || F < 0 || ldx < F) return PYGAMD_ERR_INVALID_ARG;
  if (F == 0) return PYGAMD_OK;
  if (!out || (n_rows > 0 && !x)) return PYGAMD_ERR_INVALID_ARG;
  return launch_colsum(x, ldx, nullptr, 0, nullptr, 0, n_rows, F, out, as_stream(stream));
}

int pygamd_relu_backward_colsum(const float* grad, int64_t ldg, const float* act, int64_t lda,
                                int64_t n_rows, int64_t F, float* grad_in, int64_t ldo,
                                float* colsum_out, void* stream) {
  if (n_rows < 0 || F < 0 || ldg < F || lda < F || ldo < F) return PYGAMD_ERR_INVALID_ARG;
  if (F == 0) return PYGAMD_OK;
  if (n_rows > 0 && (!grad || !act || !grad_in)) return PYGAMD_ERR_INVALID_ARG;
  return launch_colsum(grad, ldg, act, lda, grad_in, ldo, n_rows, F, colsum_out,
                       as_stream(stream));
}

int pygamd_gather_rows(const float* x, int64_t ldx, int64_t n_src, const void* index,
                       int idx_dtype, int64_t n, int64_t F, float* out, int64_t ldo,
                       int32_t* err_flag, void* stream) {
  if (n < 0 || F < 0 || n_src < 0 || ldx < F || ldo < F) return PYGAMD_ERR_INVALID_ARG;
  if (n == 0 || F == 0) return PYGAMD_OK;
  if (!x || !index || !out) return PYGAMD_ERR_INVALID_ARG;
  const bool v4 = (F % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && aligned16(x) &&
                  aligned16(out);
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    if (v4) {
      const int64_t units = F / 4;
      hipLaunchKernelGGL((gather_rows_kernel<IdxT, 4>), dim3(flat_grid(n * units)), dim3(kBlock),
                         0, as_stream(stream), x, ldx, n_src, static_cast<const IdxT*>(index), n,
                         units, F, out, ldo, err_flag);
    } else {
      hipLaunchKernelGGL((gather_rows_kernel<IdxT, 1>), dim3(flat_grid(n * F)), dim3(kBlock), 0,
                         as_stream(stream), x, ldx, n_src, static_cast<const IdxT*>(index), n, F,
                         F, out, ldo, err_flag);
    }
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_gather_scatter_add(const float* x, int64_t ldx, const void* gather_idx,
                              const void* scatter_idx, int idx_dtype, const float* scale,
                              const float* w, int64_t n_edges, const int64_t* n_valid, int64_t F,
                              float* out, int64_t ldo, void* stream) {
  if (n_edges < 0 || F < 0 || ldx < F || ldo < F) return PYGAMD_ERR_INVALID_ARG;
  if (n_edges == 0 || F == 0) return PYGAMD_OK;
  if (!x || !gather_idx || !scatter_idx || !out) return PYGAMD_ERR_INVALID_ARG;
  const bool v4 = (F % 4 == 0) && (ldx % 4 == 0) && aligned16(x);
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    if (v4) {
      const int64_t units = F / 4;
      hipLaunchKernelGGL((gather_scatter_add_kernel<IdxT, 4>), dim3(flat_grid(n_edges * units)),
                         dim3(kBlock), 0, as_stream(stream), x, ldx,
                         static_cast<const IdxT*>(gather_idx),
                         static_cast<const IdxT*>(scatter_idx), scale, w, n_edges, n_valid, units,
                         F, out, ldo);
    } else {
      hipLaunchKernelGGL((gather_scatter_add_kernel<IdxT, 1>), dim3(flat_grid(n_edges * F)),
                         dim3(kBlock), 0, as_stream(stream), x, ldx,
                         static_cast<const IdxT*>(gather_idx),
                         static_cast<const IdxT*>(scatter_idx), scale, w, n_edges, n_valid, F, F,
                         out, ldo);
    }
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_scatter_init(float* out, int64_t ldo, int64_t dim_size, int64_t F, int reduce,
                        float* count, void* stream) {
  if (dim_size < 0 || F < 0 || ldo < F) return PYGAMD_ERR_INVALID_ARG;
  if (reduce < PYGAMD_SUM || reduce > PYGAMD_ANY) return PYGAMD_ERR_INVALID_ARG;
  if (dim_size == 0) return PYGAMD_OK;
  hipStream_t st = as_stream(stream);
  if (count) PYGAMD_HIP_CHECK(hipMemsetAsync(count, 0, sizeof(float) * dim_size, st));
  if (F == 0) return PYGAMD_OK;
  if (!out) return PYGAMD_ERR_INVALID_ARG;
  float v = 0.f;
  if (reduce == PYGAMD_MAX) v = -INFINITY;
  if (reduce == PYGAMD_MIN) v = INFINITY;
  if (reduce == PYGAMD_MUL) v = 1.f;
  hipLaunchKernelGGL(fill_rows_kernel, dim3(flat_grid(dim_size * F)), dim3(kBlock), 0, st, out,
                     ldo, dim_size, F, v);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

int pygamd_scatter_rows(const float* src, int64_t lds, const void* index, int idx_dtype,
                        int64_t n, int64_t F, float* out, int64_t ldo, int64_t dim_size,
                        int reduce, float* count, int32_t* err_flag, void* stream) {
  if (n < 0 || F < 0 || dim_size < 0 || lds < F || ldo < F) return PYGAMD_ERR_INVALID_ARG;
  if (reduce < PYGAMD_SUM || reduce > PYGAMD_ANY) return PYGAMD_ERR_INVALID_ARG;
  if ((reduce == PYGAMD_MEAN || reduce == PYGAMD_MIN || reduce == PYGAMD_MAX) && !count &&
      dim_size > 0)
    return PYGAMD_ERR_INVALID_ARG;
  if (n == 0 || F == 0) return PYGAMD_OK;
  if (!src || !index || !out) return PYGAMD_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  const bool v4 = (F % 4 == 0) && (lds % 4 == 0) && aligned16(src);
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    const IdxT* idx = static_cast<const IdxT*>(index);
    return v4 ? launch_scatter<IdxT, 4>(src, lds, idx, n, F, out, ldo, dim_size, reduce, count,
                                        err_flag, st)
              : launch_scatter<IdxT, 1>(src, lds, idx, n, F, out, ldo, dim_size, reduce, count,
                                        err_flag, st);
  });
}

int pygamd_scatter_finalize(float* out, int64_t ldo, int64_t dim_size, int64_t F, int reduce,
                            const float* count, void* stream) {
  if (dim_size < 0 || F < 0 || ldo < F) return PYGAMD_ERR_INVALID_ARG;
  if (dim_size == 0 || F == 0) return PYGAMD_OK;
  if (reduce != PYGAMD_MEAN && reduce != PYGAMD_MIN && reduce != PYGAMD_MAX) return PYGAMD_OK;
  if (!out || !count) return PYGAMD_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  const dim3 grid(flat_grid(dim_size * F));
  if (reduce == PYGAMD_MEAN) {
    hipLaunchKernelGGL((scatter_finalize_kernel<PYGAMD_MEAN>), grid, dim3(kBlock), 0, st, out,
                       ldo, dim_size, F, count);
  } else {
    hipLaunchKernelGGL((scatter_finalize_kernel<PYGAMD_MAX>), grid, dim3(kBlock), 0, st, out,
                       ldo, dim_size, F, count);
  }
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

int pygamd_scatter_minmax_tie_count(const float* src, int64_t lds, const void* index,
                                    int idx_dtype, int64_t n, int64_t F, const float* out,
                                    int64_t ldo, int64_t dim_size, float* ntie, void* stream) {
  if (n < 0 || F < 0 || dim_size < 0) return PYGAMD_ERR_INVALID_ARG;
  if (dim_size == 0 || F == 0) return PYGAMD_OK;
  if (!out || !ntie) return PYGAMD_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(tie_init_kernel, dim3(flat_grid(dim_size * F)), dim3(kBlock), 0, st, out,
                     ldo, dim_size, F, ntie);
  PYGAMD_LAUNCH_CHECK();
  if (n == 0) return PYGAMD_OK;
  if (!src || !index) return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    hipLaunchKernelGGL((scatter_tie_count_kernel<IdxT>), dim3(flat_grid(n * F)), dim3(kBlock), 0,
                       st, src, lds, static_cast<const IdxT*>(index), n, F, out, ldo, dim_size,
                       ntie);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_scatter_minmax_backward(const float* src, int64_t lds, const void* index,
                                   int idx_dtype, int64_t n, int64_t F, const float* out,
                                   const float* grad_out, const float* ntie, int64_t ldo,
                                   int64_t dim_size, float* grad_src, int64_t ldg,
                                   void* stream) {
  if (n < 0 || F < 0 || dim_size < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n == 0 || F == 0) return PYGAMD_OK;
  if (!src || !index || !grad_src) return PYGAMD_ERR_INVALID_ARG;
  if (dim_size > 0 && (!out || !grad_out || !ntie)) return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    hipLaunchKernelGGL((scatter_minmax_bwd_kernel<IdxT>), dim3(flat_grid(n * F)), dim3(kBlock),
                       0, as_stream(stream), src, lds, static_cast<const IdxT*>(index), n, F, out,
                       grad_out, ntie, ldo, dim_size, grad_src, ldg);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_scatter_mul_backward_workspace_bytes(int64_t dim_size, int64_t F, size_t* bytes) {
  if (!bytes || dim_size < 0 || F < 0) return PYGAMD_ERR_INVALID_ARG;
  *bytes = static_cast<size_t>(dim_size) * static_cast<size_t>(F) * 8;  // int32 count + float
  return PYGAMD_OK;
}

int pygamd_scatter_mul_backward(const float* src, int64_t lds, const void* index, int idx_dtype,
                                int64_t n, int64_t F, const float* out, const float* grad_out,
                                int64_t ldo, int64_t dim_size, float* grad_src, int64_t ldg,
                                void* workspace, size_t workspace_bytes, void* stream) {
  if (n < 0 || F < 0 || dim_size < 0 || lds < F || ldo < F || ldg < F)
    return PYGAMD_ERR_INVALID_ARG;
  if (n == 0 || F == 0) return PYGAMD_OK;
  if (!src || !index || !grad_src) return PYGAMD_ERR_INVALID_ARG;
  if (dim_size > 0 && (!out || !grad_out)) return PYGAMD_ERR_INVALID_ARG;
  size_t need = 0;
  pygamd_scatter_mul_backward_workspace_bytes(dim_size, F, &need);
  if (need > 0 && (!workspace || workspace_bytes < need)) return PYGAMD_ERR_WORKSPACE;
  hipStream_t st = as_stream(stream);
  int32_t* nzero = static_cast<int32_t*>(workspace);
  float* others = reinterpret_cast<float*>(nzero + dim_size * F);
  if (dim_size > 0) {
    PYGAMD_HIP_CHECK(hipMemsetAsync(nzero, 0, sizeof(int32_t) * dim_size * F, st));
    hipLaunchKernelGGL(fill_rows_kernel, dim3(flat_grid(dim_size * F)), dim3(kBlock), 0, st,
                       others, F, dim_size, F, 1.f);
    PYGAMD_LAUNCH_CHECK();
  }
  const dim3 grid(flat_grid(n * F));
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    const IdxT* idx = static_cast<const IdxT*>(index);
    hipLaunchKernelGGL((scatter_mul_zero_count_kernel<IdxT>), grid, dim3(kBlock), 0, st, src, lds,
                       idx, n, F, dim_size, nzero);
    PYGAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL((scatter_mul_others_kernel<IdxT>), grid, dim3(kBlock), 0, st, src, lds,
                       idx, n, F, dim_size, nzero, others);
    PYGAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL((scatter_mul_bwd_kernel<IdxT>), grid, dim3(kBlock), 0, st, src, lds, idx,
                       n, F, out, grad_out, ldo, dim_size, nzero, others, grad_src, ldg);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_scatter_argmax(const float* src, const void* index, int idx_dtype, int64_t n,
                          int64_t dim_size, float* gmax, void* arg_out, void* stream) {
  if (n < 0 || dim_size < 0) return PYGAMD_ERR_INVALID_ARG;
  if (dim_size == 0) return PYGAMD_OK;
  if (!gmax || !arg_out) return PYGAMD_ERR_INVALID_ARG;
  if (n > 0 && (!src || !index)) return PYGAMD_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    const IdxT* idx = static_cast<const IdxT*>(index);
    IdxT* arg = static_cast<IdxT*>(arg_out);
    const dim3 ggrid(static_cast<unsigned>(ceil_div(dim_size, kBlock)));
    hipLaunchKernelGGL((argmax_init_kernel<IdxT>), ggrid, dim3(kBlock), 0, st, gmax, arg,
                       dim_size);
    PYGAMD_LAUNCH_CHECK();
    if (n > 0) {
      const dim3 egrid(static_cast<unsigned>(ceil_div(n, kBlock)));
      hipLaunchKernelGGL((argmax_max_kernel<IdxT>), egrid, dim3(kBlock), 0, st, src, idx, n,
                         dim_size, gmax);
      PYGAMD_LAUNCH_CHECK();
      hipLaunchKernelGGL((argmax_pick_kernel<IdxT>), egrid, dim3(kBlock), 0, st, src, idx, n,
                         dim_size, gmax, arg);
      PYGAMD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((argmax_fix_kernel<IdxT>), ggrid, dim3(kBlock), 0, st, arg, dim_size);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

}  // extern "C"
