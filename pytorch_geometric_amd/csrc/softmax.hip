// softmax.hip — per-destination ("edge") softmax on a sorted handle, and the fused GAT logits.
// Reference: torch_geometric/utils/_softmax.py:60-81 (ptr branch; max of the detached input,
// +1e-16 on the denominator) and nn/conv/gat_conv.py:387-406.
//
// One wavefront owns one segment.  A segment's [len, H] block is contiguous, so for H a power of
// two <= 64 lane l always sees column l % H and walks the block with stride 64 (fully coalesced);
// column statistics are combined across the lanes that share a column with xor-shuffles at
// distances H, 2H, ... 32.  Other H use one lane per column.  Three passes over the block; the
// second and third hit L1/L2 (a block is deg*H*4 bytes).
#include "common.h"

namespace pygamd {

__device__ __forceinline__ bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

template <bool IS_MAX>
__device__ __forceinline__ float column_reduce(float v, int H) {
  for (int off = H; off < kWave; off <<= 1) {
    const float o = __shfl_xor(v, off, kWave);
    v = IS_MAX ? fmaxf(v, o) : v + o;
  }
  return v;
}

// Generic "values of segment `seg`" accessor: Plain reads src[k*H+h]; Gat builds the logit.
template <typename IdxT>
struct PlainLoader {
  const float* __restrict__ src;
  __device__ __forceinline__ void begin(int64_t) const {}
  __device__ __forceinline__ float at(int64_t k, int h, int64_t H) const {
    return src[k * H + h];
  }
};

template <typename IdxT>
struct GatLoader {
  const IdxT* __restrict__ col;
  const float* __restrict__ alpha_src;
  const float* __restrict__ alpha_dst;
  float slope;
  int64_t row;
  __device__ __forceinline__ void begin(int64_t r) { row = r; }
  __device__ __forceinline__ float pre(int64_t k, int h, int64_t H) const {
    return alpha_src[static_cast<int64_t>(col[k]) * H + h] + alpha_dst[row * H + h];
  }
  __device__ __forceinline__ float at(int64_t k, int h, int64_t H) const {
    const float p = pre(k, h, H);
    return p > 0.f ? p : p * slope;
  }
};

template <typename IdxT, typename Loader>
__global__ void __launch_bounds__(kBlock)
    segment_softmax_fwd_kernel(Loader ld, const IdxT* __restrict__ ptr, int64_t n_seg, int64_t H,
                               float* __restrict__ out) {
  const int lane = lane_id();
  const int64_t seg = xcd_logical_block() * kWavesPerBlock + wave_in_block();
  if (seg >= n_seg) return;
  const int64_t s = ptr[seg];
  const int64_t e = ptr[seg + 1];
  if (e <= s) return;
  ld.begin(seg);
  if (H <= kWave && is_pow2(H)) {
    const int h = lane % static_cast<int>(H);
    const int64_t k0 = s + lane / H;
    const int64_t kstep = kWave / H;
    float m = -INFINITY;
    for (int64_t k = k0; k < e; k += kstep) m = fmaxf(m, ld.at(k, h, H));
    m = column_reduce<true>(m, static_cast<int>(H));
    float sum = 0.f;
    for (int64_t k = k0; k < e; k += kstep) sum += expf(ld.at(k, h, H) - m);
    sum = column_reduce<false>(sum, static_cast<int>(H)) + 1e-16f;
    for (int64_t k = k0; k < e; k += kstep) out[k * H + h] = expf(ld.at(k, h, H) - m) / sum;
  } else {
    for (int64_t h = lane; h < H; h += kWave) {
      float m = -INFINITY;
      for (int64_t k = s; k < e; ++k) m = fmaxf(m, ld.at(k, static_cast<int>(h), H));
      float sum = 0.f;
      for (int64_t k = s; k < e; ++k) sum += expf(ld.at(k, static_cast<int>(h), H) - m);
      sum += 1e-16f;
      for (int64_t k = s; k < e; ++k)
        out[k * H + h] = expf(ld.at(k, static_cast<int>(h), H) - m) / sum;
    }
  }
}

// grad_src[k,h] = out[k,h] * (g[k,h] - sum_seg(out*g)[h]);  GAT: additionally through the
// leaky-relu and into grad_alpha_dst (row-owned, plain store) / grad_alpha_src (atomics).
template <typename IdxT, bool GAT>
__global__ void __launch_bounds__(kBlock)
    segment_softmax_bwd_kernel(const float* __restrict__ out, const float* __restrict__ g,
                               const IdxT* __restrict__ ptr, int64_t n_seg, int64_t H,
                               float* __restrict__ grad_src, GatLoader<IdxT> gat,
                               float* __restrict__ grad_alpha_src,
                               float* __restrict__ grad_alpha_dst) {
  const int lane = lane_id();
  const int64_t seg = xcd_logical_block() * kWavesPerBlock + wave_in_block();
  if (seg >= n_seg) return;
  const int64_t s = ptr[seg];
  const int64_t e = ptr[seg + 1];
  if (GAT) gat.begin(seg);
  if (H <= kWave && is_pow2(H)) {
    const int h = lane % static_cast<int>(H);
    const int64_t k0 = s + lane / H;
    const int64_t kstep = kWave / H;
    float dot = 0.f;
    for (int64_t k = k0; k < e; k += kstep) dot = fmaf(out[k * H + h], g[k * H + h], dot);
    dot = column_reduce<false>(dot, static_cast<int>(H));
    float dsum = 0.f;
    for (int64_t k = k0; k < e; k += kstep) {
      float gs = out[k * H + h] * (g[k * H + h] - dot);
      if (GAT) {
        const float p = gat.pre(k, h, H);
        gs = p > 0.f ? gs : gs * gat.slope;
        dsum += gs;
        atomicAdd(grad_alpha_src + static_cast<int64_t>(gat.col[k]) * H + h, gs);
      } else {
        grad_src[k * H + h] = gs;
      }
    }
    if (GAT) {
      dsum = column_reduce<false>(dsum, static_cast<int>(H));
      if (lane < H) grad_alpha_dst[seg * H + h] = dsum;
    }
  } else {
    for (int64_t h = lane; h < H; h += kWave) {
      float dot = 0.f;
      for (int64_t k = s; k < e; ++k) dot = fmaf(out[k * H + h], g[k * H + h], dot);
      float dsum = 0.f;
      for (int64_t k = s; k < e; ++k) {
        float gs = out[k * H + h] * (g[k * H + h] - dot);
        if (GAT) {
          const float p = gat.pre(k, static_cast<int>(h), H);
          gs = p > 0.f ? gs : gs * gat.slope;
          dsum += gs;
          atomicAdd(grad_alpha_src + static_cast<int64_t>(gat.col[k]) * H + h, gs);
        } else {
          grad_src[k * H + h] = gs;
        }
      }
      if (GAT) grad_alpha_dst[seg * H + h] = dsum;
    }
  }
}

}  // namespace pygamd

using namespace pygamd;

extern "C" {

int pygamd_segment_softmax_forward(const float* src, const void* ptr, int idx_dtype,
                                   int64_t n_seg, int64_t H, float* out, void* stream) {
  if (n_seg < 0 || H < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n_seg == 0 || H == 0) return PYGAMD_OK;
  if (!src || !ptr || !out) return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    PlainLoader<IdxT> ld{src};
    hipLaunchKernelGGL((segment_softmax_fwd_kernel<IdxT, PlainLoader<IdxT>>),
                       dim3(wave_grid(n_seg)), dim3(kBlock), 0, as_stream(stream), ld,
                       static_cast<const IdxT*>(ptr), n_seg, H, out);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_segment_softmax_backward(const float* out, const float* grad_out, const void* ptr,
                                    int idx_dtype, int64_t n_seg, int64_t H, float* grad_src,
                                    void* stream) {
  if (n_seg < 0 || H < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n_seg == 0 || H == 0) return PYGAMD_OK;
  if (!out || !grad_out || !ptr || !grad_src) return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    GatLoader<IdxT> none{nullptr, nullptr, nullptr, 0.f, 0};
    hipLaunchKernelGGL((segment_softmax_bwd_kernel<IdxT, false>), dim3(wave_grid(n_seg)),
                       dim3(kBlock), 0, as_stream(stream), out, grad_out,
                       static_cast<const IdxT*>(ptr), n_seg, H, grad_src, none,
                       static_cast<float*>(nullptr), static_cast<float*>(nullptr));
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_gat_edge_softmax_forward(const void* rowptr, const void* col, int idx_dtype,
                                    const float* alpha_src, const float* alpha_dst,
                                    int64_t n_rows, int64_t H, float slope, float* alpha_out,
                                    void* stream) {
  if (n_rows < 0 || H < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n_rows == 0 || H == 0) return PYGAMD_OK;
  if (!rowptr || !col || !alpha_src || !alpha_dst || !alpha_out) return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    GatLoader<IdxT> ld{static_cast<const IdxT*>(col), alpha_src, alpha_dst, slope, 0};
    hipLaunchKernelGGL((segment_softmax_fwd_kernel<IdxT, GatLoader<IdxT>>),
                       dim3(wave_grid(n_rows)), dim3(kBlock), 0, as_stream(stream), ld,
                       static_cast<const IdxT*>(rowptr), n_rows, H, alpha_out);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_gat_edge_softmax_backward(const void* rowptr, const void* col, int idx_dtype,
                                     const float* alpha_src, const float* alpha_dst,
                                     const float* alpha_out, const float* grad_alpha,
                                     int64_t n_rows, int64_t H, float slope,
                                     float* grad_alpha_src, float* grad_alpha_dst,
                                     void* stream) {
  if (n_rows < 0 || H < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n_rows == 0 || H == 0) return PYGAMD_OK;
  if (!rowptr || !col || !alpha_src || !alpha_dst || !alpha_out || !grad_alpha ||
      !grad_alpha_src || !grad_alpha_dst)
    return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    GatLoader<IdxT> ld{static_cast<const IdxT*>(col), alpha_src, alpha_dst, slope, 0};
    hipLaunchKernelGGL((segment_softmax_bwd_kernel<IdxT, true>), dim3(wave_grid(n_rows)),
                       dim3(kBlock), 0, as_stream(stream), alpha_out, grad_alpha,
                       static_cast<const IdxT*>(rowptr), n_rows, H,
                       static_cast<float*>(nullptr), ld, grad_alpha_src, grad_alpha_dst);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

}  // extern "C"
