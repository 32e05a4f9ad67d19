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

// kSegPer segments per wave, one after the other — but with every segment's pointer reads, then
// every segment's first kSegKeep passes of values issued TOGETHER and kept in registers: a
// segment is a chain of dependent reads (ptr -> col -> alpha_src[col] for the GAT loader) and a
// graph of short rows (13 edges on average at the arxiv shape, 8 heads: two passes of a wave) is
// bound by how many chains the chip has in flight.  Longer segments re-read their tail.
constexpr int kSegPer = 4;
constexpr int kSegKeep = 2;

template <typename IdxT, typename Loader>
__global__ void __launch_bounds__(kBlock)
    segment_softmax_fwd_kernel(Loader ld, const IdxT* __restrict__ ptr, int64_t n_seg, int64_t H,
                               float* __restrict__ out) {
  const int lane = lane_id();
  if (H <= kWave && is_pow2(H)) {
    const int64_t seg0 = (xcd_logical_block() * kWavesPerBlock + wave_in_block()) * kSegPer;
    if (seg0 >= n_seg) return;
    const int h = lane % static_cast<int>(H);
    const int64_t kstep = kWave / H;
    int64_t sb[kSegPer], se[kSegPer];
#pragma unroll
    for (int q = 0; q < kSegPer; ++q) {
      const int64_t sg = seg0 + q < n_seg ? seg0 + q : n_seg - 1;
      sb[q] = ptr[sg];
      se[q] = seg0 + q < n_seg ? static_cast<int64_t>(ptr[sg + 1]) : sb[q];  // (past the end: empty)
    }
    float v[kSegPer][kSegKeep];
#pragma unroll
    for (int q = 0; q < kSegPer; ++q) {
      Loader lq = ld;
      lq.begin(seg0 + q < n_seg ? seg0 + q : n_seg - 1);
#pragma unroll
      for (int i = 0; i < kSegKeep; ++i) {
        const int64_t k = sb[q] + lane / H + i * kstep;
        v[q][i] = k < se[q] ? lq.at(k, h, H) : -INFINITY;
      }
    }
#pragma unroll
    for (int q = 0; q < kSegPer; ++q) {
      const int64_t s = sb[q], e = se[q];
      if (e <= s) continue;  // (wave-uniform)
      Loader lq = ld;
      lq.begin(seg0 + q);
      const int64_t k0 = s + lane / H;
      const int64_t kt = k0 + kSegKeep * kstep;  // first k behind the kept passes
      float m = -INFINITY;
#pragma unroll
      for (int i = 0; i < kSegKeep; ++i) m = fmaxf(m, v[q][i]);
      for (int64_t k = kt; k < e; k += kstep) m = fmaxf(m, lq.at(k, h, H));
      m = column_reduce<true>(m, static_cast<int>(H));
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < kSegKeep; ++i)
        if (k0 + i * kstep < e) sum += expf(v[q][i] - m);
      for (int64_t k = kt; k < e; k += kstep) sum += expf(lq.at(k, h, H) - m);
      sum = column_reduce<false>(sum, static_cast<int>(H)) + 1e-16f;
#pragma unroll
      for (int i = 0; i < kSegKeep; ++i) {
        const int64_t k = k0 + i * kstep;
        if (k < e) out[k * H + h] = expf(v[q][i] - m) / sum;
      }
      for (int64_t k = kt; k < e; k += kstep) out[k * H + h] = expf(lq.at(k, h, H) - m) / sum;
    }
    return;
  }
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

// ---- segment_logsumexp (utils/_segment.py:53-80) -------------------------------------------------
// out[seg, h] = log(sum_{k in seg} exp(src[k, h] - m)) + m with m = the segment maximum (0 for an
// empty segment, whose result is 0: log(0) = -inf is mapped to 0 by the reference's
// nan_to_num(neginf=0)).  Same lane mapping as the softmax: for narrow power-of-two H the wave
// covers 64 / H rows per pass.  BWD: grad_src[k, h] = exp(src[k, h] - out[seg, h]) * g[seg, h].
template <typename IdxT, bool BWD>
__global__ void __launch_bounds__(kBlock)
    segment_logsumexp_kernel(const float* __restrict__ src, const IdxT* __restrict__ ptr,
                             int64_t n_seg, int64_t H, float* __restrict__ out,
                             const float* __restrict__ grad_out, float* __restrict__ grad_src) {
  const int lane = lane_id();
  const int64_t seg = xcd_logical_block() * kWavesPerBlock + wave_in_block();
  if (seg >= n_seg) return;
  const int64_t s = ptr[seg];
  const int64_t e = ptr[seg + 1];
  const bool narrow = H <= kWave && is_pow2(H);
  const int64_t hstep = narrow ? H : kWave;
  const int64_t kstep = narrow ? kWave / H : 1;
  const int64_t koff = narrow ? lane / H : 0;
  for (int64_t h = narrow ? lane % H : lane; h < H; h += hstep) {
    if (BWD) {
      const float lse = out[seg * H + h];
      const float g = grad_out[seg * H + h];
      for (int64_t k = s + koff; k < e; k += kstep)
        grad_src[k * H + h] = expf(src[k * H + h] - lse) * g;
    } else {
      float m = -INFINITY;
      for (int64_t k = s + koff; k < e; k += kstep) m = fmaxf(m, src[k * H + h]);
      if (narrow) m = column_reduce<true>(m, static_cast<int>(H));
      if (e <= s) m = 0.f;
      float sum = 0.f;
      for (int64_t k = s + koff; k < e; k += kstep) sum += expf(src[k * H + h] - m);
      if (narrow) sum = column_reduce<false>(sum, static_cast<int>(H));
      const float l = logf(sum);
      if (koff == 0) out[seg * H + h] = (l == -INFINITY ? 0.f : l) + m;
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
  if (H <= kWave && is_pow2(H)) {
    // kSegPer segments per wave, their first kSegKeep passes loaded together (see the forward)
    const int64_t seg0 = (xcd_logical_block() * kWavesPerBlock + wave_in_block()) * kSegPer;
    if (seg0 >= n_seg) return;
    const int h = lane % static_cast<int>(H);
    const int64_t kstep = kWave / H;
    int64_t sb[kSegPer], se[kSegPer];
#pragma unroll
    for (int q = 0; q < kSegPer; ++q) {
      const int64_t sg = seg0 + q < n_seg ? seg0 + q : n_seg - 1;
      sb[q] = ptr[sg];
      se[q] = seg0 + q < n_seg ? static_cast<int64_t>(ptr[sg + 1]) : sb[q];
    }
    float vo[kSegPer][kSegKeep], vg[kSegPer][kSegKeep], vp[kSegPer][kSegKeep];
    int64_t vc[kSegPer][kSegKeep];
#pragma unroll
    for (int q = 0; q < kSegPer; ++q) {
#pragma unroll
      for (int i = 0; i < kSegKeep; ++i) {
        const int64_t k = sb[q] + lane / H + i * kstep;
        const bool ok = k < se[q];
        vo[q][i] = ok ? out[k * H + h] : 0.f;
        vg[q][i] = ok ? g[k * H + h] : 0.f;
        vc[q][i] = (GAT && ok) ? static_cast<int64_t>(gat.col[k]) : 0;
      }
    }
    if (GAT) {
#pragma unroll
      for (int q = 0; q < kSegPer; ++q) {
        const int64_t sg = seg0 + q < n_seg ? seg0 + q : n_seg - 1;
        const float ad = gat.alpha_dst[sg * H + h];
#pragma unroll
        for (int i = 0; i < kSegKeep; ++i) vp[q][i] = gat.alpha_src[vc[q][i] * H + h] + ad;
      }
    }
#pragma unroll
    for (int q = 0; q < kSegPer; ++q) {
      const int64_t s = sb[q], e = se[q];
      if (seg0 + q >= n_seg) break;  // (wave-uniform)
      if (GAT) gat.begin(seg0 + q);
      const int64_t k0 = s + lane / H;
      const int64_t kt = k0 + kSegKeep * kstep;
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < kSegKeep; ++i)
        if (k0 + i * kstep < e) dot = fmaf(vo[q][i], vg[q][i], dot);
      for (int64_t k = kt; k < e; k += kstep) dot = fmaf(out[k * H + h], g[k * H + h], dot);
      dot = column_reduce<false>(dot, static_cast<int>(H));
      float dsum = 0.f;
#pragma unroll
      for (int i = 0; i < kSegKeep; ++i) {
        const int64_t k = k0 + i * kstep;
        if (k < e) {
          float gs = vo[q][i] * (vg[q][i] - dot);
          if (GAT) {
            gs = vp[q][i] > 0.f ? gs : gs * gat.slope;
            dsum += gs;
            atomicAdd(grad_alpha_src + vc[q][i] * H + h, gs);
          } else {
            grad_src[k * H + h] = gs;
          }
        }
      }
      for (int64_t k = kt; k < e; k += kstep) {
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
        if (lane < H) grad_alpha_dst[(seg0 + q) * H + h] = dsum;
      }
    }
    return;
  }
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

// a[n,h] = sum_c x[n,h,c] att_a[h,c]  (and b with att_b): one wave per node, `lph` (a power of
// two) lanes per head, so a wave covers 64 / lph heads at once — for the 8-head GAT layers every
// lane is busy, the row is read once with coalesced (VW = 4: 16-byte) loads and each head costs
// log2(lph) shuffles instead of a full 64-lane butterfly.
template <int VW>
__global__ void __launch_bounds__(kBlock)
    head_dot_fwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ att_a,
                        const float* __restrict__ att_b, int64_t n_rows, int H, int C, int lph,
                        float* __restrict__ out_a, float* __restrict__ out_b) {
  const int lane = lane_id();
  const int64_t n = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block();
  if (n >= n_rows) return;
  const float* __restrict__ xr = x + n * ldx;
  const int heads_per_pass = kWave / lph;
  const int j = lane & (lph - 1);
  for (int h0 = 0; h0 < H; h0 += heads_per_pass) {
    const int h = h0 + lane / lph;
    float pa = 0.f, pb = 0.f;
    if (h < H) {
      for (int c = j * VW; c < C; c += lph * VW) {
        const int f = h * C + c;
        if constexpr (VW == 4) {
          const float4 v = *reinterpret_cast<const float4*>(xr + f);
          const float4 wa = *reinterpret_cast<const float4*>(att_a + f);
          pa = fmaf(v.x, wa.x, fmaf(v.y, wa.y, fmaf(v.z, wa.z, fmaf(v.w, wa.w, pa))));
          if (att_b) {
            const float4 wb = *reinterpret_cast<const float4*>(att_b + f);
            pb = fmaf(v.x, wb.x, fmaf(v.y, wb.y, fmaf(v.z, wb.z, fmaf(v.w, wb.w, pb))));
          }
        } else {
          const float v = xr[f];
          pa = fmaf(v, att_a[f], pa);
          if (att_b) pb = fmaf(v, att_b[f], pb);
        }
      }
    }
    for (int off = lph >> 1; off > 0; off >>= 1) {
      pa += __shfl_xor(pa, off, kWave);
      pb += __shfl_xor(pb, off, kWave);
    }
    if (h < H && j == 0) {
      out_a[n * H + h] = pa;
      if (att_b) out_b[n * H + h] = pb;
    }
  }
}

// grad_x[n,f] = ga[n,h(f)] att_a[f] + gb[n,h(f)] att_b[f];  grad_att_a[f] = sum_n ga[n,h(f)] x[n,f]
// Lane t of a 256-lane group owns column f = f0 + t for a strip of rows (coalesced row reads and
// writes, four rows in flight per lane); four such groups share a workgroup, their column sums
// meet in LDS and leave through one set of atomics (grad_att_* zeroed by the host wrapper).
constexpr int kHdSub = 4;  // row sub-strips per workgroup (kBlock lanes each)

__global__ void __launch_bounds__(kBlock * kHdSub)
    head_dot_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ att_a,
                        const float* __restrict__ att_b, const float* __restrict__ ga,
                        const float* __restrict__ gb, int64_t n_rows, int H, int C,
                        int64_t rows_per_block, float* __restrict__ grad_x, int64_t ldg,
                        int accumulate, float* __restrict__ grad_att_a,
                        float* __restrict__ grad_att_b) {
  __shared__ float red[2][kHdSub][kBlock];
  const int64_t F = static_cast<int64_t>(H) * C;
  const int tid = threadIdx.x % kBlock, sub = threadIdx.x / kBlock;
  const int64_t b0 = static_cast<int64_t>(blockIdx.x) * rows_per_block;
  int64_t b1 = b0 + rows_per_block;
  if (b1 > n_rows) b1 = n_rows;
  // the workgroup's strip in kHdSub pieces: as many rows in flight as four small workgroups, one
  // set of closing atomics (all workgroups of a resident grid finish together: 1024 workgroups x
  // 2 F atomics on the same addresses were a quarter of the launch)
  const int64_t piece = (b1 - b0 + kHdSub - 1) / kHdSub;
  const int64_t r0 = b0 + sub * piece;
  int64_t r1 = r0 + piece;
  if (r1 > b1) r1 = b1;
  for (int64_t fb = 0; fb < F; fb += kBlock) {
    const int64_t f = fb + tid;
    const bool live = f < F;
    float sa = 0.f, sb = 0.f;
    if (live) {
      const int h = static_cast<int>(f / C);
      const float wa = att_a[f];
      const float wb = att_b ? att_b[f] : 0.f;
      int64_t n = r0;
      for (; n + 4 <= r1; n += 4) {
        float xv[4], va[4], vb[4], old[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          xv[u] = x[(n + u) * ldx + f];
          va[u] = ga[(n + u) * H + h];
          vb[u] = gb ? gb[(n + u) * H + h] : 0.f;
          // (the values to add to, loaded with the operands: four rows in flight, not a
          // read-modify-write per row)
          old[u] = (grad_x && accumulate) ? grad_x[(n + u) * ldg + f] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          sa = fmaf(va[u], xv[u], sa);
          sb = fmaf(vb[u], xv[u], sb);
          if (grad_x) grad_x[(n + u) * ldg + f] = old[u] + (va[u] * wa + vb[u] * wb);
        }
      }
      for (; n < r1; ++n) {
        const float xv = x[n * ldx + f];
        const float va = ga[n * H + h];
        const float vb = gb ? gb[n * H + h] : 0.f;
        sa = fmaf(va, xv, sa);
        sb = fmaf(vb, xv, sb);
        if (grad_x) {
          float* gp = grad_x + n * ldg + f;
          const float v = va * wa + vb * wb;
          *gp = accumulate ? *gp + v : v;
        }
      }
    }
    red[0][sub][tid] = sa;
    red[1][sub][tid] = sb;
    __syncthreads();
    if (sub == 0 && live) {
      float ta = 0.f, tb = 0.f;
#pragma unroll
      for (int q = 0; q < kHdSub; ++q) {
        ta += red[0][q][tid];
        tb += red[1][q][tid];
      }
      atomicAdd(grad_att_a + f, ta);
      if (grad_att_b) atomicAdd(grad_att_b + f, tb);
    }
    __syncthreads();
  }
}

// ---- softmax over an UNSORTED index (utils/_softmax.py:82-88) ---------------------------------------
// The reference's index branch is scatter-max (detached) -> gather -> exp -> scatter-sum -> gather ->
// div: six passes over the [n, H] values (ten launches with the scatter prologues / epilogues).  Here:
//   init      gmax = -inf, gsum = 0                                   [N, H] each (workspace)
//   max       atomic float max of src[k, h] into gmax[index[k], h]
//   exp_sum   e = exp(src - gmax[index]); out = e; atomicAdd(gsum[index], e)
//   div       out /= gsum[index] + 1e-16
// and for the backward  s = scatter_sum(out * g);  grad = out * (g - s[index])  as init + 2 passes.
// One thread per (k, h) element; the group statistics are N x H floats (L2-resident for the
// attention shapes this serves).  Sums use fp32 atomics (order-dependent rounding, like the
// reference's scatter_add_ on a GPU).
__global__ void __launch_bounds__(kBlock)
    softmax_index_init_kernel(float* __restrict__ gmax, float* __restrict__ gsum, int64_t total) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= total) return;
  if (gmax) gmax[t] = -INFINITY;
  gsum[t] = 0.f;
}

template <typename IdxT, int PHASE>
__global__ void __launch_bounds__(kBlock)
    softmax_index_kernel(const float* __restrict__ src, const IdxT* __restrict__ index, int64_t n,
                         int64_t H, int64_t N, float* __restrict__ gmax, float* __restrict__ gsum,
                         float* __restrict__ out, const float* __restrict__ grad_out,
                         int32_t* __restrict__ err) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= n * H) return;
  const int64_t k = t / H;
  const int64_t h = t - k * H;
  const int64_t g = static_cast<int64_t>(index[k]);
  if (g < 0 || g >= N) {  // out of range: skipped and flagged, like the scatter kernels
    if (PHASE == 0 && err != nullptr && h == 0) *err = 1;
    return;
  }
  const int64_t s = g * H + h;
  if constexpr (PHASE == 0) {         // group maxima
    atomic_max_f32(gmax + s, src[t]);
  } else if constexpr (PHASE == 1) {  // exp + group sums
    const float e = expf(src[t] - gmax[s]);
    out[t] = e;
    atomicAdd(gsum + s, e);
  } else if constexpr (PHASE == 2) {  // normalise
    out[t] = out[t] / (gsum[s] + 1e-16f);
  } else if constexpr (PHASE == 3) {  // backward: group sums of out * g
    atomicAdd(gsum + s, src[t] * grad_out[t]);
  } else {                            // backward: grad = out * (g - sum)
    out[t] = src[t] * (grad_out[t] - gsum[s]);
  }
}

}  // namespace pygamd

using namespace pygamd;

// waves of segment_softmax_{fwd,bwd}_kernel: kSegPer segments each on the narrow power-of-two route
static int64_t softmax_fwd_waves(int64_t n_seg, int64_t H) {
  return (H > 0 && H <= kWave && (H & (H - 1)) == 0) ? ceil_div(n_seg, static_cast<int64_t>(kSegPer)) : n_seg;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" {

int pygamd_segment_softmax_forward(const float* src, const void* ptr, int idx_dtype,
                                   int64_t n_seg, int64_t H, float* out, void* stream) {
  if (n_seg < 0 || H < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n_seg == 0 || H == 0) return PYGAMD_OK;
  if (!src || !ptr || !out) return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    PlainLoader<IdxT> ld{src};
    hipLaunchKernelGGL((segment_softmax_fwd_kernel<IdxT, PlainLoader<IdxT>>),
                       dim3(wave_grid(softmax_fwd_waves(n_seg, H))), dim3(kBlock), 0,
                       as_stream(stream), ld,
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
    hipLaunchKernelGGL((segment_softmax_bwd_kernel<IdxT, false>),
                       dim3(wave_grid(softmax_fwd_waves(n_seg, H))),
                       dim3(kBlock), 0, as_stream(stream), out, grad_out,
                       static_cast<const IdxT*>(ptr), n_seg, H, grad_src, none,
                       static_cast<float*>(nullptr), static_cast<float*>(nullptr));
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_softmax_index_forward(const float* src, const void* index, int idx_dtype, int64_t n,
                                 int64_t H, int64_t N, float* workspace, float* out,
                                 int32_t* err, void* stream) {
  if (n < 0 || H < 0 || N < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n == 0 || H == 0) return PYGAMD_OK;
  if (!src || !index || !out || (N > 0 && !workspace)) return PYGAMD_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  if (N == 0) {  // every index is out of range
    if (err) PYGAMD_HIP_CHECK(hipMemsetAsync(err, 1, 1, st));
    return PYGAMD_OK;
  }
  float* gmax = workspace;
  float* gsum = workspace + N * H;
  const dim3 ginit(static_cast<unsigned>(ceil_div(N * H, kBlock)));
  const dim3 grid(static_cast<unsigned>(ceil_div(n * H, kBlock)));
  hipLaunchKernelGGL(softmax_index_init_kernel, ginit, dim3(kBlock), 0, st, gmax, gsum, N * H);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    const IdxT* idx = static_cast<const IdxT*>(index);
    int32_t* const no_flag = nullptr;
    hipLaunchKernelGGL((softmax_index_kernel<IdxT, 0>), grid, dim3(kBlock), 0, st, src, idx, n, H,
                       N, gmax, gsum, out, static_cast<const float*>(nullptr), err);
    hipLaunchKernelGGL((softmax_index_kernel<IdxT, 1>), grid, dim3(kBlock), 0, st, src, idx, n, H,
                       N, gmax, gsum, out, static_cast<const float*>(nullptr), no_flag);
    hipLaunchKernelGGL((softmax_index_kernel<IdxT, 2>), grid, dim3(kBlock), 0, st, src, idx, n, H,
                       N, gmax, gsum, out, static_cast<const float*>(nullptr), no_flag);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_softmax_index_backward(const float* out, const float* grad_out, const void* index,
                                  int idx_dtype, int64_t n, int64_t H, int64_t N,
                                  float* workspace, float* grad_src, void* stream) {
  if (n < 0 || H < 0 || N < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n == 0 || H == 0) return PYGAMD_OK;
  if (!out || !grad_out || !index || !grad_src || (N > 0 && !workspace))
    return PYGAMD_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  if (N == 0) {  // every index is out of range: zero gradient
    PYGAMD_HIP_CHECK(hipMemsetAsync(grad_src, 0, sizeof(float) * n * H, st));
    return PYGAMD_OK;
  }
  const dim3 ginit(static_cast<unsigned>(ceil_div(N * H, kBlock)));
  const dim3 grid(static_cast<unsigned>(ceil_div(n * H, kBlock)));
  PYGAMD_HIP_CHECK(hipMemsetAsync(grad_src, 0, sizeof(float) * n * H, st));  // skipped rows: 0
  hipLaunchKernelGGL(softmax_index_init_kernel, ginit, dim3(kBlock), 0, st,
                     static_cast<float*>(nullptr), workspace, N * H);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    const IdxT* idx = static_cast<const IdxT*>(index);
    int32_t* const no_flag = nullptr;
    hipLaunchKernelGGL((softmax_index_kernel<IdxT, 3>), grid, dim3(kBlock), 0, st, out, idx, n, H,
                       N, static_cast<float*>(nullptr), workspace, grad_src, grad_out, no_flag);
    hipLaunchKernelGGL((softmax_index_kernel<IdxT, 4>), grid, dim3(kBlock), 0, st, out, idx, n, H,
                       N, static_cast<float*>(nullptr), workspace, grad_src, grad_out, no_flag);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_segment_logsumexp_forward(const float* src, const void* ptr, int idx_dtype,
                                     int64_t n_seg, int64_t H, float* out, void* stream) {
  if (n_seg < 0 || H < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n_seg == 0 || H == 0) return PYGAMD_OK;
  if (!ptr || !out) return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    hipLaunchKernelGGL((segment_logsumexp_kernel<IdxT, false>), dim3(wave_grid(n_seg)),
                       dim3(kBlock), 0, as_stream(stream), src, static_cast<const IdxT*>(ptr),
                       n_seg, H, out, static_cast<const float*>(nullptr),
                       static_cast<float*>(nullptr));
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_segment_logsumexp_backward(const float* src, const float* out, const float* grad_out,
                                      const void* ptr, int idx_dtype, int64_t n_seg, int64_t H,
                                      float* grad_src, void* stream) {
  if (n_seg < 0 || H < 0) return PYGAMD_ERR_INVALID_ARG;
  if (n_seg == 0 || H == 0) return PYGAMD_OK;
  if (!ptr || !out || !grad_out) return PYGAMD_ERR_INVALID_ARG;
  return PYGAMD_DISPATCH_IDX(idx_dtype, [&]() -> int {
    hipLaunchKernelGGL((segment_logsumexp_kernel<IdxT, true>), dim3(wave_grid(n_seg)),
                       dim3(kBlock), 0, as_stream(stream), src, static_cast<const IdxT*>(ptr),
                       n_seg, H, const_cast<float*>(out), grad_out, grad_src);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

int pygamd_head_dot_forward(const float* x, int64_t ldx, const float* att_a, const float* att_b,
                            int64_t n_rows, int64_t H, int64_t C, float* out_a, float* out_b,
                            void* stream) {
  if (n_rows < 0 || H < 1 || C < 1 || ldx < H * C) return PYGAMD_ERR_INVALID_ARG;
  if (n_rows == 0) return PYGAMD_OK;
  if (!x || !att_a || !out_a || (att_b && !out_b)) return PYGAMD_ERR_INVALID_ARG;
  int lph = 1;  // lanes per head: the largest power of two with H * lph <= 64
  while (lph * 2 * H <= kWave && lph * 2 <= C) lph *= 2;
  const bool vec4 = (C % 4 == 0) && (ldx % 4 == 0) && aligned16(x) && aligned16(att_a) &&
                    (!att_b || aligned16(att_b));
  const dim3 grid(static_cast<unsigned>(ceil_div(n_rows, kWavesPerBlock)));
  if (vec4) {
    hipLaunchKernelGGL((head_dot_fwd_kernel<4>), grid, dim3(kBlock), 0, as_stream(stream), x, ldx,
                       att_a, att_b, n_rows, static_cast<int>(H), static_cast<int>(C), lph, out_a,
                       out_b);
  } else {
    hipLaunchKernelGGL((head_dot_fwd_kernel<1>), grid, dim3(kBlock), 0, as_stream(stream), x, ldx,
                       att_a, att_b, n_rows, static_cast<int>(H), static_cast<int>(C), lph, out_a,
                       out_b);
  }
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

int pygamd_head_dot_backward(const float* x, int64_t ldx, const float* att_a, const float* att_b,
                             const float* grad_a, const float* grad_b, int64_t n_rows, int64_t H,
                             int64_t C, float* grad_x, int64_t ldg, int accumulate,
                             float* grad_att_a, float* grad_att_b, void* stream) {
  if (n_rows < 0 || H < 1 || C < 1 || ldx < H * C) return PYGAMD_ERR_INVALID_ARG;
  if (!att_a || !grad_att_a || (att_b && (!grad_b || !grad_att_b)))
    return PYGAMD_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  PYGAMD_HIP_CHECK(hipMemsetAsync(grad_att_a, 0, sizeof(float) * H * C, st));
  if (grad_att_b) PYGAMD_HIP_CHECK(hipMemsetAsync(grad_att_b, 0, sizeof(float) * H * C, st));
  if (n_rows == 0) return PYGAMD_OK;
  if (!x || !grad_a || (grad_x && ldg < H * C)) return PYGAMD_ERR_INVALID_ARG;
  int64_t blocks = ceil_div(n_rows, 64);
  if (blocks > 256) blocks = 256;   // one 1024-lane workgroup per CU: few closing atomics
  const int64_t rows_per_block = ceil_div(n_rows, blocks);
  blocks = ceil_div(n_rows, rows_per_block);
  hipLaunchKernelGGL(head_dot_bwd_kernel, dim3(static_cast<unsigned>(blocks)),
                     dim3(kBlock * kHdSub), 0,
                     st, x, ldx, att_a, att_b, grad_a, grad_b, n_rows, static_cast<int>(H),
                     static_cast<int>(C), rows_per_block, grad_x, ldg, accumulate ? 1 : 0,
                     grad_att_a, grad_att_b);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
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
                       dim3(wave_grid(softmax_fwd_waves(n_rows, H))), dim3(kBlock), 0,
                       as_stream(stream), ld,
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
    hipLaunchKernelGGL((segment_softmax_bwd_kernel<IdxT, true>),
                       dim3(wave_grid(softmax_fwd_waves(n_rows, H))),
                       dim3(kBlock), 0, as_stream(stream), alpha_out, grad_alpha,
                       static_cast<const IdxT*>(rowptr), n_rows, H,
                       static_cast<float*>(nullptr), ld, grad_alpha_src, grad_alpha_dst);
    PYGAMD_LAUNCH_CHECK();
    return PYGAMD_OK;
  });
}

}  // extern "C"
