// train.hip — the two ends of a captured mini-batch training step that are not graph work
// (BASELINE config 4; the reference's loop: examples/multi_gpu/distributed_sampling.py:104-117):
//
//   loss = F.cross_entropy(out, y[seeds]); loss.backward()   -> pygamd_cross_entropy_step
//   optimizer.step()  (torch.optim.Adam)                      -> pygamd_adam_step
//
// Inside a hipGraph a launch costs ~5 us whatever it does, and a static-shape batch step is bound
// by its launch count (round 6: 60 launches in 1.09 ms, 35 of them 5-us ATen kernels): the loss of
// 1024 seed rows was seven launches (label gather, log-softmax, NLL, their two backward kernels, the
// mean's scale, a copy of the scalar), the fused multi-tensor Adam 42 us for 0.2 M parameters, and
// the layers' weights were concatenated / transposed by a launch each before every use.  Here:
//   * ONE pass reads the logits once and writes d loss / d logits and the row losses (+ a
//     one-workgroup launch that adds them up in row order);
//   * ONE launch updates every parameter of the model in its flat buffer (the layout the layer
//     kernels read: [W_l | W_r] row blocks + bias, slots.SlotTrainer) and refreshes the TRANSPOSED
//     copy of the weights the input-gradient GEMMs read — so no layer launches a concatenation or a
//     transpose again.
// HBM-bound elementwise work; both are latency-sized (a few microseconds).
#include "common.h"

namespace pygamd {

constexpr int kCeRpw = 4;                                   // rows per wave (<= 256 classes)
constexpr int kCeRowsPerBlock = kWavesPerBlock * kCeRpw;

// One wave per row: max, sum of exp, the row's loss, the row's gradient.  Labels:
// y[label_idx[r]] (label_idx NULL: y[r]); a label outside [0, C) contributes no loss and a zero
// gradient row and raises *err_flag (the reference's device assert), the mean still divides by B
// (no ignore_index).
// A wave takes kCeRpw consecutive rows and issues every row's loads TOGETHER — the row indices,
// then the labels and the rows themselves: a row is a chain of dependent reads (index -> row,
// index -> label) and the 196 k rows of a full-batch training split were bound by how many chains
// the chip keeps in flight, not by their 74 MB.
__global__ void __launch_bounds__(kBlock)
    cross_entropy_rows_kernel(const float* __restrict__ logits, int64_t ld, int64_t B, int C,
                              const int64_t* __restrict__ row_idx, int64_t n_logit_rows,
                              const int64_t* __restrict__ y,
                              const int64_t* __restrict__ label_idx,
                              float* __restrict__ grad, int64_t ldg, float* __restrict__ row_loss,
                              int32_t* __restrict__ err_flag) {
  const int lane = lane_id();
  const int64_t r0 = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block()) * kCeRpw;
  if (r0 >= B) return;
  const float inv_b = 1.f / static_cast<float>(B);
  // (row_idx: the loss of a ROW SUBSET of the logits — a full-batch model's training split,
  // `F.cross_entropy(out[train_idx], y[train_idx])` — without the gathered copy; an index outside
  // the logits is treated like a bad label)
  int64_t src[kCeRpw], li[kCeRpw], lab[kCeRpw];
  bool live[kCeRpw], src_ok[kCeRpw];
#pragma unroll
  for (int q = 0; q < kCeRpw; ++q) {
    live[q] = r0 + q < B;
    const int64_t r = live[q] ? r0 + q : B - 1;
    src[q] = row_idx ? row_idx[r] : r;
    li[q] = label_idx ? label_idx[r] : r;
  }
#pragma unroll
  for (int q = 0; q < kCeRpw; ++q) {
    src_ok[q] = src[q] >= 0 && src[q] < n_logit_rows;
    src[q] = src_ok[q] ? src[q] : 0;
    lab[q] = y[li[q]];
  }
  if (C <= 4 * kWave) {
    // up to 256 classes: every row is read ONCE (all loads of the wave issued together) and kept
    // in registers
    float v[kCeRpw][4];
#pragma unroll
    for (int q = 0; q < kCeRpw; ++q) {
      const float* __restrict__ row = logits + src[q] * ld;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        v[q][u] = lane + u * kWave < C ? row[lane + u * kWave] : -INFINITY;
    }
#pragma unroll
    for (int q = 0; q < kCeRpw; ++q) {
      if (!live[q]) continue;  // (wave-uniform)
      const bool lab_ok = src_ok[q] && lab[q] >= 0 && lab[q] < C;
      float mx = fmaxf(fmaxf(v[q][0], v[q][1]), fmaxf(v[q][2], v[q][3]));
#pragma unroll
      for (int o = kWave / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, kWave));
      float se = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (lane + u * kWave < C) se += expf(v[q][u] - mx);
#pragma unroll
      for (int o = kWave / 2; o > 0; o >>= 1) se += __shfl_xor(se, o, kWave);
      const float lse = mx + logf(se);
      float* __restrict__ grow = grad + (r0 + q) * ldg;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = lane + u * kWave;
        if (c < C) {
          const float p = expf(v[q][u] - lse);
          grow[c] = lab_ok ? (p - (c == lab[q] ? 1.f : 0.f)) * inv_b : 0.f;
        }
      }
      if (lane == 0) {
        row_loss[r0 + q] = lab_ok ? lse - logits[src[q] * ld + (lab_ok ? lab[q] : 0)] : 0.f;
        if (!lab_ok && err_flag) atomicOr(err_flag, 1);
      }
    }
    return;
  }
  for (int q = 0; q < kCeRpw; ++q) {
    if (!live[q]) break;
    const float* __restrict__ row = logits + src[q] * ld;
    const bool lab_ok = src_ok[q] && lab[q] >= 0 && lab[q] < C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += kWave) mx = fmaxf(mx, row[c]);
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, kWave));
    float se = 0.f;
    for (int c = lane; c < C; c += kWave) se += expf(row[c] - mx);
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) se += __shfl_xor(se, o, kWave);
    const float lse = mx + logf(se);
    float* __restrict__ grow = grad + (r0 + q) * ldg;
    for (int c = lane; c < C; c += kWave) {
      const float p = expf(row[c] - lse);
      grow[c] = lab_ok ? (p - (c == lab[q] ? 1.f : 0.f)) * inv_b : 0.f;
    }
    if (lane == 0) {
      row_loss[r0 + q] = lab_ok ? lse - row[lab_ok ? lab[q] : 0] : 0.f;
      if (!lab_ok && err_flag) atomicOr(err_flag, 1);
    }
  }
}

// The row losses added up in row order by ONE workgroup (deterministic), behind the row kernel on
// the stream.  (One launch with a last-workgroup-done ticket was measured first: the release /
// acquire fences around the ticket write the L2 back on this chip — 28 us for 1,024 rows against
// 6 + 5 us for the two launches.)
constexpr int kCeMeanBlock = 1024;
__global__ void __launch_bounds__(kCeMeanBlock)
    cross_entropy_mean_kernel(const float* __restrict__ row_loss, int64_t B,
                              float* __restrict__ loss, int64_t* __restrict__ step_counter) {
  __shared__ float part[kCeMeanBlock];
  // thread t owns rows t, t + 1024, ... (coalesced reads, a fixed order), then a fixed tree
  float s = 0.f;
  // (eight loads in flight, added in the same order: the training split of a full-batch model is
  // 196 k rows = 192 trips per thread, one at a time 52 us)
  int64_t i = threadIdx.x;
  for (; i + 7 * kCeMeanBlock < B; i += 8 * kCeMeanBlock) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = row_loss[i + u * kCeMeanBlock];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < B; i += kCeMeanBlock) s += row_loss[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = kCeMeanBlock / 2; w > 0; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *loss = part[0] / static_cast<float>(B);
    if (step_counter) *step_counter += 1;  // the optimizer's step count of a captured step
  }
}

struct AdamSegs {
  // weight blocks whose transposed copy is kept: element i of [off, off + rows * cols) is
  // w[r][c] with r = (i - off) / cols; its copy goes to wt[t_off + c * rows + r]
  int n;
  int64_t off[PYGAMD_ADAM_MAX_SEGMENTS];
  int rows[PYGAMD_ADAM_MAX_SEGMENTS];
  int cols[PYGAMD_ADAM_MAX_SEGMENTS];
  int64_t t_off[PYGAMD_ADAM_MAX_SEGMENTS];
};

// torch.optim.Adam's update (amsgrad = False, maximize = False) in float32, the step count read
// from the device: step = *step_dev - step_base (a captured step bumps *step_dev itself).
__global__ void __launch_bounds__(kBlock)
    adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                     float* __restrict__ v, int64_t n, const int64_t* __restrict__ step_dev,
                     int64_t step_base, float lr, float beta1, float beta2, float om_beta1,
                     float om_beta2, float eps, float weight_decay, float grad_scale,
                     float* __restrict__ wt, AdamSegs segs) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const float step = static_cast<float>(*step_dev - step_base);
  // (torch: bias_correction = 1 - beta ** step, evaluated on the device in the step tensor's
  // float32 by the capturable form this replaces; om_beta = 1 - beta rounded from double)
  const float bc1 = 1.f - powf(beta1, step);
  const float bc2 = 1.f - powf(beta2, step);
  float gi = g[i] * grad_scale;
  float pi = p[i];
  if (weight_decay != 0.f) gi += weight_decay * pi;
  const float mi = beta1 * m[i] + om_beta1 * gi;
  const float vi = beta2 * v[i] + om_beta2 * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float step_size = lr / bc1;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  pi -= step_size * (mi / denom);
  p[i] = pi;
  if (wt) {
#pragma unroll 1
    for (int s = 0; s < segs.n; ++s) {
      const int64_t j = i - segs.off[s];
      if (j >= 0 && j < static_cast<int64_t>(segs.rows[s]) * segs.cols[s]) {
        const int r = static_cast<int>(j / segs.cols[s]);
        const int c = static_cast<int>(j - static_cast<int64_t>(r) * segs.cols[s]);
        wt[segs.t_off[s] + static_cast<int64_t>(c) * segs.rows[s] + r] = pi;
        break;
      }
    }
  }
}

}  // namespace pygamd

using namespace pygamd;

extern "C" {

int pygamd_cross_entropy_step_workspace_bytes(int64_t B, size_t* bytes) {
  if (B < 0 || !bytes) return PYGAMD_ERR_INVALID_ARG;
  *bytes = static_cast<size_t>(B) * sizeof(float);  // the row losses
  return PYGAMD_OK;
}

int pygamd_cross_entropy_step(const float* logits, int64_t ld, const int64_t* row_idx,
                              int64_t n_logit_rows, int64_t B, int64_t C, const int64_t* y,
                              const int64_t* label_idx, float* grad,
                              int64_t ldg, float* loss, void* workspace, size_t workspace_bytes,
                              int32_t* err_flag, int64_t* step_counter, void* stream) {
  if (B < 1 || C < 1 || C > (1 << 24) || ld < C || ldg < C || n_logit_rows < 1)
    return PYGAMD_ERR_INVALID_ARG;
  // (B == 0: the reference's mean over no rows is NaN — refused)
  if (!loss || !logits || !y || !grad) return PYGAMD_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < static_cast<size_t>(B) * sizeof(float))
    return PYGAMD_ERR_WORKSPACE;
  float* row_loss = static_cast<float*>(workspace);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(cross_entropy_rows_kernel,
                     dim3(static_cast<unsigned>(ceil_div(B, kCeRowsPerBlock))), dim3(kBlock), 0,
                     st, logits, ld, B, static_cast<int>(C), row_idx, n_logit_rows, y, label_idx,
                     grad, ldg, row_loss, err_flag);
  PYGAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(cross_entropy_mean_kernel, dim3(1), dim3(kCeMeanBlock), 0, st, row_loss, B, loss,
                     step_counter);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

int pygamd_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                     const int64_t* step_dev, int64_t step_base, double lr, double beta1,
                     double beta2, double eps, double weight_decay, double grad_scale,
                     float* transposed, int n_segments, const int64_t* seg_off,
                     const int32_t* seg_rows, const int32_t* seg_cols, const int64_t* seg_t_off,
                     void* stream) {
  if (n < 0 || n_segments < 0 || n_segments > PYGAMD_ADAM_MAX_SEGMENTS)
    return PYGAMD_ERR_INVALID_ARG;
  if (n == 0) return PYGAMD_OK;
  if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev) return PYGAMD_ERR_INVALID_ARG;
  if (!(lr >= 0.) || !(eps >= 0.) || !(beta1 >= 0. && beta1 < 1.) || !(beta2 >= 0. && beta2 < 1.) ||
      !(weight_decay >= 0.))
    return PYGAMD_ERR_INVALID_ARG;
  AdamSegs segs = {};
  segs.n = transposed ? n_segments : 0;
  if (segs.n > 0 && (!seg_off || !seg_rows || !seg_cols || !seg_t_off))
    return PYGAMD_ERR_INVALID_ARG;
  for (int s = 0; s < segs.n; ++s) {
    if (seg_off[s] < 0 || seg_rows[s] < 1 || seg_cols[s] < 1 || seg_t_off[s] < 0 ||
        seg_off[s] + static_cast<int64_t>(seg_rows[s]) * seg_cols[s] > n)
      return PYGAMD_ERR_INVALID_ARG;
    segs.off[s] = seg_off[s];
    segs.rows[s] = seg_rows[s];
    segs.cols[s] = seg_cols[s];
    segs.t_off[s] = seg_t_off[s];
  }
  hipLaunchKernelGGL(adam_step_kernel, dim3(static_cast<unsigned>(ceil_div(n, kBlock))),
                     dim3(kBlock), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, n,
                     step_dev, step_base, static_cast<float>(lr), static_cast<float>(beta1),
                     static_cast<float>(beta2), static_cast<float>(1. - beta1),
                     static_cast<float>(1. - beta2), static_cast<float>(eps),
                     static_cast<float>(weight_decay), static_cast<float>(grad_scale),
                     segs.n > 0 ? transposed : nullptr, segs);
  PYGAMD_LAUNCH_CHECK();
  return PYGAMD_OK;
}

}  // extern "C"
