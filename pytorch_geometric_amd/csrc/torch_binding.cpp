// torch_binding.cpp — the compiled PyTorch binding of the C ABI (VERDICT r2 "missing #2").
//
// north_star: "exposed to Python via a PyTorch C++/HIP extension".  The reference reaches its native
// code through dispatcher operators (torch.ops.torch_sparse.* / pyg_lib ops,
// torch_geometric/edge_index.py:1798-1810, torch_geometric/typing.py:45-174).  This translation
// unit registers the hot entry points of include/pyg_amd.h as operators of the `pyg_amd_c`
// namespace (TORCH_LIBRARY + a HIP-key implementation each; shape functions for FakeTensor /
// torch.compile are registered from Python, pytorch_geometric_amd/_compiled.py): tensors in,
// tensors out, the marshalling (row views, leading dimensions, the argument blocks, workspaces,
// the current HIP stream) done HERE instead of in Python + ctypes.  It holds no kernel: everything
// it launches lives in libpyg_amd.so, which it links (the C ABI stays the drop-in boundary).
//
// Built by pytorch_geometric_amd/_build.py with hipcc (host code only) into
// lib/libpyg_amd_torch.so and loaded with torch.ops.load_library; _native.py routes a call here
// when the operands are plain HIP fp32 / int tensors and falls back to its own (validating,
// error-typed) Python path otherwise.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include "../../include/pyg_amd.h"

namespace {

using at::Tensor;
using OptTensor = c10::optional<Tensor>;

inline void* ptr(const Tensor& t) { return t.defined() ? t.data_ptr() : nullptr; }
inline void* ptr(const OptTensor& t) {
  return (t.has_value() && t->defined()) ? t->data_ptr() : nullptr;
}
inline const float* fptr(const Tensor& t) { return static_cast<const float*>(ptr(t)); }
inline bool has(const OptTensor& t) { return t.has_value() && t->defined(); }

inline void* cur_stream(const Tensor& ref) {
  return c10::hip::getCurrentHIPStream(ref.get_device()).stream();
}

inline int idx_dtype(const Tensor& t) {
  TORCH_CHECK(t.scalar_type() == at::kLong || t.scalar_type() == at::kInt,
              "index tensors must be int32 or int64 (got ", t.scalar_type(), ")");
  return t.scalar_type() == at::kLong ? PYGAMD_IDX_I64 : PYGAMD_IDX_I32;
}

inline void check(int rc, const char* what) {
  TORCH_CHECK(rc == PYGAMD_OK, "pyg_amd: ", what, ": ", pygamd_status_string(rc));
}

// 2-D fp32 view with unit inner stride (the row stride may exceed the width)
inline Tensor rows_f32(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), "pytorch_geometric_amd kernels need HIP device tensors ('", name,
              "' is on ", t.device(), "); there is no CPU fallback on this path");
  TORCH_CHECK(t.scalar_type() == at::kFloat, "'", name, "' must be float32");
  TORCH_CHECK(t.dim() == 2, "'", name, "' must be two-dimensional");
  if (t.size(1) > 0 && t.size(0) > 0 && (t.stride(1) != 1 || t.stride(0) < t.size(1)))
    return t.contiguous();
  return t;
}
inline int64_t ld(const Tensor& t) {
  return (t.size(0) > 1 && t.size(1) > 0) ? t.stride(0) : std::max<int64_t>(t.size(1), 1);
}
inline Tensor contig(const OptTensor& t) { return has(t) ? t->contiguous() : Tensor(); }

// index tensors reach the kernels as raw pointers: same device and dtype as `like`, contiguous
inline void check_index(const Tensor& t, const Tensor& like, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.device() == like.device(), "'", name, "' must live on ",
              like.device());
  TORCH_CHECK(t.is_contiguous(), "'", name, "' must be contiguous");
  TORCH_CHECK(t.scalar_type() == like.scalar_type(), "'", name, "' must have the index dtype of '",
              "rowptr' (", like.scalar_type(), "), got ", t.scalar_type());
}
inline void check_index(const OptTensor& t, const Tensor& like, const char* name) {
  if (has(t)) check_index(*t, like, name);
}

// ---- CSR SpMM (pygamd_spmm_csr) ---------------------------------------------------------------
// Every operator WRITES into tensors the caller allocated (the `out=` convention: mutable arguments
// in the schema, nothing returned) — an operator that may or may not alias its result to an
// argument has no valid schema.
void spmm_csr(const Tensor& rowptr, const OptTensor& col, const Tensor& x, int64_t reduce,
              int64_t n_rows, const OptTensor& eid, const OptTensor& w,
              const OptTensor& src_scale, const OptTensor& hub_rows, const OptTensor& hub_cptr,
              int64_t n_hub, int64_t n_chunks, int64_t hub_threshold, int64_t hub_chunk,
              Tensor out, bool accumulate, int64_t hub_phase, const OptTensor& arg32,
              const OptTensor& relu_mask, const OptTensor& relu_bits) {
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(rowptr.device());
  check_index(rowptr, rowptr, "rowptr");
  check_index(col, rowptr, "col");
  check_index(eid, rowptr, "eid");
  check_index(hub_rows, rowptr, "hub_rows");
  check_index(hub_cptr, rowptr, "hub_chunk_ptr");
  const Tensor x2 = rows_f32(x, "x");
  const int64_t F = x2.size(1);
  if (n_rows < 0) n_rows = rowptr.numel() - 1;
  TORCH_CHECK(out.size(0) == n_rows && out.size(1) == F && out.scalar_type() == at::kFloat &&
                  (F <= 1 || out.stride(1) == 1),
              "'out' must be a float32 [n_rows, F] tensor with unit inner stride");
  pygamd_spmm_args a = {};
  a.rowptr = ptr(rowptr);
  a.col = ptr(col);
  a.eid = ptr(eid);
  Tensor wc = contig(w), sc = contig(src_scale);
  int64_t w_heads = 1, head_dim = F;
  if (wc.defined()) {
    TORCH_CHECK(wc.scalar_type() == at::kFloat, "edge weights must be float32");
    w_heads = wc.dim() == 1 ? 1 : wc.size(1);
    if (w_heads > 1) {
      TORCH_CHECK(F % w_heads == 0, "feature width must be divisible by the number of heads");
      head_dim = F / w_heads;
    }
    a.w = fptr(wc);
  }
  a.src_scale = fptr(sc);
  a.x = fptr(x2);
  a.out = static_cast<float*>(ptr(out));
  if (has(arg32) && (reduce == PYGAMD_MIN || reduce == PYGAMD_MAX)) {
    TORCH_CHECK(arg32->is_contiguous() && arg32->scalar_type() == at::kInt &&
                    arg32->numel() == n_rows * F,
                "'arg32' must be a contiguous int32 [n_rows, F] tensor");
    a.arg32_out = static_cast<int32_t*>(ptr(arg32));
  }
  a.n_rows = n_rows;
  a.n_src = x2.size(0);
  a.F = F;
  a.ldx = ld(x2);
  a.ldo = ld(out);
  a.idx_dtype = idx_dtype(rowptr);
  a.reduce = static_cast<int32_t>(reduce);
  a.w_heads = static_cast<int32_t>(w_heads);
  a.head_dim = static_cast<int32_t>(head_dim);
  a.accumulate = accumulate ? 1 : 0;
  a.hub_phase = static_cast<int32_t>(hub_phase);
  Tensor m2;
  if (has(relu_mask)) {
    m2 = rows_f32(*relu_mask, "relu_mask");
    TORCH_CHECK(m2.size(0) == n_rows && m2.size(1) == F, "'relu_mask' must be [n_rows, F]");
    a.relu_mask = fptr(m2);
    a.ld_mask = ld(m2);
  }
  if (has(relu_bits)) {
    a.relu_bits = static_cast<const uint32_t*>(ptr(relu_bits));
    a.ld_bits = relu_bits->size(1);
  }
  Tensor ws;
  size_t ws_bytes = 0;
  if (n_hub > 0 && has(hub_rows) && has(hub_cptr)) {
    a.hub_rows = ptr(hub_rows);
    a.hub_chunk_ptr = ptr(hub_cptr);
    a.n_hub = n_hub;
    a.n_chunks = n_chunks;
    a.hub_threshold = hub_threshold;
    a.hub_chunk = hub_chunk;
    if (hub_phase != 1 && (reduce == PYGAMD_SUM || reduce == PYGAMD_MEAN)) {
      ws_bytes = static_cast<size_t>(n_chunks) * static_cast<size_t>(F) * 4;
      ws = at::empty({static_cast<int64_t>(ws_bytes)}, x2.options().dtype(at::kByte));
    }
  }
  check(pygamd_spmm_csr(&a, ptr(ws), ws_bytes, cur_stream(x2)), "spmm_csr");
}

// ---- dense transform (csrc/gemm.hip) --------------------------------------------------------------
void linear_forward(const Tensor& x, const Tensor& w, const OptTensor& bias, bool relu, Tensor out,
                    bool accumulate) {
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(x.device());
  const Tensor x2 = rows_f32(x, "x"), w2 = rows_f32(w, "weight");
  const int64_t M = x2.size(0), K = x2.size(1), N = w2.size(0);
  TORCH_CHECK(w2.size(1) == K, "'x' has ", K, " columns but 'weight' expects ", w2.size(1));
  TORCH_CHECK(out.size(0) == M && out.size(1) == N && out.scalar_type() == at::kFloat &&
                  (N <= 1 || out.stride(1) == 1),
              "'out' must be a float32 [M, N] tensor with unit inner stride");
  const Tensor b = contig(bias);
  size_t ws_bytes = 0;  // split over the reduction when the output has few row tiles
  check(pygamd_linear_nt_workspace_bytes(M, N, K, &ws_bytes), "linear_forward");
  Tensor ws;
  if (ws_bytes > 0) ws = at::empty({static_cast<int64_t>(ws_bytes)}, x2.options().dtype(at::kByte));
  check(pygamd_linear_forward(fptr(x2), ld(x2), fptr(w2), ld(w2), fptr(b), M, K, N, relu ? 1 : 0,
                              accumulate ? 1 : 0, static_cast<float*>(ptr(out)), ld(out), ptr(ws),
                              ws_bytes, cur_stream(x2)),
        "linear_forward");
}

void linear_dgrad(const Tensor& g, const Tensor& w_t, const OptTensor& row_scale, int64_t n_scaled,
                  Tensor out, bool accumulate, const OptTensor& relu_mask,
                  const OptTensor& relu_bits, const OptTensor& out_scaled) {
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(g.device());
  const Tensor g2 = rows_f32(g, "grad"), w2 = rows_f32(w_t, "weight_t");
  const int64_t M = g2.size(0), N = g2.size(1), K = w2.size(0);
  TORCH_CHECK(w2.size(1) == N, "'grad' has ", N, " columns but 'weight_t' expects ", w2.size(1));
  TORCH_CHECK(!(has(relu_mask) && has(relu_bits)), "pass at most one of 'relu_mask' / 'relu_bits'");
  TORCH_CHECK(out.size(0) == M && out.size(1) == K && out.scalar_type() == at::kFloat &&
                  (K <= 1 || out.stride(1) == 1),
              "'out' must be a float32 [M, K] tensor with unit inner stride");
  const Tensor rs = contig(row_scale);
  if (has(out_scaled)) {
    TORCH_CHECK(rs.defined() && !accumulate, "'out_scaled' needs 'row_scale' and no 'accumulate'");
    TORCH_CHECK(out_scaled->size(0) == M && out_scaled->size(1) == K &&
                    out_scaled->scalar_type() == at::kFloat &&
                    (K <= 1 || out_scaled->stride(1) == 1),
                "'out_scaled' must be a float32 [M, K] tensor with unit inner stride");
  }
  Tensor m2;
  if (has(relu_mask)) {
    m2 = rows_f32(*relu_mask, "relu_mask");
    TORCH_CHECK(m2.size(0) == M && m2.size(1) == K, "'relu_mask' must be [M, K]");
  }
  size_t ws_bytes = 0;
  check(pygamd_linear_nt_workspace_bytes(M, K, N, &ws_bytes), "linear_dgrad");
  Tensor ws;
  if (ws_bytes > 0) ws = at::empty({static_cast<int64_t>(ws_bytes)}, g2.options().dtype(at::kByte));
  check(pygamd_linear_dgrad2(
            fptr(g2), ld(g2), fptr(w2), ld(w2), fptr(rs), rs.defined() ? n_scaled : 0, M, N, K,
            accumulate ? 1 : 0, fptr(m2), m2.defined() ? ld(m2) : 0,
            static_cast<const uint32_t*>(ptr(relu_bits)), has(relu_bits) ? relu_bits->size(1) : 0,
            static_cast<float*>(ptr(out)), ld(out), static_cast<float*>(ptr(out_scaled)),
            has(out_scaled) ? ld(*out_scaled) : 0, ptr(ws), ws_bytes, cur_stream(g2)),
        "linear_dgrad");
}

// out [N, K1 + K2]; grad_b [N] (optional: the column sums of g from the same pass)
void linear_wgrad(const Tensor& g, const Tensor& x, Tensor out, bool accumulate,
                  int64_t wgs_per_cu, const OptTensor& grad_b, const OptTensor& x2_) {
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(g.device());
  const Tensor g2 = rows_f32(g, "grad"), first = rows_f32(x, "x");
  Tensor second;
  if (has(x2_)) second = rows_f32(*x2_, "x2");
  const int64_t M = g2.size(0), N = g2.size(1), K1 = first.size(1);
  const int64_t K2 = second.defined() ? second.size(1) : 0, K = K1 + K2;
  TORCH_CHECK(first.size(0) == M && (!second.defined() || second.size(0) == M),
              "'grad' and the operands must have the same number of rows");
  TORCH_CHECK(!second.defined() || (K1 > 0 && K2 > 0),
              "both operands of a two-operand weight gradient need columns");
  TORCH_CHECK(!(has(grad_b) && K == 0), "the bias gradient needs a weight tile to ride on");
  TORCH_CHECK(out.size(0) == N && out.size(1) == K && out.scalar_type() == at::kFloat &&
                  (K <= 1 || out.stride(1) == 1),
              "'out' must be a float32 [N, K] tensor with unit inner stride");
  TORCH_CHECK(!has(grad_b) || (grad_b->numel() == N && grad_b->is_contiguous() &&
                               grad_b->scalar_type() == at::kFloat),
              "'grad_b' must be a contiguous float32 [N] tensor");
  size_t nbytes = 0;
  check(pygamd_linear_wgrad_workspace_bytes(M, N, K, &nbytes), "linear_wgrad workspace");
  Tensor ws = at::empty({static_cast<int64_t>(std::max<size_t>(nbytes, 4))},
                        g2.options().dtype(at::kByte));
  check(pygamd_linear_wgrad2(fptr(g2), ld(g2), fptr(first), ld(first), K1, fptr(second),
                             second.defined() ? ld(second) : 0, K2, M, N, accumulate ? 1 : 0,
                             static_cast<int>(wgs_per_cu), static_cast<float*>(ptr(out)), ld(out),
                             static_cast<float*>(ptr(grad_b)), ptr(ws), nbytes, cur_stream(g2)),
        "linear_wgrad");
}

// ---- SAGEConv layer as one kernel (csrc/sage_fused.hip) ---------------------------------------------
void sage_layer_fused(const Tensor& rowptr, const OptTensor& col, const Tensor& x_gather,
                        const Tensor& x_root, const Tensor& w, const OptTensor& bias,
                        int64_t reduce, bool relu, Tensor agg, Tensor out,
                        const OptTensor& hub_rows, const OptTensor& hub_cptr, int64_t n_hub,
                        int64_t n_chunks, int64_t hub_threshold, int64_t hub_chunk, bool save_agg,
                        const OptTensor& relu_bits, const OptTensor& mask_bits,
                        const OptTensor& row_scale, const OptTensor& out_scaled) {
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(rowptr.device());
  check_index(rowptr, rowptr, "rowptr");
  check_index(col, rowptr, "col");
  check_index(hub_rows, rowptr, "hub_rows");
  check_index(hub_cptr, rowptr, "hub_chunk_ptr");
  const Tensor xg = rows_f32(x_gather, "x"), xr = rows_f32(x_root, "x_root");
  const Tensor w2 = rows_f32(w, "weight");
  const int64_t n_rows = rowptr.numel() - 1, F = xg.size(1), Fo = w2.size(0);
  TORCH_CHECK(xr.size(0) == n_rows && xr.size(1) == F && w2.size(1) == 2 * F &&
                  agg.size(0) == n_rows && agg.size(1) == F && out.size(0) == n_rows &&
                  out.size(1) == Fo,
              "shape mismatch in sage_layer_forward");
  pygamd_spmm_args a = {};
  a.rowptr = ptr(rowptr);
  a.col = ptr(col);
  a.x = fptr(xg);
  a.out = static_cast<float*>(ptr(agg));
  a.n_rows = n_rows;
  a.n_src = xg.size(0);
  a.F = F;
  a.ldx = ld(xg);
  a.ldo = ld(agg);
  a.idx_dtype = idx_dtype(rowptr);
  a.reduce = static_cast<int32_t>(reduce);
  a.w_heads = 1;
  a.head_dim = static_cast<int32_t>(F);
  if (n_hub > 0 && has(hub_rows) && has(hub_cptr)) {
    a.hub_rows = ptr(hub_rows);
    a.hub_chunk_ptr = ptr(hub_cptr);
    a.n_hub = n_hub;
    a.n_chunks = n_chunks;
    a.hub_threshold = hub_threshold;
    a.hub_chunk = hub_chunk;
  }
  const Tensor b = contig(bias), rs = contig(row_scale);
  pygamd_sage_fused_args f = {};
  f.x_root = fptr(xr);
  f.ld_root = ld(xr);
  f.w = fptr(w2);
  f.ldw = ld(w2);
  f.bias = fptr(b);
  f.Fo = Fo;
  f.relu = relu ? 1 : 0;
  f.save_agg = save_agg ? 1 : 0;
  f.y = static_cast<float*>(ptr(out));
  f.ldy = ld(out);
  if (has(relu_bits)) {
    f.relu_bits_out = static_cast<uint32_t*>(ptr(relu_bits));
    f.ld_bits_out = relu_bits->size(1);
  }
  if (has(mask_bits)) {
    f.mask_bits = static_cast<const uint32_t*>(ptr(mask_bits));
    f.ld_mask_bits = mask_bits->size(1);
  }
  if (has(out_scaled)) {
    TORCH_CHECK(rs.defined() && rs.numel() == n_rows && rs.scalar_type() == at::kFloat,
                "'out_scaled' needs a float32 'row_scale' with one entry per row");
    TORCH_CHECK(out_scaled->size(0) == n_rows && out_scaled->size(1) == Fo &&
                    out_scaled->scalar_type() == at::kFloat &&
                    (Fo <= 1 || out_scaled->stride(1) == 1),
                "'out_scaled' must be a float32 [n_rows, Fo] tensor with unit inner stride");
    f.row_scale = fptr(rs);
    f.y_scaled = static_cast<float*>(ptr(out_scaled));
    f.ldy_scaled = ld(*out_scaled);
  }
  // hub partials + (split arithmetic) the weight's bf16 term planes
  size_t ws_bytes = 0;
  check(pygamd_sage_layer_fused_workspace_bytes(&a, &f, &ws_bytes), "sage_layer_forward");
  Tensor ws;
  if (ws_bytes > 0)
    ws = at::empty({static_cast<int64_t>(ws_bytes)}, xg.options().dtype(at::kByte));
  check(pygamd_sage_layer_fused(&a, &f, ptr(ws), ws_bytes, cur_stream(xg)), "sage_layer_forward");
}

// ---- index / gather side -------------------------------------------------------------------------------
Tensor index2ptr(const Tensor& index, int64_t size) {
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(index.device());
  const Tensor idx = index.contiguous();
  Tensor out = at::empty({size + 1}, idx.options());
  check(pygamd_index2ptr(ptr(idx), idx_dtype(idx), idx.numel(), size, ptr(out), cur_stream(idx)),
        "index2ptr");
  return out;
}

Tensor ptr2index(const Tensor& p, int64_t n) {
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(p.device());
  const Tensor pc = p.contiguous();
  Tensor out = at::empty({n}, pc.options());
  check(pygamd_ptr2index(ptr(pc), idx_dtype(pc), pc.numel() - 1, n, ptr(out), cur_stream(pc)),
        "ptr2index");
  return out;
}

Tensor gather_rows(const Tensor& x, const Tensor& index) {
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(x.device());
  const Tensor x2 = rows_f32(x, "x"), idx = index.contiguous();
  const int64_t n = idx.numel(), F = x2.size(1);
  Tensor out = at::empty({n, F}, x2.options());
  check(pygamd_gather_rows(fptr(x2), ld(x2), x2.size(0), ptr(idx), idx_dtype(idx), n, F,
                           static_cast<float*>(ptr(out)), ld(out), nullptr, cur_stream(x2)),
        "gather_rows");
  return out;
}

void gather_scatter_add(const Tensor& x, const Tensor& gather_idx, const Tensor& scatter_idx,
                        const OptTensor& scale, const OptTensor& w, Tensor out,
                        const OptTensor& n_valid) {
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(x.device());
  const Tensor x2 = rows_f32(x, "x");
  const int64_t F = x2.size(1);
  TORCH_CHECK(out.dim() == 2 && out.size(1) == F && out.scalar_type() == at::kFloat &&
                  (F <= 1 || out.stride(1) == 1),
              "'out' must be a float32 [n_out, F] tensor with unit inner stride");
  const Tensor gi = gather_idx.contiguous(), si = scatter_idx.contiguous();
  const Tensor sc = contig(scale), wc = contig(w);
  check(pygamd_gather_scatter_add(fptr(x2), ld(x2), ptr(gi), ptr(si), idx_dtype(gi), fptr(sc),
                                  fptr(wc), gi.numel(), static_cast<const int64_t*>(ptr(n_valid)),
                                  F, static_cast<float*>(ptr(out)), ld(out), cur_stream(x2)),
        "gather_scatter_add");
}

Tensor sddmm_csr(const Tensor& rowptr, const OptTensor& col, const OptTensor& eid,
                 const Tensor& grad_out, const Tensor& x, int64_t n_edges, int64_t w_heads) {
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(rowptr.device());
  const Tensor g2 = rows_f32(grad_out, "grad_out"), x2 = rows_f32(x, "x");
  const int64_t F = x2.size(1);
  Tensor grad_w = at::zeros({n_edges, w_heads}, x2.options());
  check(pygamd_sddmm_csr(ptr(rowptr), ptr(col), ptr(eid), idx_dtype(rowptr), fptr(g2), ld(g2),
                         fptr(x2), ld(x2), rowptr.numel() - 1, F, static_cast<int32_t>(w_heads),
                         static_cast<int32_t>(F / std::max<int64_t>(w_heads, 1)),
                         static_cast<float*>(ptr(grad_w)), cur_stream(x2)),
        "sddmm_csr");
  return grad_w;
}

// ---- segment softmax -----------------------------------------------------------------------------------
Tensor segment_softmax_forward(const Tensor& src, const Tensor& p) {
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(src.device());
  const Tensor s2 = src.contiguous();
  Tensor out = at::empty_like(s2);
  check(pygamd_segment_softmax_forward(fptr(s2), ptr(p), idx_dtype(p), p.numel() - 1, s2.size(1),
                                       static_cast<float*>(ptr(out)), cur_stream(s2)),
        "segment_softmax_forward");
  return out;
}

Tensor segment_softmax_backward(const Tensor& out, const Tensor& grad_out, const Tensor& p) {
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(out.device());
  const Tensor o2 = out.contiguous(), g2 = grad_out.contiguous();
  Tensor grad_src = at::empty_like(o2);
  check(pygamd_segment_softmax_backward(fptr(o2), fptr(g2), ptr(p), idx_dtype(p), p.numel() - 1,
                                        o2.size(1), static_cast<float*>(ptr(grad_src)),
                                        cur_stream(o2)),
        "segment_softmax_backward");
  return grad_src;
}

int64_t abi_version() { return pygamd_abi_version(); }

// ---- autograd nodes in C++ ----------------------------------------------------
// The three operators a launch-bound layer stack spends its host time in — F.linear
// (nn/dense/linear.py:121-127), the weighted sum / mean aggregation of message_and_aggregate
// (utils/_spmm.py:12-136, edge_index.py:1849-1900 for the transposed backward) and the layer tail
// `act(out + bias)` (gcn_conv.py:278-281 + basic_gnn.py:262-263) — as torch::autograd::Function s:
// forward AND backward run without re-entering Python (an eager 2-layer GCN step at the Cora shape
// spent ~0.4 of its 0.5 ms in the Python autograd Functions and their ctypes / marshalling glue
// around 0.12 ms of GPU work, scripts/eager_probe.py).  Same entry points of the C ABI, same order,
// as pytorch_geometric_amd/_functions.py {LinearFunction, SpmmFunction, BiasActFunction}; that
// module routes here when the operands are plain float32 HIP tensors and nothing exotic is asked
// for (no gradient w.r.t. the edge weights, sum / mean only), and stays the reference
// implementation for everything else.
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

inline Tensor rows2d(const Tensor& t) { return t.reshape({-1, t.size(-1)}); }

struct LinearAG : public torch::autograd::Function<LinearAG> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& weight,
                        const OptTensor& bias) {
    at::AutoDispatchBelowADInplaceOrView guard;
    const Tensor x2 = rows2d(x);
    Tensor out = at::empty({x2.size(0), weight.size(0)}, x2.options());
    linear_forward(x2, weight, bias, false, out, false);
    ctx->save_for_backward({x2, weight});
    ctx->saved_data["has_bias"] = has(bias);
    ctx->saved_data["x_shape"] = x.sizes().vec();
    std::vector<int64_t> shape = x.sizes().vec();
    shape.back() = weight.size(0);
    return out.view(shape);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    at::AutoDispatchBelowADInplaceOrView guard;
    const auto saved = ctx->get_saved_variables();
    const Tensor &x2 = saved[0], &weight = saved[1];
    const Tensor g2 = rows2d(grads[0]);
    Tensor gx, gw, gb;
    if (ctx->needs_input_grad(0)) {
      const Tensor wt = weight.t().contiguous();
      gx = at::empty({g2.size(0), weight.size(1)}, g2.options());
      linear_dgrad(g2, wt, c10::nullopt, 0, gx, false, c10::nullopt, c10::nullopt, c10::nullopt);
      gx = gx.view(ctx->saved_data["x_shape"].toIntVector());
    }
    const bool need_b = ctx->saved_data["has_bias"].toBool() && ctx->needs_input_grad(2);
    if (ctx->needs_input_grad(1)) {
      gw = at::empty({weight.size(0), weight.size(1)}, g2.options());
      if (need_b) gb = at::empty({weight.size(0)}, g2.options());
      linear_wgrad(g2, x2, gw, false, 2, need_b ? OptTensor(gb) : c10::nullopt, c10::nullopt);
    } else if (need_b) {
      gb = at::empty({weight.size(0)}, g2.options());
      const Tensor g = rows_f32(g2, "grad");
      check(pygamd_colsum(fptr(g), ld(g), g.size(0), g.size(1), static_cast<float*>(ptr(gb)),
                          cur_stream(g)), "colsum");
    }
    return {gx, gw, gb};
  }
};

Tensor linear_ag(const Tensor& x, const Tensor& weight, const OptTensor& bias) {
  return LinearAG::apply(x, weight, bias);
}

struct BiasActAG : public torch::autograd::Function<BiasActAG> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const OptTensor& bias, bool relu) {
    at::AutoDispatchBelowADInplaceOrView guard;
    const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(x.device());
    const Tensor x2 = rows_f32(rows2d(x), "x");
    const Tensor b = contig(bias);
    TORCH_CHECK(!b.defined() || (b.scalar_type() == at::kFloat && b.numel() == x2.size(1)),
                "'bias' must be float32 with ", x2.size(1), " entries");
    Tensor out = at::empty({x2.size(0), x2.size(1)}, x2.options());
    check(pygamd_bias_act(fptr(x2), ld(x2), fptr(b), x2.size(0), x2.size(1), relu ? 1 : 0,
                          static_cast<float*>(ptr(out)), ld(out), cur_stream(x2)), "bias_act");
    out = out.view(x.sizes());
    ctx->saved_data["relu"] = relu;
    ctx->saved_data["has_bias"] = b.defined();
    if (relu) ctx->save_for_backward({out});
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    at::AutoDispatchBelowADInplaceOrView guard;
    const Tensor g2 = rows_f32(rows2d(grads[0]), "grad");
    const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(g2.device());
    const bool need_b = ctx->saved_data["has_bias"].toBool() && ctx->needs_input_grad(1);
    Tensor gb;
    if (need_b) gb = at::empty({g2.size(1)}, g2.options());
    if (ctx->saved_data["relu"].toBool()) {
      const Tensor act = rows_f32(rows2d(ctx->get_saved_variables()[0]), "act");
      Tensor gin = at::empty({g2.size(0), g2.size(1)}, g2.options());
      check(pygamd_relu_backward_colsum(fptr(g2), ld(g2), fptr(act), ld(act), g2.size(0),
                                        g2.size(1), static_cast<float*>(ptr(gin)), ld(gin),
                                        static_cast<float*>(ptr(gb)), cur_stream(g2)),
            "relu_backward_colsum");
      return {gin.view(grads[0].sizes()), gb, Tensor()};
    }
    if (need_b)
      check(pygamd_colsum(fptr(g2), ld(g2), g2.size(0), g2.size(1), static_cast<float*>(ptr(gb)),
                          cur_stream(g2)), "colsum");
    return {grads[0], gb, Tensor()};
  }
};

Tensor bias_act_ag(const Tensor& x, const OptTensor& bias, bool relu) {
  return BiasActAG::apply(x, bias, relu);
}

// out[i] = reduce_{slots k of row i} w[eid[k]] * x[col[k]] (sum / mean; w, eid optional); the
// backward w.r.t. x runs the same kernel on the transposed CSR (rowptr_t, col_t, eid_t), with the
// mean's 1/deg folded into the gathered gradient rows (src_scale) or, with weights, applied to
// the gradient first — exactly SpmmFunction's schedule.  No gradient w.r.t. w here.
struct SpmmAG : public torch::autograd::Function<SpmmAG> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const OptTensor& w,
                        const Tensor& rowptr, const Tensor& col, const OptTensor& eid,
                        const OptTensor& hub_rows, const OptTensor& hub_cptr, int64_t n_hub,
                        int64_t n_chunks, const Tensor& rowptr_t, const Tensor& col_t,
                        const OptTensor& eid_t, const OptTensor& hub_rows_t,
                        const OptTensor& hub_cptr_t, int64_t n_hub_t, int64_t n_chunks_t,
                        const OptTensor& inv_deg, int64_t reduce, int64_t hub_threshold,
                        int64_t hub_chunk) {
    at::AutoDispatchBelowADInplaceOrView guard;
    TORCH_CHECK(reduce == PYGAMD_SUM || reduce == PYGAMD_MEAN, "spmm_ag: sum / mean only");
    const Tensor x2 = x.reshape({x.size(0), -1});
    const int64_t n_rows = rowptr.numel() - 1;
    Tensor out = at::empty({n_rows, x2.size(1)}, x2.options());
    spmm_csr(rowptr, col, x2, reduce, n_rows, eid, w, c10::nullopt, hub_rows, hub_cptr, n_hub,
             n_chunks, hub_threshold, hub_chunk, out, false, 0, c10::nullopt, c10::nullopt,
             c10::nullopt);
    ctx->save_for_backward({rowptr_t, col_t, has(eid_t) ? *eid_t : Tensor(),
                            has(hub_rows_t) ? *hub_rows_t : Tensor(),
                            has(hub_cptr_t) ? *hub_cptr_t : Tensor(),
                            has(w) ? *w : Tensor(), has(inv_deg) ? *inv_deg : Tensor()});
    ctx->saved_data["n_hub_t"] = n_hub_t;
    ctx->saved_data["n_chunks_t"] = n_chunks_t;
    ctx->saved_data["reduce"] = reduce;
    ctx->saved_data["hub_threshold"] = hub_threshold;
    ctx->saved_data["hub_chunk"] = hub_chunk;
    ctx->saved_data["x_shape"] = x.sizes().vec();
    std::vector<int64_t> shape = x.sizes().vec();
    shape[0] = n_rows;
    return out.view(shape);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    at::AutoDispatchBelowADInplaceOrView guard;
    variable_list res(20);
    if (!ctx->needs_input_grad(0)) return res;
    const auto sv = ctx->get_saved_variables();
    const Tensor &rowptr_t = sv[0], &col_t = sv[1];
    auto opt = [](const Tensor& t) { return t.defined() ? OptTensor(t) : c10::nullopt; };
    const int64_t reduce = ctx->saved_data["reduce"].toInt();
    Tensor g2 = grads[0].reshape({grads[0].size(0), -1});
    OptTensor scale = c10::nullopt;
    if (reduce == PYGAMD_MEAN && sv[6].defined()) {
      if (sv[5].defined()) {
        g2 = g2 * sv[6].view({-1, 1});   // (weights AND mean: the rare form, one ATen pass)
      } else {
        scale = sv[6];
      }
    }
    const int64_t n_src = rowptr_t.numel() - 1;
    Tensor gx = at::empty({n_src, g2.size(1)}, g2.options());
    spmm_csr(rowptr_t, col_t, g2, PYGAMD_SUM, n_src, opt(sv[2]), opt(sv[5]), scale, opt(sv[3]),
             opt(sv[4]), ctx->saved_data["n_hub_t"].toInt(), ctx->saved_data["n_chunks_t"].toInt(),
             ctx->saved_data["hub_threshold"].toInt(), ctx->saved_data["hub_chunk"].toInt(), gx,
             false, 0, c10::nullopt, c10::nullopt, c10::nullopt);
    res[0] = gx.view(ctx->saved_data["x_shape"].toIntVector());
    return res;
  }
};

Tensor spmm_ag(const Tensor& x, const OptTensor& w, const Tensor& rowptr, const Tensor& col,
               const OptTensor& eid, const OptTensor& hub_rows, const OptTensor& hub_cptr,
               int64_t n_hub, int64_t n_chunks, const Tensor& rowptr_t, const Tensor& col_t,
               const OptTensor& eid_t, const OptTensor& hub_rows_t, const OptTensor& hub_cptr_t,
               int64_t n_hub_t, int64_t n_chunks_t, const OptTensor& inv_deg, int64_t reduce,
               int64_t hub_threshold, int64_t hub_chunk) {
  return SpmmAG::apply(x, w, rowptr, col, eid, hub_rows, hub_cptr, n_hub, n_chunks, rowptr_t,
                       col_t, eid_t, hub_rows_t, hub_cptr_t, n_hub_t, n_chunks_t, inv_deg, reduce,
                       hub_threshold, hub_chunk);
}

}  // namespace

TORCH_LIBRARY(pyg_amd_c, m) {
  m.def("abi_version() -> int", &abi_version);
  m.def(
      "spmm_csr(Tensor rowptr, Tensor? col, Tensor x, int reduce, int n_rows, Tensor? eid, "
      "Tensor? w, Tensor? src_scale, Tensor? hub_rows, Tensor? hub_cptr, int n_hub, "
      "int n_chunks, int hub_threshold, int hub_chunk, Tensor(a!) out, bool accumulate, "
      "int hub_phase, Tensor(b!)? arg32, Tensor? relu_mask, Tensor? relu_bits) -> ()");
  m.def(
      "linear_forward(Tensor x, Tensor w, Tensor? bias, bool relu, Tensor(a!) out, "
      "bool accumulate) -> ()");
  m.def(
      "linear_dgrad(Tensor g, Tensor w_t, Tensor? row_scale, int n_scaled, Tensor(a!) out, "
      "bool accumulate, Tensor? relu_mask, Tensor? relu_bits, Tensor(b!)? out_scaled) -> ()");
  m.def(
      "linear_wgrad(Tensor g, Tensor x, Tensor(a!) out, bool accumulate, int wgs_per_cu, "
      "Tensor(b!)? grad_b, Tensor? x2) -> ()");
  m.def(
      "sage_layer_fused(Tensor rowptr, Tensor? col, Tensor x_gather, Tensor x_root, Tensor w, "
      "Tensor? bias, int reduce, bool relu, Tensor(a!) agg, Tensor(b!) out, Tensor? hub_rows, "
      "Tensor? hub_cptr, int n_hub, int n_chunks, int hub_threshold, int hub_chunk, "
      "bool save_agg, Tensor(c!)? relu_bits, Tensor? mask_bits, Tensor? row_scale, "
      "Tensor(d!)? out_scaled) -> ()");
  m.def("index2ptr(Tensor index, int size) -> Tensor");
  m.def("ptr2index(Tensor ptr, int n) -> Tensor");
  m.def("gather_rows(Tensor x, Tensor index) -> Tensor");
  m.def(
      "gather_scatter_add(Tensor x, Tensor gather_idx, Tensor scatter_idx, Tensor? scale, "
      "Tensor? w, Tensor(a!) out, Tensor? n_valid) -> ()");
  m.def(
      "sddmm_csr(Tensor rowptr, Tensor? col, Tensor? eid, Tensor grad_out, Tensor x, "
      "int n_edges, int w_heads) -> Tensor");
  m.def("segment_softmax_forward(Tensor src, Tensor ptr) -> Tensor");
  m.def("segment_softmax_backward(Tensor out, Tensor grad_out, Tensor ptr) -> Tensor");
  // autograd nodes in C++ (forward and backward without re-entering Python)
  m.def("linear_ag(Tensor x, Tensor weight, Tensor? bias) -> Tensor");
  m.def("bias_act_ag(Tensor x, Tensor? bias, bool relu) -> Tensor");
  m.def(
      "spmm_ag(Tensor x, Tensor? w, Tensor rowptr, Tensor col, Tensor? eid, Tensor? hub_rows, "
      "Tensor? hub_cptr, int n_hub, int n_chunks, Tensor rowptr_t, Tensor col_t, Tensor? eid_t, "
      "Tensor? hub_rows_t, Tensor? hub_cptr_t, int n_hub_t, int n_chunks_t, Tensor? inv_deg, "
      "int reduce, int hub_threshold, int hub_chunk) -> Tensor");
}

// "CUDA" is the dispatch key of HIP tensors in a ROCm build of PyTorch
TORCH_LIBRARY_IMPL(pyg_amd_c, CUDA, m) {
  m.impl("spmm_csr", &spmm_csr);
  m.impl("linear_forward", &linear_forward);
  m.impl("linear_dgrad", &linear_dgrad);
  m.impl("linear_wgrad", &linear_wgrad);
  m.impl("sage_layer_fused", &sage_layer_fused);
  m.impl("index2ptr", &index2ptr);
  m.impl("ptr2index", &ptr2index);
  m.impl("gather_rows", &gather_rows);
  m.impl("gather_scatter_add", &gather_scatter_add);
  m.impl("sddmm_csr", &sddmm_csr);
  m.impl("segment_softmax_forward", &segment_softmax_forward);
  m.impl("segment_softmax_backward", &segment_softmax_backward);
}

// the same three operators below the autograd key (inference_mode, calls from inside another
// node's backward): forward only
namespace {
Tensor linear_plain(const Tensor& x, const Tensor& weight, const OptTensor& bias) {
  const Tensor x2 = rows2d(x);
  Tensor out = at::empty({x2.size(0), weight.size(0)}, x2.options());
  linear_forward(x2, weight, bias, false, out, false);
  std::vector<int64_t> shape = x.sizes().vec();
  shape.back() = weight.size(0);
  return out.view(shape);
}
Tensor bias_act_plain(const Tensor& x, const OptTensor& bias, bool relu) {
  const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(x.device());
  const Tensor x2 = rows_f32(rows2d(x), "x");
  const Tensor b = contig(bias);
  TORCH_CHECK(!b.defined() || (b.scalar_type() == at::kFloat && b.numel() == x2.size(1)),
              "'bias' must be float32 with ", x2.size(1), " entries");
  Tensor out = at::empty({x2.size(0), x2.size(1)}, x2.options());
  check(pygamd_bias_act(fptr(x2), ld(x2), fptr(b), x2.size(0), x2.size(1), relu ? 1 : 0,
                        static_cast<float*>(ptr(out)), ld(out), cur_stream(x2)), "bias_act");
  return out.view(x.sizes());
}
Tensor spmm_plain(const Tensor& x, const OptTensor& w, const Tensor& rowptr, const Tensor& col,
                  const OptTensor& eid, const OptTensor& hub_rows, const OptTensor& hub_cptr,
                  int64_t n_hub, int64_t n_chunks, const Tensor&, const Tensor&, const OptTensor&,
                  const OptTensor&, const OptTensor&, int64_t, int64_t, const OptTensor&,
                  int64_t reduce, int64_t hub_threshold, int64_t hub_chunk) {
  TORCH_CHECK(reduce == PYGAMD_SUM || reduce == PYGAMD_MEAN, "spmm_ag: sum / mean only");
  const Tensor x2 = x.reshape({x.size(0), -1});
  const int64_t n_rows = rowptr.numel() - 1;
  Tensor out = at::empty({n_rows, x2.size(1)}, x2.options());
  spmm_csr(rowptr, col, x2, reduce, n_rows, eid, w, c10::nullopt, hub_rows, hub_cptr, n_hub,
           n_chunks, hub_threshold, hub_chunk, out, false, 0, c10::nullopt, c10::nullopt,
           c10::nullopt);
  std::vector<int64_t> shape = x.sizes().vec();
  shape[0] = n_rows;
  return out.view(shape);
}
}  // namespace

TORCH_LIBRARY_IMPL(pyg_amd_c, CUDA, m) {
  m.impl("linear_ag", &linear_plain);
  m.impl("bias_act_ag", &bias_act_plain);
  m.impl("spmm_ag", &spmm_plain);
}

// the autograd key: these operators build their own graph nodes (torch::autograd::Function)
TORCH_LIBRARY_IMPL(pyg_amd_c, AutogradCUDA, m) {
  m.impl("linear_ag", &linear_ag);
  m.impl("bias_act_ag", &bias_act_ag);
  m.impl("spmm_ag", &spmm_ag);
}
