/*
 * pyg_amd.h — C ABI of the MI355X (gfx950) message-passing / aggregation backend.
 *
 * This header is the drop-in boundary.  The reference (PyG 2.9, pure Python) has no C ABI of its
 * own; its native seam is the set of `torch.ops.torch_sparse.*`, `torch_scatter.*` and
 * `pyg_lib.ops.*` symbols it *expects to exist* (SURVEY.md §2.3, §8(b) S2).  Every entry point
 * below names the reference call site (file:line under the reference root) it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked [host];
 *   - `stream` is a `hipStream_t` passed as `void*` (NULL = the null stream); every call is
 *     asynchronous on that stream unless marked [sync];
 *   - index tensors are int32 or int64, selected by `idx_dtype` (PYGAMD_IDX_I32 / _I64) — the
 *     reference accepts both (torch_geometric/typing.py:32-36);
 *   - features are fp32, row-major, with an explicit leading dimension in ELEMENTS
 *     (`ldx`, `ldo` ≥ F) so callers can aggregate in/out of wider buffers without copies;
 *   - every function returns a pygamd_status (0 = ok).  No function allocates device memory:
 *     workspaces are passed in, their size comes from the matching `*_workspace_bytes` query;
 *   - outputs never alias inputs.
 */
#ifndef PYG_AMD_H
#define PYG_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYGAMD_API __attribute__((visibility("default")))

#define PYGAMD_ABI_VERSION 10

typedef enum {
  PYGAMD_OK = 0,
  PYGAMD_ERR_INVALID_ARG = 1,   /* bad size / dtype / NULL pointer                     */
  PYGAMD_ERR_UNSUPPORTED = 2,   /* valid in the reference but not implemented natively */
  PYGAMD_ERR_WORKSPACE = 3,     /* workspace too small                                 */
  PYGAMD_ERR_HIP = 4            /* a HIP runtime call failed (see pygamd_last_hip_error) */
} pygamd_status;

typedef enum { PYGAMD_IDX_I32 = 0, PYGAMD_IDX_I64 = 1 } pygamd_idx_dtype;

/* layout of the gathered block of an aggregation (pygamd_spmm_args.x_format) */
typedef enum { PYGAMD_X_DENSE = 0, PYGAMD_X_COMPRESSED = 1 } pygamd_x_format;

/* reduce ids follow torch_geometric/utils/_scatter.py:14-138 */
typedef enum {
  PYGAMD_SUM = 0,
  PYGAMD_MEAN = 1,
  PYGAMD_MIN = 2,
  PYGAMD_MAX = 3,
  PYGAMD_MUL = 4,
  PYGAMD_ANY = 5
} pygamd_reduce;

/* ---- library info ------------------------------------------------------------------------- */
PYGAMD_API int pygamd_abi_version(void);
PYGAMD_API const char* pygamd_status_string(int status);
PYGAMD_API int pygamd_last_hip_error(void);          /* hipError_t of the last failing call */
PYGAMD_API const char* pygamd_build_arch(void);      /* "gfx950" */

/* ---- a12: index_sort ------------------------------------------------------------------------
 * Replaces `pyg_lib.ops.index_sort(inputs, max_value)` / `inputs.sort(stable=True)`
 * (torch_geometric/utils/_index_sort.py:10-32).  Stable LSD radix sort of non-negative integer
 * keys; `perm_out` (always int64, like torch.sort's indices) satisfies
 * keys_out[i] == keys_in[perm_out[i]] and is bit-identical to a stable sort.
 * `max_value` (>= every key, or <0 = unknown) bounds the radix passes (8 bits each; own kernels,
 * csrc/graph.hip).  Out of place (keys_in != keys_out), n < 2^32.                              */
PYGAMD_API int pygamd_index_sort_workspace_bytes(int idx_dtype, int64_t n, size_t* bytes /*[host]*/);
PYGAMD_API int pygamd_index_sort(const void* keys_in, int idx_dtype, int64_t n, int64_t max_value,
                                 void* keys_out, int64_t* perm_out, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* ---- a11: index2ptr / ptr2index -------------------------------------------------------------
 * `torch._convert_indices_from_coo_to_csr(index, size)` and
 * `arange(n).repeat_interleave(ptr.diff())` (torch_geometric/index.py:27-37).
 * `index` must be sorted ascending with values in [0, size).  ptr has size+1 entries.          */
PYGAMD_API int pygamd_index2ptr(const void* index, int idx_dtype, int64_t n, int64_t size,
                                void* ptr_out, void* stream);
PYGAMD_API int pygamd_ptr2index(const void* ptr, int idx_dtype, int64_t size, int64_t n,
                                void* index_out, void* stream);

/* Range check used by maybe_num_nodes / _index_select_safe
 * (torch_geometric/utils/num_nodes.py, nn/conv/message_passing.py:269-290).
 * Writes {min, max} (int64) to `minmax_out` (device, 2 entries); n == 0 gives {INT64_MAX, -1}. */
PYGAMD_API int pygamd_index_minmax(const void* index, int idx_dtype, int64_t n,
                                   int64_t* minmax_out, void* stream);

/* out[i] = src[perm[i]] for integer payloads (COO -> CSR column permutation,
 * torch_geometric/edge_index.py:605-623).  perm is int64.                                      */
PYGAMD_API int pygamd_permute_index(const void* src, int idx_dtype, const int64_t* perm,
                                    int64_t n, void* out, void* stream);
/* out[i] = (idx_dtype) perm[i]; narrows an int64 permutation to the graph's index dtype.      */
PYGAMD_API int pygamd_cast_index(const int64_t* src, int64_t n, int idx_dtype, void* out,
                                 void* stream);

/* out[i] = index[i] if 0 <= index[i] < size, else `size` (a sentinel group behind the last real
 * one); *err (device int32, may be NULL) is set to 1 when any entry was replaced.  The sorted route
 * of a large unsorted `scatter` (torch_geometric/utils/_scatter.py:68-100) sorts these keys, so an
 * out-of-range index can neither corrupt the plan nor cost a blocking min / max read: such rows
 * fall behind the last group and are skipped — as the atomic kernels skip them — and the flag
 * travels to the host as theirs does (ABI 9).  `out` has `out_dtype` (int32 needs
 * size < 2^31: narrower keys sort in fewer bytes), out != index.                               */
PYGAMD_API int pygamd_index_guard(const void* index, int idx_dtype, int64_t n, int64_t size,
                                  void* out, int out_dtype, int32_t* err, void* stream);

/* out[i] = in[0] + ... + in[i] over int32 / int64 counts: `torch.cumsum(x, 0)` where the
 * reference builds offsets from counts (torch_geometric/utils/functions.py:5-26 `cumsum`,
 * utils/_coalesce.py:186, the samplers' per-hop offsets).  `out` may be `in`.  One launch up to
 * 32 k elements, three above.  The sum must fit the dtype.                                    */
PYGAMD_API int pygamd_cumsum_workspace_bytes(int idx_dtype, int64_t n, size_t* bytes /*[host]*/);
PYGAMD_API int pygamd_cumsum(const void* in, int idx_dtype, int64_t n, void* out, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ---- (f)-4: sort_edge_index / coalesce (torch_geometric/utils/_sort_edge_index.py:105-113,
 * utils/_coalesce.py:131-176).  Like the reference, both sort the compound key
 * major * num_nodes + minor (major = row when by_row) with index_sort; these entry points build
 * the keys, flag the first entry of every run of equal sorted keys, and decode sorted keys back
 * into (row, col) — compacted to one entry per run when `scan` (the inclusive scan of the run
 * flags) is given, in which case gid_orig[perm[i]] (perm == NULL: gid_orig[i]) receives the run
 * slot of every original edge so that edge attributes can be merged with one scatter.
 * num_nodes^2 must fit int64 (the reference raises the same condition, _coalesce.py:134-135).   */
PYGAMD_API int pygamd_edge_key(const void* row, const void* col, int idx_dtype, int64_t E,
                               int64_t num_nodes, int by_row, int64_t* key_out, void* stream);
PYGAMD_API int pygamd_run_flags(const int64_t* key_sorted, int64_t E, int64_t* flag_out,
                                void* stream);
PYGAMD_API int pygamd_edge_unkey(const int64_t* key_sorted, const int64_t* scan,
                                 const int64_t* perm, int64_t E, int64_t num_nodes, int by_row,
                                 int idx_dtype, void* out_row, void* out_col, int64_t* gid_orig,
                                 void* stream);

/* ---- a10: hub plan for a CSR handle ---------------------------------------------------------
 * Rows with more than `threshold` stored entries ("hubs", power-law graphs) are split into
 * chunks of `chunk` entries so no single wavefront walks an unbounded row.  Produces, in
 * ascending row order, hub_rows[n_hub] and hub_chunk_ptr[n_hub+1] (exclusive scan of the chunk
 * counts).  Capacity of both outputs is `cap` (+1).  [sync]: n_hub / n_chunks are returned to
 * the host.  No reference equivalent (the reference never splits rows); internal to a10's cache
 * (torch_geometric/edge_index.py:677-696).                                                     */
PYGAMD_API int pygamd_hub_plan_workspace_bytes(int idx_dtype, int64_t n_rows, size_t* bytes);
PYGAMD_API int pygamd_hub_plan(const void* rowptr, int idx_dtype, int64_t n_rows,
                               int64_t threshold, int64_t chunk, void* hub_rows,
                               void* hub_chunk_ptr, int64_t cap, int64_t* n_hub_out /*[host]*/,
                               int64_t* n_chunks_out /*[host]*/, void* workspace,
                               size_t workspace_bytes, void* stream);

/* ---- a9/a13/a14: CSR SpMM (fused gather -> message -> segmented reduce) ---------------------
 * Replaces torch.ops.torch_sparse.spmm_{sum,mean,min,max}
 * (torch_geometric/edge_index.py:1798-1810) and the unfused
 * index_select -> message -> scatter chain of MessagePassing.propagate
 * (nn/conv/message_passing.py:421-563, utils/_scatter.py:68-80):
 *
 *   out[i, f] = post(i) * REDUCE_{k in [rowptr[i], rowptr[i+1])}
 *                   m(k, f) * x[col[k], f]
 *   m(k, f)  = (w ? w[e(k) * w_heads + f / head_dim] : 1) * (src_scale ? src_scale[col[k]] : 1)
 *   e(k)     = eid ? eid[k] : k
 *   post(i)  = reduce == MEAN ? 1 / max(deg(i), 1) : 1
 *
 * `col == NULL` means col[k] = k (segment reduce over contiguous rows: torch_scatter.segment_csr,
 * torch_geometric/utils/_segment.py:11-50).  Empty rows give 0 for every reduce.
 * For MIN/MAX `arg_out` (optional, same idx dtype, [n_rows, ldo]) receives the winning slot k
 * (first on ties) or -1 for an empty row; w/src_scale must be NULL.
 * The hub_* arrays come from pygamd_hub_plan (n_hub == 0: no splitting); `workspace` must hold
 * n_chunks * F floats (pygamd_spmm_csr_workspace_bytes).  Per-row accumulation order is the
 * slot order (deterministic; no atomics).                                                      */
typedef struct {
  const void* rowptr;        /* [n_rows + 1]                         */
  const void* col;           /* [nnz] or NULL                        */
  const void* eid;           /* [nnz] or NULL                        */
  const float* w;            /* [n_edges, w_heads] or NULL           */
  const float* src_scale;    /* [n_src] or NULL                      */
  const float* x;            /* [n_src, ldx]                         */
  float* out;                /* [n_rows, ldo]                        */
  void* arg_out;             /* [n_rows, ldo] or NULL (MIN/MAX only) */
  int64_t n_rows;
  int64_t n_src;
  int64_t F;
  int64_t ldx;
  int64_t ldo;
  int32_t idx_dtype;
  int32_t reduce;            /* PYGAMD_SUM | MEAN | MIN | MAX        */
  int32_t w_heads;           /* >= 1                                 */
  int32_t head_dim;          /* F / w_heads when w_heads > 1         */
  const void* hub_rows;      /* [n_hub] or NULL                      */
  const void* hub_chunk_ptr; /* [n_hub + 1] or NULL                  */
  int64_t n_hub;
  int64_t n_chunks;
  int64_t hub_threshold;
  int64_t hub_chunk;
  int32_t accumulate;        /* SUM/MEAN only: out += result (fused "x_root + aggregate") */
  int32_t hub_phase;         /* 0: rows + hubs; 1: non-hub rows only; 2: hub rows only (lets a
                                caller run row ranges as separate launches and the hubs once)  */
  int32_t* arg32_out;        /* MIN/MAX only, [n_rows, F] contiguous or NULL: per output the slot
                                OFFSET inside its row of the attaining neighbour (>= 0) when it
                                is the unique extremum; -1 for an empty row; -2 when the
                                gradient must be split (ties, or an extremum of exactly 0, which
                                ties with the zero-initialised output of the reference's
                                scatter_reduce_).  What pygamd_spmm_csr_minmax_backward_arg
                                consumes.                                                       */
  const float* relu_mask;    /* SUM/MEAN only, [n_rows, ld_mask] or NULL: the final value of
                                out[i, f] (after `accumulate`) is replaced by 0 where
                                relu_mask[i, f] <= 0 — aten::threshold_backward of the ReLU whose
                                OUTPUT relu_mask is, applied as the epilogue of the transposed
                                aggregation that produces that activation's gradient
                                (nn/models/basic_gnn.py:262-263 backward)                        */
  int64_t ld_mask;
  const uint32_t* relu_bits; /* the same mask as one BIT per element (32 x less to read), in
                                tiles of 32 rows x 32 columns: bit (f & 31) of word
                                relu_bits[((i >> 5) * ld_bits + (f >> 5)) * 32 + (i & 31)] is set
                                where the activation [i, f] is positive (the 32 words of a tile are
                                one 128-byte line: the producer's wave writes it with one store).
                                What pygamd_sage_layer_forward writes next to its output.  At most
                                one of relu_mask / relu_bits.                                      */
  int64_t ld_bits;           /* column blocks per row tile, >= ceil(F / 32); the array holds
                                ceil(n_rows / 32) * ld_bits * 32 words                             */
  const uint32_t* src_bits;  /* SUM/MEAN without w / src_scale (ignored with col == NULL, where
                                every row is read once anyway); NULL or one bit per
                                SOURCE row (bit j & 31 of word j >> 5, ceil(n_src / 32) words):
                                a clear bit promises that x[j, :] is all zero, and the row is not
                                read.  What pygamd_rows_pack writes.  The case: the gradient of a
                                loss taken on a training split (`out[train_idx]`, 8 % of the rows
                                of ogbn-products) entering the transposed aggregation of the last
                                layer — the reference gathers every zero row
                                (index_select on the full [E] index, message_passing.py:283-292)   */
  const int64_t* src_bits_set; /* NULL or a DEVICE counter of the set bits: with more than half of
                                the n_src rows live the kernel ignores src_bits (the lookup then
                                costs more than it saves) — decided on the device, no host sync    */
  int32_t x_format;          /* PYGAMD_X_DENSE (0), or PYGAMD_X_COMPRESSED: `x` is the block written
                                by pygamd_rows_compress / pygamd_sage_layer_fused.compressed_out
                                ([n_src, ldx] 32-bit words, F <= 256, F % 4 == 0, ldx >= F + 12);
                                SUM/MEAN without w / src_scale / src_bits; same sums, bit for bit,
                                as the dense rows                                                 */
  int32_t reserved0;         /* 0 */
  const void* rowend;        /* NULL, or [n_rows]: the slots of row r are [rowptr[r], rowend[r])
                                instead of [rowptr[r], rowptr[r + 1]) — rows that own fixed-stride,
                                partly filled slot blocks (a sampled batch at its static fan-out
                                capacity: pygamd_sample_slots).  SUM / MEAN without hubs, dense
                                rows; also honoured by pygamd_sage_layer_fused.                  */
  int64_t accumulate_rows;   /* with accumulate: only rows < accumulate_rows have an old value, the
                                others start from 0 (0 = every row) — the root gradient of a
                                sampled layer exists for its destination rows only.  No hubs.   */
} pygamd_spmm_args;

/* Lossless compression of a [n_rows, F <= 256] block with many exact zeros (the output of a ReLU)
 * for the row GATHER of the next aggregation, whose cost on this chip is proportional to the
 * 128-byte lines a row touches (scripts/gather_lines_probe.py): row i of `out` (32-bit words, row
 * pitch ld_out >= F + 12; 288 for F = 256) = [8 mask words | the kept values in column order],
 * bit (c & 31) of mask word (c >> 5) set where x[i, c] is kept = has a bit pattern other than
 * +0.0 (-0.0 and NaN are kept).  Words past the last kept value are unspecified.  A half-empty
 * 256-float row then occupies 4.4 lines instead of 8.  Consumed by pygamd_spmm_csr /
 * pygamd_sage_layer_fused with x_format = PYGAMD_X_COMPRESSED; pygamd_sage_layer_fused writes the
 * same format from its epilogue (compressed_out).  The reference keeps activations dense
 * (nn/models/basic_gnn.py:262-263).                                                              */
PYGAMD_API int pygamd_rows_compress(const float* x, int64_t ldx, int64_t n_rows, int64_t F,
                                    uint32_t* out, int64_t ld_out, void* stream);

/* Row-sparsity of a gradient block, found in the pass that lays it out for the backward of a
 * transform-then-aggregate layer: for every row i < n_rows of g ([n_rows, ldg], F columns)
 *   row_bits bit i   = [g[i, :] has an entry != 0]  (NaN counts; ceil(n_rows / 32) words, the
 *                      bits past n_rows are clear);
 *   scaled[i, f]     = g[i, f] * row_scale[i] (row_scale NULL: 1) for f < F, 0 for F <= f <
 *                      F_scaled (scaled NULL: skipped);
 *   copy[i, f]       = g[i, f] for f < F, 0 for F <= f < F_copy (copy NULL: skipped);
 *   *n_set           = the number of set bits (device int64, NULL: skipped; zeroed here).
 * `scaled` / `copy` may not alias g.  One read of g; replaces aten::mul + aten::copy_ + two fills
 * in the backward of the 256 -> 47 output layer and feeds pygamd_spmm_args.src_bits.             */
PYGAMD_API int pygamd_rows_pack(const float* g, int64_t ldg, int64_t n_rows, int64_t F,
                                const float* row_scale, float* scaled, int64_t ld_scaled,
                                int64_t F_scaled, float* copy, int64_t ld_copy, int64_t F_copy,
                                uint32_t* row_bits, int64_t* n_set, void* stream);

PYGAMD_API int pygamd_spmm_csr_workspace_bytes(const pygamd_spmm_args* args, size_t* bytes);
PYGAMD_API int pygamd_spmm_csr(const pygamd_spmm_args* args, void* workspace,
                               size_t workspace_bytes, void* stream);

/* Tie statistics for the backward of MIN/MAX, following the reference's CPU path
 * (zeros.scatter_reduce_(amax/amin, include_self=False), utils/_scatter.py:84-100): the
 * incoming gradient of out[i,f] is split evenly over all k with x[col[k],f] == out[i,f], and the
 * zero-initialised self entry counts as one more tie when out[i,f] == 0.
 * ntie_out[i,f] = (count_self ? [out[i,f] == 0] : 0) + #{k in row i : x[col[k],f] == out[i,f]}.
 * count_self = 0 gives the rule of torch._segment_reduce's backward (utils/_segment.py:37-50),
 * which has no self entry.                                                                      */
PYGAMD_API int pygamd_spmm_csr_tie_count(const void* rowptr, const void* col, int idx_dtype,
                                         const float* x, int64_t ldx, const float* out,
                                         int64_t ldo, int64_t n_rows, int64_t F, int count_self,
                                         float* ntie_out, void* stream);
/* grad_x[j,f] = sum over slots k of the TRANSPOSED handle (rowptr_t over sources, col_t = dst)
 *   [x[j,f] == out[col_t[k],f]] * grad_out[col_t[k],f] / ntie[col_t[k],f]                       */
PYGAMD_API int pygamd_spmm_csr_minmax_backward(const void* rowptr_t, const void* col_t,
                                               int idx_dtype, const float* x, int64_t ldx,
                                               const float* out, const float* grad_out,
                                               const float* ntie, int64_t ldo, int64_t n_src,
                                               int64_t F, float* grad_x, int64_t ldg,
                                               void* stream);
/* The same gradient, destination-driven and in ONE launch (no tie tensor, no by-source sort):
 * (rowptr, col) is the FORWARD's handle (rows = destinations, col = sources; col == NULL: slot k
 * is source k, i.e. segment reduce).  Per destination row the neighbours attaining out[i, f] are
 * counted (+1 for the zero-initialised self when count_self and out == 0), then every attaining
 * source receives grad_out[i, f] / ties through one fp32 atomic.  grad_x ([n_src, F], ld ldg) is
 * zeroed internally; sources reached from several destinations are summed in arbitrary order.  */
PYGAMD_API int pygamd_spmm_csr_minmax_backward_dst(const void* rowptr, const void* col,
                                                   int idx_dtype, const float* x, int64_t ldx,
                                                   const float* out, int64_t ldo,
                                                   const float* grad_out, int64_t ldgo,
                                                   int64_t n_rows, int64_t n_src, int64_t F,
                                                   int count_self, float* grad_x, int64_t ldg,
                                                   void* stream);
/* The fast path of the same gradient with the forward's saved `arg32` (pygamd_spmm_args.arg32_out):
 * every output whose extremum is unique sends its whole gradient to ONE source row with one fp32
 * atomic (no edge pass at all: arg32 and grad_out are read once, col is looked up inside the
 * row's slot range); outputs marked -2 are left to the two-pass kernel, which then only visits
 * rows that contain such an output.  grad_x is zeroed internally.  x / out are only read for the
 * marked outputs.                                                                                */
PYGAMD_API int pygamd_spmm_csr_minmax_backward_arg(const void* rowptr, const void* col,
                                                   int idx_dtype, const int32_t* arg32,
                                                   const float* x, int64_t ldx, const float* out,
                                                   int64_t ldo, const float* grad_out,
                                                   int64_t ldgo, int64_t n_rows, int64_t n_src,
                                                   int64_t F, int count_self, float* grad_x,
                                                   int64_t ldg, void* stream);
/* The same gradient WITHOUT the N x F scattered atomics (round 3).  Every output with a unique
 * extremum has exactly one winning edge: a destination-driven pass regroups the (feature, value)
 * pairs of grad_out by edge (`workspace`: n_rows x F pairs + one 32-bit (offset, count) word per
 * edge, pygamd_minmax_backward_src_workspace_bytes), then a source-driven pass over the TRANSPOSED
 * CSR (rowptr_t / col_t over the n_src sources; slot_map[e] = the by-destination slot of by-source
 * slot e, cf. EdgeIndex.src_slot_to_dst_slot) reads each out-edge's pairs — contiguous — and
 * sums them into the source's row in LDS, in slot order (deterministic for the unique extrema);
 * grad_x is written once, no memset.  Outputs marked -2 are then added by the two-pass tie kernel
 * like in pygamd_spmm_csr_minmax_backward_arg.  Needs F % 4 == 0, F <= 8192 and 16-byte aligned
 * rows (PYGAMD_ERR_UNSUPPORTED otherwise: use _arg).  Reference semantics: utils/_scatter.py:84-100
 * + ATen's amax / amin backward (ties share the gradient evenly).                               */
PYGAMD_API size_t pygamd_minmax_backward_src_workspace_bytes(int64_t n_rows, int64_t nnz,
                                                             int64_t F);
PYGAMD_API int pygamd_spmm_csr_minmax_backward_src(
    const void* rowptr, const void* col, const void* rowptr_t, const void* col_t,
    const void* slot_map, int idx_dtype, const int32_t* arg32, const float* x, int64_t ldx,
    const float* out, int64_t ldo, const float* grad_out, int64_t ldgo, int64_t n_rows,
    int64_t n_src, int64_t nnz, int64_t F, int count_self, void* workspace,
    size_t workspace_bytes, float* grad_x, int64_t ldg, void* stream);


/* ---- SDDMM: gradient w.r.t. edge weights ----------------------------------------------------
 * grad_w[e(k), h] = sum_{f in head h} grad_out[i, f] * x[col[k], f] * (src_scale? ...)  for k in
 * row i — the `value.requires_grad` branch of EdgeIndex._spmm (edge_index.py:1950-1953,
 * 1903-1922) and GATConv.message's alpha gradient (nn/conv/gat_conv.py:408-409).              */
PYGAMD_API int pygamd_sddmm_csr(const void* rowptr, const void* col, const void* eid,
                                int idx_dtype, const float* grad_out, int64_t ldg,
                                const float* x, int64_t ldx, int64_t n_rows, int64_t F,
                                int32_t w_heads, int32_t head_dim, float* grad_w, void* stream);
/* The two halves of a weighted aggregation's backward from ONE gather (ABI 9): over a CSR whose
 * row r holds `rows[r, :]` in registers and gathers x[col[k], :] for its slots k (e(k) = eid[k],
 * or k when eid is NULL),
 *   grad_w[e(k), h] = <rows[r, head h], x[col[k], head h]>          (the SDDMM above) and
 *   agg[r, f]       = sum_k w[e(k), head(f)] * x[col[k], f]          (pygamd_spmm_csr, per-head w).
 * GATConv's backward over the by-source form (nn/conv/gat_conv.py:387-409: rows = the projected
 * source features, x = the destinations' gradient rows, w = the attention coefficients) gets the
 * coefficient gradients and the aggregation's input gradient at the cost of one of them.  grad_w
 * must be zeroed by the caller when F > 256 (a head may span two lane groups: atomic adds).     */
PYGAMD_API int pygamd_sddmm_spmm_csr(const void* rowptr, const void* col, const void* eid,
                                     int idx_dtype, const float* rows, int64_t ldr,
                                     const float* x, int64_t ldx, int64_t n_rows, int64_t F,
                                     int32_t w_heads, int32_t head_dim, const float* w,
                                     float* grad_w, float* agg, int64_t lda, void* stream);

/* ---- a17: bias gradient of Linear --------------------------------------------------------------
 * out[f] = sum_r x[r, f]  (grad_bias = grad_out.sum(0), nn/dense/linear.py:121-127 backward).
 * `out` is zeroed internally; partial sums are combined with fp32 atomics.                     */
PYGAMD_API int pygamd_colsum(const float* x, int64_t ldx, int64_t n_rows, int64_t F, float* out,
                             void* stream);
/* ReLU backward fused with the bias gradient of the layer below (nn/models/basic_gnn.py:262-263
 * `self.act(x)` followed by a conv with bias):  grad_in[r,f] = act[r,f] <= 0 ? 0 : grad[r,f]
 * (aten::threshold_backward with threshold 0; `act` is the ReLU OUTPUT), and, when colsum_out is
 * not NULL, colsum_out[f] = sum_r grad_in[r,f] (zeroed internally, fp32 atomics).  grad_in may
 * alias grad.                                                                                   */
/* out = act(x + bias) in one pass (bias NULL: none; relu != 0: ReLU, NaN propagating) — the tail
 * of a conv layer: `out + self.bias` (gat_conv.py:378-385, gcn_conv.py:278-281) and the model's
 * activation (basic_gnn.py:262-263).  Its backward is pygamd_relu_backward_colsum on the OUTPUT. */
PYGAMD_API int pygamd_bias_act(const float* x, int64_t ldx, const float* bias, int64_t n_rows,
                               int64_t F, int relu, float* out, int64_t ldo, void* stream);
PYGAMD_API int pygamd_relu_backward_colsum(const float* grad, int64_t ldg, const float* act,
                                           int64_t lda, int64_t n_rows, int64_t F, float* grad_in,
                                           int64_t ldo, float* colsum_out, void* stream);

/* ---- a16/a17: segment_matmul (grouped GEMM over row segments, fp32 MFMA) ---------------------
 * Replaces pyg_lib.ops.segment_matmul(x, ptr, weight) (nn/conv/rgcn_conv.py:288,
 * nn/dense/linear.py:255):  out[ptr[g]:ptr[g+1]] = x[ptr[g]:ptr[g+1]] @ W[g].
 * `tiles` is a device array of int32 triples (segment, first row, rows <= tile_rows) covering
 * every non-empty segment in chunks of pygamd_segment_matmul_tile_rows() rows; W[g] is addressed
 * as w[g * w_seg_stride + k * w_stride_k + n * w_stride_n] so the same entry point computes the
 * input gradient (W^T: swap the two strides).  _wgrad: grad_w[g] = x[seg]^T @ g[seg] ([K, N]
 * row-major per segment; empty segments give 0); `chunks` uses the same triple format with any
 * chunk length (long segments are split so no wave walks a whole relation; fp32 atomics).
 * `blocks` > 1 (block-diagonal weights, RGCNConv(num_blocks=), rgcn_conv.py:222-244): group
 * g = segment * blocks + b multiplies the column block [b*K, (b+1)*K) of its rows of x by W[g]
 * into the column block [b*N, (b+1)*N) of out (ldx >= blocks*K, ldo >= blocks*N); no transposed
 * copies of the activations or the weights are needed.  blocks = 1: plain segment_matmul.      */
PYGAMD_API int pygamd_segment_matmul_tile_rows(void);
/* `n_groups` = number of weight matrices (segments * blocks).  `workspace` (device, 16-byte
 * aligned, pygamd_segment_matmul_workspace_bytes(n_groups, K, N); may be NULL): with it, in
 * PYGAMD_GEMM_SPLIT_BF16 mode, for K <= 128, K % 4 == 0 and 16-byte-aligned rows of x (weights
 * in any strides: W^T is read where it lies), the product runs on the convert-once split kernel
 * (csrc/segmm.hip: the weights' bf16 term planes are written there by a pre-pass, every call);
 * otherwise on the exact fp32 kernel (ABI 9).                                                  */
PYGAMD_API int pygamd_segment_matmul_workspace_bytes(int64_t n_groups, int64_t K, int64_t N,
                                                     size_t* bytes /*[host]*/);
/* `x_rows` / `g_rows` (device int64, or NULL): operand row r is x[x_rows[r]] (g[g_rows[r]]) — the
 * rows of a smaller tensor gathered on the fly.  RGCNConv sums its transformed (relation,
 * destination) rows per destination (rgcn_conv.py:284-290 + aggregate); the backward of that sum
 * hands every pair row the gradient row of its destination, and with the index the input- and
 * weight-gradient launches read those rows where they are instead of from a gathered [pairs, F]
 * copy (ABI 9).                                                                                */
PYGAMD_API int pygamd_segment_matmul(const float* x, int64_t ldx, const int64_t* x_rows,
                                     const float* w, int64_t w_seg_stride, int64_t w_stride_k,
                                     int64_t w_stride_n, int64_t n_groups, const int32_t* tiles,
                                     int64_t n_tiles, int64_t K, int64_t N, int64_t blocks,
                                     float* out, int64_t ldo, void* workspace,
                                     size_t workspace_bytes, void* stream);
PYGAMD_API int pygamd_segment_matmul_wgrad(const float* x, int64_t ldx, const float* g,
                                           int64_t ldg, const int64_t* g_rows,
                                           const int32_t* chunks, int64_t n_chunks,
                                           int64_t n_seg, int64_t K, int64_t N, int64_t blocks,
                                           float* grad_w, void* stream);

/* ---- a2: gather (index_select along dim 0) --------------------------------------------------
 * out[e, :] = x[index[e], :]  (nn/conv/message_passing.py:263-290, collect.jinja:118-127).
 * Out-of-range indices set *err_flag (device int32, optional) to 1 and read row 0.             */
PYGAMD_API int pygamd_gather_rows(const float* x, int64_t ldx, int64_t n_src, const void* index,
                                  int idx_dtype, int64_t n, int64_t F, float* out, int64_t ldo,
                                  int32_t* err_flag, void* stream);

/* Fused gather -> scale -> scatter-add on an unsorted edge list (the COO
 * fallback of the SpMM for graphs used once, e.g. the backward of a sampled mini-batch):
 *   out[scatter_idx[e], :] += (scale ? scale[gather_idx[e]] : 1) * (w ? w[e] : 1) * x[gather_idx[e], :]
 * `out` must be initialised by the caller (usually zeros); fp32 atomics.  `n_valid` (device int64,
 * may be NULL): only the first min(*n_valid, n_edges) entries of the fixed-capacity edge list are
 * real (a sampled hop at its static capacity: no host read, hipGraph-capturable).               */
PYGAMD_API int pygamd_gather_scatter_add(const float* x, int64_t ldx, const void* gather_idx,
                                         const void* scatter_idx, int idx_dtype,
                                         const float* scale, const float* w, int64_t n_edges,
                                         const int64_t* n_valid, int64_t F, float* out,
                                         int64_t ldo, void* stream);

/* ---- a5: scatter (unsorted COO, atomics) ----------------------------------------------------
 * out[index[e], :] (reduce)= src[e, :]   (utils/_scatter.py:14-138).  `out` must be
 * pre-initialised by pygamd_scatter_init; MEAN/MIN/MAX need `count` ([dim_size] float, zeroed)
 * and a pygamd_scatter_finalize call (mean: divide by clamp(count,1); min/max: untouched -> 0). */
PYGAMD_API int pygamd_scatter_init(float* out, int64_t ldo, int64_t dim_size, int64_t F,
                                   int reduce, float* count, void* stream);
PYGAMD_API int pygamd_scatter_rows(const float* src, int64_t lds, const void* index,
                                   int idx_dtype, int64_t n, int64_t F, float* out, int64_t ldo,
                                   int64_t dim_size, int reduce, float* count,
                                   int32_t* err_flag, void* stream);
PYGAMD_API int pygamd_scatter_finalize(float* out, int64_t ldo, int64_t dim_size, int64_t F,
                                       int reduce, const float* count, void* stream);
/* backward of MIN/MAX/MUL-free scatter: see utils/_scatter.py:84-100 (same tie rule as above).
 * ntie[g,f] = [out[g,f]==0] + #{e: index[e]==g, src[e,f]==out[g,f]}  (initialised internally);
 * grad_src[e,f] = [src[e,f]==out[g,f]] * grad_out[g,f] / ntie[g,f].                             */
PYGAMD_API int pygamd_scatter_minmax_tie_count(const float* src, int64_t lds, const void* index,
                                               int idx_dtype, int64_t n, int64_t F,
                                               const float* out, int64_t ldo, int64_t dim_size,
                                               float* ntie, void* stream);
PYGAMD_API int pygamd_scatter_minmax_backward(const float* src, int64_t lds, const void* index,
                                              int idx_dtype, int64_t n, int64_t F,
                                              const float* out, const float* grad_out,
                                              const float* ntie, int64_t ldo, int64_t dim_size,
                                              float* grad_src, int64_t ldg, void* stream);

/* Backward of scatter(reduce='mul'), i.e. of ones.scatter_reduce_('prod', include_self=True)
 * (utils/_scatter.py:119-133; ATen scatter_reduce_backward): with z zeros scattered into
 * (group, f):  z == 0: g * out / src_e;  z == 1: the zero element gets g * prod(others), all
 * other elements 0;  z >= 2: 0.  Rows with an out-of-range index get 0.  Workspace =
 * dim_size * F * 8 bytes.                                                                      */
PYGAMD_API int pygamd_scatter_mul_backward_workspace_bytes(int64_t dim_size, int64_t F,
                                                           size_t* bytes /*[host]*/);
PYGAMD_API int pygamd_scatter_mul_backward(const float* src, int64_t lds, const void* index,
                                           int idx_dtype, int64_t n, int64_t F, const float* out,
                                           const float* grad_out, int64_t ldo, int64_t dim_size,
                                           float* grad_src, int64_t ldg, void* workspace,
                                           size_t workspace_bytes, void* stream);

/* ---- a6: scatter_argmax (1-D) ---------------------------------------------------------------
 * utils/_scatter.py:147-184: out[g] = the LAST e (largest e) with src[e] == max of group g,
 * dim_size-1 for empty groups.  `gmax` is a [dim_size] float scratch.                          */
PYGAMD_API int pygamd_scatter_argmax(const float* src, const void* index, int idx_dtype,
                                     int64_t n, int64_t dim_size, float* gmax, void* arg_out,
                                     void* stream);

/* ---- a8: segment softmax (edge softmax) -----------------------------------------------------
 * utils/_softmax.py:12-92, ptr branch :60-81 (also pyg_lib.ops.softmax_csr): for each segment
 * s and column h:  out[k,h] = exp(src[k,h] - max_s) / (sum_s exp(src - max_s) + 1e-16).
 * src/out are [n, H] contiguous.  Backward: grad_src = out * (grad_out - sum_s(out*grad_out)).  */
PYGAMD_API int pygamd_segment_softmax_forward(const float* src, const void* ptr, int idx_dtype,
                                              int64_t n_seg, int64_t H, float* out,
                                              void* stream);
PYGAMD_API int pygamd_segment_softmax_backward(const float* out, const float* grad_out,
                                               const void* ptr, int idx_dtype, int64_t n_seg,
                                               int64_t H, float* grad_src, void* stream);

/* The same softmax over an UNSORTED `index` ([n], values in [0, N)) — the index branch of
 * utils/_softmax.py:82-88 (scatter-max, gather, exp, scatter-sum, gather, div) as four launches:
 * group maxima (atomic float max), exp + group sums (fp32 atomics), normalisation.  `workspace`:
 * 2 x N x H floats (forward), N x H floats (backward).  Rows whose index is out of range are
 * skipped (their output is left untouched by the forward, their gradient is 0) and, like in the
 * scatter kernels, reported: `*err` ([dev] int32, may be NULL, zeroed by the caller) becomes
 * non-zero — the reference's scatter / index_select path raises there (ABI 8).               */
PYGAMD_API int pygamd_softmax_index_forward(const float* src, const void* index, int idx_dtype,
                                            int64_t n, int64_t H, int64_t N, float* workspace,
                                            float* out, int32_t* err, void* stream);
PYGAMD_API int pygamd_softmax_index_backward(const float* out, const float* grad_out,
                                             const void* index, int idx_dtype, int64_t n,
                                             int64_t H, int64_t N, float* workspace,
                                             float* grad_src, void* stream);

/* segment_logsumexp (utils/_segment.py:53-80): out[s,h] = log(sum_{k in s} exp(src[k,h])),
 * evaluated with the segment maximum subtracted; an empty segment gives 0.  src is [n, H]
 * contiguous, out [n_seg, H].  Backward: grad_src[k,h] = exp(src[k,h] - out[s,h]) * grad_out[s,h]. */
PYGAMD_API int pygamd_segment_logsumexp_forward(const float* src, const void* ptr, int idx_dtype,
                                                int64_t n_seg, int64_t H, float* out,
                                                void* stream);
PYGAMD_API int pygamd_segment_logsumexp_backward(const float* src, const float* out,
                                                 const float* grad_out, const void* ptr,
                                                 int idx_dtype, int64_t n_seg, int64_t H,
                                                 float* grad_src, void* stream);

/* ---- a15: GAT node terms ----------------------------------------------------------------------
 * out_a[n,h] = sum_c x[n, h*C + c] * att_a[h*C + c]  (and out_b with att_b when given) — the
 * `(x * att).sum(-1)` pair of nn/conv/gat_conv.py:330-332 in one pass over x.  Backward:
 * grad_x[n,f] (+)= grad_a[n,h(f)] att_a[f] + grad_b[n,h(f)] att_b[f]  (grad_x may be NULL;
 * `accumulate` != 0 adds to what grad_x holds — the gradient of the same x through the
 * aggregation, so that the two do not meet in a separate add pass; ABI 8),
 * grad_att_a[f] = sum_n grad_a[n,h(f)] x[n,f]  (zeroed internally, atomics).                     */
PYGAMD_API int pygamd_head_dot_forward(const float* x, int64_t ldx, const float* att_a,
                                       const float* att_b, int64_t n_rows, int64_t H, int64_t C,
                                       float* out_a, float* out_b, void* stream);
PYGAMD_API int pygamd_head_dot_backward(const float* x, int64_t ldx, const float* att_a,
                                        const float* att_b, const float* grad_a,
                                        const float* grad_b, int64_t n_rows, int64_t H,
                                        int64_t C, float* grad_x, int64_t ldg, int accumulate,
                                        float* grad_att_a, float* grad_att_b, void* stream);

/* ---- a15: fused GAT edge logits ------------------------------------------------------------
 * nn/conv/gat_conv.py:387-406 on a dst-sorted handle: for slot k in row i,
 *   logit = leaky_relu(alpha_src[col[k],h] + alpha_dst[i,h], slope); softmax over the row.
 * alpha_out is [nnz, H] in SLOT order.  Backward returns grad_alpha_src / grad_alpha_dst
 * accumulated with atomics (grad buffers must be zeroed).                                      */
PYGAMD_API int pygamd_gat_edge_softmax_forward(const void* rowptr, const void* col,
                                               int idx_dtype, const float* alpha_src,
                                               const float* alpha_dst, int64_t n_rows,
                                               int64_t H, float slope, float* alpha_out,
                                               void* stream);
PYGAMD_API int pygamd_gat_edge_softmax_backward(const void* rowptr, const void* col,
                                                int idx_dtype, const float* alpha_src,
                                                const float* alpha_dst, const float* alpha_out,
                                                const float* grad_alpha, int64_t n_rows,
                                                int64_t H, float slope, float* grad_alpha_src,
                                                float* grad_alpha_dst, void* stream);

/* ---- §8(f)-1 (next): one hop of uniform neighbour sampling ------------------------------------
 * Device-side counterpart of torch.ops.pyg.neighbor_sample (sampler/neighbor_sampler.py:550-577)
 * on a CSC graph (colptr over destinations, row = source of every slot).  For frontier node
 * frontier[f] the caller provides offsets[f] / offsets[f+1] with offsets[f+1]-offsets[f] =
 * min(deg, k) (or deg for k < 0); the kernel writes, for every sampled slot, the global source
 * id, f (the position of the destination in the frontier) and the CSC slot (-> edge id through
 * the handle's permutation).  deg <= count: all neighbours; else a uniform subset (Floyd's
 * algorithm, counter-based hash of (seed, node, draw): reproducible).  max_per_node = the largest
 * bounded count requested (<= pygamd_sample_max_fanout(); pass 0 when every node takes all).
 * flags bit 0 (the reference's `replace=True`, loader/neighbor_loader.py:209; needs a bounded
 * fan-out): every frontier node with at least one in-neighbour gets exactly k independent uniform
 * draws (offsets from pygamd_sample_counts with replace set).  flags bit 1: the draws of a node
 * depend on its position in the frontier as well (disjoint sampling: one tree per seed).
 * seed_dev (device uint64, may be NULL) is added to `seed` on the device: a captured graph draws a
 * fresh batch at every replay by bumping that word.                                            */
PYGAMD_API int pygamd_sample_max_fanout(void);
PYGAMD_API int pygamd_sample_neighbors(const void* colptr, const void* row, int idx_dtype,
                                       const void* frontier, int64_t n_frontier,
                                       const void* offsets, int64_t max_per_node, uint64_t seed,
                                       int flags, const uint64_t* seed_dev, void* src_out,
                                       void* dstpos_out, void* slot_out, void* stream);

/* cnt[f] = min(deg(frontier[f]), k) (k < 0: deg; replace != 0 and k >= 0: k wherever deg > 0, else
 * 0) — the per-node sample counts of one hop.
 * `n_valid` (device int64, may be NULL): only the first *n_valid entries of the fixed-capacity
 * `frontier` are real; the rest (which must hold valid node ids, e.g. 0) get count 0.  With it a
 * hop can be sized by the static bound frontier x fan-out and run WITHOUT a host sync.           */
PYGAMD_API int pygamd_sample_counts(const void* colptr, int idx_dtype, const void* frontier,
                                    int64_t n, int64_t k, int replace, const int64_t* n_valid,
                                    void* cnt_out, void* stream);
/* Relabelling of the sampled sources (global -> local ids, new nodes in order of first
 * appearance, deterministic).  `local_map` has one entry per graph node; entries
 * not in the batch hold a value below -(m+1) (the host side uses the type's minimum).
 * phase 0 "claim":  local[src[e]] = max(local[src[e]], -(e+2));
 * phase 1 "flag":   flag_or_scan[e] = (local[src[e]] == -(e+2));        (caller scans inclusively)
 * phase 2 "assign": claimants write local[s] = base + rank and out[rank] = s (out = new nodes);
 * phase 3 "lookup": out[e] = local[src[e]].
 * `m_dev` / `base_dev` (device int64, may be NULL) override min(m, *m_dev) / *base_dev: the
 * number of sampled edges and of batch nodes so far stay on the device, `m` is then the static
 * capacity of the hop (entries past *m_dev flag 0 / look up 0).                                  */
PYGAMD_API int pygamd_relabel(int phase, const void* src, int idx_dtype, int64_t m,
                              const int64_t* m_dev, void* local_map, int64_t* flag_or_scan,
                              int64_t base, const int64_t* base_dev, void* out, void* stream);

/* ---- a18: one-pass multi-reduce (FusedAggregation) ---------------------------------------------
 * nn/aggr/fused.py:191-336 shares the group count, the sum and the sum of squares between
 * sum / mean / var / std / min / max.  Here ONE read of the rows produces all requested statistics
 * of every group: rows of group g are x[perm[k]] for k in [rowptr[g], rowptr[g+1]) (`perm` =
 * the stable sort permutation of the aggregation index, or NULL when the rows are already
 * grouped, i.e. the `ptr` form).  Null outputs are skipped; empty groups give 0.  Outputs are
 * [n_rows, F] with leading dimension ldo.                                                      */
PYGAMD_API int pygamd_multi_reduce_csr(const void* rowptr, const void* perm, int idx_dtype,
                                       const float* x, int64_t ldx, int64_t n_rows, int64_t F,
                                       float* out_sum, float* out_sq, float* out_min,
                                       float* out_max, int64_t ldo, void* stream);

/* ---- a17 / f3: the dense feature transform on the fp32 matrix cores ---------------------------
 * `F.linear(x, weight, bias)` of nn/dense/linear.py:121-127 — what SAGEConv's `lin_l(agg) +
 * lin_r(x)` (sage_conv.py:134-139), GCNConv's `lin(x)` (gcn_conv.py:260) etc. end in — and its
 * two gradients, written for v_mfma_f32_32x32x2_f32 (exact fp32).  Row-major, explicit leading
 * dimensions in elements, so halves of wider `[agg | x]` buffers are read / written in place.
 *   forward : out[M, N] = act(x[M, K] @ w[N, K]^T + bias[N]);  relu != 0 fuses the activation
 *             (basic_gnn.py:258-266), accumulate != 0 adds onto `out`.
 *   dgrad   : out[M, K] = g[M, N] @ w[N, K], the weight handed over TRANSPOSED (w_t[K, N],
 *             contiguous rows); columns [0, n_scaled) are multiplied by row_scale[row] in the
 *             epilogue (the 1/deg of a mean aggregation that follows, utils/_scatter.py:72-80);
 *             relu_mask ([M, ld_mask] or NULL): out[r, c] = 0 where relu_mask[r, c] <= 0, applied
 *             last — the backward of the ReLU whose output relu_mask is (basic_gnn.py:262-263);
 *             relu_bits (or NULL) is the same mask as one bit per element in the tiled layout of
 *             pygamd_spmm_args.relu_bits (ld_bits >= ceil(K / 32)); at most one of the two.
 *   wgrad   : out[N, K] = g[M, N]^T @ x[M, K]; deterministic (split over M, slabs summed in
 *             order); workspace from the _workspace_bytes query.  bias_grad ([N] or NULL)
 *             receives the column sums of g (the bias gradient of the same Linear,
 *             nn/dense/linear.py:121-127 backward) from the same pass over g: per-split sums taken
 *             while the rows are staged, added in split order (deterministic).                 */
/* Arithmetic of the three entry points below and of pygamd_sage_layer_forward (process-wide):
 *   PYGAMD_GEMM_FP32       (default) v_mfma_f32_32x32x2_f32: IEEE fp32 products and sums, bitwise
 *                          an fmaf chain over k;
 *   PYGAMD_GEMM_SPLIT_BF16 every fp32 operand as the exact sum of three bf16 terms, the six
 *                          leading cross products on v_mfma_f32_32x32x16_bf16 with fp32
 *                          accumulation: the same fp32 inputs and outputs, error against fp64 at
 *                          or below the fmaf chain's (profiles/r02_split_bf16_accuracy_probe.txt),
 *                          not bitwise equal to it; an Inf operand yields NaN.               */
#define PYGAMD_GEMM_FP32 0
#define PYGAMD_GEMM_SPLIT_BF16 1
PYGAMD_API int pygamd_set_gemm_mode(int mode);
PYGAMD_API int pygamd_get_gemm_mode(void);
/* Workspace of the forward / dgrad2 entry points (both are one "NT" product out[M, N_out] over a
 * reduction of K_red columns): 0 when the output has enough row tiles to fill the chip (every
 * full-batch layer); for few rows (sampled blocks of 1 k .. 16 k rows, Cora-sized inputs) room for
 * the partial tiles of a split over the reduction, combined in slice order by a second pass
 * (deterministic).  Without it (NULL) the launch runs unsliced: same result, fewer workgroups.  */
PYGAMD_API int pygamd_linear_nt_workspace_bytes(int64_t M, int64_t N_out, int64_t K_red,
                                                size_t* bytes /*[host]*/);
PYGAMD_API int pygamd_linear_forward(const float* x, int64_t ldx, const float* w, int64_t ldw,
                                     const float* bias, int64_t M, int64_t K, int64_t N, int relu,
                                     int accumulate, float* out, int64_t ldo, void* workspace,
                                     size_t workspace_bytes, void* stream);
PYGAMD_API int pygamd_linear_dgrad(const float* g, int64_t ldg, const float* w_t, int64_t ldwt,
                                   const float* row_scale, int64_t n_scaled, int64_t M, int64_t N,
                                   int64_t K, int accumulate, const float* relu_mask,
                                   int64_t ld_mask, const uint32_t* relu_bits, int64_t ld_bits,
                                   float* out, int64_t ldo, void* stream);
/* pygamd_linear_dgrad with a second output: out_scaled[M, K] (or NULL) = out * row_scale[row] in
 * every column (row_scale required then; not with accumulate).  The one-kernel layer backward
 * (pygamd_sage_layer_fused with mask_bits) gathers the 1/deg-scaled gradient rows and takes the
 * unscaled ones as its root operand: this writes both from one pass.                             */
PYGAMD_API int pygamd_linear_dgrad2(const float* g, int64_t ldg, const float* w_t, int64_t ldwt,
                                    const float* row_scale, int64_t n_scaled, int64_t M,
                                    int64_t N, int64_t K, int accumulate, const float* relu_mask,
                                    int64_t ld_mask, const uint32_t* relu_bits, int64_t ld_bits,
                                    float* out, int64_t ldo, float* out_scaled, int64_t ld_scaled,
                                    void* workspace, size_t workspace_bytes, void* stream);
PYGAMD_API int pygamd_linear_wgrad_workspace_bytes(int64_t M, int64_t N, int64_t K,
                                                   size_t* bytes /*[host]*/);
PYGAMD_API int pygamd_linear_wgrad(const float* g, int64_t ldg, const float* x, int64_t ldx,
                                   int64_t M, int64_t N, int64_t K, int accumulate,
                                   int wgs_per_cu /* 0 / 2: fill the chip; 1: leave room for a
                                   bandwidth-bound kernel running on another stream */,
                                   float* out, int64_t ldo, float* bias_grad, void* workspace,
                                   size_t workspace_bytes, void* stream);
/* The same against two operands side by side: out[N, K1 + K2] = g^T @ [x | x2] (x2 NULL, K2 = 0:
 * identical to pygamd_linear_wgrad) — SAGEConv's `[grad W_l | grad W_r]` (sage_conv.py:134-139
 * backward) from the aggregated rows and the layer input where they are, without a concatenated
 * copy.  Workspace as for K = K1 + K2.                                                           */
PYGAMD_API int pygamd_linear_wgrad2(const float* g, int64_t ldg, const float* x, int64_t ldx,
                                    int64_t K1, const float* x2, int64_t ldx2, int64_t K2,
                                    int64_t M, int64_t N, int accumulate, int wgs_per_cu,
                                    float* out, int64_t ldo, float* bias_grad, void* workspace,
                                    size_t workspace_bytes, void* stream);

/* ---- f1b: static-shape ("slot") sampled batches — csrc/minibatch.hip ---------------------------------
 * The sampling contract of pygamd_sample_neighbors (sampler/neighbor_sampler.py:550-577: per
 * frontier node a uniform min(deg, k)-subset of its in-neighbours, no replacement) with an output
 * layout in which a whole training step has the same shapes and row ranges every batch and reads
 * nothing back to the host.  Batch-local node ids are BLOCK POSITIONS: block 0 = the B seeds,
 * block h + 1 = the fanout[h] slots of every position of block h; bases[b] = first id of block b
 * (bases[hops + 1] = all rows R); slot e (global slot index, hop by hop) is row B + e.
 *   node_g  [R]   int64  graph node held by row r, -1 = hole (a duplicate, or nothing sampled)
 *   src_g   [R-B] int64  graph node sampled into slot e, -1 = slot not filled
 *   src_id  [R-B] int32  row that holds the features of slot e's source (the earliest occurrence)
 *   row_end [bases[hops]] int32  row r's slots are [row_begin[r], row_end[r]) with the static
 *           row_begin[r] = (bases[b + 1] - B) + (r - bases[b]) * fanout[b] for r in block b:
 *           pass them as pygamd_spmm_args.rowptr / .rowend with col = src_id
 *   inv_cnt [bases[hops]] float  1 / max(row_end - row_begin, 1)
 *   local_map [num_nodes] int64, zero-initialised ONCE: duplicates resolve through atomicMax of
 *           epoch << 32 | (2^32 - 1 - row) — never reset; *epoch_dev (device int64, >= 1) must
 *           grow from batch to batch and also salts the draws.
 * Call order per batch: seed, then per hop sample + resolve (resolve also counts, per source row,
 * the entries of the transposed CSRs the hop belongs to: counts[c], c < n_counts), then transpose
 * (one scan launch + one fill launch for all n_csr transposed CSRs; CSR c covers the slots
 * [0, n_slots[c]) with sources in rows [0, n_src_rows[c]); counts / cursor zeroed by the caller),
 * then gather (x[node_g] -> out, holes = zero rows).                                               */
PYGAMD_API int pygamd_slots_max_fanout(void);
PYGAMD_API int pygamd_slots_max_hops(void);
PYGAMD_API int pygamd_slots_seed(const void* seeds, int idx_dtype, int64_t B,
                                 const int64_t* epoch_dev, int64_t* local_map, int64_t* node_g,
                                 void* stream);
PYGAMD_API int pygamd_slots_sample(const void* colptr, const void* row, int idx_dtype,
                                   const int64_t* node_g, int64_t frontier_base,
                                   int64_t n_frontier, int fanout, int64_t slot_base, int64_t B,
                                   uint64_t seed, int hop, const int64_t* epoch_dev,
                                   int64_t* local_map, int64_t* src_g, int32_t* row_end,
                                   float* inv_cnt, void* stream);
PYGAMD_API int pygamd_slots_resolve(const int64_t* src_g, int64_t slot_base, int64_t n_slots,
                                    int64_t B, const int64_t* epoch_dev,
                                    const int64_t* local_map, int32_t* src_id,
                                    int64_t* node_g, int32_t* const* counts /*[host]*/,
                                    int n_counts, void* stream);
PYGAMD_API int pygamd_slots_gather(const float* x, int64_t ldx, int64_t F, const int64_t* node_g,
                                   int64_t n_rows, float* out, int64_t ldo, void* stream);
PYGAMD_API int pygamd_slots_transpose(const int64_t* src_g, const int32_t* src_id, int hops,
                                      const int32_t* fanout /*[host]*/,
                                      const int64_t* bases /*[host, hops + 2]*/, int n_csr,
                                      const int64_t* n_slots /*[host]*/,
                                      const int64_t* n_src_rows /*[host]*/,
                                      int32_t* const* counts /*[host]*/,
                                      int32_t* const* cursor /*[host]*/,
                                      int32_t* const* ptr /*[host]*/,
                                      int32_t* const* col /*[host]*/, void* stream);

/* ---- f1c: the loss and optimizer ends of a captured batch step — csrc/train.hip ----------------------
 * The reference's mini-batch loop (examples/multi_gpu/distributed_sampling.py:104-117) ends every
 * batch with `loss = F.cross_entropy(out, batch.y[:batch.batch_size]); loss.backward();
 * optimizer.step()` (torch.optim.Adam).  Inside a hipGraph every launch costs ~5 us, so these two
 * ends are 2 + 1 launches here (before: 7 ATen launches for the loss of 1024 rows, 42 us of
 * multi-tensor Adam for 0.2 M parameters, a concatenation / transpose launch per weight use).
 *
 * pygamd_cross_entropy_step: loss[0] = mean_r (logsumexp(logits[r, :]) - logits[r, label_r]),
 * grad[r, c] = (softmax(logits[r, :])[c] - [c == label_r]) / B — F.cross_entropy (reduction
 * 'mean', no class weights, no label smoothing) and its backward for an upstream gradient of 1.
 * label_r = y[label_idx[r]] (label_idx NULL: y[r]) — the seeds' labels are read from the graph's
 * label vector, no gathered copy.  row_idx (optional): row r of the loss is logits[row_idx[r], :]
 * (n_logit_rows = the rows `logits` holds) — `F.cross_entropy(out[train_idx], y[train_idx])` of a
 * full-batch model without the gathered copy of the rows; grad stays compact ([B, C]).  A label outside [0, C) sets *err_flag (device int32, optional)
 * and contributes neither loss nor gradient (no ignore_index: the mean divides by B).  The row
 * losses are added up in row order by a second, one-workgroup launch (deterministic; a
 * last-workgroup-done ticket inside one launch cost 28 us in fences for 1,024 rows).  step_counter
 * (device int64, optional) is incremented by one per call — the optimizer's step count of a
 * captured step (pygamd_adam_step's step_dev), kept by a launch the step has anyway.
 * Workspace: ..._workspace_bytes(B) (the row losses).                                             */
PYGAMD_API int pygamd_cross_entropy_step_workspace_bytes(int64_t B, size_t* bytes /*[host]*/);
PYGAMD_API int pygamd_cross_entropy_step(const float* logits, int64_t ld, const int64_t* row_idx,
                                         int64_t n_logit_rows, int64_t B, int64_t C,
                                         const int64_t* y, const int64_t* label_idx, float* grad,
                                         int64_t ldg, float* loss, void* workspace,
                                         size_t workspace_bytes, int32_t* err_flag,
                                         int64_t* step_counter, void* stream);
/* pygamd_adam_step: torch.optim.Adam (amsgrad / maximize off; weight_decay = the L2 form) over ONE
 * flat float32 buffer of n parameters with gradients `grad * grad_scale` (grad_scale = 1 / world
 * size after a SUM all-reduce).  The step count is read from the device: step = *step_dev -
 * step_base >= 1 (a captured step bumps *step_dev itself; bias corrections in float32 as the
 * capturable optimizer evaluates them).  `transposed` (optional): for each of n_segments (<=
 * PYGAMD_ADAM_MAX_SEGMENTS) row-major blocks param[seg_off .. + rows * cols) the updated values
 * are also stored transposed at transposed[seg_t_off + c * rows + r] — the W^T operand of
 * pygamd_linear_dgrad, kept current without a transpose launch per use.                          */
#define PYGAMD_ADAM_MAX_SEGMENTS 8
PYGAMD_API int pygamd_adam_step(float* param, const float* grad, float* exp_avg,
                                float* exp_avg_sq, int64_t n, const int64_t* step_dev,
                                int64_t step_base, double lr, double beta1, double beta2,
                                double eps, double weight_decay, double grad_scale,
                                float* transposed, int n_segments,
                                const int64_t* seg_off /*[host]*/,
                                const int32_t* seg_rows /*[host]*/,
                                const int32_t* seg_cols /*[host]*/,
                                const int64_t* seg_t_off /*[host]*/, void* stream);

/* ---- f3: SAGEConv layer forward in one kernel ----------------------------------------------------
 * y[i, :] = act([aggr_{j->i} x[j] | x_root[i]] @ w[Fo, 2F]^T + bias) — `propagate` + `lin_l(agg)
 * + lin_r(x)` of nn/conv/sage_conv.py:134-139 with the aggregated 32-row tile handed from the
 * gather phase to the MFMA loop through LDS.  `graph` describes the aggregation exactly as for
 * pygamd_spmm_csr (reduce SUM or MEAN, no edge weights; x / ldx = the rows that are gathered;
 * out / ldo = the global agg buffer, always required: hub rows are aggregated there first by the
 * two-stage hub kernels, and with save_agg != 0 every row is stored there once for the weight
 * gradient).  Supported: F % 4 == 0, F <= 256, Fo <= 256, 16-byte aligned operands (query below);
 * otherwise PYGAMD_ERR_UNSUPPORTED and the caller runs pygamd_spmm_csr + pygamd_linear_forward.
 * relu_bits_out (ceil(n_rows / 32) * ld_bits * 32 words or NULL; needs relu != 0): [y > 0] as
 * one bit per element in the tiled layout of pygamd_spmm_args.relu_bits — the form of the ReLU mask
 * that the backward's pygamd_spmm_csr / pygamd_linear_dgrad epilogues take
 * (ld_bits >= ceil(Fo / 32)).
 * Workspace: pygamd_sage_layer_fused_workspace_bytes (pygamd_spmm_csr_workspace_bytes(graph)
 * suffices in PYGAMD_GEMM_FP32 mode).                                                              */
PYGAMD_API int pygamd_sage_layer_forward_supported(int64_t F, int64_t Fo, int reduce);
PYGAMD_API int pygamd_sage_layer_forward(const pygamd_spmm_args* graph, const float* x_root,
                                         int64_t ld_root, const float* w, int64_t ldw,
                                         const float* bias, int64_t Fo, int relu, int save_agg,
                                         float* y, int64_t ldy, uint32_t* relu_bits_out,
                                         int64_t ld_bits, void* workspace,
                                         size_t workspace_bytes, void* stream);

/* The same kernel with every epilogue it has, as one argument block (zero-initialise, fill what is
 * needed).  Beyond pygamd_sage_layer_forward:
 *  - mask_bits: y = bit ? y : 0 from a one-bit-per-element ReLU mask in the tiled layout of
 *    pygamd_spmm_args.relu_bits.  With the transposed graph, x = the (degree-scaled) gradient rows
 *    and w = [W_l^T | W_r^T] this launch IS a SAGEConv layer's input gradient,
 *      grad_x = [relu'(x)] * ((A^T D^-1 g) W_l + g W_r)
 *    (sage_conv.py:134-139 differentiated; the aggregation commutes with the right-multiplication
 *    by W_l), i.e. pygamd_linear_dgrad + the transposed pygamd_spmm_csr in one pass over the graph.
 *  - y_scaled / row_scale: a second copy y * row_scale[row] of the output (the next such launch
 *    gathers the 1/deg-scaled rows and takes the unscaled ones as its root operand).
 *  - arithmetic: the transform phase follows pygamd_set_gemm_mode.  PYGAMD_GEMM_SPLIT_BF16 runs
 *    the split schedule (csrc/sage_fused.hip (B): every operand element converted to three bf16
 *    terms ONCE where it is produced, the weight by a pre-pass into the workspace) for dense rows
 *    in and out; it needs the workspace of pygamd_sage_layer_fused_workspace_bytes
 *    (PYGAMD_ERR_WORKSPACE otherwise).
 * (The schedules that were measured and not adopted, and the kernels' timing probes, are reachable
 * through include/pyg_amd_lab.h only — not part of this boundary.)                                */
typedef struct pygamd_sage_fused_args {
  const float* x_root;       /* [n_rows, F] root rows                   */
  int64_t ld_root;
  const float* w;            /* [Fo, 2F] = [W_l | W_r]                  */
  int64_t ldw;
  const float* bias;         /* [Fo] or NULL                            */
  int64_t Fo;
  int32_t relu;
  int32_t save_agg;
  float* y;                  /* [n_rows, Fo]                            */
  int64_t ldy;
  uint32_t* relu_bits_out;   /* NULL or [y > 0] as bits (needs relu)    */
  int64_t ld_bits_out;
  const uint32_t* mask_bits; /* NULL or the mask applied to y           */
  int64_t ld_mask_bits;
  const float* row_scale;    /* [n_rows], with y_scaled                 */
  float* y_scaled;           /* NULL or [n_rows, Fo]                    */
  int64_t ldy_scaled;
  uint32_t* compressed_out;  /* NULL or [n_rows, ld_compressed] words: y once more, in the format
                                of pygamd_rows_compress (the next layer's gather source).  Needs
                                Fo % 32 == 0.  With graph->x_format = PYGAMD_X_COMPRESSED the
                                gather source itself is such a block (F % 4 == 0, F <= 256).
                                Both run the fp32-instruction schedule whatever the mode.      */
  int64_t ld_compressed;
} pygamd_sage_fused_args;
/* hub partials (pygamd_spmm_csr_workspace_bytes) + the weight's bf16 term planes of the split
 * schedule (6 bytes per padded weight element; 768 KiB at F = Fo = 256)                          */
PYGAMD_API int pygamd_sage_layer_fused_workspace_bytes(const pygamd_spmm_args* graph,
                                                       const pygamd_sage_fused_args* f,
                                                       size_t* bytes /*[host]*/);
PYGAMD_API int pygamd_sage_layer_fused(const pygamd_spmm_args* graph,
                                       const pygamd_sage_fused_args* f, void* workspace,
                                       size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PYG_AMD_H */
