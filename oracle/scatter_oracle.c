/*
 * scatter_oracle.c — plain C restatement of the reference's CPU scatter path.
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it (as the checker).  The product never links or calls it.
 *
 * Each function restates, as scalar loops, what the reference computes with ATen CPU ops; the
 * citations are paths under the reference root (PyG 2.9.0).  Sequential accumulation in edge
 * order, exactly like a single-threaded scatter_add_.
 *
 * Parity status: pinned.  tests/test_oracle_c.py checks every function against
 * oracle/pyg_oracle.py and against tests/golden/golden_v1.pt / golden_preproc_v1.pt (outputs of
 * the real reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { R_SUM = 0, R_MEAN = 1, R_MIN = 2, R_MAX = 3, R_MUL = 4 };

/* utils/_index_sort.py:10-32 -> inputs.sort(stable=True): stable counting sort of keys in
 * [0, max_value].  perm[i] = original position of the i-th smallest key. */
int oracle_index_sort(const int64_t* keys, int64_t n, int64_t max_value, int64_t* sorted,
                      int64_t* perm) {
  int64_t* count = (int64_t*)calloc((size_t)max_value + 2, sizeof(int64_t));
  if (!count) return 1;
  for (int64_t i = 0; i < n; ++i) {
    if (keys[i] < 0 || keys[i] > max_value) { free(count); return 2; }
    count[keys[i] + 1]++;
  }
  for (int64_t v = 0; v <= max_value; ++v) count[v + 1] += count[v];
  for (int64_t i = 0; i < n; ++i) {
    const int64_t pos = count[keys[i]]++;
    sorted[pos] = keys[i];
    perm[pos] = i;
  }
  free(count);
  return 0;
}

/* index.py:32-37 index2ptr == torch._convert_indices_from_coo_to_csr (index sorted). */
void oracle_index2ptr(const int64_t* index, int64_t n, int64_t size, int64_t* ptr) {
  for (int64_t v = 0; v <= size; ++v) ptr[v] = 0;
  for (int64_t i = 0; i < n; ++i) ptr[index[i] + 1]++;
  for (int64_t v = 0; v < size; ++v) ptr[v + 1] += ptr[v];
}

/* utils/_sort_edge_index.py:105-113 / utils/_coalesce.py:131-176, integer part: stable sort of
 * the compound key major * n + minor (major = row when by_row), then — for coalesce — keep the
 * first edge of every run of equal keys.  out_row / out_col hold E entries (the first *n_out are
 * valid), perm[i] = original position of the i-th sorted edge, group[e] = output slot of the
 * ORIGINAL edge e (what the attribute scatter uses).  dedup == 0: plain sort_edge_index. */
int oracle_sort_edges(const int64_t* row, const int64_t* col, int64_t E, int64_t n, int by_row,
                      int dedup, int64_t* out_row, int64_t* out_col, int64_t* perm,
                      int64_t* group, int64_t* n_out) {
  /* two stable counting-sort passes (minor digit first) == one stable sort of the compound key */
  int64_t* tmp = (int64_t*)malloc((size_t)(E > 0 ? E : 1) * sizeof(int64_t));
  int64_t* count = (int64_t*)malloc((size_t)(n + 2) * sizeof(int64_t));
  if (!tmp || !count) { free(tmp); free(count); return 1; }
  const int64_t* major = by_row ? row : col;
  const int64_t* minor = by_row ? col : row;
  for (int pass = 0; pass < 2; ++pass) {
    const int64_t* digit = pass == 0 ? minor : major;
    memset(count, 0, (size_t)(n + 2) * sizeof(int64_t));
    for (int64_t e = 0; e < E; ++e) {
      if (digit[e] < 0 || digit[e] >= n) { free(tmp); free(count); return 2; }
      count[digit[e] + 1]++;
    }
    for (int64_t v = 0; v < n; ++v) count[v + 1] += count[v];
    if (pass == 0) {
      for (int64_t e = 0; e < E; ++e) tmp[count[digit[e]]++] = e;
    } else {
      for (int64_t i = 0; i < E; ++i) perm[count[digit[tmp[i]]]++] = tmp[i];
    }
  }
  int64_t k = 0;
  for (int64_t i = 0; i < E; ++i) {
    const int64_t e = perm[i];
    const int first = !dedup || i == 0 || row[e] != row[perm[i - 1]] || col[e] != col[perm[i - 1]];
    if (first) {
      out_row[k] = row[e];
      out_col[k] = col[e];
      ++k;
    }
    group[e] = k - 1;
  }
  *n_out = k;
  free(tmp);
  free(count);
  return 0;
}

/* index.py:27-29 ptr2index == arange(size).repeat_interleave(ptr.diff()). */
void oracle_ptr2index(const int64_t* ptr, int64_t size, int64_t* index) {
  for (int64_t r = 0; r < size; ++r)
    for (int64_t k = ptr[r]; k < ptr[r + 1]; ++k) index[k] = r;
}

/* nn/conv/message_passing.py:263-290: x_j = x.index_select(0, index).  Returns 1 on a bad index. */
int oracle_gather(const float* x, int64_t n_src, int64_t F, const int64_t* index, int64_t n,
                  float* out) {
  for (int64_t e = 0; e < n; ++e) {
    if (index[e] < 0 || index[e] >= n_src) return 1;
    memcpy(out + e * F, x + index[e] * F, sizeof(float) * (size_t)F);
  }
  return 0;
}

/* utils/_scatter.py:14-138 (dim = 0): zeros.scatter_add_ / count.clamp(min=1) division /
 * zeros.scatter_reduce_(amin|amax, include_self=False) / ones.scatter_reduce_(prod).
 * Empty groups: 0 (1 for mul). */
int oracle_scatter(const float* src, const int64_t* index, int64_t n, int64_t F,
                   int64_t dim_size, int reduce, float* out) {
  int64_t* count = (int64_t*)calloc((size_t)dim_size + 1, sizeof(int64_t));
  if (!count) return 1;
  const float init = (reduce == R_MUL) ? 1.f : 0.f;
  for (int64_t i = 0; i < dim_size * F; ++i) out[i] = init;
  for (int64_t e = 0; e < n; ++e) {
    const int64_t g = index[e];
    if (g < 0 || g >= dim_size) { free(count); return 2; }
    float* o = out + g * F;
    const float* s = src + e * F;
    const int first = (count[g]++ == 0);
    for (int64_t f = 0; f < F; ++f) {
      switch (reduce) {
        case R_SUM:
        case R_MEAN: o[f] += s[f]; break;
        case R_MUL: o[f] *= s[f]; break;
        case R_MIN: o[f] = (first || s[f] < o[f] || s[f] != s[f]) ? s[f] : o[f]; break;
        case R_MAX: o[f] = (first || s[f] > o[f] || s[f] != s[f]) ? s[f] : o[f]; break;
        default: free(count); return 3;
      }
    }
  }
  if (reduce == R_MEAN) {
    for (int64_t g = 0; g < dim_size; ++g) {
      const float c = (float)(count[g] < 1 ? 1 : count[g]);
      for (int64_t f = 0; f < F; ++f) out[g * F + f] = out[g * F + f] / c;
    }
  }
  free(count);
  return 0;
}

/* utils/_segment.py:37-50: torch._segment_reduce over ptr ranges; empty segments -> 0. */
void oracle_segment(const float* src, const int64_t* ptr, int64_t n_seg, int64_t F, int reduce,
                    float* out) {
  for (int64_t s = 0; s < n_seg; ++s) {
    float* o = out + s * F;
    for (int64_t f = 0; f < F; ++f) o[f] = 0.f;
    for (int64_t k = ptr[s]; k < ptr[s + 1]; ++k) {
      const float* v = src + k * F;
      const int first = (k == ptr[s]);
      for (int64_t f = 0; f < F; ++f) {
        if (reduce == R_SUM || reduce == R_MEAN) o[f] += v[f];
        else if (reduce == R_MIN) o[f] = (first || v[f] < o[f]) ? v[f] : o[f];
        else o[f] = (first || v[f] > o[f]) ? v[f] : o[f];
      }
    }
    if (reduce == R_MEAN && ptr[s + 1] > ptr[s]) {
      const float c = (float)(ptr[s + 1] - ptr[s]);
      for (int64_t f = 0; f < F; ++f) o[f] = o[f] / c;
    }
  }
}

/* utils/_softmax.py:82-88 (index branch): max per group, exp, sum + 1e-16, divide. */
int oracle_softmax(const float* src, const int64_t* index, int64_t n, int64_t H, int64_t N,
                   float* out) {
  float* gmax = (float*)malloc(sizeof(float) * (size_t)(N * H + 1));
  float* gsum = (float*)calloc((size_t)(N * H + 1), sizeof(float));
  char* seen = (char*)calloc((size_t)N + 1, 1);
  if (!gmax || !gsum || !seen) return 1;
  for (int64_t i = 0; i < N * H; ++i) gmax[i] = 0.f; /* scatter(max) gives 0 for empty groups */
  for (int64_t e = 0; e < n; ++e) {
    const int64_t g = index[e];
    for (int64_t h = 0; h < H; ++h) {
      const float v = src[e * H + h];
      if (!seen[g] || v > gmax[g * H + h]) gmax[g * H + h] = v;
    }
    seen[g] = 1;
  }
  for (int64_t e = 0; e < n; ++e)
    for (int64_t h = 0; h < H; ++h) {
      out[e * H + h] = expf(src[e * H + h] - gmax[index[e] * H + h]);
      gsum[index[e] * H + h] += out[e * H + h];
    }
  for (int64_t e = 0; e < n; ++e)
    for (int64_t h = 0; h < H; ++h)
      out[e * H + h] = out[e * H + h] / (gsum[index[e] * H + h] + 1e-16f);
  free(gmax); free(gsum); free(seen);
  return 0;
}

/* The whole unfused propagate of nn/conv/message_passing.py:421-563 for one layer:
 * gather on edge_index[0] -> optional w_e * x_j message -> scatter(reduce) onto edge_index[1]. */
int oracle_propagate(const float* x, int64_t n_src, int64_t F, const int64_t* src_idx,
                     const int64_t* dst_idx, const float* w, int64_t n_edges, int64_t n_dst,
                     int reduce, float* out) {
  float* msg = (float*)malloc(sizeof(float) * (size_t)(n_edges * F + 1));
  if (!msg) return 1;
  int rc = oracle_gather(x, n_src, F, src_idx, n_edges, msg);
  if (rc == 0 && w)
    for (int64_t e = 0; e < n_edges; ++e)
      for (int64_t f = 0; f < F; ++f) msg[e * F + f] = w[e] * msg[e * F + f];
  if (rc == 0) rc = oracle_scatter(msg, dst_idx, n_edges, F, n_dst, reduce, out);
  free(msg);
  return rc;
}

/* nn/conv/sage_conv.py:120-144: out = lin_l(aggr_j x_j) + lin_r(x_i); W row-major [Fo, Fi]. */
int oracle_sage_conv(const float* x, int64_t N, int64_t Fi, const int64_t* src_idx,
                     const int64_t* dst_idx, int64_t n_edges, const float* w_l, const float* b_l,
                     const float* w_r, int64_t Fo, int reduce, float* out) {
  float* agg = (float*)malloc(sizeof(float) * (size_t)(N * Fi + 1));
  if (!agg) return 1;
  int rc = oracle_propagate(x, N, Fi, src_idx, dst_idx, NULL, n_edges, N, reduce, agg);
  if (rc == 0) {
    for (int64_t i = 0; i < N; ++i)
      for (int64_t o = 0; o < Fo; ++o) {
        float acc = 0.f;
        for (int64_t k = 0; k < Fi; ++k) acc += agg[i * Fi + k] * w_l[o * Fi + k];
        if (b_l) acc += b_l[o];
        if (w_r) {
          float r = 0.f;
          for (int64_t k = 0; k < Fi; ++k) r += x[i * Fi + k] * w_r[o * Fi + k];
          acc += r;
        }
        out[i * Fo + o] = acc;
      }
  }
  free(agg);
  return rc;
}
