/* pyg_amd_lab.h — LABORATORY entry points, exported by libpyg_amd_lab.so ONLY (a second build of
 * the library's sources plus csrc/sage_fused_lab.hip and the weight-gradient probes of gemm.hip
 * under -DPYGAMD_LAB; pytorch_geometric_amd/_build.py).  NOT part of the drop-in boundary
 * (include/pyg_amd.h) and not in the product library libpyg_amd.so: schedules that were measured
 * and not adopted, and the timing probes of the production kernels, kept runnable for scripts/,
 * for the parity tests that pin them to the production results and for the copy-rate side figure
 * of bench.py.  Nothing a maintainer binds lives here; signatures may change without an ABI
 * bump.                                                                                          */
#ifndef PYG_AMD_LAB_H
#define PYG_AMD_LAB_H
#include "pyg_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* pygamd_sage_layer_fused (same argument blocks, same results up to rounding) on a chosen schedule:
 *   variant 1  the production fp32 schedule with `probe` honoured;
 *   variant 2  streamed gather phase (column indices of the tile staged in LDS, row loads
 *              software-pipelined across rows; needs n_src < 2^31) — bitwise variant 1;
 *   variant 3 / 4  one persistent workgroup per CU whose gather waves feed 4 / 8 transform waves
 *              through LDS tiles (sums run root half first: equal to rounding, not bitwise);
 *   variant 5  the production split-arithmetic kernel regardless of pygamd_set_gemm_mode
 *              (workspace: pygamd_sage_layer_fused_workspace_bytes), probe bits 0 / 1 honoured;
 *   variant 6  the production fp32 kernel regardless of pygamd_set_gemm_mode.
 * probe: timing only (results undefined when non-zero): bit 0 = skip the gather loop, bit 1 =
 * skip the matrix loop, further bits per schedule (csrc/sage_fused_device.h).                    */
PYGAMD_API int pygamd_lab_sage_layer_fused(const pygamd_spmm_args* graph,
                                           const pygamd_sage_fused_args* f, int variant,
                                           int probe, void* workspace, size_t workspace_bytes,
                                           void* stream);

/* Schedule of pygamd_linear_wgrad / _wgrad2 in the split arithmetic (process-wide, like
 * pygamd_set_gemm_mode):
 *   variant 0  production: the staging threads split every operand element once on its way into
 *              LDS (k-contiguous bf16 planes, one ds_read_b128 per fragment and term);
 *   variant 1  round 3's schedule: fp32 rows in LDS, every wave splits its own fragments in
 *              registers next to the matrix instructions.  Bitwise variant 0 for the weight
 *              gradient (same terms, same order); the bias gradient differs in summation order;
 *   2 / 4 / 8 (or-ed) timing probes of variant 0 on 16-byte-aligned operands, results
 *              undefined: no matrix products / no conversion and LDS stores / no global loads;
 *   32 (+ 8)   variant 0 with the residual subtractions as v_pk_add_f32 (what the compiler
 *              emits unforced; same results);
 *   16         variant 0 without gemm_tn_skinny_kernel (narrow g, one wide x, >= 32 k rows: the
 *              tiled kernel for that shape too — the A/B of scripts/wgrad_probe.py);
 *   64 + bits  timing probes of gemm_tn_skinny_kernel<3> (g 65..96 columns wide), bits as for
 *              2 .. 14 above: 66 = no products, 68 = no conversion, 72 = no loads, ...          */
PYGAMD_API int pygamd_lab_set_wgrad_variant(int variant);

/* Streaming device-to-device copy of n_bytes (a multiple of 16, both pointers 16-byte aligned):
 * the measured HBM copy rate bench.py reports next to the 8 TB/s specification (SURVEY 8(d)'s
 * secondary denominator).  variant bit 0 = non-temporal loads and stores, bit 1 = eight instead
 * of four 16-byte loads in flight per lane, bit 2 = the read half alone (the loads are summed per
 * lane and nothing is written for finite data: n_bytes of traffic instead of 2 n_bytes; dst needs
 * 4 KiB); blocks_per_cu workgroups of 256 lanes per compute unit walk the buffers grid-stride.
 * A measurement probe, never on the product path.                                              */
PYGAMD_API int pygamd_lab_copy(const void* src, void* dst, int64_t n_bytes, int variant,
                               int blocks_per_cu, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PYG_AMD_LAB_H */
