/* pyg_amd_lab.h — LABORATORY entry points of libpyg_amd.so.  NOT part of the drop-in boundary
 * (include/pyg_amd.h): schedules that were measured and not adopted, and the timing probes of the
 * production kernels, kept runnable for scripts/fused_probe.py and for the parity tests that pin
 * them to the production results.  Nothing a maintainer binds lives here; signatures may change
 * without an ABI bump.                                                                           */
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

#ifdef __cplusplus
}
#endif
#endif /* PYG_AMD_LAB_H */
