"""Pre-tuned GEMM solution choices for the dense feature transforms (the MFMA part of the model).

The GEMMs stay plain library calls (rocBLAS / hipBLASLt through ``torch.mm``); PyTorch's TunableOp
only picks WHICH library solution runs for a given shape.  ``gemm_mi355x_products.csv`` was
produced on an MI355X by ``PYTORCH_TUNABLEOP_TUNING=1`` over one ``bench.py`` step
(scripts/gpu_tune_and_pmc.sh) and covers the eight GEMM shapes of the ogbn-products-shaped
GraphSAGE step: 24.4 ms of GEMM per step instead of 33.6 ms with the default heuristics, plus
the shapes of the GCN/Cora, GAT/arxiv and RGCN/FB15k-237 configurations of scripts/time_configs.py
(scripts/gpu_tune_configs.sh; GAT step 7.2 -> 6.6 ms).  Other shapes fall through to the default
heuristics.  The file
carries validators (PyTorch / ROCm / hipBLASLt / rocBLAS versions, gfx950); TunableOp ignores it on
any mismatch, which simply restores the default heuristics."""
import os

CSV = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'gemm_mi355x_products.csv')


def enable_tuned_gemms(path: str = CSV) -> bool:
    """Turn TunableOp on in read-only mode with the shipped table.  Returns False (and changes
    nothing) if this PyTorch build has no TunableOp or the file cannot be read."""
    try:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(False)
        if hasattr(tunable, 'record_untuned_enable'):
            tunable.record_untuned_enable(False)
        if hasattr(tunable, 'write_file_on_exit'):
            tunable.write_file_on_exit(False)  # read-only: never rewrite the shipped table
        tunable.set_filename(path, insert_device_ordinal=False)
        return bool(tunable.read_file(path))
    except Exception:
        return False
