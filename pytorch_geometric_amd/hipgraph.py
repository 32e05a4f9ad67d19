"""Capture a launch-bound training step into one HIP graph.

On small graphs (Cora: 2,708 nodes) a forward+backward is ~40 kernels of a few microseconds each
and the step is bound by launch latency, not by HBM or MFMA.  Everything this package launches goes
to the current stream with outputs from torch's allocator and, once the sorted graph handles are
cached, without any host synchronisation — so a whole step can be recorded once and replayed
(``hipGraphLaunch``) instead of re-issuing every kernel: 0.48 -> 0.12 ms per step on the Cora-shaped
GCN (CHANGELOG.md §6b)."""
from typing import Any, Callable

import torch


class CapturedStep:
    """``replay = CapturedStep(fn)`` runs ``fn`` eagerly ``warmup`` times on a side stream (this is
    where handles get sorted and cached; 0 = the caller has warmed ``fn`` up itself), captures one
    more call and then replays it.  ``fn`` must
    read its inputs from fixed buffers (update them in place between replays), keep ``.grad``
    tensors allocated (``zero_grad(set_to_none=False)``) and must not synchronise with the host."""

    def __init__(self, fn: Callable[[], Any], warmup: int = 3, capture_error_mode: str = None):
        """``capture_error_mode``: hipStreamCaptureMode of the recording (``torch.cuda.graph``'s
        argument).  Default: ``'global'`` — except when a ``torch.distributed`` process group
        exists: its watchdog thread polls events of earlier collectives while this thread records,
        which a GLOBAL capture reports as an illegal call from another thread (seen as an
        intermittent ``ProcessGroupNCCL`` watchdog abort when the step's all-reduce is recorded
        too); such steps record ``'thread_local'``."""
        if capture_error_mode is None:
            import torch.distributed as dist
            pg = dist.is_available() and dist.is_initialized()
            capture_error_mode = 'thread_local' if pg else 'global'
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 0)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        if capture_error_mode != 'global':
            torch.cuda.synchronize()   # the warm-up collectives are done before recording starts
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode):
            self.output = fn()

    def __call__(self):
        self.graph.replay()
        return self.output
