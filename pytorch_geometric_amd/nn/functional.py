"""Loss helpers that stay on this package's kernels (the reference leaves the loss to
``torch.nn.functional``; its full-batch examples take it on a row subset — ``out[train_idx]``)."""
from .._functions import cross_entropy

__all__ = ['cross_entropy']
