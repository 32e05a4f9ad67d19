"""How a model asks ONE layer call to apply the model's activation itself (bias + ReLU in the
layer's own epilogue pass instead of an ATen add in the layer and an ATen clamp in the model).

The request is call-time state, not module state: it lives in a thread-local slot, names the conv
OBJECT it is meant for, and is consumed by the first read from that object — so a conv shared
between models or called from two threads, a conv re-entered from inside its own forward, and a
conv called on its own all keep the reference's semantics (``out + bias``, no activation)."""
import contextlib
import threading

import torch


class _Slot(threading.local):
    conv = None
    act = None


_slot = _Slot()


@contextlib.contextmanager
def request_activation(conv, act):
    """While the block runs, the next ``requested_activation(conv)`` on this thread returns
    ``act`` (once)."""
    prev = (_slot.conv, _slot.act)
    _slot.conv, _slot.act = (conv, act) if act is not None else (None, None)
    try:
        yield
    finally:
        _slot.conv, _slot.act = prev


def requested_activation(conv):
    """``'relu'`` if the caller of this layer call asked the layer to apply it, else ``None``.
    Reading consumes the request."""
    if _slot.conv is conv:
        act = _slot.act
        _slot.conv = _slot.act = None
        return act
    return None


def has_forward_hooks(module) -> bool:
    """A forward hook observes (or replaces) the layer's output: it must see what the reference's
    layer returns — pre-activation — so layers with hooks are never asked to fuse."""
    from torch.nn.modules import module as _m
    return bool(module._forward_hooks or module._forward_pre_hooks
                or getattr(module, '_forward_hooks_with_kwargs', None)
                or _m._global_forward_hooks or _m._global_forward_pre_hooks
                or getattr(_m, '_global_forward_hooks_always_called', None))
