"""ctypes loader for oracle/_build/liboracle.so (the plain C restatement).  Checker only."""
import ctypes
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'liboracle.so')
REDUCE = {'sum': 0, 'add': 0, 'mean': 1, 'min': 2, 'max': 3, 'mul': 4}
_lib = None


def load():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, 'scatter_oracle.c')
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
            subprocess.run(['make', '-C', HERE, '-s'], check=True)
        _lib = ctypes.CDLL(LIB)
    return _lib


def _f(t):
    a = np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _i(t):
    a = np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.int64)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _out(shape, dtype=np.float32):
    a = np.empty(shape, dtype=dtype)
    return a, a.ctypes.data_as(ctypes.c_void_p)


I64 = ctypes.c_int64


def index_sort(keys, max_value):
    k, kp = _i(keys)
    s, sp = _out(k.shape, np.int64)
    p, pp = _out(k.shape, np.int64)
    assert load().oracle_index_sort(kp, I64(k.size), I64(max_value), sp, pp) == 0
    return torch.from_numpy(s), torch.from_numpy(p)


def sort_edges(edge_index, num_nodes, by_row=True, dedup=False):
    """(edge_index_out, perm, group): sort_edge_index (dedup=False) / coalesce (dedup=True)."""
    r, rp = _i(edge_index[0])
    c, cp = _i(edge_index[1])
    E = r.size
    orow, orp = _out((E, ), np.int64)
    ocol, ocp = _out((E, ), np.int64)
    perm, pp = _out((E, ), np.int64)
    group, gp = _out((E, ), np.int64)
    n_out = I64(0)
    rc = load().oracle_sort_edges(rp, cp, I64(E), I64(num_nodes), int(by_row), int(dedup), orp,
                                  ocp, pp, gp, ctypes.byref(n_out))
    assert rc == 0, rc
    k = n_out.value
    out = torch.stack([torch.from_numpy(orow[:k].copy()), torch.from_numpy(ocol[:k].copy())])
    return out, torch.from_numpy(perm), torch.from_numpy(group)


def index2ptr(index, size):
    k, kp = _i(index)
    o, op = _out((size + 1, ), np.int64)
    load().oracle_index2ptr(kp, I64(k.size), I64(size), op)
    return torch.from_numpy(o)


def ptr2index(ptr):
    k, kp = _i(ptr)
    o, op = _out((int(k[-1]), ), np.int64)
    load().oracle_ptr2index(kp, I64(k.size - 1), op)
    return torch.from_numpy(o)


def scatter(src, index, dim_size, reduce):
    s, sp = _f(src.reshape(src.size(0), -1))
    k, kp = _i(index)
    o, op = _out((dim_size, s.shape[1]))
    rc = load().oracle_scatter(sp, kp, I64(k.size), I64(s.shape[1]), I64(dim_size),
                               REDUCE[reduce], op)
    assert rc == 0, rc
    return torch.from_numpy(o).view(dim_size, *src.shape[1:])


def segment(src, ptr, reduce):
    s, sp = _f(src)
    k, kp = _i(ptr)
    o, op = _out((k.size - 1, s.shape[1]))
    load().oracle_segment(sp, kp, I64(k.size - 1), I64(s.shape[1]), REDUCE[reduce], op)
    return torch.from_numpy(o)


def softmax(src, index, num_nodes):
    s, sp = _f(src)
    k, kp = _i(index)
    o, op = _out(s.shape)
    assert load().oracle_softmax(sp, kp, I64(k.size), I64(s.shape[1]), I64(num_nodes), op) == 0
    return torch.from_numpy(o)


def propagate(x, edge_index, num_dst, reduce, edge_weight=None):
    xa, xp = _f(x)
    s, sp = _i(edge_index[0])
    d, dp = _i(edge_index[1])
    wp = None
    if edge_weight is not None:
        w, wp = _f(edge_weight)
    o, op = _out((num_dst, xa.shape[1]))
    rc = load().oracle_propagate(xp, I64(xa.shape[0]), I64(xa.shape[1]), sp, dp, wp,
                                 I64(s.size), I64(num_dst), REDUCE[reduce], op)
    assert rc == 0, rc
    return torch.from_numpy(o)


def sage_conv(x, edge_index, w_l, b_l, w_r, aggr='mean'):
    xa, xp = _f(x)
    s, sp = _i(edge_index[0])
    d, dp = _i(edge_index[1])
    wl, wlp = _f(w_l)
    blp = wrp = None
    if b_l is not None:
        bl, blp = _f(b_l)
    if w_r is not None:
        wr, wrp = _f(w_r)
    o, op = _out((xa.shape[0], wl.shape[0]))
    rc = load().oracle_sage_conv(xp, I64(xa.shape[0]), I64(xa.shape[1]), sp, dp, I64(s.size),
                                 wlp, blp, wrp, I64(wl.shape[0]), REDUCE[aggr], op)
    assert rc == 0, rc
    return torch.from_numpy(o)
