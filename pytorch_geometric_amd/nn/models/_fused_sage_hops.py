"""Hop-aware whole-stack GraphSAGE for sampled mini-batches (SURVEY.md §8(f) items 2 + 3).

A NeighborLoader batch lists its nodes hop by hop and its edges hop by hop, destination-sorted
inside a hop.  With ``L`` layers, layer ``l`` only has to produce the rows the next layer consumes
(nodes of hops ``<= L-1-l``) from the edges of hops ``<= L-1-l`` — exactly what the reference's
``trim_to_layer`` feeds the NEXT layer (torch_geometric/utils/_trim_to_layer.py:44-127,
nn/models/basic_gnn.py:229-243); the reference still evaluates every layer on all of its input
rows and throws the tail away.  Here every layer is ONE prefix-SpMM (the batch handle's pointer
clipped at the layer's edge count: no sort, no host sync) writing into the left half of a
``[rows, 2F]`` buffer, ONE GEMM against ``[W_l | W_r]`` over the needed rows only, ReLU in place;
the backward uses the edge-parallel atomic kernel (a batch is used once) accumulating onto the
root-gradient rows.  The output has the reference's shape (rows of its last trimmed input);
``out[:batch_size]`` is what a training step uses.
"""
from typing import List, Optional

import torch
from torch import Tensor
from torch.autograd import Function

from ... import _native
from ...edge_index import EdgeIndex
from . import _fused_sage


class FusedSageHopStack(Function):
    @staticmethod
    def forward(ctx, x: Tensor, graph: EdgeIndex, aggr: str, nodes_per_hop: List[int],
                edges_per_hop: List[int], *params: Optional[Tensor]):
        L = len(params) // 3
        if len(edges_per_hop) < L or len(nodes_per_hop) < L + 1:
            raise ValueError('need one sampled hop per layer for the hop-aware stack')
        fwd = graph.by_dst()
        dev = x.device
        N, E = x.size(0), graph.num_edges
        in_rows = [N - sum(nodes_per_hop[len(nodes_per_hop) - j] for j in range(1, l + 1))
                   for l in range(L)]
        n_edges = [E - sum(edges_per_hop[len(edges_per_hop) - j] for j in range(1, l + 1))
                   for l in range(L)]
        out_rows = [in_rows[l + 1] if l < L - 1 else in_rows[l] for l in range(L)]
        Fi = x.size(1)
        cat = torch.empty(N, 2 * Fi, dtype=torch.float32, device=dev)
        cat[:, Fi:].copy_(x)
        cats, wmats, ptrs = [], [], []
        out = None
        for l in range(L):
            W_l, b, W_r = params[3 * l:3 * l + 3]
            Fo = W_l.size(0)
            m = out_rows[l]
            ptr = fwd.ptr.narrow(0, 0, m + 1).clamp(max=n_edges[l])
            _native.spmm_csr(ptr, fwd.idx, cat[:, Fi:], aggr, n_rows=m, out=cat[:m, :Fi])
            wmat = torch.cat([W_l, W_r], dim=1)
            last = l == L - 1
            if last:
                nxt, dst = None, torch.empty(m, Fo, dtype=torch.float32, device=dev)
                out = dst
            else:
                nxt = torch.empty(m, 2 * Fo, dtype=torch.float32, device=dev)
                dst = nxt[:, Fo:]
            if _fused_sage.own_gemm(m):
                _native.linear_forward(cat[:m], wmat, b, relu=not last, out=dst)
            elif b is not None and not last and _fused_sage.RELU_EPILOGUE:
                torch._addmm_activation(b, cat[:m], wmat.t(), use_gelu=False, out=dst)
            else:
                if b is not None:
                    torch.addmm(b, cat[:m], wmat.t(), out=dst)
                else:
                    torch.mm(cat[:m], wmat.t(), out=dst)
                if not last:
                    dst.relu_()
            cats.append(cat)
            wmats.append(wmat)
            ptrs.append(ptr)
            cat, Fi = nxt, Fo
        ctx.graph, ctx.aggr, ctx.L = graph, aggr, L
        ctx.in_rows, ctx.out_rows, ctx.n_edges = in_rows, out_rows, n_edges
        ctx.has_bias = [params[3 * i + 1] is not None for i in range(L)]
        ctx.save_for_backward(*cats, *wmats, *ptrs)
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        L, graph, aggr = ctx.L, ctx.graph, ctx.aggr
        saved = ctx.saved_tensors
        cats, wmats, ptrs = saved[:L], saved[L:2 * L], saved[2 * L:]
        ei = graph.edge_index
        grads: List[Optional[Tensor]] = [None] * (3 * L)
        g = grad_out if grad_out.stride(1) == 1 else grad_out.contiguous()
        grad_x = None
        for l in reversed(range(L)):
            cat, wmat, ptr = cats[l], wmats[l], ptrs[l]
            Fi = cat.size(1) // 2
            m = ctx.out_rows[l]
            if l < L - 1:
                h_next = cats[l + 1][:, cats[l + 1].size(1) // 2:]  # post-ReLU rows [0, m)
                g, grads[3 * l + 1] = _native.relu_backward_colsum(g, h_next, ctx.has_bias[l])
            elif ctx.has_bias[l]:
                grads[3 * l + 1] = _native.colsum(g)
            own = _fused_sage.own_gemm(m)
            gw = _native.linear_wgrad(g, cat[:m]) if own else torch.mm(g.t(), cat[:m])
            grads[3 * l], grads[3 * l + 2] = gw[:, :Fi], gw[:, Fi:]
            if l > 0 or ctx.needs_input_grad[0]:
                # [m, 2 Fi] = [grad_agg | grad_root]
                gcat = (_native.linear_dgrad(g, wmat.t().contiguous()) if own
                        else torch.mm(g, wmat))
                g_in = torch.zeros(ctx.in_rows[l], Fi, dtype=torch.float32, device=g.device)
                g_in[:m].copy_(gcat[:, Fi:])
                scale = None
                if aggr == 'mean':
                    scale = 1.0 / (ptr[1:] - ptr[:-1]).clamp(min=1).to(torch.float32)
                e = ctx.n_edges[l]
                _native.gather_scatter_add(gcat[:, :Fi], ei[1].narrow(0, 0, e),
                                           ei[0].narrow(0, 0, e), ctx.in_rows[l], scale=scale,
                                           out=g_in)
                g = g_in
                if l == 0:
                    grad_x = g
        return (grad_x, None, None, None, None, *grads)


class FusedSagePaddedHopStack(Function):
    """The hop-aware stack on a STATIC-SHAPE batch (``NeighborSampler.sample_padded(...,
    padded_ids=True)``): node ids are block positions (block 0 = seeds, block h + 1 = the new nodes
    of hop h at its capacity ``batch x k_0 x ... x k_h``), every hop brings its own CSR pointer over
    its destination block, and the number of real edges of a hop lives on the device.  Every tensor
    shape and every row range below is a function of (batch size, fan-outs) only, nothing is read
    back to the host: forward + backward of a batch are capturable into ONE hipGraph together with
    the sampling and the optimizer step (bench.py --mode minibatch --capture).

    Layer ``l`` reads the rows of blocks ``0 .. L-l`` and produces blocks ``0 .. L-l-1`` from hops
    ``0 .. L-l-1`` (trim_to_layer, utils/_trim_to_layer.py:44-127): one SpMM per hop into that
    hop's destination block of the ``[agg | x]`` buffer, ONE GEMM over the produced rows.  Padding
    rows (past a block's valid count) have no in-edges and feed nothing: they carry finite garbage
    forward and exactly zero gradient backward, so the weight gradients over the padded row ranges
    equal those over the real rows."""

    @staticmethod
    def forward(ctx, x: Tensor, batch, aggr: str, *params: Optional[Tensor]):
        L = len(params) // 3
        bases, ptrs, rows = batch.bases, batch.ptrs, batch.rows
        if bases is None or len(bases) != L + 2 or len(ptrs) < L:
            raise ValueError('the padded hop stack needs a padded-id batch with one hop per layer')
        dev = x.device
        if x.size(0) != bases[-1]:
            raise ValueError(f"'x' must hold the {bases[-1]} padded rows of the batch")
        Fi = x.size(1)
        cat = torch.empty(bases[-1], 2 * Fi, dtype=torch.float32, device=dev)
        cat[:, Fi:].copy_(x)
        cats, wmats = [], []
        out = None
        for l in range(L):
            W_l, b, W_r = params[3 * l:3 * l + 3]
            Fo = W_l.size(0)
            m = bases[L - l]                      # rows produced = blocks 0 .. L-l-1
            for h in range(L - l):                # hop h aggregates into block h
                _native.spmm_csr(ptrs[h], rows[h], cat[:, Fi:], aggr,
                                 n_rows=bases[h + 1] - bases[h],
                                 out=cat[bases[h]:bases[h + 1], :Fi])
            wmat = torch.cat([W_l, W_r], dim=1)
            last = l == L - 1
            if last:
                nxt, dst = None, torch.empty(m, Fo, dtype=torch.float32, device=dev)
                out = dst
            else:
                nxt = torch.empty(m, 2 * Fo, dtype=torch.float32, device=dev)
                dst = nxt[:, Fo:]
            if _fused_sage.own_gemm(m):
                _native.linear_forward(cat[:m], wmat, b, relu=not last, out=dst)
            elif b is not None and not last and _fused_sage.RELU_EPILOGUE:
                torch._addmm_activation(b, cat[:m], wmat.t(), use_gelu=False, out=dst)
            else:
                if b is not None:
                    torch.addmm(b, cat[:m], wmat.t(), out=dst)
                else:
                    torch.mm(cat[:m], wmat.t(), out=dst)
                if not last:
                    dst.relu_()
            cats.append(cat)
            wmats.append(wmat)
            cat, Fi = nxt, Fo
        inv_deg = None
        if aggr == 'mean':  # per destination row of blocks 0 .. L-1 (padding rows: 1)
            inv_deg = torch.cat([1.0 / (p[1:] - p[:-1]).clamp(min=1).to(torch.float32)
                                 for p in ptrs[:L]])
        ctx.batch, ctx.aggr, ctx.L, ctx.inv_deg = batch, aggr, L, inv_deg
        ctx.has_bias = [params[3 * i + 1] is not None for i in range(L)]
        ctx.save_for_backward(*cats, *wmats)
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        L, batch, aggr = ctx.L, ctx.batch, ctx.aggr
        bases = batch.bases
        saved = ctx.saved_tensors
        cats, wmats = saved[:L], saved[L:]
        grads: List[Optional[Tensor]] = [None] * (3 * L)
        g = grad_out if grad_out.stride(1) == 1 else grad_out.contiguous()
        grad_x = None
        for l in reversed(range(L)):
            cat, wmat = cats[l], wmats[l]
            Fi = cat.size(1) // 2
            m = bases[L - l]
            r_in = bases[L - l + 1]
            if l < L - 1:
                h_next = cats[l + 1][:, cats[l + 1].size(1) // 2:]  # post-ReLU rows [0, m)
                g, grads[3 * l + 1] = _native.relu_backward_colsum(g, h_next, ctx.has_bias[l])
            elif ctx.has_bias[l]:
                grads[3 * l + 1] = _native.colsum(g)
            own = _fused_sage.own_gemm(m)
            gw = _native.linear_wgrad(g, cat[:m]) if own else torch.mm(g.t(), cat[:m])
            grads[3 * l], grads[3 * l + 2] = gw[:, :Fi], gw[:, Fi:]
            if l > 0 or ctx.needs_input_grad[0]:
                gcat = (_native.linear_dgrad(g, wmat.t().contiguous()) if own
                        else torch.mm(g, wmat))   # [m, 2 Fi] = [grad_agg | grad_root]
                g_in = torch.zeros(r_in, Fi, dtype=torch.float32, device=g.device)
                g_in[:m].copy_(gcat[:, Fi:])
                for h in range(L - l):  # each hop's real edges (device-side count), atomics
                    _native.gather_scatter_add(gcat[:, :Fi], batch.cols[h], batch.rows[h], r_in,
                                               scale=ctx.inv_deg, out=g_in,
                                               n_valid=batch.n_edges[h])
                g = g_in
                if l == 0:
                    grad_x = g
        return (grad_x, None, None, *grads)


def eligible(model, x, edge_index, nodes_per_hop, edges_per_hop) -> bool:
    from ..conv import SAGEConv
    from ._fused_sage import params_ready
    if not getattr(model, 'fuse_stack', True) or nodes_per_hop is None or edges_per_hop is None:
        return False
    if not (isinstance(edge_index, EdgeIndex) and edge_index.sort_order == 'col'
            and edge_index.atomic_backward):
        return False  # needs the sampler's destination-sorted batch handle
    if not (isinstance(x, Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
        return False
    if not isinstance(model.act, torch.nn.ReLU) or (model.dropout.p > 0 and model.training):
        return False
    if len(edges_per_hop) < len(model.convs) or len(nodes_per_hop) < len(model.convs) + 1:
        return False
    if edge_index.sparse_size != (x.size(0), x.size(0)):
        return False
    aggr = None
    for conv in model.convs:
        if not isinstance(conv, SAGEConv) or not conv.fuse or not conv.root_weight:
            return False
        if conv.aggr not in ('mean', 'sum', 'add') or conv.normalize or conv.project:
            return False
        if conv.flow != 'source_to_target' or (aggr is not None and conv.aggr != aggr):
            return False
        aggr = conv.aggr
        if not params_ready(conv, x):
            return False
    return True


def run(model, x: Tensor, graph: EdgeIndex, nodes_per_hop, edges_per_hop) -> Tensor:
    params = []
    for conv in model.convs:
        params += [conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight]
    aggr = model.convs[0].aggr
    return FusedSageHopStack.apply(x, graph, 'sum' if aggr == 'add' else aggr,
                                   list(nodes_per_hop), list(edges_per_hop), *params)


def run_padded(model, x: Tensor, batch) -> Tensor:
    """``model`` (a 3-layer-style GraphSAGE of plain SAGEConv layers, cf. :func:`eligible`) on a
    padded-id batch; returns the rows of block 0 .. (all of them when the model has as many layers
    as the batch has hops: the seed rows)."""
    params = []
    for conv in model.convs:
        params += [conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight]
    aggr = model.convs[0].aggr
    return FusedSagePaddedHopStack.apply(x, batch, 'sum' if aggr == 'add' else aggr, *params)
