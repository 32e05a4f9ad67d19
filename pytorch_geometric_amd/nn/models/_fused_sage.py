"""Whole-stack fusion of GraphSAGE (aggregate -> lin_l/lin_r GEMM -> bias
-> ReLU without the extra ``[N, F]`` HBM round trips).

Two per-layer schedules, chosen by width (both compute ``lin_l(aggr_j x_j) + lin_r(x_i) + b``):

* ``post`` (``F_in <= F_out``, the reference's order — aggregate at the input width): the
  aggregation writes straight into the LEFT half of a ``[N, 2 F_in]`` buffer whose RIGHT half
  already holds the layer input (the previous layer's GEMM wrote it there, ReLU applied in place),
  so ``lin_l(agg) + lin_r(x)`` is ONE GEMM against ``[W_l | W_r]`` with the bias folded in and no
  ``cat`` / ``add`` pass ever runs.  Backward: one GEMM for both weight gradients, one for
  ``[grad_agg | grad_root]``, and the transposed SpMM ACCUMULATES into the ``grad_root`` half
  (``accumulate = 1`` in ``pygamd_spmm_csr``), which is then the next layer's incoming gradient.

* ``pre`` (``F_out < F_in``, e.g. the 256 -> 47 output layer): mean/sum aggregation is linear, so
  ``aggr_j(x_j) W_l^T == aggr_j(x_j W_l^T)``; transform first with ``[W_l ; W_r]`` in one GEMM and
  aggregate at the (4-padded) OUTPUT width, accumulating onto the root term.  Same math, 256/48 x
  fewer gathered bytes in both directions.  (GCNConv in the reference uses exactly this
  transform-then-propagate order, nn/conv/gcn_conv.py:260-263.)

Numerics: identical operations up to fp32 summation order; tests/test_gpu_layers.py pins both
schedules against the oracle and the layer-by-layer path at 1e-5.

GEMMs run on this repo's fp32-MFMA kernels (csrc/gemm.hip: ``linear_forward`` with the bias + ReLU
epilogue writing straight into the next ``[agg | x]`` buffer, ``linear_dgrad`` whose epilogue
applies the mean's ``1/deg`` to the ``grad_agg`` half so that the transposed SpMM needs no
per-edge scale gather, ``linear_wgrad`` as a deterministic split reduction).
``PYGAMD_GEMM=lib`` switches back to rocBLAS / hipBLASLt through ``torch.mm`` for comparison."""
import os
from typing import List, Optional

import torch
from torch import Tensor
from torch.autograd import Function

from ... import _native
from ...edge_index import EdgeIndex

# ReLU as the epilogue of the bias GEMM (hipBLASLt's RELU_BIAS epilogue through
# torch._addmm_activation, which itself falls back to addmm + relu_ where the epilogue is not
# available) instead of a separate in-place pass over the layer output: -1 ms per products step.
# PYGAMD_RELU_EPILOGUE=0 restores the separate pass.
RELU_EPILOGUE = os.environ.get('PYGAMD_RELU_EPILOGUE', '1') != '0'
# 'own' = csrc/gemm.hip (default), 'lib' = torch.mm (rocBLAS / hipBLASLt)
GEMM_BACKEND = os.environ.get('PYGAMD_GEMM', 'own')
# aggregation and transform of a 'post' layer in ONE kernel (csrc/sage_fused.hip: the aggregated
# tile goes from the gather phase to the MFMA loop through LDS); PYGAMD_FUSE_LAYER=0 runs the SpMM
# and the GEMM as two launches
FUSE_LAYER = os.environ.get('PYGAMD_FUSE_LAYER', '1') != '0'
# backward of a 'post' layer's input gradient as ONE launch of the same kernel on the transposed
# graph: grad_x = [x > 0] * ((A^T D^-1 g) W_l + g W_r) — aggregation and right-multiplication
# commute, so the dgrad GEMM's flops run under the gather of the transposed SpMM instead of in
# front of it, and the [N, 2 F] `[grad_agg | grad_root]` buffer is never written.
# PYGAMD_FUSE_BWD=0 restores dgrad GEMM + transposed SpMM as two launches.
FUSE_BWD = os.environ.get('PYGAMD_FUSE_BWD', '1') != '0'
# backward: weight-gradient GEMMs (MFMA-bound, needed only by the optimizer) on a side stream with
# HALF the usual workgroups (one per CU), under the transposed SpMM of the same layer (HBM-bound,
# matrix cores idle).  0 = everything on one stream.  (Round 1 measured the same idea with a
# library GEMM that takes every wave slot: slower.  The own kernel's footprint is a parameter.)
OVERLAP_WGRAD = os.environ.get('PYGAMD_OVERLAP_WGRAD', '0') != '0'
OVERLAP_WGS = int(os.environ.get('PYGAMD_OVERLAP_WGS', '1'))  # workgroups per CU of that launch
# OPT-IN (PYGAMD_COMPRESS_ROWS=1): a one-kernel layer with a ReLU writes its output a second time
# as compressed rows (8 mask words + the non-zero values, csrc/spmm_device.h) and the next layer
# gathers those: a row gather costs what the 128-byte lines it touches cost (10.1 ms for 8 lines per
# row, 6.0 ms for 5: scripts/gather_lines_probe.py) and half of a ReLU output is exact zeros.
# Lossless, same sums bit for bit — but NOT faster inside the one-kernel layer on this chip
# (products shape: 16.8 ms against 12.9 ms for the layer that gathers, +0.8 ms for the layer that
# writes): decoding a row is ~35 VALU instructions per lane against 4 adds, and the mask word has to
# arrive before the values can be addressed (two dependent loads per source row with 4 waves per
# SIMD).  CHANGELOG.md §5a.
COMPRESS_ROWS = os.environ.get('PYGAMD_COMPRESS_ROWS', '0') != '0'
# backward of a 'pre' layer: find the all-zero rows of the incoming gradient in the pass that lays
# it out and skip them in the transposed aggregation (PYGAMD_SPARSE_GRAD=0: read every row)
SPARSE_GRAD = os.environ.get('PYGAMD_SPARSE_GRAD', '1') != '0'
_side_streams = {}
# below this many rows the library GEMM stays (cf. _functions.OWN_GEMM_MIN_ROWS: from 1 k rows up
# the own kernels fill the chip with 64 x 64 tiles and a split over the reduction)
OWN_GEMM_MIN_ROWS = int(os.environ.get('PYGAMD_OWN_GEMM_MIN_ROWS', '1024'))


def own_gemm(rows: int) -> bool:
    return GEMM_BACKEND == 'own' and rows >= OWN_GEMM_MIN_ROWS


def _side_stream(device):
    st = _side_streams.get(device)
    if st is None:
        st = _side_streams[device] = torch.cuda.Stream(device)
    return st


def _pad4(n: int) -> int:
    return (n + 3) // 4 * 4


class FusedSageStack(Function):
    @staticmethod
    def forward(ctx, x: Tensor, graph: EdgeIndex, aggr: str, reorder: bool,
                *params: Optional[Tensor]):
        # params = (W_l, b_l | None, W_r) per layer
        L = len(params) // 3
        fwd = graph.by_dst()
        N = x.size(0)
        if fwd.n_rows != N or fwd.n_cols != N:
            raise ValueError('the fused GraphSAGE stack needs a square (non-bipartite) graph')
        dev = x.device
        dims = [(params[3 * i].size(1), params[3 * i].size(0)) for i in range(L)]  # (Fi, Fo)
        modes = ['pre' if (reorder and _pad4(Fo) < Fi) else 'post' for Fi, Fo in dims]

        def new_input(layer):
            """(buffer to save, view that receives this layer's input)"""
            Fi = dims[layer][0]
            if modes[layer] == 'post':
                buf = torch.empty(N, 2 * Fi, dtype=torch.float32, device=dev)
                return buf, buf[:, Fi:]
            buf = torch.empty(N, Fi, dtype=torch.float32, device=dev)
            return buf, buf

        agg_src0 = None
        # A first 'post' layer that runs as one kernel reads its root rows from `x` itself and keeps
        # its aggregated rows in a buffer of their own: no `[agg | x]` copy of the input (0.5 ms at
        # the products shape); the weight gradient then takes the two operands side by side.
        split0 = (modes[0] == 'post' and FUSE_LAYER and GEMM_BACKEND == 'own' and x.is_contiguous()
                  and dims[0][0] % 4 == 0 and x.data_ptr() % 16 == 0
                  and _native.sage_layer_forward_supported(dims[0][0], dims[0][1], aggr))
        if split0:
            buf = torch.empty(N, dims[0][0], dtype=torch.float32, device=dev)  # agg only
            inp = x
        elif modes[0] == 'pre' and x.is_contiguous():
            buf, inp = x, x
        else:
            buf, inp = new_input(0)
            inp.copy_(x)
            if x.is_contiguous():
                agg_src0 = x  # gather from the dense original: rows are not split by the
                #               2F-stride of the [agg | x] buffer (matters for F = 100)
        bufs: List[Tensor] = []
        wmats: List[Tensor] = []
        # bits[l]: [input of layer l > 0] as one bit per element, written by the one-kernel layer
        # forward next to the activation (the backward's ReLU epilogues then read 1/32 of it)
        bits: List[Optional[Tensor]] = [None] * L
        out = None
        zsrc = None  # the current layer's input once more, as compressed rows (or None)

        def one_kernel_post(k):
            return (k < L and modes[k] == 'post' and FUSE_LAYER and GEMM_BACKEND == 'own'
                    and _native.sage_layer_forward_supported(dims[k][0], dims[k][1], aggr))

        for layer in range(L):
            W_l, b, W_r = params[3 * layer:3 * layer + 3]
            Fi, Fo = dims[layer]
            last = layer == L - 1
            if last:
                nbuf = None
                dst = (None if modes[layer] == 'pre'
                       else torch.empty(N, Fo, dtype=torch.float32, device=dev))
                out = dst
            else:
                nbuf, dst = new_input(layer + 1)
            if modes[layer] == 'post':
                src = agg_src0 if (layer == 0 and agg_src0 is not None) else inp
                wmat = torch.cat([W_l, W_r], dim=1)  # [Fo, 2 Fi]
                relu_done = False
                lone = layer == 0 and split0  # buf = the aggregated rows alone, root rows = x
                one_kernel = lone or (FUSE_LAYER and GEMM_BACKEND == 'own'
                                      and _native.sage_layer_forward_supported(Fi, Fo, aggr)
                                      and src.stride(0) % 4 == 0 and buf.stride(0) % 4 == 0)
                znext = None
                if one_kernel:
                    if not last:
                        bits[layer + 1] = _native.relu_bits_like(N, Fo, dev)
                        if (COMPRESS_ROWS and Fo % 32 == 0 and _native.SAGE_FUSED_VARIANT <= 1
                                and one_kernel_post(layer + 1)):
                            znext = torch.empty(N, _native.compressed_pitch(Fo),
                                                dtype=torch.int32, device=dev)
                    # the aggregated rows are stored once (write-only) for the weight gradient
                    _native.sage_layer_forward(fwd.ptr, fwd.idx, src if zsrc is None else zsrc,
                                               inp if lone else buf[:, Fi:],
                                               wmat, b, aggr, not last,
                                               buf if lone else buf[:, :Fi], dst, hub=fwd.hub,
                                               save_agg=True,
                                               relu_bits=None if last else bits[layer + 1],
                                               gather_width=None if zsrc is None else Fi,
                                               compressed_out=znext)
                    relu_done = True
                else:
                    _native.spmm_csr(fwd.ptr, fwd.idx, src, aggr, n_rows=N, hub=fwd.hub,
                                     out=buf[:, :Fi])
                if one_kernel:
                    pass
                elif GEMM_BACKEND == 'own':
                    _native.linear_forward(buf, wmat, b, relu=not last, out=dst)
                    relu_done = True
                elif b is not None and not last and RELU_EPILOGUE:
                    torch._addmm_activation(b, buf, wmat.t(), use_gelu=False, out=dst)
                    relu_done = True
                elif b is not None:
                    torch.addmm(b, buf, wmat.t(), out=dst)
                else:
                    torch.mm(buf, wmat.t(), out=dst)
            else:
                relu_done = False
                Fp = _pad4(Fo)
                wmat = torch.zeros(2 * Fp, Fi, dtype=torch.float32, device=dev)  # [W_l ; W_r]
                wmat[:Fo] = W_l
                wmat[Fp:Fp + Fo] = W_r
                bias = None
                if b is not None:
                    bias = torch.zeros(2 * Fp, dtype=torch.float32, device=dev)
                    bias[Fp:Fp + Fo] = b
                if GEMM_BACKEND == 'own':
                    y = _native.linear_forward(inp, wmat, bias)
                elif bias is not None:
                    y = torch.addmm(bias, inp, wmat.t())
                else:
                    y = torch.mm(inp, wmat.t())
                # y = [x W_l^T | x W_r^T + b];  right += aggr(left)
                _native.spmm_csr(fwd.ptr, fwd.idx, y[:, :Fp], aggr, n_rows=N, hub=fwd.hub,
                                 out=y[:, Fp:], accumulate=True)
                if last:  # the result IS the right half of y: hand it out as a (row-strided) view
                    out = dst = y[:, Fp:Fp + Fo]
                else:
                    dst.copy_(y[:, Fp:Fp + Fo])
            if not last and not relu_done:
                dst.relu_()
            bufs.append(buf)
            wmats.append(wmat)
            buf, inp = nbuf, dst
            zsrc = znext if modes[layer] == 'post' else None
        ctx.graph, ctx.aggr, ctx.L, ctx.dims, ctx.modes = graph, aggr, L, dims, modes
        ctx.has_bias = [params[3 * i + 1] is not None for i in range(L)]
        ctx.has_bits = [t is not None for t in bits]
        ctx.split0 = split0
        ctx.save_for_backward(*bufs, *wmats, *[t for t in bits if t is not None],
                              *([x] if split0 else []))
        return out

    @staticmethod
    def _input_view(ctx, bufs, layer) -> Tensor:
        Fi = ctx.dims[layer][0]
        return bufs[layer][:, Fi:] if ctx.modes[layer] == 'post' else bufs[layer]

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        L, graph, aggr = ctx.L, ctx.graph, ctx.aggr
        saved = ctx.saved_tensors
        bufs, wmats = saved[:L], saved[L:2 * L]
        packed = list(saved[2 * L:])
        bits = [packed.pop(0) if has else None for has in ctx.has_bits]
        x0 = packed.pop(0) if ctx.split0 else None  # layer 0: buf = agg rows, root rows = x0
        bwd = graph.by_src()
        scale = graph.by_dst().inv_degree() if aggr == 'mean' else None
        N = grad_out.size(0)
        grads: List[Optional[Tensor]] = [None] * (3 * L)
        # rows must be addressable with one leading dimension >= the width: an expanded gradient
        # (`out.sum(0)` / `out.mean(0)` upstream hand over strides (0, 1)) is materialised
        g = grad_out
        if g.dim() != 2 or g.stride(1) != 1 or (g.size(0) > 1 and g.stride(0) < g.size(1)):
            g = g.contiguous()
        grad_x = None
        pending = []  # side streams with weight-gradient launches in flight
        own = GEMM_BACKEND == 'own'
        # Own kernels: no stand-alone ReLU-backward / bias-gradient pass.  The kernel that PRODUCES
        # a layer's input gradient (transposed SpMM for 'post', dgrad GEMM for 'pre') zeroes it
        # where that input — the ReLU output of the layer below — is not positive, and the weight
        # gradient GEMM returns the column sums of its `g` operand (= the bias gradient) from the
        # pass it makes over `g` anyway.
        masked = True  # grad_out itself has no activation behind it
        needs_in = [layer > 0 or ctx.needs_input_grad[0] for layer in range(L)]

        def fused_bwd(layer: int) -> bool:
            """this layer's input gradient runs as one launch of the fused kernel"""
            Fi, Fo = ctx.dims[layer]
            return (own and FUSE_BWD and FUSE_LAYER and ctx.modes[layer] == 'post'
                    and needs_in[layer] and (layer == 0 or bits[layer] is not None)
                    and _native.sage_layer_forward_supported(Fo, Fi, 'sum'))

        g_scaled = None  # g * (1 / deg) per row, when the producer of `g` wrote it as well
        for layer in reversed(range(L)):
            buf, wmat = bufs[layer], wmats[layer]
            Fi, Fo = ctx.dims[layer]
            want_b = ctx.has_bias[layer]
            if not masked:  # ReLU backward and this layer's bias gradient in one pass
                h_next = FusedSageStack._input_view(ctx, bufs, layer + 1)  # post-ReLU output
                g, grads[3 * layer + 1] = _native.relu_backward_colsum(g, h_next, want_b)
                g_scaled = None
            elif want_b and not own:
                grads[3 * layer + 1] = _native.colsum(g)
            need_input_grad = needs_in[layer]
            mask_in = FusedSageStack._input_view(ctx, bufs, layer) if (own and layer > 0) else None
            bits_in = bits[layer] if mask_in is not None else None
            mask_f = mask_in if bits_in is None else None  # the float form only without the bits
            # the layer below takes the gradient this layer produces: 1/deg-scaled as well?
            scaled_for_next = layer > 0 and scale is not None and fused_bwd(layer - 1)
            if ctx.modes[layer] == 'post':
                # [Fo, 2 Fi] = [grad W_l | grad W_r]
                one_launch = fused_bwd(layer) and g.stride(0) % 4 == 0 and g.data_ptr() % 16 == 0
                overlap = (own and OVERLAP_WGRAD and need_input_grad and not one_launch
                           and not torch.cuda.is_current_stream_capturing())
                lone_x = x0 if layer == 0 else None  # [agg | x] as two operands side by side
                # the same idea with the one-launch input gradient: its gather phase leaves the
                # matrix cores about half idle, the weight gradient is pure MFMA work that nobody
                # waits for before the optimizer step
                overlap_one = (own and OVERLAP_WGRAD and one_launch
                               and not torch.cuda.is_current_stream_capturing())
                if overlap_one:
                    cur = torch.cuda.current_stream(g.device)
                    side = _side_stream(g.device)
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        gw = _native.linear_wgrad(g, buf, wgs_per_cu=OVERLAP_WGS, bias_grad=want_b,
                                                  x2=lone_x)
                        if want_b:
                            gw, grads[3 * layer + 1] = gw
                    for t in (g, buf) + ((lone_x, ) if lone_x is not None else ()):
                        t.record_stream(side)
                    pending.append(side)
                elif not overlap:
                    if own:
                        gw = _native.linear_wgrad(g, buf, bias_grad=want_b, x2=lone_x)
                        if want_b:
                            gw, grads[3 * layer + 1] = gw
                    else:
                        gw = torch.mm(g.t(), buf)
                if one_launch:
                    if scale is None:
                        gsrc = g
                    elif g_scaled is not None:
                        gsrc = g_scaled
                    else:  # (grad_out itself: nobody upstream could write the scaled copy)
                        gsrc = g * scale.view(-1, 1)
                    # w[i, :] = [W_l[:, i] | W_r[:, i]]
                    wc = torch.cat([wmat[:, :Fi].t(), wmat[:, Fi:].t()], dim=1)
                    gin = torch.empty(N, Fi, dtype=torch.float32, device=g.device)
                    gin_s = torch.empty_like(gin) if scaled_for_next else None
                    # hub rows of the transposed graph are aggregated into a global buffer first;
                    # the output itself serves when the widths agree (a tile reads its hub rows
                    # before it writes its result)
                    scratch = gin if Fi == Fo else torch.empty(N, Fo, dtype=torch.float32,
                                                               device=g.device)
                    _native.sage_layer_forward(bwd.ptr, bwd.idx, gsrc, g, wc, None, 'sum', False,
                                               scratch, gin, hub=bwd.hub, save_agg=False,
                                               mask_bits=bits_in,
                                               row_scale=scale if gin_s is not None else None,
                                               out_scaled=gin_s)
                    grads[3 * layer] = gw[:, :Fi]
                    grads[3 * layer + 2] = gw[:, Fi:]
                    g, g_scaled = gin, gin_s
                    masked = mask_in is not None
                    if layer == 0:
                        grad_x = g
                    continue
                if need_input_grad:
                    # [N, 2 Fi] = [grad_agg | grad_root]
                    if own:  # grad_agg rows leave the GEMM already divided by their degree
                        gcat = _native.linear_dgrad(g, wmat.t().contiguous(), scale,
                                                    Fi if scale is not None else 0)
                        pre_scaled = True
                    else:
                        gcat = torch.mm(g, wmat)
                        pre_scaled = False
                    if overlap:
                        # the weight gradient starts together with the transposed SpMM: behind the
                        # dgrad GEMM in program order (both want the matrix cores), on its own
                        # stream, one workgroup per CU
                        cur = torch.cuda.current_stream(g.device)
                        side = _side_stream(g.device)
                        side.wait_stream(cur)
                        with torch.cuda.stream(side):
                            gw = _native.linear_wgrad(g, buf, wgs_per_cu=1, bias_grad=want_b,
                                                      x2=lone_x)
                            if want_b:
                                gw, grads[3 * layer + 1] = gw
                        g.record_stream(side)
                        pending.append(side)
                    _native.spmm_csr(bwd.ptr, bwd.idx, gcat[:, :Fi], 'sum', n_rows=N,
                                     src_scale=None if pre_scaled else scale, hub=bwd.hub,
                                     out=gcat[:, Fi:], accumulate=True, relu_mask=mask_f,
                                     relu_bits=bits_in)
                grads[3 * layer] = gw[:, :Fi]
                grads[3 * layer + 2] = gw[:, Fi:]
                if need_input_grad:
                    g, g_scaled = gcat[:, Fi:], None
            else:
                Fp = _pad4(Fo)
                gy = torch.empty(N, 2 * Fp, dtype=torch.float32, device=g.device)
                # grad wrt (x W_l^T) = A^T (g / deg);  grad wrt (x W_r^T) = g.  The 1/deg factor
                # is applied ONCE per row into a dense [N, Fp] copy instead of once per gathered
                # slot inside the SpMM (a random 4-byte read per edge, 64-byte sectors: the scaled
                # F = 48 launch ran at 0.50 of HBM peak against 0.73 for its unscaled twin).
                # One pass over `g` writes that copy, the root half of `gy`, the zero padding of
                # both and one bit per row "has a non-zero entry": the gradient of a loss taken on
                # `out[train_idx]` is zero outside the training split (92 % of the rows of the
                # products workload), and the transposed aggregation below does not read rows
                # whose bit is clear (decided on the device from the count of set bits, so a dense
                # gradient costs nothing).
                if SPARSE_GRAD:
                    gsrc = (torch.empty(N, Fp, dtype=torch.float32, device=g.device)
                            if scale is not None else None)
                    row_bits, n_set = _native.rows_pack(g, scale, scaled=gsrc, copy=gy[:, Fp:])
                    if gsrc is None:
                        gsrc = gy[:, Fp:]
                else:
                    row_bits = n_set = None
                    if Fp != Fo:
                        gy[:, Fp + Fo:].zero_()
                    gy[:, Fp:Fp + Fo].copy_(g)
                    if scale is not None:
                        gsrc = torch.empty(N, Fp, dtype=torch.float32, device=g.device)
                        if Fp != Fo:
                            gsrc[:, Fo:].zero_()
                        torch.mul(g, scale.view(-1, 1), out=gsrc[:, :Fo])
                    else:
                        gsrc = gy[:, Fp:]
                _native.spmm_csr(bwd.ptr, bwd.idx, gsrc, 'sum', n_rows=N, hub=bwd.hub,
                                 out=gy[:, :Fp], src_bits=row_bits, src_bits_set=n_set)
                x_in = buf
                # [2 Fp, Fi]
                if own and OVERLAP_WGRAD and not torch.cuda.is_current_stream_capturing():
                    cur = torch.cuda.current_stream(g.device)
                    side = _side_stream(g.device)
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        gw = _native.linear_wgrad(gy, x_in, wgs_per_cu=1, bias_grad=want_b)
                    gy.record_stream(side)
                    pending.append(side)
                elif own:
                    gw = _native.linear_wgrad(gy, x_in, bias_grad=want_b)
                else:
                    gw = torch.mm(gy.t(), x_in)
                if own and want_b:  # the bias sits on the root half of y = [x W_l^T | x W_r^T + b]
                    gw, gb = gw
                    grads[3 * layer + 1] = gb[Fp:Fp + Fo]
                grads[3 * layer] = gw[:Fo]
                grads[3 * layer + 2] = gw[Fp:Fp + Fo]
                g_scaled = None
                if need_input_grad:  # [N, Fi]
                    if own:
                        if scaled_for_next:
                            g_scaled = torch.empty(N, Fi, dtype=torch.float32, device=g.device)
                        g = _native.linear_dgrad(gy, wmat.t().contiguous(),
                                                 row_scale=scale if scaled_for_next else None,
                                                 relu_mask=mask_f, relu_bits=bits_in,
                                                 out_scaled=g_scaled)
                    else:
                        g = torch.mm(gy, wmat)
            masked = mask_in is not None
            if layer == 0 and need_input_grad:
                grad_x = g.contiguous()
        for side in pending:  # the optimizer (main stream) consumes the weight gradients
            torch.cuda.current_stream(grad_out.device).wait_stream(side)
        return (grad_x, None, None, None, *grads)


def _linears(conv):
    """(neighbour map, root map): ``SAGEConv.lin_l / lin_r`` (sage_conv.py:101-107) or
    ``GraphConv.lin_rel / lin_root`` (graph_conv.py:63-64) — the same layer up to names."""
    if type(conv).__name__ == 'GraphConv':
        return conv.lin_rel, conv.lin_root
    return conv.lin_l, conv.lin_r


def params_ready(conv, x: Tensor) -> bool:
    """The layer's weights exist (a lazily initialised ``SAGEConv(-1, ...)`` materialises them in
    its first ORIGINAL forward, nn/dense/linear.py:139-150 in the reference), are fp32 and live on
    ``x``'s device."""
    for lin in _linears(conv):
        ps = [lin.weight] + ([lin.bias] if getattr(lin, 'bias', None) is not None else [])
        for p in ps:
            if isinstance(p, torch.nn.parameter.UninitializedParameter):
                return False
            if p.dtype != torch.float32 or p.device != x.device:
                return False
    return True


def eligible(model, x, edge_index, trim: bool) -> bool:
    """Conditions under which the fused stack computes exactly what the layer loop does.  Duck-typed
    on purpose: ``backend.install()`` routes the REFERENCE's ``GraphSAGE`` here too, whose layers
    are ``torch_geometric.nn.SAGEConv`` (same attribute names, nn/conv/sage_conv.py:72-116)."""
    if trim or not getattr(model, 'fuse_stack', True):
        return False
    if not (isinstance(x, Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
        return False
    if not isinstance(model.act, torch.nn.ReLU):  # act_first is moot with Identity norms
        return False
    if model.dropout.p > 0 and model.training:
        return False
    # the reference's BasicGNN extras (basic_gnn.py:99-160): norm layers, jumping knowledge and
    # the trailing Linear are not part of the fused schedule
    if getattr(model, 'jk_mode', None) is not None or hasattr(model, 'lin'):
        return False
    norms = getattr(model, 'norms', None)
    if norms is not None and any(not isinstance(n, torch.nn.Identity) for n in norms):
        return False
    aggr = None
    for conv in model.convs:
        if not _conv_ok(conv, x) or conv.normalize:
            return False
        if conv._forward_hooks or conv._forward_pre_hooks:
            return False
        if aggr is not None and conv.aggr != aggr:
            return False
        aggr = conv.aggr
    return _graph_ok(edge_index, x.size(0))


def _conv_ok(conv, x: Tensor, kinds=('SAGEConv', )) -> bool:
    """A plain mean / sum ``SAGEConv`` with a root weight (or, for a single layer, an unweighted
    ``GraphConv``) whose ``propagate`` nobody observes."""
    kind = type(conv).__name__
    if kind not in kinds or not getattr(conv, 'fuse', True):
        return False
    if conv.aggr not in ('mean', 'sum', 'add') or conv.flow != 'source_to_target':
        return False
    if kind == 'SAGEConv' and (not conv.root_weight or conv.project):
        return False
    if getattr(conv, 'explain', False) or getattr(conv, 'decomposed_layers', 1) != 1:
        return False
    if (getattr(conv, '_propagate_forward_pre_hooks', None)
            or getattr(conv, '_propagate_forward_hooks', None)
            or getattr(conv, '_message_and_aggregate_forward_pre_hooks', None)
            or getattr(conv, '_message_and_aggregate_forward_hooks', None)
            or getattr(conv, '_message_forward_pre_hooks', None)
            or getattr(conv, '_message_forward_hooks', None)
            or getattr(conv, '_aggregate_forward_pre_hooks', None)
            or getattr(conv, '_aggregate_forward_hooks', None)):
        return False
    return params_ready(conv, x)


# A single-use batch handle (the loader's, `EdgeIndex.from_sorted_batch`) never reaches the
# whole-stack schedule (hop-aware batches have their own, `_fused_sage_hops`); a single LAYER takes
# it from this many edges on: the by-source sort of the batch (once per handle, shared by the
# layers that use it) and its hub plan (one host read) against the atomic backward of every layer
# — a 3-layer SAGEConv model on whole [15, 10, 5] subgraphs of 0.6 M nodes / 0.69 M edges:
# 16.0 -> 10.3 ms per batch step (scripts/time_minibatch_layers.py); small batches keep the atomics.
SINGLE_USE_MIN_EDGES = 1 << 16


def _graph_ok(edge_index, n: int, single_use: bool = False) -> bool:
    if isinstance(edge_index, EdgeIndex):
        if edge_index.atomic_backward and not (
                single_use and edge_index.size(1) >= SINGLE_USE_MIN_EDGES):
            return False
        return edge_index.sparse_size == (n, n)
    return (isinstance(edge_index, Tensor) and type(edge_index) is Tensor and edge_index.is_cuda
            and not edge_index.is_sparse and edge_index.dim() == 2 and edge_index.size(0) == 2
            and edge_index.dtype in (torch.int32, torch.int64))


# One SAGEConv layer of ANY model (the usual PyG script builds its network from conv layers, not
# from `GraphSAGE`) as a one-layer stack: aggregation + both linear maps + bias in the one-kernel
# layer, the backward as the one-launch input gradient + one weight-gradient GEMM over
# `[agg | x]` — instead of SpMM + two GEMMs + an add forward and their five backward launches.
# PYGAMD_SAGE_LAYER_NODE=0 (or `conv.fuse = False`) keeps the propagate + Linear path.
LAYER_NODE = os.environ.get('PYGAMD_SAGE_LAYER_NODE', '1') != '0'


def layer_eligible(conv, x, edge_index, size) -> bool:
    """``conv(x, edge_index)`` is what a one-layer :class:`FusedSageStack` computes (duck-typed:
    ``backend.install()`` routes the reference's ``SAGEConv.forward`` / ``GraphConv.forward``
    here as well; a ``GraphConv`` call with edge weights is not asked)."""
    if not LAYER_NODE or not FUSE_LAYER or GEMM_BACKEND != 'own':
        return False
    if not (isinstance(x, Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
        return False
    if x.size(0) == 0 or torch.is_autocast_enabled() or torch.compiler.is_compiling():
        return False
    if torch.jit.is_scripting() or torch.cuda.is_current_stream_capturing():
        return False
    if size is not None and tuple(size) != (x.size(0), x.size(0)):
        return False
    if (not _conv_ok(conv, x, ('SAGEConv', 'GraphConv'))
            or not _graph_ok(edge_index, x.size(0), single_use=True)):
        return False
    return edge_index.size(1) > 0  # (an edgeless graph: nothing to fuse, the general path knows it)


def run_layer(conv, x: Tensor, edge_index) -> Tensor:
    from ...edge_index import as_edge_index
    graph = as_edge_index(edge_index, x.size(0), x.size(0))
    aggr = 'sum' if conv.aggr == 'add' else conv.aggr
    lin_l, lin_r = _linears(conv)
    return FusedSageStack.apply(x, graph, aggr, True, lin_l.weight, lin_l.bias, lin_r.weight)


def run(model, x: Tensor, edge_index) -> Tensor:
    from ...edge_index import as_edge_index
    graph = as_edge_index(edge_index, x.size(0), x.size(0))
    params = []
    for conv in model.convs:
        params += [conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight]
    aggr = model.convs[0].aggr
    reorder = bool(getattr(model, 'reorder_narrow_layers', True))
    return FusedSageStack.apply(x, graph, 'sum' if aggr == 'add' else aggr, reorder, *params)
