"""Static-shape ("slot") neighbour-sampled batches and the GraphSAGE stack that trains on them
(BASELINE config 4) — the host side of ``csrc/minibatch.hip``.

What the reference does per batch (``examples/multi_gpu/distributed_sampling.py:64-115``):
``NeighborLoader`` samples ``[15, 10, 5]`` neighbours around 1024 seeds, gathers ``x[n_id]``, and the
model runs layer by layer on the sampled subgraph (``trim_to_layer``,
``utils/_trim_to_layer.py:167-215``: layer ``l`` only needs the rows the next layer consumes).  Here
the same computation is laid out so that every tensor shape and every row range is a function of
(batch size, fan-outs) only and nothing is read back to the host — the whole step captures into one
hipGraph — with as few, as fat kernels as the path allows (round 3: 181 kernels / 3.0 ms per batch):

* sampling: 1 launch for the seeds + 2 per hop (``SlotSampler.sample``); the duplicate-resolving
  node map is never reset (epoch-stamped claims);
* feature gather: 1 launch, straight into the right half of the first ``[agg | x]`` buffer —
  round 6: of the DESTINATION rows only; layer 0 gathers its neighbours' rows (the last hop: 80 % of
  the batch) from the graph's feature matrix by graph node id (``gather(direct=True)``: 1.18 ->
  1.09 ms per captured batch at the papers100M shape);
* forward: ONE launch per layer — the one-kernel SAGE layer (gather -> LDS -> MFMA, bias + ReLU +
  ReLU bits in the epilogue) over all destination blocks of the layer at once (``rowptr`` =
  the static ``row_begin``, ``rowend`` = the sampler's ``row_end``);
* backward: per layer the weight gradient (+ bias gradient from the same pass), two half-width
  dgrad GEMMs (``g W_l`` scaled by 1/deg in the epilogue -> the rows to scatter; ``g W_r`` straight
  into the destination rows of the input gradient) and ONE transposed SpMM over the batch's
  transposed CSR (built by 3 launches per batch) that accumulates onto the root gradient and applies
  the ReLU mask bits — no atomics, no zero-fill, no stand-alone ReLU-backward / column-sum pass.
Padding rows (holes) hold finite values forward (zero features, or an aggregation of nothing) and
receive exactly zero gradient backward: no entry of a transposed CSR points at them."""
import ctypes
import weakref
from dataclasses import dataclass
from typing import List, Optional

import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib, _native
from ._lib import check


def _i64p(t: Optional[Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class SlotPlan:
    """Everything static about a batch of ``batch_size`` seeds and bounded fan-outs ``fanouts``."""

    def __init__(self, batch_size: int, fanouts: List[int], device):
        lib = _lib.load()
        if not fanouts or len(fanouts) > lib.pygamd_slots_max_hops():
            raise ValueError(f'1 .. {lib.pygamd_slots_max_hops()} hops')
        if any(k < 1 or k > lib.pygamd_slots_max_fanout() for k in fanouts):
            raise ValueError(f'fan-outs must be in 1 .. {lib.pygamd_slots_max_fanout()} '
                             f'(slot batches need bounded fan-outs)')
        self.B, self.fanouts, self.L = int(batch_size), [int(k) for k in fanouts], len(fanouts)
        self.device = torch.device(device)
        cap = [self.B]
        for k in self.fanouts:
            cap.append(cap[-1] * k)
        self.cap = cap
        self.bases = [0]
        for c in cap:
            self.bases.append(self.bases[-1] + c)          # len L + 2
        self.R = self.bases[-1]                              # all rows
        self.R_dst = self.bases[self.L]                      # rows that are ever a destination
        self.S = self.R - self.B                             # all slots
        if self.R >= 2 ** 31:
            raise ValueError('batch too large for 32-bit batch-local ids')
        begin = [(self.bases[b + 1] - self.B) + torch.arange(cap[b], dtype=torch.int64) * k
                 for b, k in enumerate(self.fanouts)]
        self.row_begin = torch.cat(begin).to(torch.int32).to(self.device)
        # (the same pointer as int64: layer 0 in DIRECT mode gathers from the graph's feature
        # matrix by graph node id, SlotBatch.x_global)
        self.row_begin64 = self.row_begin.to(torch.int64)
        # transposed CSR c serves the backward of layer c + 1: hops 0 .. L - c - 2
        self.n_csr = self.L - 1
        self.t_slots = [self.bases[self.L - c] - self.B for c in range(self.n_csr)]
        self.t_rows = [self.bases[self.L - c] for c in range(self.n_csr)]

    def ebase(self, h: int) -> int:
        return self.bases[h + 1] - self.B


@dataclass
class SlotBatch:
    plan: SlotPlan
    node_g: Tensor      # [R] int64: graph node of a row, -1 = hole
    src_g: Tensor       # [S] int64: graph node sampled into a slot, -1 = empty
    src_id: Tensor      # [S] int32: row holding the slot's source
    row_end: Tensor     # [R_dst] int32
    inv_cnt: Tensor     # [R_dst] float32
    t_ptr: List[Tensor]  # per transposed CSR: [rows + 1] int32
    t_col: List[Tensor]  # per transposed CSR: [slots] int32 (filled prefix = ptr[-1])
    x: Optional[Tensor] = None   # [R, 2 F] = [ (aggregation target) | x[node_g] ]
    y: Optional[Tensor] = None
    # DIRECT mode (SlotSampler.gather(direct=True)): `x` holds the DESTINATION rows only
    # ([R_dst, 2 F]) and layer 0 gathers its neighbours straight from the graph's feature matrix
    # `x_global` through `src_g` (graph node ids) — the rows of the last hop, 80 % of the batch,
    # are then read once by the layer instead of being copied into the batch first and read back.
    x_global: Optional[Tensor] = None
    row_end64: Optional[Tensor] = None
    # The index tensors above are the SAMPLER's buffers, overwritten by its next `sample()`:
    # `stamp` is the sampler's call counter when this batch was drawn, `owner` the sampler.  The
    # stack's backward reads `t_ptr / t_col / inv_cnt` again and refuses a batch whose sampler has
    # moved on (prefetching the next batch before `backward()` needs a second SlotSampler).
    stamp: int = 0
    owner: Optional[object] = None

    def check_current(self, where: str) -> None:
        owner = self.owner() if self.owner is not None else None
        if owner is not None and owner.calls != self.stamp:
            raise RuntimeError(
                f'{where}: this SlotBatch was drawn by sample() call {self.stamp} of its '
                f'SlotSampler, which has sampled again since (call {owner.calls}) into the same '
                f'buffers; run backward() before sampling the next batch, or prefetch with a '
                f'second SlotSampler')


class SlotSampler:
    """Device-side k-hop sampler over the destination-sorted (CSC) form of a graph that writes
    :class:`SlotBatch` es.  ``colptr`` / ``row`` as in :class:`pytorch_geometric_amd.sampler.
    NeighborSampler` (``graph.by_dst()``)."""

    def __init__(self, colptr: Tensor, row: Tensor, num_nodes: int, plan: SlotPlan,
                 seed: int = 0, local: Optional[Tensor] = None):
        """``local``: the claim map of ANOTHER sampler over the same graph to share (int64
        ``[num_nodes]``): claims are epoch-stamped, so samplers that are fed one growing epoch
        sequence between them — the two halves of a double-buffered loop — need only one map."""
        _native._require_device(colptr, row)
        if colptr.dtype != row.dtype or colptr.dtype not in (torch.int32, torch.int64):
            raise ValueError("'colptr' / 'row' must share an int32 / int64 dtype")
        self.colptr, self.row, self.plan, self.seed = colptr, row, plan, int(seed)
        self.calls = 0   # sample() calls so far (host side): the stamp of the batch it returns
        dev = colptr.device
        p = plan
        # the claim map: zero = "never claimed"; epochs >= 1 always beat it — never reset
        if local is not None and (local.dtype != torch.int64 or local.numel() != num_nodes
                                  or local.device != dev):
            raise ValueError("'local' must be an int64 [num_nodes] tensor on the graph's device")
        self.local = torch.zeros(num_nodes, dtype=torch.int64, device=dev) if local is None \
            else local
        self.node_g = torch.empty(p.R, dtype=torch.int64, device=dev)
        self.src_g = torch.empty(p.S, dtype=torch.int64, device=dev)
        self.src_id = torch.empty(p.S, dtype=torch.int32, device=dev)
        self.row_end = torch.empty(p.R_dst, dtype=torch.int32, device=dev)
        self.row_end64 = torch.empty(p.R_dst, dtype=torch.int64, device=dev)  # (direct gather)
        self.inv_cnt = torch.empty(p.R_dst, dtype=torch.float32, device=dev)
        # counts and cursors of every transposed CSR in ONE buffer (one memset per batch)
        self._tbuf = torch.zeros(max(2 * sum(p.t_rows), 1), dtype=torch.int32, device=dev)
        self.t_counts, self.t_cursor, off = [], [], 0
        for n in p.t_rows:
            self.t_counts.append(self._tbuf[off:off + n])
            self.t_cursor.append(self._tbuf[off + n:off + 2 * n])
            off += 2 * n
        self.t_ptr = [torch.empty(n + 1, dtype=torch.int32, device=dev) for n in p.t_rows]
        self.t_col = [torch.zeros(max(n, 1), dtype=torch.int32, device=dev) for n in p.t_slots]
        self._fan_host = (ctypes.c_int32 * p.L)(*p.fanouts)
        self._bases_host = (ctypes.c_int64 * (p.L + 2))(*p.bases)
        self._slots_host = (ctypes.c_int64 * max(p.n_csr, 1))(*p.t_slots)
        self._rows_host = (ctypes.c_int64 * max(p.n_csr, 1))(*p.t_rows)

        def ptrs(ts):
            return (ctypes.c_void_p * max(len(ts), 1))(*[t.data_ptr() for t in ts])
        self._counts_host, self._cursor_host = ptrs(self.t_counts), ptrs(self.t_cursor)
        self._ptr_host, self._col_host = ptrs(self.t_ptr), ptrs(self.t_col)

    @torch.no_grad()
    def sample(self, seeds: Tensor, epoch_dev: Tensor) -> SlotBatch:
        """``seeds`` [B] graph node ids (device, dtype of ``colptr``); ``epoch_dev`` int64 [1] on the
        device, >= 1 and larger than for every earlier batch of this sampler (a captured step bumps
        it before every replay).  No host synchronisation; the returned tensors are the sampler's
        own buffers (valid until the next call)."""
        p, lib = self.plan, _lib.load()
        if seeds.numel() != p.B or seeds.dtype != self.colptr.dtype or not seeds.is_contiguous():
            raise ValueError(f"'seeds' must be {p.B} contiguous {self.colptr.dtype} node ids")
        if epoch_dev.dtype != torch.int64 or epoch_dev.numel() != 1 or not epoch_dev.is_cuda:
            raise ValueError("'epoch_dev' must be one int64 on the device")
        st = _native._stream(seeds)
        idt = _native._idx_dtype(self.colptr)
        ep = _i64p(epoch_dev)
        self._tbuf.zero_()
        check(lib.pygamd_slots_seed(_i64p(seeds), idt, p.B, ep, _i64p(self.local),
                                    _i64p(self.node_g), st), 'slots_seed')
        for h, k in enumerate(p.fanouts):
            check(lib.pygamd_slots_sample(
                _i64p(self.colptr), _i64p(self.row), idt, _i64p(self.node_g), p.bases[h],
                p.cap[h], k, p.ebase(h), p.B, self.seed & 0xFFFFFFFFFFFFFFFF, h, ep,
                _i64p(self.local), _i64p(self.src_g), _i64p(self.row_end), _i64p(self.inv_cnt),
                st), 'slots_sample')
            n_counts = max(p.L - 1 - h, 0)   # transposed CSRs c = 0 .. L - 2 - h contain hop h
            check(lib.pygamd_slots_resolve(
                _i64p(self.src_g), p.ebase(h), p.cap[h + 1], p.B, ep, _i64p(self.local),
                _i64p(self.src_id), _i64p(self.node_g), self._counts_host, n_counts, st),
                'slots_resolve')
        if p.n_csr > 0:
            check(lib.pygamd_slots_transpose(
                _i64p(self.src_g), _i64p(self.src_id), p.L, self._fan_host, self._bases_host,
                p.n_csr, self._slots_host, self._rows_host, self._counts_host, self._cursor_host,
                self._ptr_host, self._col_host, st), 'slots_transpose')
        self.calls += 1
        return SlotBatch(p, self.node_g, self.src_g, self.src_id, self.row_end, self.inv_cnt,
                         self.t_ptr, self.t_col, stamp=self.calls, owner=weakref.ref(self))

    @torch.no_grad()
    def gather(self, x: Tensor, batch: SlotBatch, out: Optional[Tensor] = None,
               direct: bool = False) -> Tensor:
        """``[R, 2 F]`` with ``x[node_g]`` in the right half (holes: zero rows); the left half is
        where layer 0 stores its aggregated rows.  ``direct``: only the ``R_dst`` rows that are ever
        a destination are copied (``[R_dst, 2 F]``) and the batch remembers ``x`` itself
        (``x_global``) and an int64 ``row_end``: layer 0 of the stack then reads its neighbours'
        rows from ``x`` by graph node id."""
        p = self.plan
        F = x.size(1)
        rows = p.R_dst if direct else p.R
        if direct:
            if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
                raise ValueError("'x' must be a float32 [N, F] matrix with unit inner stride")
            batch.x_global = x
            batch.row_end64 = self.row_end64.copy_(batch.row_end)   # (no allocation: a captured
                                                                    # side branch may run this)
        if out is None:
            out = torch.empty(rows, 2 * F, dtype=torch.float32, device=x.device)
        check(_lib.load().pygamd_slots_gather(
            _i64p(x), _native._ld(x), F, _i64p(batch.node_g), rows,
            ctypes.c_void_p(out.data_ptr() + 4 * F), _native._ld(out), _native._stream(x)),
            'slots_gather')
        return out


class FusedSageSlotStack(Function):
    """``L``-layer GraphSAGE (mean / sum aggregation, ReLU between layers, root weight + bias) on a
    :class:`SlotBatch` whose ``x`` is the gathered ``[R, 2 F]`` buffer; returns the seed rows
    ``[B, out_channels]``.  Layer ``l`` produces the rows of blocks ``0 .. L-l-1`` from the rows of
    blocks ``0 .. L-l`` (trim_to_layer)."""

    @staticmethod
    def forward(ctx, cat0: Tensor, batch: SlotBatch, aggr: str, *params: Optional[Tensor]):
        p = batch.plan
        batch.check_current('FusedSageSlotStack.forward')
        L = len(params) // 3
        if L != p.L:
            raise ValueError(f'{L} layers on a batch of {p.L} hops')
        direct = batch.x_global is not None
        rows0 = p.R_dst if direct else p.R
        if cat0.shape[0] != rows0 or cat0.dtype != torch.float32 or cat0.size(1) % 2:
            raise ValueError(f"'x' must be the float32 [{rows0}, 2 F] buffer of "
                             f"SlotSampler.gather")
        if direct and batch.x_global.size(1) != cat0.size(1) // 2:
            raise ValueError("'x_global' and the gathered buffer disagree about the width")
        dev = cat0.device
        Fi = cat0.size(1) // 2
        cat = cat0
        cats, wmats, bits = [], [], []
        out = None
        for l in range(L):
            W_l, b, W_r = params[3 * l:3 * l + 3]
            Fo = W_l.size(0)
            m = p.bases[L - l]                       # rows produced = blocks 0 .. L-l-1
            if not _native.sage_layer_forward_supported(Fi, Fo, aggr):
                raise ValueError(f'layer {l} ({Fi} -> {Fo}, {aggr}) is outside the one-kernel '
                                 f'layer (F % 4 == 0, F <= 256, Fo <= 256, sum / mean)')
            wmat = torch.cat([W_l, W_r], dim=1)
            last = l == L - 1
            if last:
                nxt, dst = None, torch.empty(m, Fo, dtype=torch.float32, device=dev)
                out = dst
                rb = None
            else:
                nxt = torch.empty(m, 2 * Fo, dtype=torch.float32, device=dev)
                dst = nxt[:, Fo:]
                rb = _native.relu_bits_like(m, Fo, dev)
            if l == 0 and direct:
                # neighbours by GRAPH node id from the feature matrix itself (slots past a row's
                # end hold -1 and are never dereferenced; a duplicate's slot names the same graph
                # node as the row `src_id` points at: same features)
                _native.sage_layer_forward(p.row_begin64, batch.src_g, batch.x_global,
                                           cat[:m, Fi:], wmat, b, aggr, not last, cat[:m, :Fi],
                                           dst, save_agg=True, relu_bits=rb,
                                           rowend=batch.row_end64)
            else:
                _native.sage_layer_forward(p.row_begin, batch.src_id, cat[:, Fi:], cat[:m, Fi:],
                                           wmat, b, aggr, not last, cat[:m, :Fi], dst,
                                           save_agg=True, relu_bits=rb, rowend=batch.row_end)
            cats.append(cat)
            wmats.append(wmat)
            bits.append(rb)
            cat, Fi = nxt, Fo
        ctx.batch, ctx.aggr, ctx.L = batch, aggr, L
        ctx.has_bias = [params[3 * i + 1] is not None for i in range(L)]
        ctx.n_bits = sum(b is not None for b in bits)
        ctx.save_for_backward(*cats, *wmats, *[b for b in bits if b is not None])
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        L, batch, aggr = ctx.L, ctx.batch, ctx.aggr
        batch.check_current('FusedSageSlotStack.backward')
        p = batch.plan
        saved = ctx.saved_tensors
        cats, wmats = saved[:L], saved[L:2 * L]
        bits = list(saved[2 * L:]) + [None]        # bits[l]: ReLU mask of layer l's output
        grads: List[Optional[Tensor]] = [None] * (3 * L)
        g = grad_out
        if g.dim() != 2 or g.stride(1) != 1 or (g.size(0) > 1 and g.stride(0) < g.size(1)):
            g = g.contiguous()
        grad_x = None
        for l in reversed(range(L)):
            cat, wmat = cats[l], wmats[l]
            Fi = cat.size(1) // 2
            m = p.bases[L - l]
            gw = _native.linear_wgrad(g, cat[:m], bias_grad=ctx.has_bias[l])
            if ctx.has_bias[l]:
                gw, grads[3 * l + 1] = gw
            grads[3 * l], grads[3 * l + 2] = gw[:, :Fi], gw[:, Fi:]
            if l == 0:
                if ctx.needs_input_grad[0]:
                    raise NotImplementedError('the slot stack does not differentiate its '
                                              'gathered input features')
                break
            c = l - 1                                # this layer's transposed CSR
            r_in = p.bases[L - l + 1]                # rows of the layer input = rows layer l-1 made
            w_t = wmat.t().contiguous()              # [2 Fi, Fo]: ONE transpose, both halves
            scale = batch.inv_cnt[:m] if aggr == 'mean' else None   # are row blocks of it
            gagg = _native.linear_dgrad(g, w_t[:Fi], row_scale=scale,
                                        n_scaled=Fi if scale is not None else 0)
            g_in = torch.empty(r_in, Fi, dtype=torch.float32, device=g.device)
            _native.linear_dgrad(g, w_t[Fi:], out=g_in[:m])
            # g_in = relu'(h) * (A^T gagg + [groot ; 0]): rows past m have no root part
            _native.spmm_csr(batch.t_ptr[c], batch.t_col[c], gagg, 'sum', n_rows=r_in, out=g_in,
                             accumulate=True, accumulate_rows=m, relu_bits=bits[l - 1])
            g = g_in
        return (grad_x, None, None, *grads)


def check_slot_model(model, batch: Optional[SlotBatch] = None) -> None:
    """Raises ``ValueError`` naming the first thing about ``model`` the slot stack would silently
    ignore or trip over (cf. ``nn.models._fused_sage_hops.eligible``, which steps aside instead):
    the stack computes ``relu(lin_l(aggr_j x_j) + lin_r(x_i))`` per layer and nothing else."""
    from .nn.conv import SAGEConv
    convs = list(getattr(model, 'convs', []))
    if not convs:
        raise ValueError('run_slot_stack needs a model with a `convs` list of SAGEConv layers')
    if batch is not None and len(convs) != batch.plan.L:
        raise ValueError(f'{len(convs)} layers on a batch of {batch.plan.L} hops')
    act = getattr(model, 'act', None)
    if act is not None and not isinstance(act, torch.nn.ReLU):
        raise ValueError(f'the slot stack applies ReLU between layers (model.act is '
                         f'{type(act).__name__})')
    drop = getattr(model, 'dropout', None)
    p_drop = drop.p if isinstance(drop, torch.nn.Dropout) else (drop or 0.0)
    if p_drop > 0 and model.training:
        raise ValueError(f'the slot stack has no dropout (model.dropout = {p_drop} in training '
                         f'mode)')
    if getattr(model, 'norms', None) is not None and any(
            not isinstance(n, torch.nn.Identity) for n in model.norms):
        raise ValueError('the slot stack has no normalisation layers (model.norms)')
    if getattr(model, 'jk_mode', None) not in (None, 'last'):
        raise ValueError(f"the slot stack has no jumping knowledge (jk='{model.jk_mode}')")
    aggr = convs[0].aggr
    for l, conv in enumerate(convs):
        if not isinstance(conv, SAGEConv):
            raise ValueError(f'convs[{l}] is a {type(conv).__name__}, not a SAGEConv')
        if conv.aggr not in ('mean', 'sum', 'add'):
            raise ValueError(f"convs[{l}].aggr = '{conv.aggr}': the slot stack aggregates with "
                             f"mean or sum")
        if conv.aggr != aggr:
            raise ValueError(f"convs[{l}].aggr = '{conv.aggr}' but convs[0].aggr = '{aggr}': one "
                             f"aggregation for the whole stack")
        if not conv.root_weight:
            raise ValueError(f'convs[{l}] has root_weight=False (no lin_r): unsupported')
        if getattr(conv, 'normalize', False):
            raise ValueError(f'convs[{l}] has normalize=True: unsupported')
        if getattr(conv, 'project', False):
            raise ValueError(f'convs[{l}] has project=True: unsupported')
        if getattr(conv, 'flow', 'source_to_target') != 'source_to_target':
            raise ValueError(f"convs[{l}].flow = '{conv.flow}': unsupported")


def run_slot_stack(model, batch: SlotBatch) -> Tensor:
    """``model``: a GraphSAGE of plain ``SAGEConv`` layers (mean / sum aggregation, root weight,
    ReLU, no norm / dropout / jk — anything else raises, :func:`check_slot_model`)."""
    check_slot_model(model, batch)
    params = []
    for conv in model.convs:
        params += [conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight]
    aggr = model.convs[0].aggr
    return FusedSageSlotStack.apply(batch.x, batch, 'sum' if aggr == 'add' else aggr, *params)


class SlotTrainer:
    r"""The reference's mini-batch loop for a GraphSAGE (``examples/multi_gpu/
    distributed_sampling.py:104-117``: sample, ``model(...)``, ``F.cross_entropy``, ``backward()``,
    ``Adam.step()``, under DDP the gradient all-reduce) as ONE static-shape step per batch that
    replays as one hipGraph — with every launch that only re-arranged parameters removed:

    * the parameters live in ONE flat buffer in the layout the layer kernels read (per layer the
      row block ``[W_l | W_r]`` then the bias); ``conv.lin_l.weight`` / ``conv.lin_r.weight`` /
      ``conv.lin_l.bias`` are re-bound as VIEWS of it (``state_dict`` / checkpoints keep working),
      their ``.grad`` as views of the flat gradient buffer, which is also the single all-reduce
      bucket — no concatenation per forward, no ``grad +=`` per parameter, no zero-fill;
    * the weight-gradient launch of a layer writes its ``[grad W_l | grad W_r]`` and bias gradient
      straight into that buffer;
    * ``pygamd_cross_entropy_step``: loss and ``d loss / d logits`` of the seed rows in one launch,
      labels read from the graph's label vector through the seed ids;
    * ``pygamd_adam_step``: one launch for all parameters, which also refreshes the transposed
      weights the input-gradient GEMMs read.

    ``trainer = SlotTrainer(model, loader)``; per batch ``loss = trainer.step(seeds)`` (``seeds``:
    ``loader.batch_size`` graph node ids on the device; the returned device scalar is the
    trainer's own buffer).  ``capture=True`` (default) records the step on its first call and
    replays it afterwards; the recording's warm-up runs leave parameters and optimizer state
    untouched.  With a ``torch.distributed`` process group the gradients are SUM-all-reduced and
    divided by the world size inside the optimizer launch (DDP's mean); over RCCL the collective
    is part of the recorded graph.  Arithmetic: the package's GEMM mode (``set_gemm_mode``);
    the update is ``torch.optim.Adam``'s (``amsgrad=False``).

    ``pipeline=True``: sampling + feature gather of the NEXT batch run on a second stream (their
    own recording), beside the forward / backward / optimizer of the current one (two sets of sampler
    buffers over one claim map; the draws are latency-bound integer work, the training chain is
    matrix / bandwidth work — what the reference's loader workers overlap from the host).
    ``step(seeds)`` then trains on the batch drawn from the PREVIOUS call's seeds while it draws
    the batch of ``seeds``: the first call only draws (its return value is meaningless),
    :meth:`finish` trains on the last drawn batch."""

    def __init__(self, model, loader, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, process_group=None, capture: bool = True,
                 collective_in_graph: Optional[bool] = None, pipeline: bool = False):
        check_slot_model(model)
        if loader.y is None:
            raise ValueError('SlotTrainer needs the loader to hold the label vector (y=...)')
        if loader.y.dtype != torch.int64 or loader.y.dim() != 1:
            raise ValueError("'y' must be a 1-D int64 label vector")
        smp = loader.sampler
        if smp.replace or smp.disjoint or smp.subgraph_type != 'directional' \
                or any(k < 1 for k in smp.num_neighbors):
            raise ValueError('slot batches cover bounded fan-outs, directional, non-disjoint, '
                             'without replacement')
        self.model, self.loader = model, loader
        self.lr, self.betas, self.eps, self.weight_decay = lr, tuple(betas), eps, weight_decay
        self.group = process_group
        convs = list(model.convs)
        self.L = len(convs)
        if self.L != len(smp.num_neighbors):
            raise ValueError(f'{self.L} layers on batches of {len(smp.num_neighbors)} hops')
        aggr = convs[0].aggr
        self.aggr = 'sum' if aggr == 'add' else aggr
        dev = convs[0].lin_l.weight.device
        self.device = dev
        _native._require_device(convs[0].lin_l.weight, loader.x, loader.y)
        # ---- the flat layout: per layer [Fo, 2 Fi] then [Fo] (each block padded to 16 bytes)
        off, t_off = 0, 0
        self._shape, self._woff, self._boff, self._toff = [], [], [], []
        for l, conv in enumerate(convs):
            Fo, Fi = conv.lin_l.weight.shape
            if conv.lin_r.weight.shape != (Fo, Fi) or conv.lin_l.weight.dtype != torch.float32:
                raise ValueError(f'convs[{l}]: lin_l / lin_r must be float32 [{Fo}, {Fi}]')
            if not _native.sage_layer_forward_supported(Fi, Fo, self.aggr):
                raise ValueError(f'layer {l} ({Fi} -> {Fo}, {self.aggr}) is outside the one-kernel '
                                 f'layer (F % 4 == 0, F <= 256, Fo <= 256, sum / mean)')
            self._shape.append((Fo, Fi))
            self._woff.append(off)
            off += -(-(Fo * 2 * Fi) // 4) * 4
            self._boff.append(off if conv.lin_l.bias is not None else None)
            if conv.lin_l.bias is not None:
                off += -(-Fo // 4) * 4
            self._toff.append(t_off if l > 0 else None)
            if l > 0:
                t_off += Fo * 2 * Fi
        self.n = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self.wt_flat = torch.zeros(max(t_off, 1), dtype=torch.float32, device=dev)
        self.wmat, self.gmat, self.bias, self.gbias, self.wt = [], [], [], [], []
        with torch.no_grad():
            for l, conv in enumerate(convs):
                Fo, Fi = self._shape[l]
                o = self._woff[l]
                wm = self.flat[o:o + Fo * 2 * Fi].view(Fo, 2 * Fi)
                gm = self.grad[o:o + Fo * 2 * Fi].view(Fo, 2 * Fi)
                wm[:, :Fi].copy_(conv.lin_l.weight)
                wm[:, Fi:].copy_(conv.lin_r.weight)
                conv.lin_l.weight.data = wm[:, :Fi]
                conv.lin_r.weight.data = wm[:, Fi:]
                conv.lin_l.weight.grad = gm[:, :Fi]
                conv.lin_r.weight.grad = gm[:, Fi:]
                self.wmat.append(wm)
                self.gmat.append(gm)
                if self._boff[l] is not None:
                    bo = self._boff[l]
                    bv, gb = self.flat[bo:bo + Fo], self.grad[bo:bo + Fo]
                    bv.copy_(conv.lin_l.bias)
                    conv.lin_l.bias.data = bv
                    conv.lin_l.bias.grad = gb
                    self.bias.append(bv)
                    self.gbias.append(gb)
                else:
                    self.bias.append(None)
                    self.gbias.append(None)
                if l > 0:
                    to = self._toff[l]
                    self.wt.append(self.wt_flat[to:to + Fo * 2 * Fi].view(2 * Fi, Fo))
                else:
                    self.wt.append(None)
        segs = [l for l in range(self.L) if l > 0]
        self._seg_n = len(segs)
        n = max(len(segs), 1)
        self._seg_off = (ctypes.c_int64 * n)(*[self._woff[l] for l in segs])
        self._seg_rows = (ctypes.c_int32 * n)(*[self._shape[l][0] for l in segs])
        self._seg_cols = (ctypes.c_int32 * n)(*[2 * self._shape[l][1] for l in segs])
        self._seg_toff = (ctypes.c_int64 * n)(*[self._toff[l] for l in segs])
        self.refresh()
        # ---- the samplers: one set of batch buffers, two when the next batch is drawn beside the
        # current one's training (one claim map between them)
        from .loader import SLOTS_DIRECT
        B = loader.batch_size
        self.B = B
        self._pipeline = bool(pipeline)
        self._direct = SLOTS_DIRECT
        plan = SlotPlan(B, smp.num_neighbors, dev)
        first = SlotSampler(smp.colptr, smp.row, loader.num_nodes, plan, seed=smp.seed)
        self._samplers = [first]
        if self._pipeline:
            self._samplers.append(SlotSampler(smp.colptr, smp.row, loader.num_nodes, plan,
                                              seed=smp.seed, local=first.local))
        loader._slots = first    # (what `loader.collate_slots` would have built)
        Fin = self._shape[0][1]
        if loader.x.dim() != 2 or loader.x.size(1) != Fin:
            raise ValueError(f'the loader holds {tuple(loader.x.shape)} features, the model takes '
                             f'{Fin}')
        rows = plan.R_dst if self._direct else plan.R
        self._xbuf = [torch.empty(rows, 2 * Fin, dtype=torch.float32, device=dev)
                      for _ in self._samplers]
        self._batch = [None] * len(self._samplers)
        self._cur = 0            # pipeline: the buffer set the next training branch reads
        self._primed = False
        self._side = torch.cuda.Stream(dev) if self._pipeline else None
        # ---- per-step state on the device
        self.seeds = torch.zeros(B, dtype=smp.colptr.dtype, device=dev)
        self.epoch = torch.zeros(1, dtype=torch.int64, device=dev)   # stamps / salts the draws
        self.opt_step = torch.zeros(1, dtype=torch.int64, device=dev)  # Adam's step count
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.label_err = torch.zeros(1, dtype=torch.int32, device=dev)
        self._labels_of = [torch.zeros(B, dtype=torch.int64, device=dev) for _ in self._samplers]
        nbytes = ctypes.c_size_t(0)
        check(_lib.load().pygamd_cross_entropy_step_workspace_bytes(B, ctypes.byref(nbytes)))
        self._ce_ws = torch.zeros(nbytes.value, dtype=torch.uint8, device=dev)
        self._capture = bool(capture)
        self._graphs = None
        import torch.distributed as dist
        self._dist = dist.is_available() and dist.is_initialized()
        self._world = dist.get_world_size(process_group) if self._dist else 1
        if collective_in_graph is None:
            collective_in_graph = self._dist and dist.get_backend(process_group) == 'nccl'
        self._collective_in_graph = bool(collective_in_graph) and self._dist

    # ---- host-side maintenance
    @torch.no_grad()
    def refresh(self) -> None:
        """Re-derives the transposed weights from the parameters (call after writing to them from
        outside: ``load_state_dict``, a manual initialisation)."""
        for l in range(1, self.L):
            self.wt[l].copy_(self.wmat[l].t())

    def check_labels(self) -> None:
        """One host read: raises if a step met a label outside ``[0, out_channels)`` (such rows
        contribute neither loss nor gradient)."""
        if int(self.label_err.item()) != 0:
            self.label_err.zero_()
            raise IndexError(f'a seed label was outside [0, {self._shape[-1][0]})')

    # ---- the two halves of a step
    @torch.no_grad()
    def _draw(self, k: int) -> None:
        """Sampling + feature gather of ``self.seeds`` into buffer set ``k``; allocation-free (it
        may run as a side branch of a recording)."""
        self.epoch.add_(1)
        smp = self._samplers[k]
        b = smp.sample(self.seeds, self.epoch)
        b.x = smp.gather(self.loader.x, b, out=self._xbuf[k], direct=self._direct)
        self._labels_of[k].copy_(self.seeds)   # the seed ids this batch's labels are read through
        self._batch[k] = b

    @torch.no_grad()
    def _train(self, k: int) -> None:
        """Forward, loss, backward (gradients into the flat buffer), collective, optimizer on the
        batch in buffer set ``k``."""
        p_lib = _lib.load()
        b = self._batch[k]
        p, L, aggr, dev = b.plan, self.L, self.aggr, self.device
        b.check_current('SlotTrainer.step')
        direct = b.x_global is not None
        cat = b.x
        Fi = cat.size(1) // 2
        cats, bits = [], []
        out = None
        for l in range(L):
            Fo = self._shape[l][0]
            m = p.bases[L - l]
            last = l == L - 1
            if last:
                nxt, dst, rb = None, torch.empty(m, Fo, dtype=torch.float32, device=dev), None
                out = dst
            else:
                nxt = torch.empty(m, 2 * Fo, dtype=torch.float32, device=dev)
                dst = nxt[:, Fo:]
                rb = _native.relu_bits_like(m, Fo, dev)
            if l == 0 and direct:
                _native.sage_layer_forward(p.row_begin64, b.src_g, b.x_global, cat[:m, Fi:],
                                           self.wmat[l], self.bias[l], aggr, not last,
                                           cat[:m, :Fi], dst, save_agg=True, relu_bits=rb,
                                           rowend=b.row_end64)
            else:
                _native.sage_layer_forward(p.row_begin, b.src_id, cat[:, Fi:], cat[:m, Fi:],
                                           self.wmat[l], self.bias[l], aggr, not last,
                                           cat[:m, :Fi], dst, save_agg=True, relu_bits=rb,
                                           rowend=b.row_end)
            cats.append(cat)
            bits.append(rb)
            cat, Fi = nxt, Fo
        # ---- loss of the seed rows + its gradient (+ the optimizer's step count)
        C = out.size(1)
        g = torch.empty(self.B, C, dtype=torch.float32, device=dev)
        check(p_lib.pygamd_cross_entropy_step(
            _i64p(out), _native._ld(out), None, self.B, self.B, C, _i64p(self.loader.y),
            _i64p(self._labels_of[k]), _i64p(g), _native._ld(g), _i64p(self.loss),
            _i64p(self._ce_ws), self._ce_ws.numel(), _i64p(self.label_err),
            _i64p(self.opt_step), _native._stream(out)), 'cross_entropy_step')
        # ---- backward (FusedSageSlotStack.backward with the gradients going to the flat buffer)
        for l in reversed(range(L)):
            cat = cats[l]
            Fi = cat.size(1) // 2
            m = p.bases[L - l]
            has_bias = self.bias[l] is not None
            _native.linear_wgrad(g, cat[:m], out=self.gmat[l], bias_grad=has_bias,
                                 bias_out=self.gbias[l])
            if l == 0:
                break
            c = l - 1
            r_in = p.bases[L - l + 1]
            w_t = self.wt[l]
            scale = b.inv_cnt[:m] if aggr == 'mean' else None
            gagg = _native.linear_dgrad(g, w_t[:Fi], row_scale=scale,
                                        n_scaled=Fi if scale is not None else 0)
            g_in = torch.empty(r_in, Fi, dtype=torch.float32, device=dev)
            _native.linear_dgrad(g, w_t[Fi:], out=g_in[:m])
            _native.spmm_csr(b.t_ptr[c], b.t_col[c], gagg, 'sum', n_rows=r_in, out=g_in,
                             accumulate=True, accumulate_rows=m, relu_bits=bits[l - 1])
            g = g_in
        if self._dist and self._collective_in_graph:
            self._all_reduce()
        if not self._dist or self._collective_in_graph:
            self._optimizer()

    def _all_reduce(self) -> None:
        import torch.distributed as dist
        dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.group)

    def _optimizer(self) -> None:
        check(_lib.load().pygamd_adam_step(
            _i64p(self.flat), _i64p(self.grad), _i64p(self.exp_avg), _i64p(self.exp_avg_sq),
            self.n, _i64p(self.opt_step), 0, self.lr, self.betas[0], self.betas[1],
            self.eps, self.weight_decay, 1.0 / self._world,
            _i64p(self.wt_flat) if self._seg_n else None, self._seg_n, self._seg_off,
            self._seg_rows, self._seg_cols, self._seg_toff, _native._stream(self.flat)),
            'adam_step')

    def _behind(self) -> None:   # a group that cannot be recorded (gloo): behind the graph
        if self._dist and not self._collective_in_graph:
            self._all_reduce()
            self._optimizer()

    def _body(self, k: int = 0) -> None:
        """One step on buffer set ``k``.  Plain: draw, then train.  Pipelined: train on set ``k``
        while a side stream draws ``self.seeds`` into the other set (fork / join — inside a
        recording the two become parallel branches of the graph)."""
        if not self._pipeline:
            self._draw(0)
            self._train(0)
            return
        main = torch.cuda.current_stream(self.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            self._draw(1 - k)
        self._train(k)
        main.wait_stream(self._side)

    def _state(self):
        return (self.flat, self.exp_avg, self.exp_avg_sq, self.wt_flat, self.opt_step)

    def _record(self) -> None:
        from .hipgraph import CapturedStep
        # (the warm-up runs are real steps on throw-away state: parameters, moments and Adam's
        # count are put back; the sampling epoch keeps growing — the claim map needs that)
        keep = [t.clone() for t in self._state()]
        sets = (0, 1) if self._pipeline else (0, )
        if self._pipeline:       # every training branch needs a drawn batch to run on
            self._draw(0)
            self._draw(1)

        def warm(k):
            self._body(k)
            if self._dist and not self._collective_in_graph:
                self._optimizer()   # (eagerly; no collective: throw-away state)

        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                for k in sets:
                    warm(k)
        torch.cuda.current_stream(self.device).wait_stream(side)
        if self._pipeline:
            # Two recordings per buffer set — the training chain and the draw — replayed on two
            # STREAMS (round 6: as two branches of one graph they replayed one after the other on
            # this runtime: 0.959 vs 0.967 ms per batch, DEBUG_HIP_FORCE_GRAPH_QUEUES no help)
            self._graphs = [(CapturedStep(lambda k=k: self._train(k), warmup=0),
                             CapturedStep(lambda k=k: self._draw(k), warmup=0)) for k in sets]
        else:
            self._graphs = [CapturedStep(lambda: self._body(0), warmup=0)]
        for t, k in zip(self._state(), keep):
            t.copy_(k)

    def step(self, seeds: Optional[Tensor] = None) -> Tensor:
        """One training step on ``seeds`` (None: whatever ``self.seeds`` holds).  Returns the
        loss buffer (a device scalar, overwritten by the next step).  Pipelined: see the class
        docstring (the batch of ``seeds`` is trained on by the NEXT call)."""
        if seeds is not None:
            if seeds.numel() != self.B:
                raise ValueError(f'{self.B} seeds per step (the loader\'s batch size), got '
                                 f'{seeds.numel()}')
            self.seeds.copy_(seeds)
        if self._capture and self._graphs is None:
            self._record()
        if self._pipeline and not self._primed:
            self._draw(self._cur)     # the first call only draws
            self._primed = True
            return self.loss
        if not self._capture:
            self._body(self._cur)
        elif not self._pipeline:
            self._graphs[0]()
        else:
            k = self._cur
            main = torch.cuda.current_stream(self.device)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                self._graphs[1 - k][1]()     # draw `self.seeds` into the other buffer set
            self._graphs[k][0]()             # train on this one
            main.wait_stream(self._side)
        self._behind()
        if self._pipeline:
            self._cur = 1 - self._cur
        return self.loss

    def finish(self) -> Tensor:
        """Pipelined: trains on the last drawn batch without drawing another one (eagerly)."""
        if self._pipeline and self._primed:
            self._train(self._cur)
            self._behind()
            self._primed = False
        return self.loss
