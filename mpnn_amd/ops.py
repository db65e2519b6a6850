"""torch.autograd bindings of the HIP kernels (the only callers of the C ABI).

Each Function enqueues its kernel on torch's current stream and owns nothing: torch tensors
are the buffers.  Backward passes call the matching *_bwd kernels.
"""
import os

import torch

from . import _lib


def _empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


class KernelTimer:
    """Opt-in per-launch timing with HIP events on the launch stream (bench.py's roofline leg).
    Events are recorded on torch's current stream, which is the stream every kernel here is
    launched on (`_lib.stream()`)."""

    def __init__(self, names):
        self.names = set(names)
        self.events = {n: [] for n in names}

    def launch(self, name, fn):
        if name not in self.names:
            return fn()
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.events[name].append((a, b))
        return r

    def mean_ms(self, name):
        ev = self.events[name]
        if not ev:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev)

    def reset(self):
        for v in self.events.values():
            v.clear()


_timer = None


def set_kernel_timer(timer):
    global _timer
    _timer = timer


def _timed(name, fn):
    return fn() if _timer is None else _timer.launch(name, fn)


def math_mode():
    """"fp32" (MPNN_GRU_MATH=fp32: every contraction on the fp32 matrix pipe) or "split" (default: operand splits on the
    16-bit matrix pipe with fp32 accumulation)."""
    return "fp32" if os.environ.get("MPNN_GRU_MATH") == "fp32" else "split"


def mfma_per_product(kernel=None, hidden=64):
    """16-bit MFMAs issued per fp32 product by a contraction kernel (`kernel` = its KernelTimer label; 0 = the fp32
    matrix pipe itself).  Two range-guarded fp16 pieces, three MFMAs: the fused message + sum kernels and the GRU forward
    and backward at widths 64 / 128 / 256.  Three bf16 pieces, six MFMAs: the per-edge message kernels."""
    if math_mode() == "fp32":
        return 0
    if kernel in ("gru_update", "gru_update_bwd") and hidden in (64, 128, 256):
        return 3
    if kernel == "message_aggregate":
        return 3
    return 6


def math_description():
    return {"fp32": "fp32 matrix pipe (MPNN_GRU_MATH=fp32)",
            "split": "fp32 data and accumulation; dense contractions on the 16-bit matrix pipe: range-guarded two-way fp16 "
                     "operand splits (three MFMAs per product) in the fused message+sum kernels and the GRU forward and "
                     "backward at widths 64/128/256, three-way bf16 splits (six MFMAs) in the per-edge message kernels; "
                     "parity 1e-5 as the fp32 kernels"}[math_mode()]


# --------------------------------------------------------------------------- raw launches
def segsum_raw(msg, row_ptr, w, num_rows, label="segsum"):
    """`label` names the launch for ops.KernelTimer: "segsum" is reserved for the aggregator proper (message rows
    summed per destination atom), so per-molecule sums and gradient scatters do not dilute its measured time."""
    lib = _lib.load()
    F = int(msg.shape[-1])
    if msg.shape[0] == 0 or num_rows == 0:          # no edges at all: every row is an empty sum
        return torch.zeros(num_rows, F, dtype=torch.float32, device=msg.device)
    out = _empty((num_rows, F), msg)
    _lib.check(_timed(label, lambda: lib.mpnn_segsum_f32(
        _lib.fptr(msg), _lib.iptr(row_ptr), _lib.fptr(w), _lib.fptr(out), num_rows, F, _lib.stream())),
        "mpnn_segsum_f32")
    return out


def segsum_bwd_raw(dout, row_ptr, w, num_edges):
    lib = _lib.load()
    V, F = int(dout.shape[0]), int(dout.shape[1])
    dmsg = _empty((num_edges, F), dout)
    if num_edges == 0 or V == 0:
        return dmsg
    _lib.check(lib.mpnn_segsum_bwd_f32(_lib.fptr(dout), _lib.iptr(row_ptr), _lib.fptr(w), _lib.fptr(dmsg),
                                       V, F, _lib.stream()), "mpnn_segsum_bwd_f32")
    return dmsg


def segsum_gather_raw(x, row_ptr, idx, w, num_rows):
    lib = _lib.load()
    F = int(x.shape[-1])
    if x.shape[0] == 0 or num_rows == 0 or (idx is not None and idx.shape[0] == 0):
        return torch.zeros(num_rows, F, dtype=torch.float32, device=x.device)
    out = _empty((num_rows, F), x)
    _lib.check(lib.mpnn_segsum_gather_f32(_lib.fptr(x), _lib.iptr(row_ptr), _lib.iptr(idx), _lib.fptr(w),
                                          _lib.fptr(out), num_rows, F, _lib.stream()), "mpnn_segsum_gather_f32")
    return out


def edge_message_raw(h, A, graph, gate=None):
    lib = _lib.load()
    K, mf, nf = (int(s) for s in A.shape)
    E = graph.num_edges
    msg = _empty((E, mf), h)
    if E == 0:
        return msg
    _lib.check(_timed("edge_message", lambda: lib.mpnn_edge_message_f32(
        _lib.fptr(h), _lib.fptr(A), _lib.iptr(graph.col_idx), _lib.iptr(graph.order), _lib.iptr(graph.type_ptr),
        _lib.fptr(gate), _lib.fptr(msg), graph.num_nodes, E, K, nf, mf, _lib.stream())),
        "mpnn_edge_message_f32")
    return msg


def tile_kernel_applies(A, gate, w, graph):
    """The fused message+sum tile kernel covers: no gate, unit edge weights, nf = mf = 64, a batch of separate
    molecules of at most one tile each with few bond types (graph.tile_plan), default math.  MPNN_UNFUSED_MESSAGE=1
    keeps the two-kernel path (message rows to HBM, then the segmented-sum aggregator) for A/B runs."""
    K, mf, nf = (int(s) for s in A.shape)
    if (gate is not None or w is not None or mf != 64 or nf != 64 or math_mode() == "fp32"
            or os.environ.get("MPNN_UNFUSED_MESSAGE")):
        return False
    plan = graph.tile_plan
    return plan is not None and K == graph.num_types


def message_aggregate_tile_raw(h, A, graph):
    lib = _lib.load()
    K, mf, nf = (int(s) for s in A.shape)
    V = graph.num_nodes
    plan = graph.tile_plan
    if V == 0 or graph.num_edges == 0:                  # no bonds at all: every row is an empty sum
        return torch.zeros(V, mf, dtype=torch.float32, device=h.device)
    out = _empty((V, mf), h)
    _lib.check(_timed("message_aggregate", lambda: lib.mpnn_message_aggregate_f32(
        _lib.fptr(h), _lib.fptr(A), _lib.iptr(plan.tile_rec), _lib.iptr(plan.tile_atom), _lib.iptr(plan.slots),
        _lib.fptr(out), V, plan.num_tiles, K, nf, mf, _lib.stream())), "mpnn_message_aggregate_f32")
    return out


def wide_kernel_applies(A, gate, w, graph):
    """The fused message+sum kernels on molecule tiles of up to 256 atoms (typed aggregate-then-contract,
    csrc/message_tile_wide.hip) cover: no gate, unit edge weights, nf = mf in {128, 256} with at most 8 bond types, or
    nf = mf = 64 with at most 4 (the resident-matrix form: what width-64 batches with molecules of more than 128 atoms
    take, the 128-atom tile kernel of message_tile.hip being tried first), a batch of separate molecules
    (graph.wide_plan), default math.  MPNN_UNFUSED_MESSAGE=1 keeps the two-kernel path."""
    K, mf, nf = (int(s) for s in A.shape)
    widths = (64, 128, 256) if K <= 4 else (128, 256)    # (width 64: resident matrices, K <= 4)
    if (gate is not None or w is not None or mf != nf or nf not in widths or math_mode() == "fp32"
            or os.environ.get("MPNN_UNFUSED_MESSAGE")):
        return False
    plan = graph.wide_plan
    return plan is not None and K == graph.num_types


def message_aggregate_wide_raw(h, A, graph):
    lib = _lib.load()
    K, mf, nf = (int(s) for s in A.shape)
    V = graph.num_nodes
    plan = graph.wide_plan
    if V == 0 or graph.num_edges == 0:                  # no bonds at all: every row is an empty sum
        return torch.zeros(V, mf, dtype=torch.float32, device=h.device)
    out = _empty((V, mf), h)
    ws_bytes = lib.mpnn_message_aggregate_wide_workspace_bytes(K, nf)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=h.device)
    _lib.check(_timed("message_aggregate", lambda: lib.mpnn_message_aggregate_wide_f32(
        _lib.fptr(h), _lib.fptr(A), _lib.iptr(plan.tile_rec), _lib.iptr(plan.tile_atom), _lib.iptr(plan.blk_off),
        _lib.ptr(plan.slots, torch.int16), _lib.fptr(out), _lib.ptr(ws), ws_bytes, V, plan.num_tiles, K, nf, mf,
        _lib.stream())), "mpnn_message_aggregate_wide_f32")
    return out


def wide_gated_applies(A, w, graph):
    """The fused message+sum kernel with AttEdgeNetwork's feature gate formed inside it (mpnn_message_aggregate_wide_gated_f32):
    nf = mf = 128, at most four bond types, otherwise as wide_kernel_applies."""
    K, mf, nf = (int(s) for s in A.shape)
    return nf == 128 and K <= 4 and wide_kernel_applies(A, None, w, graph)


def message_aggregate_wide_gated_raw(h, A, z_atom, q, graph, keep_workspace=False):
    lib = _lib.load()
    K, mf, nf = (int(s) for s in A.shape)
    V = graph.num_nodes
    plan = graph.wide_plan
    if V == 0 or graph.num_edges == 0:
        out = torch.zeros(V, mf, dtype=torch.float32, device=h.device)
        return (out, None) if keep_workspace else out
    out = _empty((V, mf), h)
    ws_bytes = lib.mpnn_message_aggregate_wide_gated_workspace_bytes(K, nf, plan.num_tiles)
    ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=h.device)
    _lib.check(_timed("message_aggregate", lambda: lib.mpnn_message_aggregate_wide_gated_f32(
        _lib.fptr(h), _lib.fptr(A), _lib.fptr(z_atom), _lib.fptr(q), _lib.iptr(plan.tile_rec), _lib.iptr(plan.tile_atom),
        _lib.iptr(plan.blk_off), _lib.ptr(plan.slots, torch.int16), _lib.fptr(out), _lib.ptr(ws), ws_bytes, V,
        plan.num_tiles, K, nf, mf, _lib.stream())), "mpnn_message_aggregate_wide_gated_f32")
    if keep_workspace:
        return out, ws
    return out


def message_aggregate_wide_gated_bwd_raw(h, A, z_atom, q, dout, fwd_ws, graph):
    """Backward of the gated message + sum per (atom, type) on the forward's plan -> (dA, dz_atom, dq); no (E, nf) tensor.
    Two kernels: the logits' gradient on the plan (mpnn_message_aggregate_wide_gated_bwd_f32, which also hands the softmax
    statistics over in atom order), then the matrices' gradient with the gate evaluated in flight
    (mpnn_edge_message_agg_bwd_da_att_f32); dq's first term is read off A * dA (include/mpnn_amd.h has the algebra)."""
    lib = _lib.load()
    K, mf, nf = (int(s) for s in A.shape)
    V, E = graph.num_nodes, graph.num_edges
    plan = graph.wide_plan
    dz = _empty((V, nf), h)
    parts = int(lib.mpnn_message_aggregate_wide_gated_bwd_parts())
    dq_part = torch.zeros((parts, K, nf), dtype=torch.float32, device=h.device)
    stats_atom = _empty((V, K, 2), h)
    ws_bytes = lib.mpnn_message_aggregate_wide_gated_bwd_workspace_bytes(K, nf)
    ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=h.device)
    _lib.check(_timed("att_message_bwd", lambda: lib.mpnn_message_aggregate_wide_gated_bwd_f32(
        _lib.fptr(h), _lib.fptr(A), _lib.fptr(z_atom), _lib.fptr(q), _lib.fptr(dout), _lib.ptr(fwd_ws), fwd_ws.numel() * 4,
        _lib.iptr(plan.tile_rec), _lib.iptr(plan.tile_atom), _lib.iptr(plan.blk_off), _lib.ptr(plan.slots, torch.int16),
        _lib.fptr(dz), _lib.fptr(dq_part), _lib.fptr(stats_atom), _lib.ptr(ws), ws_bytes, V, plan.num_tiles, K, nf, mf,
        _lib.stream())), "mpnn_message_aggregate_wide_gated_bwd_f32")
    dA = torch.zeros_like(A)
    _lib.check(_timed("message_aggregate_bwd", lambda: lib.mpnn_edge_message_agg_bwd_da_att_f32(
        _lib.fptr(dout), _lib.fptr(h), _lib.iptr(graph.col_idx), _lib.iptr(graph.edge_dst), _lib.iptr(graph.order),
        _lib.iptr(graph.type_ptr), _lib.fptr(z_atom), _lib.fptr(q), _lib.fptr(stats_atom), _lib.fptr(dA), V, E, K, nf, mf,
        _lib.stream())), "mpnn_edge_message_agg_bwd_da_att_f32")
    dq = (A * dA).sum(1) - dq_part.sum(0)
    return dA, dz, dq


def edge_message_bwd_raw(h, A, graph, gate, dmsg, need_dx=True, need_dA=True, dA_accum=None):
    """`dA_accum`: a (K, mf, nf) buffer the weight gradient is ADDED to (the kernels accumulate) instead of a fresh zero one."""
    lib = _lib.load()
    K, mf, nf = (int(s) for s in A.shape)
    E = graph.num_edges
    dx = _empty((E, nf), h) if need_dx else None
    dA = (dA_accum if dA_accum is not None else torch.zeros_like(A)) if need_dA else None
    if E == 0 or not (need_dx or need_dA):
        return dx, dA
    _lib.check(lib.mpnn_edge_message_bwd_f32(_lib.fptr(h), _lib.fptr(A), _lib.iptr(graph.col_idx),
                                             _lib.iptr(graph.order), _lib.iptr(graph.type_ptr), _lib.fptr(gate),
                                             _lib.fptr(dmsg), _lib.fptr(dx), _lib.fptr(dA),
                                             graph.num_nodes, E, K, nf, mf, _lib.stream()),
               "mpnn_edge_message_bwd_f32")
    return dx, dA


def gru_update_raw(m, h, mask, W_ih, W_hh, b_ih, b_hh, save):
    lib = _lib.load()
    V, H = int(h.shape[0]), int(h.shape[1])
    out = _empty((V, H), h)
    saved = _empty((V, 4 * H), h) if save else None
    ws_bytes = lib.mpnn_gru_fwd_workspace_bytes(V, H)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=h.device) if ws_bytes else None
    _lib.check(_timed("gru_update", lambda: lib.mpnn_gru_update_f32(
        _lib.fptr(m), _lib.fptr(h), _lib.fptr(mask), _lib.fptr(W_ih), _lib.fptr(W_hh), _lib.fptr(b_ih),
        _lib.fptr(b_hh), _lib.fptr(out), _lib.fptr(saved), _lib.ptr(ws), ws_bytes, V, H, _lib.stream())),
        "mpnn_gru_update_f32")
    return out, saved


def gru_weight_grad_buffers(H, like):
    """(dW_ih, dW_hh, db_ih, db_hh) as views of ONE zero-filled buffer (one fill launch instead of four: the reference driver's
    batches of 16 molecules are bound by the count of such launches); the backward kernels ACCUMULATE into them."""
    nW, nb = (3 * H * H + 3) // 4 * 4, (3 * H + 3) // 4 * 4          # (every view starts on a 16-byte boundary)
    zeros = torch.zeros(2 * nW + 2 * nb, dtype=torch.float32, device=like.device)
    return (zeros[:3 * H * H].view(H, 3 * H), zeros[nW:nW + 3 * H * H].view(H, 3 * H),
            zeros[2 * nW:2 * nW + 3 * H], zeros[2 * nW + nb:2 * nW + nb + 3 * H])


def gru_update_bwd_raw(dout, m, h, mask, W_ih, W_hh, saved, accum=None):
    """`accum`: gru_weight_grad_buffers of an earlier call -- this call's weight gradients are added to them (the T updates of
    a model share their weights: one buffer instead of T buffers and T - 1 additions per parameter)."""
    lib = _lib.load()
    V, H = int(h.shape[0]), int(h.shape[1])
    dm, dh = _empty((V, H), h), _empty((V, H), h)
    dW_ih, dW_hh, db_ih, db_hh = accum if accum is not None else gru_weight_grad_buffers(H, h)
    ws_bytes = lib.mpnn_gru_bwd_workspace_bytes(V, H)
    ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=h.device)
    _lib.check(_timed("gru_update_bwd", lambda: lib.mpnn_gru_update_bwd_f32(
        _lib.fptr(dout), _lib.fptr(m), _lib.fptr(h), _lib.fptr(mask), _lib.fptr(W_ih), _lib.fptr(W_hh),
        _lib.fptr(saved), _lib.fptr(dm), _lib.fptr(dh), _lib.fptr(dW_ih), _lib.fptr(dW_hh), _lib.fptr(db_ih),
        _lib.fptr(db_hh), _lib.ptr(ws), ws_bytes, V, H, _lib.stream())), "mpnn_gru_update_bwd_f32")
    return dm, dh, dW_ih, dW_hh, db_ih, db_hh


# --------------------------------------------------------------------------- autograd
class SegSum(torch.autograd.Function):
    """out[i] = sum_{e in row i} w[e] * msg[e]   (w optional, not differentiated)."""

    @staticmethod
    def forward(ctx, msg, row_ptr, w, num_rows):
        msg = msg.contiguous()
        ctx.save_for_backward(row_ptr, w)
        ctx.num_edges = int(msg.shape[0])
        return segsum_raw(msg, row_ptr, w, num_rows)

    @staticmethod
    def backward(ctx, dout):
        row_ptr, w = ctx.saved_tensors
        return segsum_bwd_raw(dout.contiguous(), row_ptr, w, ctx.num_edges), None, None, None


class SegSumGather(torch.autograd.Function):
    """out[i] = sum_{e in row i} w[e] * x[idx[e]];  backward runs the same kernel on the transposed
    index (`t_row_ptr`, `t_idx`, `t_w`): dx[j] = sum_{e: idx[e]=j} w[e] * dout[row(e)]."""

    @staticmethod
    def forward(ctx, x, row_ptr, idx, w, num_rows, t_row_ptr, t_idx, t_w):
        x = x.contiguous()
        ctx.save_for_backward(t_row_ptr, t_idx, t_w)
        ctx.num_src = int(x.shape[0])
        return segsum_gather_raw(x, row_ptr, idx, w, num_rows)

    @staticmethod
    def backward(ctx, dout):
        t_row_ptr, t_idx, t_w = ctx.saved_tensors
        if t_row_ptr is None:
            raise _lib.MpnnError("SegSumGather: no transposed index was supplied, cannot differentiate")
        dx = segsum_gather_raw(dout.contiguous(), t_row_ptr, t_idx, t_w, ctx.num_src)
        return dx, None, None, None, None, None, None, None


class EdgeMessage(torch.autograd.Function):
    """msg[e] = A[type(e)] . (gate[e] * h[src(e)])"""

    @staticmethod
    def forward(ctx, h, A, gate, graph):
        h = h.contiguous()
        A = A.contiguous()
        gate = gate.contiguous() if gate is not None else None
        ctx.graph = graph
        ctx.save_for_backward(h, A, gate)
        return edge_message_raw(h, A, graph, gate)

    @staticmethod
    def backward(ctx, dmsg):
        h, A, gate = ctx.saved_tensors
        g = ctx.graph
        need_dx = ctx.needs_input_grad[0] or (gate is not None and ctx.needs_input_grad[2])
        dx, dA = edge_message_bwd_raw(h, A, g, gate, dmsg.contiguous(), need_dx=need_dx,
                                      need_dA=ctx.needs_input_grad[1])    # dx = A^T dmsg per edge
        if not need_dx:      # e.g. BasicModel: the message input is the constant afm
            return None, dA, None, None
        t_row_ptr, t_eid = g.transpose
        dgate = None
        if gate is not None:
            src = g.col_idx.to(torch.int64)
            if ctx.needs_input_grad[2]:
                dgate = dx * h[src]
            if ctx.needs_input_grad[0]:
                dx = dx * gate
        dh = segsum_gather_raw(dx, t_row_ptr, t_eid, None, g.num_nodes) if ctx.needs_input_grad[0] else None
        return dh, (dA if ctx.needs_input_grad[1] else None), dgate, None


def _message_aggregate_backward(h, A, gate, w, g, dout, needs, dA_accum=None):
    """(dh, dA, dgate) of MessageAggregate for one incoming gradient; `needs` = needs_input_grad of (h, A, gate); `dA_accum`: a
    zero-initialised (or partly summed) buffer the weight gradient is added to."""
    dout = dout.contiguous()
    need_dx = needs[0] or (gate is not None and needs[2])
    K, mf, nf = (int(s) for s in A.shape)
    fused_widths = (64, 128) if os.environ.get("MPNN_GRU_MATH") == "fp32" else (64, 128, 256)   # no fp32 twin at 256
    if not need_dx and mf == nf and mf in fused_widths and K <= 64:
        if not needs[1]:
            return None, None, None
        lib = _lib.load()
        dA = dA_accum if dA_accum is not None else torch.zeros_like(A)
        if g.num_edges:
            _lib.check(_timed("message_aggregate_bwd", lambda: lib.mpnn_edge_message_agg_bwd_da_f32(
                _lib.fptr(dout), _lib.fptr(h), _lib.iptr(g.col_idx), _lib.iptr(g.edge_dst), _lib.fptr(w),
                _lib.iptr(g.order), _lib.iptr(g.type_ptr), _lib.fptr(gate), _lib.fptr(dA), g.num_nodes,
                g.num_edges, K, nf, mf, _lib.stream())), "mpnn_edge_message_agg_bwd_da_f32")
        return None, dA, None
    if (gate is not None and needs[2] and not needs[0] and mf == nf
            and mf in (64, 128) and K <= 64 and os.environ.get("MPNN_GRU_MATH") != "fp32"):
        # attention models: the node features feeding the message are constants, only the gate (and A) want
        # gradients -- both come straight from dout[dst(e)]; d(msg) and dx are never written
        lib = _lib.load()
        dA = (dA_accum if dA_accum is not None else torch.zeros_like(A)) if needs[1] else None
        dgate = _empty((g.num_edges, nf), h)
        if g.num_edges:
            if dA is not None:
                _lib.check(lib.mpnn_edge_message_agg_bwd_da_f32(
                    _lib.fptr(dout), _lib.fptr(h), _lib.iptr(g.col_idx), _lib.iptr(g.edge_dst), _lib.fptr(w),
                    _lib.iptr(g.order), _lib.iptr(g.type_ptr), _lib.fptr(gate), _lib.fptr(dA), g.num_nodes,
                    g.num_edges, K, nf, mf, _lib.stream()), "mpnn_edge_message_agg_bwd_da_f32")
            _lib.check(lib.mpnn_edge_message_agg_bwd_dgate_f32(
                _lib.fptr(dout), _lib.fptr(A), _lib.fptr(h), _lib.iptr(g.col_idx), _lib.iptr(g.edge_dst), _lib.fptr(w),
                _lib.iptr(g.order), _lib.iptr(g.type_ptr), _lib.fptr(dgate), g.num_nodes, g.num_edges, K, nf, mf,
                _lib.stream()), "mpnn_edge_message_agg_bwd_dgate_f32")
        return None, dA, dgate
    dmsg = segsum_bwd_raw(dout, g.row_ptr, w, g.num_edges)
    dx, dA = edge_message_bwd_raw(h, A, g, gate, dmsg, need_dx=need_dx, need_dA=needs[1], dA_accum=dA_accum)
    if not need_dx:
        return None, dA, None
    t_row_ptr, t_eid = g.transpose
    dgate = None
    if gate is not None:
        if needs[2]:
            dgate = dx * h[g.col_idx.to(torch.int64)]
        if needs[0]:
            dx = dx * gate
    dh = segsum_gather_raw(dx, t_row_ptr, t_eid, None, g.num_nodes) if needs[0] else None
    return dh, dA, dgate


class MessageAggregate(torch.autograd.Function):
    """out[i] = sum_{e in row i} w[e] * (A[type e] . (gate[e] * h[src e])) -- the edge message followed by
    the adjacency-weighted sum as ONE autograd node.  Forward launches the same two kernels as the
    separate ops (message, then the segmented-sum aggregator); the node exists for the backward: when no
    gradient is wanted for h / gate, dA is accumulated straight from dout[dst(e)] and the (E, mf) message
    gradient is never written or read."""

    @staticmethod
    def forward(ctx, h, A, gate, w, graph):
        h, A = h.contiguous(), A.contiguous()
        gate = gate.contiguous() if gate is not None else None
        ctx.graph = graph
        ctx.save_for_backward(h, A, gate, w)
        if tile_kernel_applies(A, gate, w, graph):
            return message_aggregate_tile_raw(h, A, graph)
        if wide_kernel_applies(A, gate, w, graph):
            return message_aggregate_wide_raw(h, A, graph)
        msg = edge_message_raw(h, A, graph, gate)
        return segsum_raw(msg, graph.row_ptr, w, graph.num_nodes)

    @staticmethod
    def backward(ctx, dout):
        h, A, gate, w = ctx.saved_tensors
        return _message_aggregate_backward(h, A, gate, w, ctx.graph, dout, ctx.needs_input_grad) + (None, None)


def message_aggregate(h, A, graph, w=None, gate=None):
    return MessageAggregate.apply(h, A, gate, w, graph)


class MessageAggregateSteps(torch.autograd.Function):
    """T evaluations of MessageAggregate on the SAME (h, A) -- BasicModel computes its message from the constant atom features
    at every step (basic_model.py:57) -- as one node: T forward launches, T weight-gradient launches, but the T gradients
    are added inside the kernels into ONE zero-filled buffer instead of T buffers that autograd sums.  h carries no gradient."""

    @staticmethod
    def forward(ctx, h, A, w, graph, steps):
        h, A = h.contiguous(), A.contiguous()
        ctx.graph, ctx.steps = graph, steps
        ctx.save_for_backward(h, A, w)
        outs = []
        for _ in range(steps):
            if tile_kernel_applies(A, None, w, graph):
                outs.append(message_aggregate_tile_raw(h, A, graph))
            elif wide_kernel_applies(A, None, w, graph):
                outs.append(message_aggregate_wide_raw(h, A, graph))
            else:
                outs.append(segsum_raw(edge_message_raw(h, A, graph, None), graph.row_ptr, w, graph.num_nodes))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        h, A, w = ctx.saved_tensors
        if not ctx.needs_input_grad[1]:
            return None, None, None, None, None
        dA = torch.zeros_like(A)
        for dout in douts:
            _message_aggregate_backward(h, A, None, w, ctx.graph, dout, (False, True, False), dA_accum=dA)
        return None, dA, None, None, None


def message_aggregate_steps(h, A, graph, w, steps):
    """-> list of `steps` tensors, each out[i] = sum_{e in row i} w[e] * A[type e] . h[src e]; h must not require a gradient."""
    if h.requires_grad:
        return [message_aggregate(h, A, graph, w) for _ in range(steps)]
    return list(MessageAggregateSteps.apply(h, A, w, graph, steps))


BN_MASKED_MEAN, BN_EPS_INSIDE, BN_USE_STATS = 1, 2, 4


class MaskedBatchNorm(torch.autograd.Function):
    """y, batch_mean, batch_var = masked batch norm of x (V,F); see include/mpnn_amd.h.  The statistics
    outputs carry no gradient (they feed the running estimates)."""

    @staticmethod
    def forward(ctx, x, mask, weight, bias, stats_mean, stats_var, eps, flags):
        lib = _lib.load()
        x = x.contiguous()
        V, F = int(x.shape[0]), int(x.shape[1])
        mask = mask.contiguous() if mask is not None else None
        y = _empty((V, F), x)
        if flags & BN_USE_STATS:
            mean, var = stats_mean.contiguous().clone(), stats_var.contiguous().clone()
        else:
            mean, var = _empty((F,), x), _empty((F,), x)
        count = _empty((1,), x)
        ws_bytes = lib.mpnn_masked_bn_workspace_bytes(F)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
        w = weight.contiguous() if weight is not None else None
        b = bias.contiguous() if bias is not None else None
        _lib.check(lib.mpnn_masked_bn_fwd_f32(_lib.fptr(x), _lib.fptr(mask), _lib.fptr(w), _lib.fptr(b), _lib.fptr(y),
                                              _lib.fptr(mean), _lib.fptr(var), _lib.fptr(count), V, F, float(eps),
                                              int(flags), _lib.ptr(ws), ws_bytes, _lib.stream()),
                   "mpnn_masked_bn_fwd_f32")
        ctx.save_for_backward(x, mask, w, mean, var, count)
        ctx.eps, ctx.flags = float(eps), int(flags)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dmean, _dvar):
        lib = _lib.load()
        x, mask, w, mean, var, count = ctx.saved_tensors
        V, F = int(x.shape[0]), int(x.shape[1])
        dy = dy.contiguous()
        if ctx.flags & BN_USE_STATS:                   # eval mode: a fixed per-column scale
            scale = 1.0 / (torch.sqrt(var + ctx.eps) if ctx.flags & BN_EPS_INSIDE else torch.sqrt(var) + ctx.eps)
            g = dy * (mask.unsqueeze(-1) if mask is not None else 1.0)
            xhat = (x - mean) * scale
            dx = g * scale * (w if w is not None else 1.0)
            return (dx, None, (g * xhat).sum(0) if w is not None else None, g.sum(0) if w is not None else None,
                    None, None, None, None)
        dx = _empty((V, F), x)
        dweight = _empty((F,), x) if w is not None else None
        dbias = _empty((F,), x) if w is not None else None
        ws_bytes = lib.mpnn_masked_bn_workspace_bytes(F)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
        _lib.check(lib.mpnn_masked_bn_bwd_f32(_lib.fptr(dy), _lib.fptr(x), _lib.fptr(mask), _lib.fptr(w), _lib.fptr(mean),
                                              _lib.fptr(var), _lib.fptr(dx), _lib.fptr(dweight), _lib.fptr(dbias), V, F,
                                              ctx.eps, ctx.flags, _lib.fptr(count), _lib.ptr(ws), ws_bytes,
                                              _lib.stream()), "mpnn_masked_bn_bwd_f32")
        return dx, None, dweight, dbias, None, None, None, None


def masked_batch_norm(x, mask, weight=None, bias=None, stats=None, eps=1e-5, flags=BN_MASKED_MEAN):
    """x (V,F), mask (V,) -> (y, batch_mean, batch_var); `stats` = (mean, var) with BN_USE_STATS."""
    sm, sv = stats if stats is not None else (None, None)
    return MaskedBatchNorm.apply(x, mask, weight, bias, sm, sv, eps, flags)


class GRUUpdateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, h, mask, W_ih, W_hh, b_ih, b_hh, grad_mode):
        m, h = m.contiguous(), h.contiguous()
        mask = mask.contiguous() if mask is not None else None
        # `grad_mode` = torch.is_grad_enabled() of the CALLER (always False in here): no gate dump
        # (4H floats per atom) on inference passes
        need = grad_mode and any(ctx.needs_input_grad)
        out, saved = gru_update_raw(m, h, mask, W_ih.contiguous(), W_hh.contiguous(), b_ih.contiguous(),
                                    b_hh.contiguous(), need)
        if need:
            ctx.save_for_backward(m, h, mask, W_ih, W_hh, saved)
        return out

    @staticmethod
    def backward(ctx, dout):
        m, h, mask, W_ih, W_hh, saved = ctx.saved_tensors
        dm, dh, dW_ih, dW_hh, db_ih, db_hh = gru_update_bwd_raw(dout.contiguous(), m, h, mask,
                                                                 W_ih.contiguous(), W_hh.contiguous(), saved)
        return dm, dh, None, dW_ih, dW_hh, db_ih, db_hh, None


# ---- masked batch norm fused into the update (models/att_model.py:58, lipo_basic_model.py:85) ----
def gru_norm_kind(H, tensor):
    """0: no fused update + norm kernel for width H on this tensor's device; 2: the wide split-precision kernels
    (H = 128 / 256); 1: the generic fp32 kernel (what runs the plain update at the remaining widths anyway)."""
    if not (tensor.is_cuda and tensor.dtype == torch.float32):
        return 0
    return int(_lib.load().mpnn_gru_update_norm_supported(int(H)))


def gru_norm_applies(H, tensor):
    """True when the fast fused update + norm kernels (hidden 128 / 256) cover width H on this tensor's device."""
    return gru_norm_kind(H, tensor) == 2


def gru_norm_costs_nothing(H, tensor):
    """True when fusing the norm into the update cannot slow the update down: the wide kernels, or a width whose plain
    update runs on the generic kernel as well (not 32 / 64 / 128, which have faster un-normed forms)."""
    k = gru_norm_kind(H, tensor)
    return k == 2 or (k == 1 and int(H) not in (32, 64, 128, 256))


class OutputMoments:
    """Column sums of an update's output and of its squares over all atoms (2H doubles, from the update kernel's
    epilogue) with the masked atom count: the batch statistics the norm after that update needs."""

    def __init__(self, sums, count):
        self.sums, self.count = sums, count               # (2H,) float64, (1,) float32 = mask.sum()
        self._mv = None

    def mean_var(self):
        if self._mv is None:
            H = self.sums.shape[0] // 2
            n = self.count.double()
            mean = self.sums[:H] / n
            var = (self.sums[H:] / n - mean * mean).clamp_min_(0.0)      # biased, as mask_batch_norm.py:14,31
            self._mv = (mean.float(), var.float())
        return self._mv


def _norm_scale(var, eps, flags):
    return torch.sqrt(var + eps) if flags & BN_EPS_INSIDE else torch.sqrt(var) + eps


class GRUUpdateNormIn(torch.autograd.Function):
    """y, sums = update(m, norm(y_prev)) with the norm applied inside the update kernel and the moments of y taken
    in its epilogue.  `y_prev` is the previous update's raw output, (mean, var) its batch statistics (None: `y_prev`
    enters as it is -- the first step).  Backward = the GRU backward kernels on the normalised state the forward
    kernel saved, then the masked-norm backward kernels (reduction + apply) for y_prev."""

    @staticmethod
    def forward(ctx, m, y_prev, mask, W_ih, W_hh, b_ih, b_hh, weight, bias, mean, var, count, eps, flags, grad_mode):
        lib = _lib.load()
        m, y_prev = m.contiguous(), y_prev.contiguous()
        mask = mask.contiguous() if mask is not None else None
        V, H = int(y_prev.shape[0]), int(y_prev.shape[1])
        W_ih, W_hh, b_ih, b_hh = W_ih.contiguous(), W_hh.contiguous(), b_ih.contiguous(), b_hh.contiguous()
        normed = mean is not None
        if normed:
            inv = 1.0 / _norm_scale(var, eps, flags)
            hs = inv * weight if weight is not None else inv
            ht = -mean * hs + (bias if bias is not None else 0.0)
            Wf = (W_hh * hs.unsqueeze(1)).contiguous()                    # hn W_hh + b_hh = y (diag(hs) W_hh) + (ht W_hh + b_hh)
            bf = torch.addmv(b_hh, W_hh.t(), ht)
        else:
            hs, ht = torch.ones(H, dtype=torch.float32, device=m.device), torch.zeros(H, dtype=torch.float32, device=m.device)
            Wf, bf = W_hh, b_hh
        need = grad_mode and any(ctx.needs_input_grad)
        out = _empty((V, H), y_prev)
        saved = _empty((V, 4 * H), y_prev) if need else None
        hn = _empty((V, H), y_prev) if need else None
        sums = torch.zeros(2 * H, dtype=torch.float64, device=m.device)
        ws_bytes = lib.mpnn_gru_fwd_workspace_bytes(V, H)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=m.device)
        _lib.check(_timed("gru_update", lambda: lib.mpnn_gru_update_norm_f32(
            _lib.fptr(m), _lib.fptr(y_prev), _lib.fptr(mask), _lib.fptr(W_ih), _lib.fptr(Wf), _lib.fptr(b_ih),
            _lib.fptr(bf), _lib.fptr(hs), _lib.fptr(ht), _lib.fptr(out), _lib.fptr(saved), _lib.fptr(hn),
            _lib.ptr(sums), _lib.ptr(ws), ws_bytes, V, H, _lib.stream())), "mpnn_gru_update_norm_f32")
        if need:
            ctx.save_for_backward(m, hn, y_prev if normed else None, mask, W_ih, W_hh, saved, weight, mean, var, count)
        ctx.normed, ctx.eps, ctx.flags = normed, float(eps), int(flags)
        ctx.mark_non_differentiable(sums)
        return out, sums

    @staticmethod
    def backward(ctx, dout, _dsums):
        lib = _lib.load()
        m, hn, y_prev, mask, W_ih, W_hh, saved, weight, mean, var, count = ctx.saved_tensors
        dm, dhn, dW_ih, dW_hh, db_ih, db_hh = gru_update_bwd_raw(dout.contiguous(), m, hn, mask, W_ih, W_hh, saved)
        dweight = dbias = None
        if ctx.normed:
            V, F = int(hn.shape[0]), int(hn.shape[1])
            dy = _empty((V, F), hn)
            if weight is not None:
                dweight, dbias = _empty((F,), hn), _empty((F,), hn)
            ws_bytes = lib.mpnn_masked_bn_workspace_bytes(F)
            ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=hn.device)
            _lib.check(lib.mpnn_masked_bn_bwd_f32(_lib.fptr(dhn), _lib.fptr(y_prev), _lib.fptr(mask), _lib.fptr(weight),
                                                  _lib.fptr(mean), _lib.fptr(var), _lib.fptr(dy), _lib.fptr(dweight),
                                                  _lib.fptr(dbias), V, F, ctx.eps, ctx.flags, _lib.fptr(count),
                                                  _lib.ptr(ws), ws_bytes, _lib.stream()), "mpnn_masked_bn_bwd_f32")
        else:
            dy = dhn
        return dm, dy, None, dW_ih, dW_hh, db_ih, db_hh, dweight, dbias, None, None, None, None, None, None


def gru_update_norm_in(m, y_prev, mask, W_ih, W_hh, b_ih, b_hh, moments=None, weight=None, bias=None, eps=1e-6,
                       flags=BN_EPS_INSIDE, count=None):
    """-> (y, OutputMoments of y).  `moments`: OutputMoments of y_prev when y_prev is to be normalised on the way in."""
    if count is None:
        count = moments.count if moments is not None else (mask.sum().reshape(1) if mask is not None else
                                                            torch.full((1,), float(y_prev.shape[0]), device=m.device))
    mean, var = moments.mean_var() if moments is not None else (None, None)
    y, sums = GRUUpdateNormIn.apply(m, y_prev, mask, W_ih, W_hh, b_ih, b_hh, weight, bias, mean, var, count, eps, flags,
                                    torch.is_grad_enabled())
    return y, OutputMoments(sums, count)


def _norm_fold(sums, count, weight, bias, W_hh, b_hh, eps, flags, given=None):
    """moments of an update's output (or, with BN_USE_STATS, the `given` (mean, var)) -> (mean, var, hs, ht, W_hh with the
    norm folded in, b_hh likewise): one launch."""
    lib = _lib.load()
    H = int(W_hh.shape[0])
    dev = W_hh.device
    mean, var, hs, ht = (torch.empty(H, dtype=torch.float32, device=dev) for _ in range(4))
    if given is not None:
        mean, var = given[0].contiguous().float(), given[1].contiguous().float()
    Wf, bf = torch.empty_like(W_hh), torch.empty_like(b_hh)
    _lib.check(lib.mpnn_norm_fold_f32(_lib.ptr(sums), _lib.fptr(count), _lib.fptr(weight), _lib.fptr(bias), _lib.fptr(W_hh),
                                      _lib.fptr(b_hh), _lib.fptr(mean), _lib.fptr(var), _lib.fptr(hs), _lib.fptr(ht),
                                      _lib.fptr(Wf), _lib.fptr(bf), H, float(eps), int(flags), _lib.stream()),
               "mpnn_norm_fold_f32")
    return mean, var, hs, ht, Wf, bf


class GRUNormChain(torch.autograd.Function):
    """state_t = norm(update(m_t, state_{t-1})), t = 1..T, state_0 = h0  (models/att_model.py:57-58 with the messages
    m_t given: there they depend on the atom features only).  No norm runs as passes of its own except the apply pass
    of the LAST one (its output is what the caller reads) and one two-sum reduction for that norm's backward:
      forward   update t takes the raw output of update t-1 with the norm folded in (mpnn_gru_update_norm_f32) and emits
                the moments of its own output;
      backward  the dm | dh kernel of update t emits the column sums the backward of the norm in front of it needs, and
                the gate-gradient kernel of update t-1 applies that backward to its incoming gradient in registers
                (mpnn_gru_update_norm_bwd_f32).
    sync=True: the statistics span all ranks of the default process group (the batch is sharded by graph): the
    2H moments, the count and the 2H backward sums are all-reduced -- five tiny collectives per step, no extra pass."""

    @staticmethod
    def forward(ctx, h0, mask, W_ih, W_hh, b_ih, b_hh, weight, bias, eps, flags, sync, grad_mode, given_mean, given_var,
                *msgs):
        lib = _lib.load()
        y = h0.contiguous()
        V, H = int(y.shape[0]), int(y.shape[1])
        dev = y.device
        mask = mask.contiguous() if mask is not None else None
        W_ih, W_hh, b_ih, b_hh = W_ih.contiguous(), W_hh.contiguous(), b_ih.contiguous(), b_hh.contiguous()
        weight = weight.contiguous() if weight is not None else None
        bias = bias.contiguous() if bias is not None else None
        count = (mask.sum() if mask is not None else torch.tensor(float(V), device=dev)).reshape(1).float()
        sync = bool(sync) and _dist_world() > 1
        if sync:
            _all_reduce(count)
        need = grad_mode and any(ctx.needs_input_grad)
        hs = torch.ones(H, dtype=torch.float32, device=dev)
        ht = torch.zeros(H, dtype=torch.float32, device=dev)
        Wf, bf = W_hh, b_hh
        ws_bytes = lib.mpnn_gru_fwd_workspace_bytes(V, H)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
        msgs = [m.contiguous() for m in msgs]
        hns, saveds, stats, raws = [], [], [], []
        mean = var = None
        for m in msgs:
            raws.append(y)                                # the raw state this update's norm took (h0 for the first)
            out = _empty((V, H), y)
            saved = _empty((V, 4 * H), y) if need else None
            hn = _empty((V, H), y) if need else None
            sums = torch.zeros(2 * H, dtype=torch.float64, device=dev)
            _lib.check(_timed("gru_update", lambda: lib.mpnn_gru_update_norm_f32(
                _lib.fptr(m), _lib.fptr(y), _lib.fptr(mask), _lib.fptr(W_ih), _lib.fptr(Wf), _lib.fptr(b_ih), _lib.fptr(bf),
                _lib.fptr(hs), _lib.fptr(ht), _lib.fptr(out), _lib.fptr(saved), _lib.fptr(hn), _lib.ptr(sums),
                _lib.ptr(ws), ws_bytes, V, H, _lib.stream())), "mpnn_gru_update_norm_f32")
            if sync:
                _all_reduce(sums)
            mean, var, hs, ht, Wf, bf = _norm_fold(sums, count, weight, bias, W_hh, b_hh, eps, flags,
                                                   None if given_mean is None else (given_mean, given_var))
            hns.append(hn)
            saveds.append(saved)
            stats.append((mean, var))
            y = out
        if not msgs:
            return (y,)
        final = _empty((V, H), y)
        bws_bytes = lib.mpnn_masked_bn_workspace_bytes(H)
        bws = torch.empty(bws_bytes // 4, dtype=torch.float32, device=dev)
        _lib.check(lib.mpnn_masked_bn_fwd_f32(_lib.fptr(y), _lib.fptr(mask), _lib.fptr(weight), _lib.fptr(bias),
                                              _lib.fptr(final), _lib.fptr(mean), _lib.fptr(var), None, V, H, float(eps),
                                              int(flags) | BN_USE_STATS, _lib.ptr(bws), bws_bytes, _lib.stream()),
                   "mpnn_masked_bn_fwd_f32")
        if need:
            # y = the last update's raw output: the last norm's backward sums run against it (not against `final`, the
            # norm's affine output, from which a zero weight entry could not be undone)
            ctx.save_for_backward(y, mask, W_ih, W_hh, weight, bias, count, *msgs, *hns, *saveds, *raws[1:],
                                  *[t for mv in stats for t in mv])
        ctx.T, ctx.eps, ctx.flags, ctx.sync = len(msgs), float(eps), int(flags), sync
        # the per-step batch statistics ride along as non-differentiable outputs (running estimates of MaskBatchNorm1d)
        flat = [t.clone() for mv in stats for t in mv]
        ctx.mark_non_differentiable(*flat)
        return (final,) + tuple(flat)

    @staticmethod
    def backward(ctx, dfinal, *_dstats):
        lib = _lib.load()
        T = ctx.T
        sv = ctx.saved_tensors
        y_last, mask, W_ih, W_hh, weight, bias, count = sv[:7]
        msgs, hns, saveds = sv[7:7 + T], sv[7 + T:7 + 2 * T], sv[7 + 2 * T:7 + 3 * T]
        raws = (None,) + tuple(sv[7 + 3 * T:7 + 4 * T - 1])   # raws[t] = raw input state of update t (t >= 1)
        stats = sv[7 + 4 * T - 1:]
        V, H = int(y_last.shape[0]), int(y_last.shape[1])
        dev = y_last.device
        affine = weight is not None
        dweight = torch.zeros(H, dtype=torch.float32, device=dev) if affine else None
        dbias = torch.zeros(H, dtype=torch.float32, device=dev) if affine else None

        def consts(sums, t):
            """out_norm_k of the norm after update t (0-based) from its two backward sums; parameter gradients from the
            LOCAL sums (the caller's gradient all-reduce adds the ranks)."""
            kn = torch.empty(3 * H, dtype=torch.float32, device=dev)
            mean, var = stats[2 * t], stats[2 * t + 1]

            def run(sm, dw, db, out):
                _lib.check(lib.mpnn_norm_bwd_consts_f32(_lib.ptr(sm), _lib.fptr(mean), _lib.fptr(var), _lib.fptr(count),
                                                        _lib.fptr(weight), _lib.fptr(out), _lib.fptr(dw),
                                                        _lib.fptr(db), H, ctx.eps, ctx.flags, _lib.stream()),
                           "mpnn_norm_bwd_consts_f32")
            if not ctx.sync:
                run(sums, dweight, dbias, kn)
                return kn
            if affine:                                  # parameter gradients: this rank's terms only
                run(sums, dweight, dbias, torch.empty_like(kn))
            _all_reduce(sums)
            scratch = (torch.zeros(H, dtype=torch.float32, device=dev),) * 2 if affine else (None, None)
            run(sums, scratch[0], scratch[1], kn)
            return kn

        # the last norm's output left this function: its two sums take one pass over (dfinal, the norm's raw input); its
        # apply happens in the gate-gradient kernel of the last update like every other norm's
        dfinal = dfinal.contiguous()
        lsums = torch.zeros(2 * H, dtype=torch.float64, device=dev)
        _lib.check(lib.mpnn_norm_bwd_sums_f32(_lib.fptr(dfinal), _lib.fptr(y_last), _lib.fptr(mask), _lib.ptr(lsums), V, H,
                                              _lib.stream()), "mpnn_norm_bwd_sums_f32")
        kn = consts(lsums, T - 1)
        dW_ih, dW_hh = torch.zeros_like(W_ih), torch.zeros_like(W_hh)
        db_ih = torch.zeros(3 * H, dtype=torch.float32, device=dev)
        db_hh = torch.zeros(3 * H, dtype=torch.float32, device=dev)
        ws_bytes = lib.mpnn_gru_norm_bwd_workspace_bytes(V, H)
        ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=dev)
        dms = [None] * T
        dout = dfinal
        for t in range(T - 1, -1, -1):
            dm, dhn = _empty((V, H), y_last), _empty((V, H), y_last)
            sums = torch.zeros(2 * H, dtype=torch.float64, device=dev) if t > 0 else None
            _lib.check(_timed("gru_update_bwd", lambda: lib.mpnn_gru_update_norm_bwd_f32(
                _lib.fptr(dout), _lib.fptr(msgs[t]), _lib.fptr(hns[t]), _lib.fptr(mask), _lib.fptr(W_ih), _lib.fptr(W_hh),
                _lib.fptr(saveds[t]), _lib.fptr(kn), _lib.fptr(dm), _lib.fptr(dhn), _lib.fptr(dW_ih), _lib.fptr(dW_hh),
                _lib.fptr(db_ih), _lib.fptr(db_hh), _lib.ptr(sums), _lib.fptr(raws[t]), _lib.ptr(ws), ws_bytes, V, H,
                _lib.stream())),
                "mpnn_gru_update_norm_bwd_f32")
            dms[t] = dm
            if t > 0:                                   # constants of the norm between update t-1 and update t
                kn = consts(sums, t - 1)
            dout = dhn
        return (dout, None, dW_ih, dW_hh, db_ih, db_hh, dweight, dbias, None, None, None, None, None, None) + tuple(dms)


def _dist_world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _all_reduce(t):
    import torch.distributed as dist
    if dist.get_backend() == "gloo" and t.is_cuda:      # rehearsal on one GPU: gloo reduces host tensors
        c = t.cpu()
        dist.all_reduce(c)
        t.copy_(c)
    else:
        dist.all_reduce(t)


def gru_norm_chain(h0, msgs, mask, W_ih, W_hh, b_ih, b_hh, weight=None, bias=None, eps=1e-6, flags=BN_EPS_INSIDE,
                   sync=False, given=None, return_stats=False):
    """`given` = (mean, var): every norm of the chain uses these statistics instead of the batch's (flags gets
    BN_USE_STATS: MaskBatchNorm1d in eval mode).  return_stats: also the list of per-step (mean, var)."""
    if given is not None:
        flags = int(flags) | BN_USE_STATS
    res = GRUNormChain.apply(h0, mask, W_ih, W_hh, b_ih, b_hh, weight, bias, eps, flags, sync, torch.is_grad_enabled(),
                             None if given is None else given[0], None if given is None else given[1], *msgs)
    out, flat = res[0], res[1:]
    if return_stats:
        return out, [(flat[2 * i], flat[2 * i + 1]) for i in range(len(flat) // 2)]
    return out


class MaskedBatchNormGiven(torch.autograd.Function):
    """Masked batch norm whose batch statistics were taken elsewhere (the producing update's epilogue): the forward is
    the apply pass alone, the backward the full batch-statistics backward (the statistics depend on x)."""

    @staticmethod
    def forward(ctx, x, mask, weight, bias, mean, var, count, eps, flags):
        lib = _lib.load()
        x = x.contiguous()
        V, F = int(x.shape[0]), int(x.shape[1])
        y = _empty((V, F), x)
        mean_c, var_c = mean.contiguous().clone(), var.contiguous().clone()
        ws_bytes = lib.mpnn_masked_bn_workspace_bytes(F)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
        _lib.check(lib.mpnn_masked_bn_fwd_f32(_lib.fptr(x), _lib.fptr(mask), _lib.fptr(weight), _lib.fptr(bias), _lib.fptr(y),
                                              _lib.fptr(mean_c), _lib.fptr(var_c), None, V, F, float(eps),
                                              int(flags) | BN_USE_STATS, _lib.ptr(ws), ws_bytes, _lib.stream()),
                   "mpnn_masked_bn_fwd_f32")
        ctx.save_for_backward(x, mask, weight, mean_c, var_c, count)
        ctx.eps, ctx.flags = float(eps), int(flags)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, mask, w, mean, var, count = ctx.saved_tensors
        V, F = int(x.shape[0]), int(x.shape[1])
        dx = _empty((V, F), x)
        dweight = _empty((F,), x) if w is not None else None
        dbias = _empty((F,), x) if w is not None else None
        ws_bytes = lib.mpnn_masked_bn_workspace_bytes(F)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
        _lib.check(lib.mpnn_masked_bn_bwd_f32(_lib.fptr(dy.contiguous()), _lib.fptr(x), _lib.fptr(mask), _lib.fptr(w),
                                              _lib.fptr(mean), _lib.fptr(var), _lib.fptr(dx), _lib.fptr(dweight),
                                              _lib.fptr(dbias), V, F, ctx.eps, ctx.flags, _lib.fptr(count), _lib.ptr(ws),
                                              ws_bytes, _lib.stream()), "mpnn_masked_bn_bwd_f32")
        return dx, None, dweight, dbias, None, None, None, None, None


def masked_batch_norm_given(x, mask, moments, weight=None, bias=None, eps=1e-6, flags=BN_EPS_INSIDE):
    mean, var = moments.mean_var()
    return MaskedBatchNormGiven.apply(x, mask, weight, bias, mean, var, moments.count, eps, flags)


def segsum(msg, row_ptr, w=None, num_rows=None):
    return SegSum.apply(msg, row_ptr, w, int(row_ptr.shape[0]) - 1 if num_rows is None else num_rows)


def neighbour_sum(x, graph, weighted=False):
    """sum over incoming edges of x[src]  (V,F) -> (V,F); differentiable through the transposed graph."""
    t_row_ptr, t_eid = graph.transpose
    w = graph.edge_weight if weighted else None
    t_w = w[t_eid.to(torch.int64)].contiguous() if w is not None else None
    t_dst = graph.edge_dst[t_eid.to(torch.int64)].contiguous()
    return SegSumGather.apply(x, graph.row_ptr, graph.col_idx, w, graph.num_nodes, t_row_ptr, t_dst, t_w)


def molecule_sum(x, graph):
    """Per-molecule sum of atom rows (V,F) -> (G,F).  Backward is a row broadcast."""
    return _MoleculeSum.apply(x, graph)


class _MoleculeSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, graph):
        ctx.graph = graph
        return segsum_raw(x.contiguous(), graph.graph_ptr, None, graph.num_graphs, label="molecule_sum")

    @staticmethod
    def backward(ctx, dout):
        g = ctx.graph
        return segsum_bwd_raw(dout.contiguous(), g.graph_ptr, None, g.num_nodes), None


class _MoleculeBroadcast(torch.autograd.Function):
    """(G,F) -> (V,F): every atom receives its molecule's row; backward is the per-molecule sum."""

    @staticmethod
    def forward(ctx, x, graph):
        ctx.graph = graph
        return segsum_bwd_raw(x.contiguous(), graph.graph_ptr, None, graph.num_nodes)

    @staticmethod
    def backward(ctx, dout):
        g = ctx.graph
        return segsum_raw(dout.contiguous(), g.graph_ptr, None, g.num_graphs, label="molecule_sum"), None


def molecule_broadcast(x, graph):
    return _MoleculeBroadcast.apply(x, graph)


class _ExpandRows(torch.autograd.Function):
    """x (V,F) -> (E,F): every edge receives the row of its DESTINATION atom.  Forward is the
    aggregator's backward kernel (row broadcast), backward is the aggregator itself -- unlike
    `x[dst]`, whose autograd falls back to a sort-based scatter that takes seconds at 6 M edges."""

    @staticmethod
    def forward(ctx, x, graph):
        ctx.graph = graph
        return segsum_bwd_raw(x.contiguous(), graph.row_ptr, None, graph.num_edges)

    @staticmethod
    def backward(ctx, dout):
        g = ctx.graph
        return segsum_raw(dout.contiguous(), g.row_ptr, None, g.num_nodes, label="row_sum_bwd"), None


def expand_rows(x, graph):
    return _ExpandRows.apply(x, graph)


# Column sums a backward kernel already has for a gradient tensor it hands to autograd: keyed by the tensor's address, valid
# only while that very tensor is alive and unmodified (a weak reference and its version counter say so).  The gated message
# backward knows sum_i dz_atom[i] = sum_k dq[k] for free; the Linear that made z_atom needs exactly that sum for its bias
# gradient -- a (V, F) reduction of 0.5 ms at c3's size, five times per step.
_KNOWN_COLSUMS = {}


def _offer_colsum(t, colsum):
    import weakref
    if len(_KNOWN_COLSUMS) > 64:
        _KNOWN_COLSUMS.clear()
    _KNOWN_COLSUMS[t.data_ptr()] = (weakref.ref(t), t._version, colsum)


def _take_colsum(t):
    hit = _KNOWN_COLSUMS.pop(t.data_ptr(), None)
    if hit is None:
        return None
    ref, version, colsum = hit
    live = ref()
    if live is None or live.data_ptr() != t.data_ptr() or live.shape != t.shape or t._version != version:
        return None
    return colsum


class _TallLinear(torch.autograd.Function):
    """x (V,F) @ W^T + b for V in the millions.  Forward is the library GEMM; the weight gradient dy^T x is a GEMM with
    a contraction of V rows and a tiny output, which the library runs on a handful of CUs (3.8 ms at V = 3 M,
    F = 128): it is cut into row chunks and issued as one batched GEMM + a sum instead."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        return torch.addmm(b, x, W.t())

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dx = dy @ W if ctx.needs_input_grad[0] else None
        dW = None
        if ctx.needs_input_grad[1]:
            V = int(x.shape[0])
            chunk = 8192
            nfull = V // chunk
            dW = torch.zeros_like(W)
            if nfull:
                dW = torch.bmm(dy[:nfull * chunk].view(nfull, chunk, -1).transpose(1, 2),
                               x[:nfull * chunk].view(nfull, chunk, -1)).sum(0)
            if V > nfull * chunk:
                dW = dW + dy[nfull * chunk:].t() @ x[nfull * chunk:]
        db = None
        if ctx.needs_input_grad[2]:
            db = _take_colsum(dy)
            if db is None:
                db = dy.sum(0)
        return dx, dW, db


def tall_linear(x, W, b):
    return _TallLinear.apply(x, W, b)


class AttGate(torch.autograd.Function):
    """gate[e] = softmax_f(z_atom[dst(e)] + q[type(e)])  (E,F); see include/mpnn_amd.h."""

    @staticmethod
    def forward(ctx, z_atom, q, graph):
        lib = _lib.load()
        z_atom, q = z_atom.contiguous(), q.contiguous()
        E, F = graph.num_edges, int(z_atom.shape[1])
        gate = _empty((E, F), z_atom)
        if E:
            _lib.check(lib.mpnn_att_gate_f32(_lib.fptr(z_atom), _lib.fptr(q), _lib.iptr(graph.edge_dst),
                                             _lib.iptr(graph.edge_type), _lib.fptr(gate), graph.num_nodes, E,
                                             int(q.shape[0]), F, _lib.stream()), "mpnn_att_gate_f32")
        ctx.save_for_backward(gate)
        ctx.graph, ctx.K = graph, int(q.shape[0])
        return gate

    @staticmethod
    def backward(ctx, dgate):
        lib = _lib.load()
        gate, = ctx.saved_tensors
        g = ctx.graph
        V, E, F = g.num_nodes, g.num_edges, int(gate.shape[1])
        dz_atom = torch.zeros((V, F), dtype=torch.float32, device=gate.device) if E == 0 else _empty((V, F), gate)
        dq = torch.zeros((ctx.K, F), dtype=torch.float32, device=gate.device)
        if V and E:
            _lib.check(lib.mpnn_att_gate_bwd_f32(_lib.fptr(gate), _lib.fptr(dgate.contiguous()), _lib.iptr(g.row_ptr),
                                                 _lib.iptr(g.edge_type), _lib.fptr(dz_atom), _lib.fptr(dq), V, E, ctx.K,
                                                 F, _lib.stream()), "mpnn_att_gate_bwd_f32")
        return dz_atom, dq, None


def att_gate(z_atom, q, graph):
    return AttGate.apply(z_atom, q, graph)


class LazyAttGate:
    """AttEdgeNetwork's feature gate, softmax_f(z_atom[dst e] + q[type e]), not yet evaluated: the sum aggregator forms it
    inside the fused message + sum kernel where that kernel applies (ops.gated_message_aggregate); every other consumer
    calls materialise() and gets the (E, F) tensor of ops.att_gate."""

    def __init__(self, z_atom, q, graph):
        self.z_atom, self.q, self.graph = z_atom, q, graph
        self._gate = None

    def materialise(self):
        if self._gate is None:
            self._gate = att_gate(self.z_atom, self.q, self.graph)
        return self._gate


# True: GatedMessageAggregate's backward on per-EDGE tensors (the gate re-evaluated into an (E, nf) tensor, then the kernels of
# the unfused path) -- what the fp32-only mode runs, and the reference the per-(atom, type) kernels are tested against
ATT_BWD_PER_EDGE = False


class GatedMessageAggregate(torch.autograd.Function):
    """out[i] = sum_{e in row i} A[type e] . (softmax_f(z_atom[i] + q[type e]) * h[src e]): AttEdgeNetwork followed by
    AdjMsgAgg (att_edge_network.py:18-31, adjacent_message_agg.py:18) as ONE kernel without an (E, F) gate tensor
    (mpnn_message_aggregate_wide_gated_f32).  Backward: per (atom, type) on the same plan
    (message_aggregate_wide_gated_bwd_raw); in fp32-only mode the gate is re-evaluated (mpnn_att_gate_f32) and the kernels of
    the unfused path -- weight gradient and gate gradient straight from dout[dst e], softmax backward -- run on it."""

    @staticmethod
    def forward(ctx, h, A, z_atom, q, graph):
        h, A, z_atom, q = h.contiguous(), A.contiguous(), z_atom.contiguous(), q.contiguous()
        ctx.graph = graph
        ctx.save_for_backward(h, A, z_atom, q)
        out, ctx.fwd_ws = message_aggregate_wide_gated_raw(h, A, z_atom, q, graph, keep_workspace=True)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        h, A, z_atom, q = ctx.saved_tensors
        g = ctx.graph
        K, mf, nf = (int(s) for s in A.shape)
        V, E = g.num_nodes, g.num_edges
        dout = dout.contiguous()
        if ctx.needs_input_grad[0] or E == 0:
            # gradient for the node features as well (not what the attention models ask for: their message input is the
            # constant afm): differentiate the unfused composition
            with torch.enable_grad():
                leaves = [t.detach().requires_grad_(need) for t, need in zip((h, A, z_atom, q), ctx.needs_input_grad[:4])]
                out = MessageAggregate.apply(leaves[0], leaves[1], att_gate(leaves[2], leaves[3], g), None, g)
            want = [t for t in leaves if t.requires_grad]
            got = iter(torch.autograd.grad(out, want, dout, allow_unused=True)) if want else iter(())
            return tuple(next(got) if t.requires_grad else None for t in leaves) + (None,)
        if (not ATT_BWD_PER_EDGE and ctx.fwd_ws is not None and os.environ.get("MPNN_GRU_MATH") != "fp32"
                and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3])):
            dA, dz_atom, dq = message_aggregate_wide_gated_bwd_raw(h, A, z_atom, q, dout, ctx.fwd_ws, g)
            if ctx.needs_input_grad[2]:
                _offer_colsum(dz_atom, dq.sum(0))              # sum_i dz_atom[i] = sum_k dq[k]
            return (None, dA if ctx.needs_input_grad[1] else None, dz_atom if ctx.needs_input_grad[2] else None,
                    dq if ctx.needs_input_grad[3] else None, None)
        gate = _empty((E, nf), h)
        _lib.check(lib.mpnn_att_gate_f32(_lib.fptr(z_atom), _lib.fptr(q), _lib.iptr(g.edge_dst), _lib.iptr(g.edge_type),
                                         _lib.fptr(gate), V, E, K, nf, _lib.stream()), "mpnn_att_gate_f32")
        dA = None
        if ctx.needs_input_grad[1]:
            dA = torch.zeros_like(A)
            _lib.check(_timed("message_aggregate_bwd", lambda: lib.mpnn_edge_message_agg_bwd_da_f32(
                _lib.fptr(dout), _lib.fptr(h), _lib.iptr(g.col_idx), _lib.iptr(g.edge_dst), None, _lib.iptr(g.order),
                _lib.iptr(g.type_ptr), _lib.fptr(gate), _lib.fptr(dA), V, E, K, nf, mf, _lib.stream())),
                "mpnn_edge_message_agg_bwd_da_f32")
        dz_atom = dq = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            dgate = _empty((E, nf), h)
            _lib.check(lib.mpnn_edge_message_agg_bwd_dgate_f32(
                _lib.fptr(dout), _lib.fptr(A), _lib.fptr(h), _lib.iptr(g.col_idx), _lib.iptr(g.edge_dst), None,
                _lib.iptr(g.order), _lib.iptr(g.type_ptr), _lib.fptr(dgate), V, E, K, nf, mf, _lib.stream()),
                "mpnn_edge_message_agg_bwd_dgate_f32")
            dz_atom = _empty((V, nf), h)
            dq = torch.zeros((K, nf), dtype=torch.float32, device=h.device)
            _lib.check(lib.mpnn_att_gate_bwd_f32(_lib.fptr(gate), _lib.fptr(dgate), _lib.iptr(g.row_ptr),
                                                 _lib.iptr(g.edge_type), _lib.fptr(dz_atom), _lib.fptr(dq), V, E, K, nf,
                                                 _lib.stream()), "mpnn_att_gate_bwd_f32")
        return None, dA, dz_atom, dq, None


def gated_message_aggregate(h, A, lazy_gate, graph, w=None):
    """AttEdgeNetwork messages summed over neighbours: the fused gated kernel where it applies, else gate + message_aggregate."""
    if w is None and wide_gated_applies(A, w, graph) and h.is_cuda:
        return GatedMessageAggregate.apply(h, A, lazy_gate.z_atom, lazy_gate.q, graph)
    return message_aggregate(h, A, graph, w, lazy_gate.materialise())


class TowerChain(torch.autograd.Function):
    """n applications of relu(x W^T) with ONE shared W (the tower's 50 aliased layers) in one kernel."""

    @staticmethod
    def forward(ctx, x, W, n):
        lib = _lib.load()
        x, W = x.contiguous(), W.contiguous()
        R, L = int(x.shape[0]), int(x.shape[1])
        acts = _empty((n + 1, R, L), x)
        _lib.check(lib.mpnn_tower_chain_f32(_lib.fptr(x), _lib.fptr(W), _lib.fptr(acts), R, L, n, _lib.stream()),
                   "mpnn_tower_chain_f32")
        ctx.save_for_backward(acts, W)
        ctx.n = n
        return acts[n]

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        acts, W = ctx.saved_tensors
        n = ctx.n
        R, L = int(acts.shape[1]), int(acts.shape[2])
        dys = _empty((n, R, L), acts)
        dx = _empty((R, L), acts)
        _lib.check(lib.mpnn_tower_chain_bwd_f32(_lib.fptr(dout.contiguous()), _lib.fptr(W), _lib.fptr(acts),
                                                _lib.fptr(dys), _lib.fptr(dx), R, L, n, _lib.stream()),
                   "mpnn_tower_chain_bwd_f32")
        dW = None
        if ctx.needs_input_grad[1]:
            if n * R <= 4096:
                # few rows (the distinct bond-feature rows of a batch): as ONE (L, n R) x (n R, L) product the library picks a
                # single 256 x 256 macro-tile -- one workgroup, 64 us for 33 MFLOP at L = 256; one small product per layer runs
                # on n workgroups at once, and the sum over the layers is a 13 MB read
                dW = torch.bmm(dys.transpose(1, 2), acts[:n]).sum(0)
            else:
                dW = dys.view(n * R, L).t() @ acts[:n].reshape(n * R, L)
        return dx, dW, None


def tower_chain(x, W, n):
    return TowerChain.apply(x, W, n)


def edge_message(h, A, graph, gate=None):
    return EdgeMessage.apply(h, A, gate, graph)


def gru_update(m, h, mask, W_ih, W_hh, b_ih, b_hh):
    return GRUUpdateFn.apply(m, h, mask, W_ih, W_hh, b_ih, b_hh, torch.is_grad_enabled())


class GRUChain(torch.autograd.Function):
    """h_T of h_{t+1} = update(msgs[t], h_t), t = 0 .. T-1, with ONE set of weights (basic_model.py:57-59: the T steps of a model
    call the same GRUUpdate).  The same kernels as T separate updates; as one node the backward adds the T weight gradients
    into one zero-filled buffer inside the kernels instead of handing autograd T tensors per parameter to sum."""

    @staticmethod
    def forward(ctx, h0, mask, W_ih, W_hh, b_ih, b_hh, grad_mode, *msgs):
        mask = mask.contiguous() if mask is not None else None
        Wc = [t.contiguous() for t in (W_ih, W_hh, b_ih, b_hh)]
        need = grad_mode and any(ctx.needs_input_grad)
        h = h0.contiguous()
        keep = []
        for m in msgs:
            m = m.contiguous()
            out, saved = gru_update_raw(m, h, mask, *Wc, need)
            keep += [m, h, saved]
            h = out
        if need:
            ctx.save_for_backward(mask, Wc[0], Wc[1], *keep)
        ctx.steps = len(msgs)
        return h

    @staticmethod
    def backward(ctx, dout):
        mask, W_ih, W_hh = ctx.saved_tensors[:3]
        keep = ctx.saved_tensors[3:]
        T = ctx.steps
        acc = gru_weight_grad_buffers(int(W_hh.shape[0]), dout)
        dh = dout.contiguous()
        dms = [None] * T
        for t in reversed(range(T)):
            m, h, saved = keep[3 * t:3 * t + 3]
            dms[t], dh, *_ = gru_update_bwd_raw(dh, m, h, mask, W_ih, W_hh, saved, accum=acc)
        return (dh, None, acc[0], acc[1], acc[2], acc[3], None, *dms)


def gru_chain(h0, msgs, mask, W_ih, W_hh, b_ih, b_hh):
    return GRUChain.apply(h0, mask, W_ih, W_hh, b_ih, b_hh, torch.is_grad_enabled(), *msgs)
