"""Sparse molecule batch resident in HBM (CSR by destination atom).

Layout (all device tensors, int32 indices, fp32 data):

    row_ptr[V+1]   edges of atom i are [row_ptr[i], row_ptr[i+1])      (destination-sorted)
    col_idx[E]     source atom of edge e
    edge_dst[E]    destination atom of edge e (expanded row_ptr; backward + gating)
    edge_weight[E] adjacency value of the pair (multiplier in AdjMsgAgg,
                   reference: mpnn_functions/message_aggregators/adjacent_message_agg.py:18)
    edge_type[E]   row of `type_feat` holding this edge's bond-feature vector
    type_feat[K,ef] distinct bond-feature rows (the tower is evaluated on K rows, not B*N*N)
    order[E], type_ptr[K+1]  edge ids stably sorted by type: the tile order of the message kernel
    graph_ptr[G+1] first atom of each molecule
    t_row_ptr[V+1], t_eid[E]  the same edges grouped by SOURCE atom (transposed graph, backward)

Two sources: a dense padded batch (the reference's wire format, atoms numbered b*N+i so every
(B,N,.) tensor is a free view of a (V,.) array), or a compact `synth.MolBatch`.
"""
import torch

from . import _lib


def _i32(t):
    return t.to(torch.int32).contiguous()

def _host_tiles(g, tv, lib):
    """Molecule-aligned tiles of at most `tv` atoms (greedy, whole molecules; mpnn_plan_tiles_host) -> (nt, host int32
    tile_ptr) or (0, None) when a molecule is larger than a tile.  Cached per tile size; the molecule boundaries are read
    from the host copy `from_molbatch` keeps, so no device -> host copy is needed for batches that came from the host."""
    import ctypes
    cache = g.__dict__.setdefault("_host_tiles_cache", {})
    if tv not in cache:
        gp = getattr(g, "_graph_ptr_host", None)
        if gp is None:
            gp = g.graph_ptr.to("cpu", torch.int32).contiguous()
            g._graph_ptr_host = gp
        tp = torch.empty(g.num_graphs + 2, dtype=torch.int32)
        nt = lib.mpnn_plan_tiles_host(ctypes.c_void_p(gp.data_ptr()), g.num_graphs, tv, ctypes.c_void_p(tp.data_ptr()))
        cache[tv] = (int(nt), tp[:nt + 1].contiguous()) if nt > 0 else (0, None)
    return cache[tv]


def _fast_path(g, K, kmax):
    """The purpose-made index kernels (csrc/plan.hip) serve device-resident batches with few bond types; everything else
    (CPU tensors: the tests' walk-throughs; continuous bond features) takes the torch builders below."""
    return g.device.type == "cuda" and 1 <= K <= kmax and g.num_nodes > 0 and g.num_graphs > 0 and g.num_edges > 0 and \
        g.edge_type.dtype == torch.int32


def _tile_layout(g, tv, lib):
    """What both tile plans start from: the batch's atoms cut into molecule-aligned tiles of at most `tv` atoms (greedy,
    whole molecules) and, inside every tile, sorted by their per-type in-degree pattern (rare types lead the key, high
    counts first).  None when the batch does not fit (no atoms, a molecule larger than a tile, an edge that leaves its
    tile).  -> (nt, tile_ptr, tp64, n_t, tile_of_atom, dst, src, et, cnt, pos_in_tile)"""
    K, E, V = g.num_types, g.num_edges, g.num_nodes
    if V == 0 or g.num_graphs == 0:
        return None
    nt, tp = _host_tiles(g, tv, lib)
    if nt <= 0:
        return None                                       # a molecule larger than a tile
    dev = g.device
    tile_ptr = tp.to(dev)
    tp64 = tile_ptr.to(torch.int64)
    n_t = tp64[1:] - tp64[:-1]
    tile_of_atom = torch.repeat_interleave(torch.arange(nt, device=dev), n_t, output_size=V)
    dst, src, et = g.edge_dst.to(torch.int64), g.col_idx.to(torch.int64), g.edge_type.to(torch.int64)
    if E and bool((tile_of_atom[src] != tile_of_atom[dst]).any().item()):
        return None                                       # an edge leaves its tile: not a batch of separate molecules
    # per-atom in-degree by type; the pattern key packs one field per type into an int64: 8 bits each up to seven types,
    # 7 bits at eight (56 bits either way -- 256 ** 8 does not fit).  The tile is NOT packed into the same word (tile *
    # 256 ** K wraps from K = 7 on): two stable sorts, pattern first, tile second.
    cnt = torch.bincount(dst * K + et, minlength=V * K).view(V, K)
    bits = 8 if K <= 7 else 7
    top = (1 << bits) - 1
    code = torch.zeros(V, dtype=torch.int64, device=dev)
    for k in reversed(range(K)):                          # rare types (high ids) first: measured best fill on c2
        code = (code << bits) + (top - cnt[:, k].clamp(max=top))
    by_code = torch.sort(code, stable=True).indices
    perm = by_code[torch.sort(tile_of_atom[by_code], stable=True).indices]            # sorted position -> atom
    pos_in_tile = torch.empty(V, dtype=torch.int64, device=dev)
    pos_in_tile[perm] = torch.arange(V, device=dev) - tp64[tile_of_atom[perm]]
    return nt, tile_ptr, tp64, n_t, tile_of_atom, dst, src, et, cnt, pos_in_tile


def _edge_rank(dst, et, cnt, K, V, E, dev):
    """rank of an edge among the edges of its (destination, type), in edge order"""
    key = dst * K + et
    order = torch.sort(key, stable=True).indices
    first = torch.zeros(V * K + 1, dtype=torch.int64, device=dev)
    first[1:] = torch.cumsum(cnt.reshape(-1), 0)
    rank = torch.empty(E, dtype=torch.int64, device=dev)
    rank[order] = torch.arange(E, device=dev) - first[key[order]]
    return rank


class TilePlan:
    """Work list of the fused message+sum kernel (csrc/message_tile.hip).

    Atoms are cut into molecule-aligned TILES of at most `tile_atoms` atoms (greedy, whole molecules).  Inside a tile
    the atoms are SORTED by their per-type in-degree pattern and dealt in BLOCKS of 16 consecutive sorted atoms; the
    blocks are then labelled so that blocks 2p and 2p+1 (wave pair p of the kernel) are the p-th heaviest and the p-th
    lightest of the tile.  A ROW-TILE is "the rank-th incoming edge of type k of each of the block's 16 atoms":
    row m of the contraction IS destination atom m of the block, so the contraction's accumulator rows are output rows
    and nothing has to be summed afterwards.  A block needs max-over-its-atoms(count of type-k edges) row-tiles for
    type k; sorting atoms by pattern keeps that close to what each atom really has (the slot fill).

        tile_rec[T,16]   first atom, atoms, first row-tile of block 0..7, end of block 7, zeros
        tile_atom[T,128] atom id of every (block, row) of the tile (index 16 * block + row), -1 = none
        slots[16*R]      row-tiles in (tile, block, type, rank) order, 16 words each:
                         (source atom - tile start) | valid << 14 | bond type << 16;
                         an empty slot reads source row `tile_atoms` (a row of zeros in the kernel's LDS image)
        slot_eid[16*R]   edge id of the slot (-1 = empty); tests use it
    """

    def __init__(self, tile_ptr, tile_atom, rt_ptr, slots, slot_eid, tile_atoms, rt_start, num_types):
        self.tile_ptr, self.tile_atom, self.rt_ptr, self.slots, self.slot_eid = tile_ptr, tile_atom, rt_ptr, slots, slot_eid
        self.num_tiles = int(tile_ptr.shape[0]) - 1
        self.num_row_tiles = int(slots.shape[0]) // 16
        self.tile_atoms = tile_atoms
        T, nb = self.num_tiles, tile_atoms // 16
        rec = torch.zeros(T, 16, dtype=torch.int32, device=tile_ptr.device)
        rec[:, 0] = tile_ptr[:-1]
        rec[:, 1] = tile_ptr[1:] - tile_ptr[:-1]
        rec[:, 2:2 + nb] = rt_ptr[:nb * T].view(T, nb)
        rec[:, 2 + nb] = rt_ptr[nb::nb]
        self.tile_rec = rec.contiguous()
        # what one launch reads: a record and the sorted-atom list per tile, every slot word once
        self.nbytes = 4 * (16 * T + tile_atoms * T + 16 * self.num_row_tiles)

    @classmethod
    def _build_kernels(cls, g, lib, tv, rtmax):
        return _tile_plan_kernels(cls, g, lib, tv, rtmax)

    @classmethod
    def build(cls, g):
        lib = _lib.load()
        tv, kmax = lib.mpnn_message_aggregate_tile_atoms(), lib.mpnn_message_aggregate_max_types()
        rtmax = lib.mpnn_message_aggregate_max_row_tiles()
        K, E, V = g.num_types, g.num_edges, g.num_nodes
        if K > kmax:
            return None
        if _fast_path(g, K, kmax) and tv == 128:
            return cls._build_kernels(g, lib, tv, rtmax)
        lay = _tile_layout(g, tv, lib)
        if lay is None:
            return None
        nt, tile_ptr, tp64, n_t, tile_of_atom, dst, src, et, cnt, pos_in_tile = lay
        dev = g.device
        src_local = src - tp64[tile_of_atom[dst]]
        nblk = tv // 16
        sblk_of_atom = tile_of_atom * nblk + pos_in_tile // 16                        # block in sorted order
        row_of_atom = pos_in_tile % 16
        # ---- row-tiles: a block needs max_count(type k) of them for type k
        need = torch.zeros(nt * nblk * K, dtype=torch.int64, device=dev)
        need.scatter_reduce_(0, (sblk_of_atom.unsqueeze(1) * K + torch.arange(K, device=dev)).reshape(-1), cnt.reshape(-1),
                             reduce="amax")
        # ---- balance: the kernel gives blocks 2p and 2p+1 of a tile to wave pair p, so relabel the sorted blocks such
        # that the p-th heaviest (most row-tiles) sits next to the p-th lightest
        load = need.view(nt, nblk, K).sum(-1)
        by_load = torch.sort(load, dim=1, descending=True, stable=True).indices       # [t][i] = i-th heaviest block
        label = torch.empty(nblk, dtype=torch.int64, device=dev)
        label[: nblk // 2] = 2 * torch.arange(nblk // 2, device=dev)
        label[nblk // 2:] = 2 * torch.arange(nblk // 2 - 1, -1, -1, device=dev) + 1
        new_of_sorted = torch.empty_like(by_load)
        new_of_sorted.scatter_(1, by_load, label.expand(nt, nblk))
        blk_of_atom = tile_of_atom * nblk + new_of_sorted.reshape(-1)[sblk_of_atom]  # global block id, kernel order
        need = torch.zeros_like(need).view(nt, nblk, K).scatter_(
            1, new_of_sorted.unsqueeze(-1).expand(nt, nblk, K), need.view(nt, nblk, K)).reshape(-1)
        tile_atom = torch.full((nt * tv,), -1, dtype=torch.int64, device=dev)
        tile_atom[blk_of_atom * 16 + row_of_atom] = torch.arange(V, device=dev)
        # memory order of the row-tiles: (t, block, k, rank)
        rt_start = torch.zeros(need.numel() + 1, dtype=torch.int64, device=dev)
        rt_start[1:] = torch.cumsum(need, 0)
        R = int(rt_start[-1].item())
        rt_ptr64 = rt_start[::K]
        if R and int((rt_ptr64[1:] - rt_ptr64[:-1]).max()) > rtmax:
            return None                                   # a block with more row-tiles than the kernel parks in LDS
        grp = torch.arange(need.numel(), device=dev)
        rt_grp = torch.repeat_interleave(grp, need, output_size=R)
        slots = (tv | ((rt_grp % K) << 16)).repeat_interleave(16)                      # empty: the zero row
        slot_eid = torch.full((16 * R,), -1, dtype=torch.int64, device=dev)
        if E:
            rank = _edge_rank(dst, et, cnt, K, V, E, dev)
            pos = (rt_start[blk_of_atom[dst] * K + et] + rank) * 16 + row_of_atom[dst]
            slots[pos] = src_local | (1 << 14) | (et << 16)
            slot_eid[pos] = torch.arange(E, device=dev)
        return cls(tile_ptr, _i32(tile_atom).view(nt, tv), _i32(rt_ptr64), _i32(slots), _i32(slot_eid), tv, rt_start, K)


def _tile_plan_kernels(cls, g, lib, tv, rtmax):
    """TilePlan.build on the index kernels (csrc/plan.hip: mpnn_tile_plan_count / _fill); same arrays, bit for bit."""
    K = g.num_types
    nt, tp = _host_tiles(g, tv, lib)
    if nt <= 0:
        return None                                       # a molecule larger than a tile
    dev = g.device
    tile_ptr = tp.to(dev)
    need, start, tile_atom, atom_slot, flags = _plan_count(lib, lib.mpnn_tile_plan_count, g, tile_ptr, nt, K, tv)
    per_block = start[::K]
    head = torch.stack([start[-1], (per_block[1:] - per_block[:-1]).max(), flags[0].to(torch.int64)]).cpu()
    R, worst, bad = int(head[0]), int(head[1]), int(head[2])
    if bad or (R and worst > rtmax):
        return None                                       # an edge leaves its tile / a block with more row-tiles than fit
    slots = torch.empty(16 * R, dtype=torch.int32, device=dev)
    slot_eid = torch.empty(16 * R, dtype=torch.int32, device=dev)
    rt_ptr = torch.empty(nt * 8 + 1, dtype=torch.int32, device=dev)
    _lib.check(lib.mpnn_tile_plan_fill(_lib.iptr(g.row_ptr), _lib.iptr(g.col_idx), _lib.iptr(g.edge_type), _lib.iptr(tile_ptr),
                                       nt, K, _lib.ptr(start), _lib.iptr(atom_slot), _lib.iptr(slots), _lib.iptr(slot_eid),
                                       _lib.iptr(rt_ptr), _lib.stream()), "mpnn_tile_plan_fill")
    return cls(tile_ptr, tile_atom, rt_ptr, slots, slot_eid, tv, start, K)


def _plan_count(lib, fn, g, tile_ptr, nt, K, tile_atoms):
    """count kernel + scan: -> (start int64 [nt*8*K + 1], tile_atom, atom_slot, total rows, flags) with ONE host read."""
    dev = g.device
    need = torch.empty(nt * 8 * K, dtype=torch.int64, device=dev)
    tile_atom = torch.empty(nt, tile_atoms, dtype=torch.int32, device=dev)
    atom_slot = torch.empty(g.num_nodes, dtype=torch.int32, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(fn(_lib.iptr(g.row_ptr), _lib.iptr(g.col_idx), _lib.iptr(g.edge_type), _lib.iptr(tile_ptr), nt, K,
                  _lib.ptr(need), _lib.iptr(tile_atom), _lib.iptr(atom_slot), _lib.iptr(flags), _lib.stream()), "plan count")
    start = torch.zeros(need.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(need, 0, out=start[1:])
    return need, start, tile_atom, atom_slot, flags


class WidePlan:
    """Work list of the fused message+sum kernel at widths 128 / 256 (csrc/message_tile_wide.hip): typed
    aggregate-then-contract on molecule-aligned tiles.

    Atoms are cut into molecule-aligned TILES of at most `tile_atoms` (256) atoms -- a 200-atom molecule fits -- sorted
    inside the tile by their per-type in-degree pattern and dealt in BLOCKS of 32 consecutive sorted atoms (one wave of
    the kernel each).  For every (block, bond type k) the kernel first sums, per atom, the h rows of its type-k
    neighbours (S_k, from the tile's rows staged in LDS), then contracts the 32 sums with A_k on the matrix cores into the
    block's output accumulators.  A (block, type) pair without edges costs nothing; sorting keeps atoms with the same
    types together.  What the kernel needs to find the neighbours is a SLOT ROW per (block, type, rank): 32 16-bit words,
    word m = (rank-th type-k source of the block's atom m) - tile start, or `tile_atoms` = a row of zeros.

        tile_rec[T,4]       first atom, atoms, first slot row (global), slot rows of the tile
        tile_atom[T,256]    atom id of every (block, row) of the tile (index 32 * block + row), -1 = none
        blk_off[T,8*K+1]    first slot row of every (block, type) relative to the tile's first, then the tile's count
        slots[32*R] int16   slot rows in (tile, block, type, rank) order
        slot_eid[32*R]      edge id of the slot (-1 = empty); tests use it
    """
    TILE_ATOMS = 256
    BLOCK = 32
    MAX_ROWS = 256           # slot rows of one tile parked in LDS (16 KB of 16-bit words)
    MAX_TYPES = 8

    def __init__(self, tile_ptr, tile_rec, tile_atom, blk_off, slots, slot_eid, num_types):
        self.tile_ptr, self.tile_rec, self.tile_atom, self.blk_off = tile_ptr, tile_rec, tile_atom, blk_off
        self.slots, self.slot_eid = slots, slot_eid
        self.num_tiles = int(tile_rec.shape[0])
        self.num_rows = int(slots.shape[0]) // self.BLOCK
        self.num_types = num_types
        self.nbytes = 4 * int(tile_rec.numel() + tile_atom.numel() + blk_off.numel()) + 2 * int(slots.numel())

    @classmethod
    def _build_kernels(cls, g, lib):
        """build() on the index kernels (csrc/plan.hip: mpnn_wide_plan_count / _fill); same arrays, bit for bit."""
        K, tv = g.num_types, cls.TILE_ATOMS
        nt, tp = _host_tiles(g, tv, lib)
        if nt <= 0:
            return None                                   # a molecule larger than a tile
        dev = g.device
        tile_ptr = tp.to(dev)
        need, start, tile_atom, atom_slot, flags = _plan_count(lib, lib.mpnn_wide_plan_count, g, tile_ptr, nt, K, tv)
        per_tile = start[::8 * K]
        head = torch.stack([start[-1], (per_tile[1:] - per_tile[:-1]).max(), flags[0].to(torch.int64)]).cpu()
        R, worst, bad = int(head[0]), int(head[1]), int(head[2])
        if bad or worst > cls.MAX_ROWS:
            return None                                   # an edge leaves its tile / more slot rows than the kernel parks in LDS
        slots = torch.empty(cls.BLOCK * R, dtype=torch.int16, device=dev)
        slot_eid = torch.empty(cls.BLOCK * R, dtype=torch.int32, device=dev)
        tile_rec = torch.empty(nt, 4, dtype=torch.int32, device=dev)
        blk_off = torch.empty(nt, 8 * K + 1, dtype=torch.int32, device=dev)
        _lib.check(lib.mpnn_wide_plan_fill(_lib.iptr(g.row_ptr), _lib.iptr(g.col_idx), _lib.iptr(g.edge_type),
                                           _lib.iptr(tile_ptr), nt, K, _lib.ptr(start), _lib.iptr(atom_slot),
                                           _lib.ptr(slots, torch.int16), _lib.iptr(slot_eid), _lib.iptr(tile_rec),
                                           _lib.iptr(blk_off), _lib.stream()), "mpnn_wide_plan_fill")
        return cls(tile_ptr, tile_rec, tile_atom, blk_off, slots, slot_eid, K)

    @classmethod
    def build(cls, g):
        lib = _lib.load()
        tv, nb32 = cls.TILE_ATOMS, cls.BLOCK
        K, E, V = g.num_types, g.num_edges, g.num_nodes
        if K > cls.MAX_TYPES:
            return None
        if _fast_path(g, K, cls.MAX_TYPES):
            return cls._build_kernels(g, lib)
        lay = _tile_layout(g, tv, lib)
        if lay is None:
            return None
        nt, tile_ptr, tp64, n_t, tile_of_atom, dst, src, et, cnt, pos_in_tile = lay
        dev = g.device
        nblk = tv // nb32
        blk_of_atom = tile_of_atom * nblk + pos_in_tile // nb32
        row_of_atom = pos_in_tile % nb32
        need = torch.zeros(nt * nblk * K, dtype=torch.int64, device=dev)
        need.scatter_reduce_(0, (blk_of_atom.unsqueeze(1) * K + torch.arange(K, device=dev)).reshape(-1), cnt.reshape(-1),
                             reduce="amax")
        start = torch.zeros(need.numel() + 1, dtype=torch.int64, device=dev)
        start[1:] = torch.cumsum(need, 0)
        R = int(start[-1].item())
        per = nblk * K
        tile_row0 = start[::per][:nt]
        rows_of_tile = start[per::per] - tile_row0
        if int(rows_of_tile.max()) > cls.MAX_ROWS:
            return None                                   # more slot rows than the kernel parks in LDS
        idx = torch.arange(nt, device=dev).unsqueeze(1) * per + torch.arange(per + 1, device=dev)
        blk_off = _i32(start[idx] - tile_row0.unsqueeze(1))
        tile_rec = _i32(torch.stack([tp64[:-1], n_t, tile_row0, rows_of_tile], dim=1))
        tile_atom = torch.full((nt * tv,), -1, dtype=torch.int64, device=dev)
        tile_atom[blk_of_atom * nb32 + row_of_atom] = torch.arange(V, device=dev)
        slots = torch.full((nb32 * R,), tv, dtype=torch.int16, device=dev)
        slot_eid = torch.full((nb32 * R,), -1, dtype=torch.int64, device=dev)
        if E:
            rank = _edge_rank(dst, et, cnt, K, V, E, dev)
            pos = (start[blk_of_atom[dst] * K + et] + rank) * nb32 + row_of_atom[dst]
            slots[pos] = (src - tp64[tile_of_atom[dst]]).to(torch.int16)
            slot_eid[pos] = torch.arange(E, device=dev)
        return cls(tile_ptr, tile_rec, _i32(tile_atom).view(nt, tv), blk_off.contiguous(), slots, _i32(slot_eid), K)


class MolGraph:
    def __init__(self, row_ptr, col_idx, edge_weight, edge_type, type_feat, graph_ptr, dense_shape=None,
                 edge_feat=None):
        self.row_ptr = row_ptr
        self.col_idx = col_idx
        self.edge_weight = edge_weight
        self.edge_type = edge_type
        self.type_feat = type_feat
        self.graph_ptr = graph_ptr
        self.dense_shape = dense_shape          # (B, N) when built from a padded batch
        self.edge_feat = edge_feat              # (E, ef) raw rows when known
        self.num_nodes = int(row_ptr.shape[0]) - 1
        self.num_edges = int(col_idx.shape[0])
        self.num_types = int(type_feat.shape[0])
        self.num_graphs = int(graph_ptr.shape[0]) - 1
        self.device = row_ptr.device
        self._order = None
        self._type_ptr = None
        self._edge_dst = None
        self._transpose = None
        self._node_graph = None
        self._pad_size = None
        self._unit_weights = None
        self._adj_ptr = None
        self._tile_plan = None
        self._wide_plan = None

    def with_type_feat(self, type_feat):
        """The same graph with another (K, ef) table of bond-feature rows (index arrays and their caches shared)."""
        import copy
        g = copy.copy(self)
        g.type_feat = type_feat
        g.edge_feat = None
        return g

    def prepare(self, tile_plan=True, wide_plan=False):
        """Build every derived index array now (type order, transposed graph, destination list and the tile plans of
        the fused message+sum kernels the caller will run: `tile_plan` for width 64, `wide_plan` for 128 / 256), so
        that none of it lands inside a timed or captured region."""
        self._index_kernels()
        self.order, self.type_ptr, self.transpose, self.edge_dst, self.agg_weight
        if tile_plan:
            self.tile_plan
        if wide_plan:
            self.wide_plan
        return self

    def _index_kernels(self):
        """Destination list, type order and transposed graph from the tile kernels of csrc/plan.hip (a device-resident batch
        of separate molecules with few bond types); leaves them to the lazy torch builders otherwise."""
        if self._order is not None and self._transpose is not None and self._edge_dst is not None:
            return
        lib = _lib.load()
        K, V, E = self.num_types, self.num_nodes, self.num_edges
        if not _fast_path(self, K, lib.mpnn_plan_index_max_types()) or E == 0:
            return
        nt, tp = _host_tiles(self, lib.mpnn_plan_index_tile_atoms(), lib)
        if nt <= 0:
            return
        dev = self.device
        tile_ptr = tp.to(dev)
        edge_dst = torch.empty(E, dtype=torch.int32, device=dev)
        t_row_ptr = torch.empty(V + 1, dtype=torch.int32, device=dev)
        t_eid = torch.empty(E, dtype=torch.int32, device=dev)
        hist = torch.empty(nt, K, dtype=torch.int32, device=dev)
        flags = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.mpnn_plan_index_tiles(_lib.iptr(self.row_ptr), _lib.iptr(self.col_idx), _lib.iptr(self.edge_type),
                                             _lib.iptr(tile_ptr), V, E, nt, K, _lib.iptr(edge_dst), _lib.iptr(t_row_ptr),
                                             _lib.iptr(t_eid), _lib.iptr(hist), _lib.iptr(flags), _lib.stream()),
                   "mpnn_plan_index_tiles")
        off = torch.zeros(K * nt + 1, dtype=torch.int64, device=dev)
        torch.cumsum(hist.t().reshape(-1), 0, out=off[1:])
        order = torch.empty(E, dtype=torch.int32, device=dev)
        type_ptr = torch.empty(K + 1, dtype=torch.int32, device=dev)
        _lib.check(lib.mpnn_plan_type_order(_lib.iptr(self.row_ptr), _lib.iptr(self.edge_type), _lib.iptr(tile_ptr),
                                            _lib.ptr(off), E, nt, K, _lib.iptr(order), _lib.iptr(type_ptr), _lib.stream()),
                   "mpnn_plan_type_order")
        if int(flags.item()) != 0:
            return                                        # not a batch of separate molecules: the generic builders
        self._edge_dst, self._transpose, self._order, self._type_ptr = edge_dst, (t_row_ptr, t_eid), order, type_ptr

    def plan_bytes(self):
        """Bytes of index data the fused message+sum kernel reads per launch (0 without a plan)."""
        p = self._tile_plan or self._wide_plan
        return p.nbytes if p else 0

    @property
    def tile_plan(self):
        """TilePlan for the fused message+sum kernels, or None when the batch does not fit them (a molecule larger than a
        tile, too many bond types).  Built once per batch, like the CSR."""
        if self._tile_plan is None:
            self._tile_plan = TilePlan.build(self) or False
        return self._tile_plan or None

    @property
    def wide_plan(self):
        """WidePlan for the fused message+sum kernel at widths 128 / 256, or None when the batch does not fit it."""
        if self._wide_plan is None:
            self._wide_plan = WidePlan.build(self) or False
        return self._wide_plan or None

    # ------------------------------------------------------------------ derived index arrays
    @property
    def order(self):
        if self._order is None:
            self._build_type_order()
        return self._order

    @property
    def type_ptr(self):
        if self._type_ptr is None:
            self._build_type_order()
        return self._type_ptr

    def _build_type_order(self):
        et = self.edge_type.to(torch.int64)
        self._order = _i32(torch.sort(et, stable=True).indices)
        counts = torch.bincount(et, minlength=self.num_types)
        tp = torch.zeros(self.num_types + 1, dtype=torch.int64, device=self.device)
        tp[1:] = torch.cumsum(counts, 0)
        self._type_ptr = _i32(tp)

    @property
    def edge_dst(self):
        if self._edge_dst is None:
            deg = (self.row_ptr[1:] - self.row_ptr[:-1]).to(torch.int64)
            self._edge_dst = _i32(torch.repeat_interleave(
                torch.arange(self.num_nodes, device=self.device), deg, output_size=self.num_edges))
        return self._edge_dst

    @property
    def transpose(self):
        """(t_row_ptr, t_eid): edge ids grouped by source atom, stable in edge order."""
        if self._transpose is None:
            src = self.col_idx.to(torch.int64)
            t_eid = _i32(torch.sort(src, stable=True).indices)
            counts = torch.bincount(src, minlength=self.num_nodes)
            tp = torch.zeros(self.num_nodes + 1, dtype=torch.int64, device=self.device)
            tp[1:] = torch.cumsum(counts, 0)
            self._transpose = (_i32(tp), t_eid)
        return self._transpose

    @property
    def edge_features(self):
        """(E, ef) bond-feature row of every edge (materialised from the type table on first use)."""
        if self.edge_feat is None:
            self.edge_feat = self.type_feat[self.edge_type.to(torch.int64)].contiguous()
        return self.edge_feat

    @property
    def agg_weight(self):
        """Per-edge multiplier for AdjMsgAgg, or None when every adjacency value is exactly 1 (the
        kernel then skips the weight stream: 4*E fewer bytes and one load less per edge)."""
        if self.edge_weight is None:
            return None
        if self._unit_weights is None:
            self._unit_weights = bool((self.edge_weight == 1.0).all().item()) if self.num_edges else True
        return None if self._unit_weights else self.edge_weight

    @property
    def node_graph(self):
        """Molecule id of every atom, int64 (V,)."""
        if self._node_graph is None:
            n = (self.graph_ptr[1:] - self.graph_ptr[:-1]).to(torch.int64)
            self._node_graph = torch.repeat_interleave(
                torch.arange(self.num_graphs, device=self.device), n, output_size=self.num_nodes)
        return self._node_graph

    @property
    def pad_size(self):
        """Per-atom padded row length N (dense batch: the batch's N; compact batch: its molecule's size)."""
        if self._pad_size is None:
            if self.dense_shape is not None:
                self._pad_size = torch.full((self.num_nodes,), float(self.dense_shape[1]), device=self.device)
            else:
                n = (self.graph_ptr[1:] - self.graph_ptr[:-1]).to(torch.float32)
                self._pad_size = n[self.node_graph]
        return self._pad_size

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_dense(cls, adj=None, bfm=None):
        """CSR of the pairs where adj != 0 or any bond feature != 0, in adj.nonzero() order.

        Runs mpnn_csr_count / mpnn_csr_fill; one host read of E in between (allocation size).
        """
        lib = _lib.load()
        ref = adj if adj is not None else bfm
        if ref is None:
            raise _lib.MpnnError("from_dense needs adj or bfm")
        B, N = int(ref.shape[0]), int(ref.shape[1])
        dev = ref.device
        ef = int(bfm.shape[-1]) if bfm is not None else 0
        adj_c = adj.contiguous().float() if adj is not None else None
        bfm_c = bfm.contiguous().float() if bfm is not None else None
        rows = B * N
        row_ptr = torch.empty(rows + 1, dtype=torch.int32, device=dev)
        ws_bytes = lib.mpnn_csr_workspace_bytes(rows)
        ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
        _lib.check(lib.mpnn_csr_count(_lib.fptr(adj_c), _lib.fptr(bfm_c), rows, N, ef, _lib.iptr(row_ptr),
                                      _lib.ptr(ws), ws_bytes, _lib.stream()), "mpnn_csr_count")
        E = int(row_ptr[-1].item())
        col_idx = torch.empty(E, dtype=torch.int32, device=dev)
        edge_weight = torch.empty(E, dtype=torch.float32, device=dev)
        edge_feat = torch.empty(E, ef, dtype=torch.float32, device=dev) if bfm is not None else None
        if E > 0:
            _lib.check(lib.mpnn_csr_fill(_lib.fptr(adj_c), _lib.fptr(bfm_c), rows, N, ef, _lib.iptr(row_ptr),
                                         _lib.iptr(col_idx), _lib.fptr(edge_weight), _lib.fptr(edge_feat),
                                         _lib.stream()), "mpnn_csr_fill")
        if edge_feat is not None and E > 0:
            type_feat, inv = torch.unique(edge_feat, dim=0, return_inverse=True)
            edge_type = _i32(inv)
        elif edge_feat is not None:
            type_feat = torch.zeros(1, ef, device=dev)
            edge_type = torch.zeros(0, dtype=torch.int32, device=dev)
        else:
            type_feat = torch.zeros(1, 1, device=dev)
            edge_type = torch.zeros(E, dtype=torch.int32, device=dev)
        graph_ptr = torch.arange(0, rows + 1, N, dtype=torch.int32, device=dev)
        g = cls(row_ptr, col_idx, edge_weight, edge_type, type_feat.contiguous(), graph_ptr,
                dense_shape=(B, N), edge_feat=edge_feat)
        g._adj_ptr = adj.data_ptr() if adj is not None else None    # edge_weight mirrors THIS adj tensor
        return g

    @classmethod
    def from_molbatch(cls, mb, device, dedupe=False):
        """Upload a compact synth.MolBatch.  `dedupe` collapses a continuous batch's per-edge
        feature rows to distinct rows (reverse edges share a row)."""
        def up(a, dt):
            return torch.from_numpy(a).to(device=device, dtype=dt)
        edge_type = up(mb.bond_type, torch.int32)
        type_feat = up(mb.type_feat, torch.float32)
        if dedupe and mb.edge_feat is not None:
            type_feat, inv = torch.unique(type_feat, dim=0, return_inverse=True)
            edge_type = _i32(inv)
        g = cls(up(mb.row_ptr, torch.int32), up(mb.col_idx, torch.int32),
                None, edge_type, type_feat.contiguous(),
                up(mb.atom_ptr, torch.int32), dense_shape=None,
                edge_feat=(up(mb.edge_feat, torch.float32) if mb.edge_feat is not None else None))
        g._graph_ptr_host = torch.from_numpy(mb.atom_ptr).to(torch.int32).contiguous()   # (the tile cutter runs on the host)
        return g

    # ------------------------------------------------------------------ views
    def node_view(self, x):
        """(B,N,F) or (V,F) -> contiguous (V,F)."""
        F = x.shape[-1]
        x2 = x.reshape(-1, F)
        if x2.shape[0] != self.num_nodes:
            raise _lib.MpnnError("node array has %d rows, graph has %d atoms" % (x2.shape[0], self.num_nodes))
        return x2.contiguous()

    def node_unview(self, x2, like=None):
        if self.dense_shape is not None:
            return x2.view(self.dense_shape[0], self.dense_shape[1], x2.shape[-1])
        return x2
