"""Drop-in ``LightGCN`` / ``LightGCNEngine`` for beta_rec/models/lightgcn.py on libhiprec.so.

Interface parity (file:line = /root/reference/beta_rec/...): ``LightGCN(config, norm_adj)``
models/lightgcn.py:7-101 (``forward(norm_adj) -> (user, item) embeddings``, ``predict``),
``LightGCNEngine(config)`` :104-191 (``train_single_batch(batch) -> float``, ``train_an_epoch``).
Same config keys (``n_users n_items emb_dim layer_size keep_pro regs norm_adj optimizer lr
device_str``), same ``state_dict`` keys, same initial weights for the same torch seed.

The sparse propagation, the loss and the whole backward are ``csrc/lightgcn.hip``.  The graph is
converted ONCE from the torch sparse COO tensor the reference passes around to CSR + transposed CSR
on the device.  Edge dropout (training only, ``keep_pro``): ``dropout_rng = "torch_cpu"`` (default)
draws ``torch.rand(nnz)`` from the global CPU generator exactly like models/lightgcn.py:32, so the
same seed drops the same edges as the reference; ``"device"`` draws the keep bytes on the GPU
(no host work, no 2 MB upload per step) for production speed.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .mf import _new_stats, raise_on_status, read_stats
from .ncf import _FlatModel, _ParamView
from .flat_engine import FlatModelEngine


def _slice_rows(rowptr, col, val, n, nnz):
    """hiprec_csr.slice_row of a CSR on the device (one launch, once per graph)."""
    lib = _lib.load()
    out = torch.empty(max(int(lib.hiprec_csr_n_slices(nnz)), 1), dtype=torch.int32, device=rowptr.device)
    csr = _lib.Csr(rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), None, n, nnz, None)
    _lib.check(lib.hiprec_csr_slice_rows(ctypes.byref(csr), _lib.ptr(out), out.numel(), _lib.stream_ptr(rowptr.device)))
    return out


def _csr_from_coo(rows, cols, vals, n, device):
    """Sorted CSR (int64 rowptr, int32 col, fp32 val) + the sort permutation."""
    order = torch.argsort(rows * n + cols, stable=True)
    r, c, v = rows[order], cols[order], vals[order]
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
    return (rowptr.to(device), c.to(torch.int32).to(device), v.to(torch.float32).to(device), order)


# Column-sliced SpMM (csrc/spmm_sliced.hip): a row is padded to a multiple of S = lane_slots slots and cut into chunks
# of 4 S (S per lane of a quad).  S = 16 always works; a factored graph may use the larger ones.
SLICED_LANE_SLOTS = (16, 24, 32, 48)
SLICED_PAD, SLICED_CHUNK = 16, 64  # the S = 16 geometry


def choose_lane_slots(lens, n_groups, factored):
    """S for a graph with these row lengths: the one that minimises a workgroup's modelled time.  A chunk costs its
    3 S VALU instructions for the slots (address + two packed adds each) plus ~60 for everything else (descriptor,
    loads, the quad's reduction, the store; ~40 more when the row has several chunks: scan and carries), a wave takes
    every 16th window of 16 chunks, and the slowest wave of a 16-wave workgroup sets the time -- so few, long chunks
    win until the rows stop filling them or the waves' trip counts quantise badly (round 6: at S = 16 the kernel was
    VALU-bound, 165 instructions per 16-slot trip)."""
    if not factored:
        return 16
    lens = np.asarray(lens, dtype=np.int64)
    best = None
    for S in SLICED_LANE_SLOTS:
        per_row = (((lens + S - 1) // S) + 3) // 4
        n_chunks = int(per_row.sum())
        multi = int(per_row[per_row > 1].sum())
        windows = -(-n_chunks // (16 * max(n_groups, 1)))          # per workgroup
        trips = -(-windows // 16)                                   # of its slowest wave
        mean_cost = 3 * S + 60 + 40 * (multi / max(n_chunks, 1))
        cost = trips * mean_cost
        if best is None or cost < best[0]:
            best = (cost, S)
    return best[1]


def factor_edge_values(rowptr, col, val, rel_tol=2e-6, max_rounds=256):
    """(row_scale, col_scale) float32 [n] with val[e] == row_scale[row(e)] * col_scale[col(e)] to rel_tol on every
    edge of a square CSR -- what every degree-normalised adjacency is (D^-1 (A + I) of the reference: row_scale =
    1 / degree, col_scale = 1; the symmetric D^-1/2 A D^-1/2 as well) -- or None.

    Solved by propagation over the bipartite row / column graph: one seed row per connected component gets scale
    1, known row scales fix the scales of their columns and the other way round, then every edge is verified."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import connected_components

    rowptr = np.asarray(rowptr, dtype=np.int64)
    n, nnz = rowptr.size - 1, int(rowptr[-1])
    col = np.asarray(col, dtype=np.int64)[:nnz]
    v = np.asarray(val, dtype=np.float64)[:nnz]
    r, c = np.ones(n), np.ones(n)
    if nnz == 0:
        return r.astype(np.float32), c.astype(np.float32)
    if not np.all(np.isfinite(v)) or np.any(v == 0):
        return None
    row = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
    both = sp.coo_matrix((np.ones(nnz, np.int8), (row, col + n)), shape=(2 * n, 2 * n))
    _, label = connected_components(both, directed=False)
    known_r, known_c = np.zeros(n, bool), np.zeros(n, bool)
    has_edges = np.diff(rowptr) > 0
    rows_with = np.nonzero(has_edges)[0]
    _, first = np.unique(label[rows_with], return_index=True)  # one seed row per component that has edges
    known_r[rows_with[first]] = True
    for _ in range(max_rounds):
        m = known_r[row] & ~known_c[col]  # rows fix columns (first such edge per column)
        if m.any():
            cj, idx = np.unique(col[m], return_index=True)
            e = np.nonzero(m)[0][idx]
            c[cj] = v[e] / r[row[e]]
            known_c[cj] = True
        m2 = known_c[col] & ~known_r[row]
        if m2.any():
            ri, idx = np.unique(row[m2], return_index=True)
            e = np.nonzero(m2)[0][idx]
            r[ri] = v[e] / c[col[e]]
            known_r[ri] = True
        if not m.any() and not m2.any():
            break
    else:
        return None
    r32, c32 = r.astype(np.float32), c.astype(np.float32)
    if not (np.all(np.isfinite(r32)) and np.all(np.isfinite(c32))):
        return None
    prod = r32[row].astype(np.float64) * c32[col].astype(np.float64)
    if np.any(np.abs(prod - v) > rel_tol * np.abs(v)):
        return None
    return r32, c32


def _windowed_chunks(start, chunk_row, clen, within, per_row, sub_chunk):
    """Chunk descriptors as csrc/spmm_sliced.hip wants them.  A wave of the kernel works on a WINDOW of 16
    consecutive chunks (one per quad of lanes) and sums the chunks of one row inside the window itself before anything
    touches the LDS accumulators; what it has to do is static, so it is worked out here:
      * every subgroup is padded with empty chunks (0 slots) to a multiple of 16, so that a window never straddles two
        subgroups and the 16 quads of a wave always work on the same window;
      * a RUN = the consecutive chunks of one row inside a window.  Per chunk: its position in the run counted inside
        its 16-lane row of the wave (bits 24-25: how many of the two DPP scan steps apply), whether the run began in an
        earlier 16-lane row (bit 26: take the carry of that row's last quad), whether it is the run's last chunk (bit
        27: it stores) and whether the run is the whole row (bit 28: a plain store, else an LDS atomic add).
    Descriptor: first slot, row | slots << 16 | flags.  Returns (chunks int32 [n, 2], sub_chunk)."""
    sizes = np.diff(sub_chunk)
    padded = (sizes + 15) // 16 * 16
    new_sub = np.concatenate([[0], np.cumsum(padded)])
    n_new = int(new_sub[-1])
    sub_of_old = np.repeat(np.arange(sizes.size), sizes)
    new_idx = new_sub[sub_of_old] + np.arange(sub_of_old.size) - sub_chunk[sub_of_old]
    # an empty chunk carries the row of its subgroup's last real chunk (descriptors stay sorted by row)
    last_row = chunk_row[np.maximum(sub_chunk[1:] - 1, 0)] if chunk_row.size else np.zeros(sizes.size, np.int64)
    row2 = np.repeat(last_row, padded)
    start2, clen2 = np.zeros(n_new, np.int64), np.zeros(n_new, np.int64)
    within2, per2 = np.zeros(n_new, np.int64), np.ones(n_new, np.int64)
    real = np.zeros(n_new, bool)
    row2[new_idx], start2[new_idx], clen2[new_idx] = chunk_row, start, clen
    within2[new_idx], per2[new_idx], real[new_idx] = within, per_row[chunk_row], True
    idx = np.arange(n_new)
    prev_real = np.concatenate([[False], real[:-1]])
    prev_row = np.concatenate([[-1], row2[:-1]])
    new_run = ((idx & 15) == 0) | (row2 != prev_row) | ~real | ~prev_real
    run_start = np.maximum.accumulate(np.where(new_run, idx, 0))
    behind, qir = idx - run_start, idx & 3
    last = np.concatenate([new_run[1:], [True]]) & real
    whole = last & (within2[run_start] == 0) & (within2 == per2 - 1)
    flags = (np.minimum(behind, qir) << 24) | ((behind > qir).astype(np.int64) << 26) | (last.astype(np.int64) << 27) \
        | (whole.astype(np.int64) << 28)
    flags[~real] = 0
    desc = (row2 | (clen2 << 16) | flags).astype(np.int64)
    return np.stack([start2, desc], axis=1).astype(np.int32), new_sub


def sliced_graph_host(rowptr, col, val, eid, n_groups, row_cap, factor=True, lane_slots=None):
    """hiprec_sliced_csr (include/hiprec.h) of a CSR given as numpy arrays; eid = keep-byte index of every edge
    (None = the edge number itself).  factor: look for the rank-one form of the values (factor_edge_values); the
    graph is then stored with row_scale / col_scale and its padding slots point at the zero row n.  lane_slots: S
    (None: choose_lane_slots).  Workgroup g of a slice takes the chunks of rows sub_row[g] .. sub_row[g + 1] (about
    equal chunk counts).  A row whose chunks are one run of a window is written by the kernel directly; the others are
    the workgroup's SPILL rows (summed in LDS: at most row_cap of them) or its EMPTY rows.

    Returns dict(col16 uint16, val float32, eid int32 [n_slots]; chunks int32 [n_chunks, 2]; sub_row, sub_chunk,
    spill_ptr, empty_ptr int32 [n_groups + 1]; spill_row, empty_row int32; subs_per_group 1; n_chunks; n_slots;
    lane_slots; pad_slot; optionally row_scale, col_scale float32 [n]) or None when a workgroup would have more than
    row_cap spill rows or the graph more than 2^22 slots."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    n, nnz = rowptr.size - 1, int(rowptr[-1])
    scales = factor_edge_values(rowptr, col, val) if factor else None
    lens = np.diff(rowptr)
    S = int(lane_slots) if lane_slots is not None else choose_lane_slots(lens, n_groups, scales is not None)
    if S not in SLICED_LANE_SLOTS or (scales is None and S != 16):
        raise ValueError(f"lane_slots {S}: 16, or one of {SLICED_LANE_SLOTS} for a factored graph")
    chunk = 4 * S
    padded = (lens + S - 1) // S * S
    slotptr = np.concatenate([[0], np.cumsum(padded)])
    pad_slot = int(slotptr[-1])                      # rows' slots, then the all-padding tail lanes without slots read
    n_slots = (pad_slot + S + 15) // 16 * 16
    if n_slots >= 1 << 22:                           # a descriptor holds 22 bits of first slot
        return None
    edge_row = np.repeat(np.arange(n, dtype=np.int64), lens)
    slot = np.arange(nnz, dtype=np.int64) + (slotptr[:-1] - rowptr[:-1])[edge_row]
    col16 = np.full(n_slots, n if scales is not None else 0, np.uint16)
    valp, eidp = np.zeros(n_slots, np.float32), np.full(n_slots, -1, np.int32)
    col16[slot] = np.asarray(col)[:nnz]
    valp[slot] = np.asarray(val)[:nnz]
    eidp[slot] = np.arange(nnz) if eid is None else np.asarray(eid)[:nnz]
    per_row = (padded + chunk - 1) // chunk
    first = np.cumsum(per_row) - per_row  # first chunk of every row
    n_chunks = int(per_row.sum())
    chunk_row = np.repeat(np.arange(n, dtype=np.int64), per_row)
    within = np.arange(n_chunks, dtype=np.int64) - first[chunk_row]
    start = slotptr[chunk_row] + chunk * within
    clen = np.minimum(chunk, padded[chunk_row] - chunk * within)
    target = np.minimum((np.arange(n_groups + 1) * n_chunks) // n_groups, max(n_chunks - 1, 0))
    sub_row = chunk_row[target] if n_chunks else np.zeros(n_groups + 1, dtype=np.int64)
    sub_row[0], sub_row[-1] = 0, n
    sub_row = np.maximum.accumulate(sub_row)
    sub_chunk = np.append(first, n_chunks)[sub_row]  # a workgroup starts at the first chunk of its first row
    chunks, sub_chunk = _windowed_chunks(start, chunk_row, clen, within, per_row, sub_chunk)
    # spill rows: the rows with a run that is not the whole row (its last chunk has bit 27 but not bit 28)
    flags = chunks[:, 1].astype(np.int64)
    part_end = ((flags >> 27) & 1).astype(bool) & ~((flags >> 28) & 1).astype(bool)
    spill_rows = np.unique(flags[part_end] & 0xFFFF)               # sorted, so grouped by workgroup
    spill_ptr = np.searchsorted(spill_rows, sub_row)
    if spill_rows.size and np.diff(spill_ptr).max() > row_cap:
        return None
    idx_of = np.zeros(n + 1, dtype=np.int64)                        # index inside its workgroup's list
    group_of = np.searchsorted(sub_row, spill_rows, side="right") - 1
    idx_of[spill_rows] = np.arange(spill_rows.size) - spill_ptr[group_of]
    chunks[:, 0] |= np.where(part_end, idx_of[flags & 0xFFFF] << 22, 0).astype(np.int32)
    empty_rows = np.nonzero(per_row == 0)[0]
    out = {"col16": col16, "val": valp, "eid": eidp, "chunks": chunks, "sub_row": sub_row.astype(np.int32),
           "sub_chunk": sub_chunk.astype(np.int32), "subs_per_group": 1, "n_chunks": int(chunks.shape[0]),
           "spill_row": spill_rows.astype(np.int32), "spill_ptr": spill_ptr.astype(np.int32),
           "empty_row": empty_rows.astype(np.int32),
           "empty_ptr": np.searchsorted(empty_rows, sub_row).astype(np.int32),
           "n_slots": n_slots, "lane_slots": S, "pad_slot": pad_slot}
    if scales is not None:
        out["row_scale"], out["col_scale"] = scales
    return out


# quads (of a wave's 16) whose lanes one ds_read_b128 serves in the same LDS cycle (MI355X_MICROARCH, LDS table:
# lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63})
_B128_QUAD_GROUPS = ((0, 3, 5, 6), (1, 2, 4, 7), (8, 11, 13, 14), (9, 10, 12, 15))


def spread_bank_conflicts(host, n_groups, quads_per_block=192):
    """Reorder the S slots inside every lane's segment of a sliced graph (in place) so that the source rows the 16
    lanes of one `ds_read_b128` cycle read at the same time fall into different bank quads (column mod 16) where
    possible.  The kernel's chunk -> (block, wave, quad, trip) assignment is static, so which lanes read together
    is known here; a lane sums its S slots anyway, so their order is free.  Greedy: read by read, lane by lane, take
    the unused slot whose bank quad is least used in this cycle.  Random columns conflict 3.0-way on average on the
    BASELINE configs[4] graph, 2.0-way after this.  Returns (mean ways before, after)."""
    chunks, k, S = host["chunks"], host["subs_per_group"], host["lane_slots"]
    col16, n_chunks = host["col16"], host["n_chunks"]
    if n_chunks == 0:
        return 1.0, 1.0
    start = (chunks[:, 0] & 0x3FFFFF).astype(np.int64)
    clen = ((chunks[:, 1] >> 16) & 0xFF).astype(np.int64)
    block_first = host["sub_chunk"][np.arange(n_groups + 1) * k].astype(np.int64)  # chunks of block g
    block_of = np.searchsorted(block_first, np.arange(n_chunks), side="right") - 1
    idx = np.arange(n_chunks) - block_first[block_of]
    trip, quad = idx // quads_per_block, idx % quads_per_block
    wave, qiw = quad // 16, quad % 16
    group_of_quad = np.empty(16, np.int64)
    pos_in_group = np.empty(16, np.int64)
    for gi, quads in enumerate(_B128_QUAD_GROUPS):
        for pi, qq in enumerate(quads):
            group_of_quad[qq], pos_in_group[qq] = gi, pi
    # instance = (block, trip, wave, lane group): 4 chunks x 4 lanes = 16 lanes of S slots
    n_trips = int(trip.max()) + 1
    inst = ((block_of * n_trips + trip) * 16 + wave) * 4 + group_of_quad[qiw]
    uniq, inst = np.unique(inst, return_inverse=True)
    n_inst = uniq.size
    slot_of = np.full((n_inst, 16, S), -1, np.int64)  # [instance, lane, j] -> slot (-1: no such slot)
    lane0 = pos_in_group[qiw] * 4
    for q in range(4):
        live = clen > S * q
        base = start[live] + S * q
        slot_of[inst[live], lane0[live] + q] = base[:, None] + np.arange(S)[None, :]
    valid = slot_of >= 0
    # bank quad of every slot; 16 = costs nothing (no slot, or a padding slot: those all read one address, a broadcast)
    real = valid & (host["eid"][np.maximum(slot_of, 0)] >= 0)
    bank = np.where(real, col16[np.maximum(slot_of, 0)].astype(np.int64) & 15, 16)

    def mean_ways(b):
        cnt = np.zeros((n_inst, S, 17), np.int64)  # [instance, j, bank]
        np.add.at(cnt, (np.arange(n_inst)[:, None, None], np.arange(S)[None, None, :], b), 1)
        ways = cnt[:, :, :16].max(2)
        busy = ways > 0
        return float(ways[busy].mean()) if busy.any() else 1.0

    before = mean_ways(bank)
    order = np.zeros((n_inst, 16, S), np.int64)
    remaining = np.ones((n_inst, 16, S), bool)
    ar = np.arange(n_inst)
    for j in range(S):
        cnt = np.zeros((n_inst, 17), np.int64)
        for lane in range(16):
            b = bank[:, lane]
            cost = np.take_along_axis(cnt, b, 1).astype(np.float64)
            cost[b == 16] = 1e6  # free slots go last
            cost[~remaining[:, lane]] = 1e9
            pick = cost.argmin(1)
            order[:, lane, j] = pick
            remaining[ar, lane, pick] = False
            pb = b[ar, pick]
            cnt[ar, pb] += pb != 16
    new_bank = np.take_along_axis(bank, order, 2)
    after = mean_ways(new_bank)
    src = np.take_along_axis(slot_of, order, 2)  # slot whose contents move to position [instance, lane, j]
    dst = slot_of
    ok = (dst >= 0) & (src >= 0)
    # a lane with fewer than S real slots: `order` may pair a real position with an empty source; such lanes do
    # not exist (segments are whole: a lane either has all S slots or none)
    assert np.array_equal(dst >= 0, src >= 0)
    for name in ("col16", "val", "eid"):
        arr = host[name]
        moved = arr[src[ok]]
        arr[dst[ok]] = moved
    return before, after


def sliced_graph_device(host, n_rows, n_groups, row_cap, device):
    """(_lib.SlicedCsr, the device tensors it points to) of sliced_graph_host's arrays."""
    hold = {k: torch.from_numpy(v.view(np.int16) if v.dtype == np.uint16 else v).to(device)
            for k, v in host.items() if isinstance(v, np.ndarray)}
    for name in ("spill_row", "empty_row", "chunks"):   # (an empty tensor has no address; the kernel reads none of it)
        if hold[name].numel() == 0:
            hold[name] = torch.zeros(2, dtype=torch.int32, device=device)
    sc = _lib.SlicedCsr(hold["chunks"].data_ptr(), hold["col16"].data_ptr(), hold["val"].data_ptr(),
                        hold["eid"].data_ptr(), hold["sub_row"].data_ptr(), hold["sub_chunk"].data_ptr(),
                        hold["spill_row"].data_ptr(), hold["spill_ptr"].data_ptr(), hold["empty_row"].data_ptr(),
                        hold["empty_ptr"].data_ptr(), _lib.ptr(hold.get("row_scale")), _lib.ptr(hold.get("col_scale")), n_rows,
                        host["n_slots"], n_groups, host["subs_per_group"], host["n_chunks"], row_cap,
                        host["lane_slots"], host["pad_slot"])
    return sc, hold


def build_sliced_graphs(csr, csr_t, order_t, n, dim, dev, mode="auto", lane_slots=None):
    """{"slice_w": W, "sliced": (SlicedCsr, tensors), "sliced_t": ..., "n_groups", "row_cap"} for a graph and its
    transpose (device CSR triples; order_t: forward edge number of every transposed edge), or {"slice_w": 0} when the
    column-sliced SpMM (csrc/spmm_sliced.hip) does not apply: 65 536 nodes or more, a slice that does not fit the
    LDS, mode "gather".  mode "sliced_values" keeps the edge values in the stream (no rank-one factoring)."""
    if mode not in ("auto", "gather", "sliced_values"):
        raise ValueError(f"unknown spmm mode {mode!r}")
    lib = _lib.load()
    w = int(lib.hiprec_sliced_width(n, dim)) if mode != "gather" else 0
    out = {"slice_w": 0}
    if w == 0:
        return out
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    n_groups = max(8, n_cu // (dim // w) // 8 * 8)  # one (slice, row group) per CU; same row group -> same XCD
    cap = int(lib.hiprec_sliced_row_cap(n, dim))
    for tag, (rowptr, col, val), eid in (("", csr, None), ("_t", csr_t, order_t)):
        host = sliced_graph_host(rowptr.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy(),
                                 None if eid is None else eid.cpu().numpy(), n_groups, cap, factor=mode == "auto",
                                 lane_slots=lane_slots)
        if host is None:
            return {"slice_w": 0}
        spread_bank_conflicts(host, n_groups)
        out["sliced" + tag] = sliced_graph_device(host, n, n_groups, cap, dev)
    out.update(slice_w=w, n_groups=n_groups, row_cap=cap)
    return out


class LightGCN(_FlatModel):
    """models/lightgcn.py:7-101."""

    def __init__(self, config, norm_adj):
        super().__init__()
        self.config = config
        self.n_users, self.n_items = int(config["n_users"]), int(config["n_items"])
        self.emb_dim = int(config["emb_dim"])
        self.layer_size = config["layer_size"]
        self.n_layers = len(self.layer_size)
        self.norm_adj = norm_adj
        self.layer_size = [self.emb_dim] + list(self.layer_size)
        v = self._build([("user_embedding.weight", (self.n_users, self.emb_dim)),
                         ("item_embedding.weight", (self.n_items, self.emb_dim))])
        # RNG order of lightgcn.py:22-25,40-44: two nn.Embedding (N(0,1)) then xavier_uniform_ x2
        v["user_embedding.weight"].normal_(0, 1)
        v["item_embedding.weight"].normal_(0, 1)
        nn.init.xavier_uniform_(v["user_embedding.weight"])
        nn.init.xavier_uniform_(v["item_embedding.weight"])
        self.user_embedding = _ParamView(v["user_embedding.weight"])
        self.item_embedding = _ParamView(v["item_embedding.weight"])
        self.f = nn.Sigmoid()
        self.dropout_rng = config["dropout_rng"] if "dropout_rng" in config else "torch_cpu"
        self.dropout_seed = int(config["dropout_seed"]) if "dropout_seed" in config else 0
        self._graph = None
        self._ws = None
        self._stats = None
        self._step = 0

    # ---- graph + workspace ------------------------------------------------------------------
    def graph(self):
        """CSR and transposed CSR of norm_adj on the parameters' device (built once)."""
        dev = self._flat.device
        if self._graph is not None and self._graph["dev"] == dev:
            return self._graph
        co = self.norm_adj.coalesce().cpu()
        N = self.n_users + self.n_items
        if tuple(co.shape) != (N, N):
            raise ValueError(f"norm_adj is {tuple(co.shape)}, expected ({N}, {N})")
        rows, cols = co.indices()[0], co.indices()[1]
        vals = co.values().to(torch.float32)
        # coalesced COO is already sorted row-major: edge e of the CSR == value e of the COO, which
        # is the order LightGCN.dropout draws its mask in (lightgcn.py:29-35)
        rp, c, v, _ = _csr_from_coo(rows, cols, vals, N, dev)
        rpt, ct, vt, order_t = _csr_from_coo(cols, rows, vals, N, dev)
        nnz = int(vals.numel())
        self._graph = {"dev": dev, "nnz": nnz, "rowptr": rp, "col": c, "val": v,
                       "rowptr_t": rpt, "col_t": ct, "val_t": vt,
                       "eid_t": order_t.to(torch.int32).to(dev)}
        if dev.type == "cuda":
            self._graph["slice_row"] = _slice_rows(rp, c, v, N, nnz)
            self._graph["slice_row_t"] = _slice_rows(rpt, ct, vt, N, nnz)
            self._graph.update(build_sliced_graphs((rp, c, v), (rpt, ct, vt), order_t, N, self.emb_dim, dev,
                                                   self.config.get("spmm", "auto"),
                                                   self.config.get("spmm_lane_slots")))
        return self._graph

    def workspace(self):
        dev = self._flat.device
        if self._ws is not None and self._ws["dev"] == dev:
            return self._ws
        N, D = self.n_users + self.n_items, self.emb_dim
        g = self.graph()
        self._ws = {"dev": dev, "keep": torch.ones(max(g["nnz"], 1), dtype=torch.uint8, device=dev)}
        self._ws["acc"] = torch.zeros(N, D, dtype=torch.float32, device=dev)
        # d_out + one output buffer per SpMM of a step, contiguous: the library clears them with one
        # fill per step (hiprec_lightgcn_plan.zero_ws) instead of one fill launch per SpMM
        self._ws["zero_ws"] = torch.zeros((1 + 2 * self.n_layers) * N * D, dtype=torch.float32, device=dev)
        if g.get("slice_w", 0) > 0:
            slots = g["sliced"][0].n_slots + g["sliced_t"][0].n_slots
            self._ws["sliced_ws"] = torch.empty(4 * N * D + slots, dtype=torch.float32, device=dev)
        return self._ws

    def plan(self, g_flat=None, decay=0.0):
        gr, ws = self.graph(), self.workspace()
        N = self.n_users + self.n_items
        p = _lib.LightGcnPlan()
        p.a = _lib.Csr(gr["rowptr"].data_ptr(), gr["col"].data_ptr(), gr["val"].data_ptr(), None, N, gr["nnz"],
                       _lib.ptr(gr.get("slice_row")))
        p.at = _lib.Csr(gr["rowptr_t"].data_ptr(), gr["col_t"].data_ptr(), gr["val_t"].data_ptr(),
                        gr["eid_t"].data_ptr(), N, gr["nnz"], _lib.ptr(gr.get("slice_row_t")))
        p.n_users, p.n_items, p.dim, p.n_layers = self.n_users, self.n_items, self.emb_dim, self.n_layers
        p.decay = float(decay)
        p.e0 = self._flat.data_ptr()
        p.g = None if g_flat is None else g_flat.data_ptr()
        p.acc = ws["acc"].data_ptr()
        p.zero_ws = ws["zero_ws"].data_ptr()
        p.zero_ws_floats = ws["zero_ws"].numel()
        if gr.get("slice_w", 0) > 0:
            p.sa, p.sat = gr["sliced"][0], gr["sliced_t"][0]
            p.slice_w = gr["slice_w"]
            p.dropped_ready = 1 if getattr(self, "_dropped_ready", False) else 0
            p.sliced_ws = ws["sliced_ws"].data_ptr()
            p.sliced_ws_floats = ws["sliced_ws"].numel()
        return p

    def draw_keep_mask(self):
        """Edge keep bytes of one training step (None in eval mode)."""
        self._dropped_ready = False
        if not self.training:
            self._staged_for = None          # an eval-mode propagation overwrites the staged buffers
            return None
        lib = self._require_hip()
        ws, gr = self.workspace(), self.graph()
        keep_prob = float(self.config["keep_pro"])
        self._step += 1
        st = _lib.stream_ptr(self._flat.device)
        sliced = gr.get("slice_w", 0) > 0  # then the step's dropped edge values are prepared here, in one launch
        if self.dropout_rng == "torch_cpu":
            # lightgcn.py:32-33: (torch.rand(len(values)) + keep_prob).int().bool()
            mask = (torch.rand(gr["nnz"]) + keep_prob).int().bool()
            ws["keep"][: gr["nnz"]].copy_(mask.to(torch.uint8), non_blocking=False)
        elif self.dropout_rng == "device":
            if not sliced:
                _lib.check(lib.hiprec_edge_dropout_mask(_lib.ptr(ws["keep"]), gr["nnz"], keep_prob, self.dropout_seed,
                                                        self._step, st))
        else:
            raise ValueError(f"unknown dropout_rng {self.dropout_rng!r}: 'torch_cpu' or 'device'")
        if sliced:
            device_draw = self.dropout_rng == "device"  # the draw is folded into the launch; no keep bytes written
            if device_draw and getattr(self, "_staged_for", None) == self._stage_key(self._step):
                # the previous step's optimizer launch drew this step's edge streams and laid the fresh E0 out
                # (hiprec_lightgcn_opt_stage), and nothing has written the weights since (torch counts in-place writes)
                self._staged_for = None      # (consumed: the passes overwrite the staged buffers)
                self._dropped_ready = True
                return ws["keep"]
            self._staged_for = None
            plan = self.plan()
            _lib.check(lib.hiprec_lightgcn_step_values(ctypes.byref(plan), None if device_draw else _lib.ptr(ws["keep"]),
                                                       keep_prob, 1 if device_draw else 0, self.dropout_seed,
                                                       self._step, st))
            self._dropped_ready = True
        return ws["keep"]

    def _stage_key(self, step):
        """What a staged next step was prepared FOR: the step number, the dropout draw's seed and keep probability, the
        graph (its sliced layout), and the state of the weights (:meth:`_weights_version`).  Anything else the caller
        changes between two steps makes the step prepare itself (seed / keep_pro used to be left out).
        Library calls that write the weights through raw pointers invalidate explicitly (``_staged_for = None``)."""
        return (step, int(self.dropout_seed), float(self.config["keep_pro"]), id(self.graph())) + self._weights_version()

    def _weights_version(self):
        """Changes whenever torch writes the weights in place -- through the flat buffer or through a parameter (they
        count separately: a parameter's ``.data`` is a view of the flat buffer with a version counter of its own) --
        or the buffer moves.  The library's own kernels write through raw pointers and do not count."""
        return (self._flat._version, self.user_embedding.weight._version, self.item_embedding.weight._version,
                self._flat.data_ptr())

    def can_stage_next_step(self):
        """True when the optimizer launch may also prepare the next training step (sliced plan of width 4, device
        draw, training mode): see LightGCNEngine._enqueue_opt."""
        return (self.training and self.dropout_rng == "device" and self.graph().get("slice_w", 0) == 4
                and self.config.get("stage_next_step", True))

    def last_keep_mask(self):
        """Keep bytes (uint8 [nnz], forward CSR order) of the last training step's edge dropout.  With the device
        draw on the sliced path nothing materialises them during the step: the stateless draw is repeated here."""
        ws, gr = self.workspace(), self.graph()
        if self.dropout_rng == "device" and gr.get("slice_w", 0) > 0:
            _lib.check(self._require_hip().hiprec_edge_dropout_mask(
                _lib.ptr(ws["keep"]), gr["nnz"], float(self.config["keep_pro"]), self.dropout_seed, self._step,
                _lib.stream_ptr(self._flat.device)))
        return ws["keep"][: gr["nnz"]]

    # ---- reference API ---------------------------------------------------------------------
    def forward(self, norm_adj=None):
        """lightgcn.py:46-78 without autograd: ``(u_g_embeddings, i_g_embeddings)``.  The graph is
        the one given at construction (the reference always passes that same tensor)."""
        lib = self._require_hip()
        keep = self.draw_keep_mask()
        plan = self.plan()
        _lib.check(lib.hiprec_lightgcn_propagate(
            ctypes.byref(plan), _lib.ptr(keep), float(self.config["keep_pro"]) if keep is not None else 1.0,
            _lib.stream_ptr(self._flat.device)))
        out = self._ws["acc"] / float(self.n_layers + 1)
        return torch.split(out, [self.n_users, self.n_items])

    def predict(self, users, items):
        """lightgcn.py:80-101: eval mode, full propagation, sigmoid of the dot product."""
        self.eval()
        self._staged_for = None              # the propagation below overwrites what an optimizer launch may have staged
        lib = self._require_hip()
        dev = self._flat.device
        users_t, items_t = (x.to(dev, torch.int64).reshape(-1).contiguous() if torch.is_tensor(x) else
                            torch.as_tensor(np.asarray(x), dtype=torch.int64).to(dev).reshape(-1).contiguous()
                            for x in (users, items))
        if self._stats is None or self._stats.device != dev:
            self._stats = _new_stats(dev)
        plan = self.plan()
        st = _lib.stream_ptr(dev)
        _lib.check(lib.hiprec_lightgcn_propagate(ctypes.byref(plan), None, 1.0, st))
        scores = torch.empty(users_t.numel(), dtype=torch.float32, device=dev)
        _lib.check(lib.hiprec_lightgcn_predict(ctypes.byref(plan), _lib.ptr(users_t), _lib.ptr(items_t),
                                               users_t.numel(), _lib.ptr(scores), _lib.ptr(self._stats), st))
        s = read_stats(self._stats)
        if s.status:
            self._stats = None
            raise_on_status(s.status)
        return scores


class LightGCNEngine(FlatModelEngine):
    """models/lightgcn.py:104-191."""

    def __init__(self, config):
        self.config = config
        self.regs = config["model"]["regs"]
        self.decay = self.regs[0]
        self.norm_adj = config["model"]["norm_adj"]
        self.model = LightGCN(config["model"], self.norm_adj)
        super(LightGCNEngine, self).__init__(config)
        self.model.to(self.device)

    def _enqueue_grad(self, batch_data):
        lib = self._setup()
        m = self.model
        dev = m.flat.device
        users, pos, neg = (torch.as_tensor(x, device=dev).to(torch.int64).reshape(-1).contiguous()
                           for x in batch_data)
        B = users.numel()
        if not (pos.numel() == B and neg.numel() == B):
            raise ValueError("batch tensors differ in length")
        if B == 0:
            raise ValueError("empty batch")
        keep = m.draw_keep_mask()
        plan = m.plan(self._g_flat, self.decay)
        _lib.check(lib.hiprec_lightgcn_grad(
            ctypes.byref(plan), _lib.ptr(keep), float(m.config["keep_pro"]) if keep is not None else 1.0,
            _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), B, self._batch_share() / B, _lib.ptr(self._stats),
            _lib.ptr(self._scratch), self._scratch.numel(), _lib.stream_ptr(dev)))

    def _enqueue_opt(self, fold_partials=True):
        """optimizer.step().  On the sliced path with the device draw the sweep ALSO prepares the next step -- its
        dropped edge streams (a function of seed and step number only) and the sliced layout of the weights it has just
        written -- in the same launch (hiprec_lightgcn_opt_stage: two launches of a step become one; a short launch
        here never costs less than ~5 us).  The model remembers what was staged; a step that finds the weights
        touched in between (load_state_dict, a checkpoint resume) prepares itself as before."""
        m = self.model
        if not m.can_stage_next_step():
            m._staged_for = None
            return super()._enqueue_opt(fold_partials)
        lib, opt = _lib.load(), self.optimizer
        m._dropped_ready = False
        plan = m.plan()
        _lib.check(lib.hiprec_lightgcn_opt_stage(
            ctypes.byref(plan), opt.kind, _lib.ptr(self._g_flat), _lib.ptr(opt.exp_avg), _lib.ptr(opt.exp_avg_sq), opt.lr,
            opt.beta1, opt.beta2, opt.eps, _lib.ptr(self._stats), _lib.ptr(self._scratch) if fold_partials else None,
            float(m.config["keep_pro"]), m.dropout_seed, m._step + 1, _lib.stream_ptr(m.flat.device)))
        m._staged_for = m._stage_key(m._step + 1)

    def train_single_batch(self, batch_data):
        """lightgcn.py:119-152: one step, returns ``batch_mf_loss + batch_reg_loss`` as a float."""
        assert hasattr(self, "model"), "Please specify the exact model !"
        self._enqueue_step(batch_data)
        return self._sync_stats().loss

    def train_an_epoch(self, train_loader, epoch_id):
        """lightgcn.py:154-169: prints the last batch's loss, logs the epoch sum."""
        assert hasattr(self, "model"), "Please specify the exact model !"
        self.model.train()
        lib = self._setup()
        _lib.check(lib.hiprec_stats_begin_epoch(_lib.ptr(self._stats),
                                                _lib.stream_ptr(self.model.flat.device)))
        for batch_data in train_loader:
            self._enqueue_step(batch_data)
        st = self._sync_stats()
        print("[Training Epoch {}], Loss {}".format(epoch_id, st.loss))
        self.writer.add_scalar("model/loss", st.loss_sum, epoch_id)
