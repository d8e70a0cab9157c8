"""Multi-GPU layer: frames shard across ranks, per-rank voxel maps merge with ONE exchange step.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  The reference is
single-process; this module is the new frame-sharded design of SURVEY.md §8e:

  merge_dense_maps   (dense mean / max modes)
     1. all-gather of the per-rank voxel keys (local id order) -> the union in GLOBAL first-touch order,
        identical on every rank: ids equal the single-process numbering
     2. each rank lays its rows out in union order (bsc_dense_gather; untouched voxels are neutral)
     3. reduce-scatter of the (U,D) accumulators and (U,) counts: rank r ends up owning the r-th
        contiguous slice of the union.  xGMI is point-to-point, so reduce-scatter (all 7 links busy)
        is preferred over a ring all-reduce followed by a broadcast.
     4. colour state (7 B / voxel / rank: slices exchanged by one all-to-all) and the top-down map (gs^2 cells: one MAX
        all-reduce of packed height | rank | colour keys) are merged
        (merge_colour_states: documented rule, or merge_colour_replay: exact; allreduce_heightmap: exact)
  gather_merged_to_root   slices -> one engine that holds the whole memory and can save a loadable directory
  localize_sharded   every rank scans its slice, all-gather of the (Q,K) local winners, K-way merge
                     with the reference's tie order (HDF5 group-name order).

The exact (token-cache) mode is order-defined and does not shard: "replicas only" (DESIGN.md).
"""
import numpy as np
import torch
import torch.distributed as dist

_SENTINEL = (1 << 62)


def pack_keys(keys):
    """(n,3) int32 [row,col,h] -> (n,) int64 sortable code."""
    k = keys.to(torch.int64)
    return (k[:, 0] << 42) | (k[:, 1] << 21) | k[:, 2]


def unpack_keys(codes):
    out = torch.stack([(codes >> 42) & 0x1FFFFF, (codes >> 21) & 0x1FFFFF, codes & 0x1FFFFF], dim=1).to(torch.int32)
    out[codes >= _SENTINEL] = -1         # padding keys fall outside every grid
    return out


def name_key_np(pos):
    """HDF5 link-name order of 'grid_r_c_h' as a sortable tuple (see localize.hip)."""
    def field(v, last):
        s = str(int(v))
        syms = [int(ch) + (1 if last else 0) for ch in s] + [0 if last else 10] * (6 - len(s))
        k = 0
        for x in syms:
            k = k * 11 + x
        return k
    return (field(pos[0], False), field(pos[1], False), field(pos[2], True))


def _world(group):
    if not _active():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def _active():
    """True when a process group exists.  Every exchange below is skipped only when there is NO group: a group of one
    rank still goes through the collectives (identity results), so a single-GPU box exercises the RCCL code path
    (tests/test_gpu_dist.py::test_rccl_world1_*)."""
    return dist.is_available() and dist.is_initialized()


def _all_gather(t, group=None):
    """dist.all_gather -> list of tensors; gloo has no device all_gather, so device tensors are staged through the
    host there (gloo is only used by the CPU / single-GPU tests, RCCL takes the direct path)."""
    rank, world = _world(group)
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        bufs = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(bufs, h, group=group)
        return [b.to(t.device) for b in bufs]
    bufs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(bufs, t, group=group)
    return bufs


def all_gather_ragged(t, group=None):
    """All-gather of 1-D tensors of different lengths -> list of tensors (one per rank)."""
    rank, world = _world(group)
    if not _active():
        return [t]
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [int(s.item()) for s in _all_gather(n, group)]
    m = max(sizes + [1])
    pad = torch.zeros(m, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    bufs = _all_gather(pad, group)
    return [b[:s] for b, s in zip(bufs, sizes)]


def global_id_order(local_codes, group=None):
    """Voxel codes of every rank, each in its LOCAL id order -> the union in GLOBAL first-touch order, identical on
    every rank, padded with sentinels to a multiple of the world size.  -> (codes, n_union, per_rank)

    Frames are sharded in contiguous blocks (rank 0 holds the earliest frames), so the order in which a single process
    would have met the voxels (memory_2.py:888-894, `max_id` numbering) is: rank 0's voxels in their local order, then
    the voxels rank 1 saw that rank 0 did not, in rank 1's local order, and so on — the first occurrence of every code
    in the rank-major concatenation of the local lists."""
    rank, world = _world(group)
    parts = all_gather_ragged(local_codes, group)
    allc = torch.cat(parts)
    if allc.numel():
        uniq, inv = torch.unique(allc, return_inverse=True)
        first = torch.full((uniq.numel(),), allc.numel(), dtype=torch.int64, device=allc.device)
        first.scatter_reduce_(0, inv, torch.arange(allc.numel(), dtype=torch.int64, device=allc.device), reduce="amin")
        union = uniq[torch.argsort(first)]
    else:
        union = allc
    n_union = union.numel()
    per = (n_union + world - 1) // world if n_union else 0
    if per * world > n_union:
        union = torch.cat([union, torch.full((per * world - n_union,), _SENTINEL, dtype=torch.int64, device=union.device)])
    return union, n_union, per


def reduce_scatter_rows(rows, op, per, group=None):
    """rows (world*per, ...) -> this rank's (per, ...) slice of the element-wise reduction."""
    rank, world = _world(group)
    if not _active():
        return rows
    out = torch.empty((per,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    backend = dist.get_backend(group)
    if backend == "gloo":                           # test path: gloo has no reduce_scatter (and no int32 max on device)
        full = rows.cpu().clone()
        dist.all_reduce(full, op=op, group=group)
        out.copy_(full[rank * per:(rank + 1) * per])
    else:
        dist.reduce_scatter_tensor(out, rows.contiguous(), op=op, group=group)
    return out


def merge_colour_states(rgbs, weights, present):
    """Colour state of one voxel set held by several ranks -> one (rgb u8 (n,3), weight f32 (n,)).

    The reference's running mean `c' = trunc((f32(c*w) + r*a) / (w + a)); w' = f32(w + a)` (memory_2.py:895-899) is
    defined by the global point order and truncates at every step, so per-rank results cannot be combined into the
    sequential answer.  The dense-mode rule: a rank's final state (c_r, w_r) enters the SAME update as one
    observation of colour c_r and weight a = w_r, ranks in rank order (= frame order), the first rank that holds the
    voxel initialises it.  Same arithmetic as the chain: the product c*w rounded in f32, the rest in f64, truncating
    store.  rgbs (R,n,3) u8, weights (R,n) f32, present (R,n) bool."""
    R, n = weights.shape
    c = torch.zeros((n, 3), dtype=torch.float64, device=weights.device)
    w = torch.zeros(n, dtype=torch.float32, device=weights.device)
    have = torch.zeros(n, dtype=torch.bool, device=weights.device)
    for r in range(R):
        cr, wr, pr = rgbs[r].to(torch.float64), weights[r], present[r]
        first, cont = pr & ~have, pr & have
        den = w.to(torch.float64) + wr.to(torch.float64)
        num = (c.to(torch.float32) * w[:, None]).to(torch.float64) + cr * wr.to(torch.float64)[:, None]
        cn = torch.trunc(num / torch.where(den > 0, den, torch.ones_like(den))[:, None])
        c = torch.where(first[:, None], cr, torch.where(cont[:, None], cn, c))
        w = torch.where(first, wr, torch.where(cont, den.to(torch.float32), w))
        have = have | pr
    return c.to(torch.uint8), w


def _all_to_all_rows(rows, dest, group=None):
    """rows (n, k) int64 with a destination rank per row -> the rows sent to this rank, concatenated in SOURCE-RANK order,
    each source's rows in their original order.  RCCL: one all_to_all_single with split sizes; gloo (tests): all-gather of
    everything, then selection."""
    rank, world = _world(group)
    order = torch.argsort(dest, stable=True)
    rows, dest = rows[order].contiguous(), dest[order]
    if dist.get_backend(group) == "gloo":
        n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
        sizes = [int(v.item()) for v in _all_gather(n, group)]
        m = max(sizes + [1])
        pad_r = torch.zeros((m, rows.shape[1]), dtype=rows.dtype, device=rows.device)
        pad_d = torch.full((m,), -1, dtype=torch.int64, device=rows.device)
        pad_r[:rows.shape[0]], pad_d[:rows.shape[0]] = rows, dest
        all_r, all_d = _all_gather(pad_r, group), _all_gather(pad_d, group)
        return torch.cat([r[d == rank] for r, d in zip(all_r, all_d)])
    send = torch.bincount(dest, minlength=world).to(torch.int64)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    send_l, recv_l = send.tolist(), recv.tolist()
    out = torch.empty((int(sum(recv_l)), rows.shape[1]), dtype=rows.dtype, device=rows.device)
    dist.all_to_all_single(out, rows, output_split_sizes=recv_l, input_split_sizes=send_l, group=group)
    return out


def merge_colour_replay(engine, union, per, group=None):
    """EXACT colour state of this rank's slice of the merged map ("rgb/weights exact via replay").

    Every rank logged (cell, alpha, rgb) of each point it ingested (engine.point_log_enable before the build).  Frames are
    sharded in contiguous blocks, so the global point order is (rank, local order).  Each record goes to the rank that owns
    its voxel (one all-to-all of 24-byte rows); the owner lines the records up voxel by voxel — arrival is in source-rank
    order and a stable sort by voxel keeps it — and replays the truncating running mean of memory_2.py:888-899 from the
    empty state (bsc_replay_colour): bit for bit what one process leaves.  -> (rgb (per,3) u8, weight (per,) f32)."""
    rank, world = _world(group)
    cells, recs = engine.point_log()
    gs, nh = engine.cfg.grid_size, engine.nh
    ok = cells >= 0
    cells, recs = cells[ok].to(torch.int64), recs[ok]
    rc, h = cells // nh, cells % nh
    codes = ((rc // gs) << 42) | ((rc % gs) << 21) | h
    su, order = torch.sort(union)                                   # union is in first-touch order: sorted view for the lookup
    if codes.numel():
        at = torch.searchsorted(su, codes).clamp_(max=su.numel() - 1)
        if not bool((su[at] == codes).all()):
            raise RuntimeError("merge_colour_replay: the point log holds voxels that are not in the merged union (a log that "
                               "belongs to another map state)")
        upos = order[at]
    else:
        upos = codes
    dest = upos // max(per, 1)
    rows = torch.stack([upos, (recs[:, 0].to(torch.int64) & 0xffffffff) | (recs[:, 1].to(torch.int64) << 32),
                        recs[:, 2].to(torch.int64)], dim=1) if codes.numel() else torch.zeros((0, 3), dtype=torch.int64, device=union.device)
    got = _all_to_all_rows(rows, dest, group) if _active() else rows
    local = got[:, 0] - rank * per
    o = torch.argsort(local, stable=True)
    got, local = got[o], local[o]
    rec = torch.stack([(got[:, 1] & 0xffffffff), (got[:, 1] >> 32) & 0xffffffff, got[:, 2]], dim=1)
    rec = torch.where(rec >= (1 << 31), rec - (1 << 32), rec).to(torch.int32)          # 32-bit words back into int32
    return engine.replay_colour(local.to(torch.int32), rec, per)


def _exchange_slices(rgb, wgt, present, per, group=None):
    """Per-rank colour state laid out in union order (world * per rows) -> what the owner of slice `rank` needs from every
    rank: (world, per, 3) u8, (world, per) f32, (world, per) bool."""
    rank, world = _world(group)
    if dist.get_backend(group) == "gloo":
        lo, hi = rank * per, (rank + 1) * per
        return (torch.stack(_all_gather(rgb, group))[:, lo:hi], torch.stack(_all_gather(wgt, group))[:, lo:hi],
                torch.stack(_all_gather(present, group))[:, lo:hi])
    packed = torch.zeros((world * per, 8), dtype=torch.uint8, device=rgb.device)          # r g b present | weight as 4 bytes
    packed[:, :3] = rgb
    packed[:, 3] = present.to(torch.uint8)
    packed[:, 4:] = wgt.contiguous().view(torch.uint8).view(-1, 4)
    out = torch.empty_like(packed)
    dist.all_to_all_single(out, packed, group=group)                                      # equal splits of `per` rows
    out = out.view(world, per, 8)
    return out[..., :3].contiguous(), out[..., 4:].contiguous().view(torch.float32).view(world, per), out[..., 3] > 0


def allreduce_heightmap(mh, cv, device, group=None):
    """max_height (gs,gs) f64 (-inf empty, else an integer height index) and cv_map (gs,gs,3) u8 of this rank -> the merged
    pair, identical on every rank: per cell the greatest height, among the ranks that reach it the highest one (its points
    come last in the global order, memory_2.py:901-903 `h >= max_height`), and that rank's colour."""
    rank, world = _world(group)
    h = np.where(np.isfinite(mh), mh, -1.0).astype(np.int64) + 1                       # 0: empty
    rgbp = cv[..., 0].astype(np.int64) | (cv[..., 1].astype(np.int64) << 8) | (cv[..., 2].astype(np.int64) << 16)
    key = torch.from_numpy((h << 40) | (np.int64(rank) << 24) | rgbp)
    if dist.get_backend(group) != "gloo":
        key = key.to(device)
    dist.all_reduce(key, op=dist.ReduceOp.MAX, group=group)
    key = key.cpu().numpy()
    hh = key >> 40
    top = np.where(hh > 0, (hh - 1).astype(np.float64), -np.inf)
    colour = np.stack([key & 0xff, (key >> 8) & 0xff, (key >> 16) & 0xff], axis=-1).astype(np.uint8)
    colour[hh == 0] = cv[hh == 0] if world == 1 else 0                                 # empty everywhere: no colour
    return top, colour


def merge_dense_maps(engine, group=None):
    """Merge per-rank dense maps into ONE memory whose rows are distributed: afterwards rank r holds slice r of the
    global voxel set, in global id order (ids of slice r start at r * per_rank).

      ids / positions   global first-touch order (global_id_order) == the single-process numbering
      features, counts  one reduce-scatter of the (U,D) sums (or maxima) and (U,) counts
      rgb, weights      merge_colour_states (documented dense-mode rule)
      top-down map      allreduce_heightmap (exact), replicated on every rank

    `engine` needs: mode, device, cfg.token_dim, keys_tensor(), dense_gather(keys), dense_gather_rgb(keys), export_heightmap(),
    import_heightmap(h, cv), dense_replace(keys, acc, cnt, rgb, weight).  Returns dict(n_union, per_rank, n_local, colour,
    phases_ms, reduce_scatter_bytes_sent_per_rank, chunk_rows): the phases are timed on every rank (a device synchronize per phase —
    microseconds against a merge of milliseconds) so that a bench line can say where the merge's time goes."""
    rank, world = _world(group)
    import os, time
    marks = []
    on_gpu = getattr(getattr(engine, "device", None), "type", "cpu") == "cuda"

    def mark(name):
        if on_gpu:
            torch.cuda.synchronize()
        marks.append((name, time.perf_counter()))

    mark("start")
    keys = engine.keys_tensor()
    mark("keys")
    codes = pack_keys(keys) if keys.numel() else torch.zeros(0, dtype=torch.int64, device=keys.device)
    union, n_union, per = global_id_order(codes, group)
    mark("global_id_order")
    if not _active():
        return dict(n_union=n_union, per_rank=n_union, n_local=n_union)
    ukeys = unpack_keys(union)
    rgb, wgt = engine.dense_gather_rgb(ukeys)
    op = dist.ReduceOp.MAX if engine.mode == "max" else dist.ReduceOp.SUM
    # The (U, D) union of feature rows is never materialised (4.3 GB per rank at 2^20 x 1024): the reduce-scatter runs over row
    # chunks — chunk c of EVERY rank's slice gathered from this rank's map (world x n rows), reduce-scattered into rows
    # [c, c + n) of this rank's slice.  Same bytes on the wire, bounded staging (BSC_MERGE_CHUNK_BYTES, default 1 GiB).
    D = int(engine.cfg.token_dim)
    budget = int(os.environ.get("BSC_MERGE_CHUNK_BYTES", str(1 << 30)))
    # every rank must issue the same number of collectives: the smallest budget of any rank's environment counts
    bt = torch.tensor([budget], dtype=torch.int64, device=ukeys.device if dist.get_backend(group) != "gloo" else "cpu")
    dist.all_reduce(bt, op=dist.ReduceOp.MIN, group=group)
    budget = int(bt.item())
    chunk = max(1, min(per, budget // max(1, world * D * 4)))
    my_acc = torch.empty((per, D), dtype=torch.float32, device=ukeys.device)
    my_cnt = torch.empty(per, dtype=torch.int32, device=ukeys.device)
    have = torch.empty(world * per, dtype=torch.bool, device=ukeys.device)            # this rank holds the voxel (colour exchange)
    uk3 = ukeys.view(world, per, 3)
    for lo in range(0, per, chunk):
        n = min(chunk, per - lo)
        acc, cnt = engine.dense_gather(uk3[:, lo:lo + n].reshape(world * n, 3).contiguous())
        have.view(world, per)[:, lo:lo + n] = (cnt > 0).view(world, n)
        my_acc[lo:lo + n] = reduce_scatter_rows(acc, op, n, group)
        my_cnt[lo:lo + n] = reduce_scatter_rows(cnt, dist.ReduceOp.SUM, n, group)
        del acc, cnt
    mark("gather_rows+reduce_scatter")
    lo, hi = rank * per, (rank + 1) * per
    if getattr(engine, "log_capacity", 0):
        # the ranks kept their points (sub-sampled modes): exact colour state by replay on the voxel's owner
        my_rgb, my_w = merge_colour_replay(engine, union, per, group)
        colour_rule = "replay (exact)"
    else:
        # colour state: 7 bytes per voxel and rank, needed only by the voxel's owner: every rank sends slice s of its
        # (rgb, weight, present) to rank s — one all-to-all of equal splits (gloo, tests: all-gather and slice)
        all_rgb, all_w, all_present = _exchange_slices(rgb, wgt, have, per, group)
        my_rgb, my_w = merge_colour_states(all_rgb, all_w, all_present)
        colour_rule = "per-rank states as observations (approximate)"
    mark("colour")
    # top-down map: gs^2 cells, replicated.  Per cell the LATEST point among those at the greatest height wins
    # (memory_2.py:901-903: `h >= max_height` in point order); points of a higher rank come later, so the winner is the
    # highest rank at the maximum: one MAX all-reduce of (h + 1) << 40 | rank << 24 | rgb carries height, winner and colour.
    mh, cv = engine.export_heightmap()
    top, colour = allreduce_heightmap(mh, cv, engine.device, group)
    mark("heightmap")
    n_local = int((union[lo:hi] < _SENTINEL).sum().item())
    engine.dense_replace(ukeys[lo:lo + n_local].contiguous(), my_acc[:n_local].contiguous(), my_cnt[:n_local].contiguous(),
                         my_rgb[:n_local].contiguous(), my_w[:n_local].contiguous())
    engine.import_heightmap(top, colour)
    mark("replace")
    phases = {n: round(1e3 * (t - marks[i][1]), 3) for i, (n, t) in enumerate(marks[1:])}
    if os.environ.get("BSC_MERGE_TIMING") is not None and rank == 0:
        print("[merge] " + " ".join(f"{n}={v:.1f}ms" for n, v in phases.items()), flush=True)
    # what a rank puts on the wire in the reduce-scatter: the rows of the other ranks' slices (sums + counts)
    sent = int(per) * (D * 4 + 4) * (world - 1)
    return dict(n_union=n_union, per_rank=per, n_local=n_local, colour=colour_rule, phases_ms=phases, chunk_rows=int(chunk),
                reduce_scatter_bytes_sent_per_rank=sent)


def _gather_to_root(t, root, group=None):
    """Same-shape tensor from every rank -> list on `root` (None elsewhere)."""
    rank, world = _world(group)
    stage = dist.get_backend(group) == "gloo" and t.is_cuda
    src = t.cpu() if stage else t
    bufs = [torch.empty_like(src) for _ in range(world)] if rank == root else None
    dist.gather(src, bufs, dst=root, group=group)
    if rank != root:
        return None
    return [b.to(t.device) for b in bufs] if stage else bufs


def gather_merged_to_root(engine, info, root=0, group=None):
    """After merge_dense_maps: collect every rank's slice on `root`, whose engine then holds the WHOLE merged memory
    (ids 0..n_union-1 in global order, features, counts, rgb, weights, top-down map) and can `save_memory` a directory
    that `load_memory` accepts (memory_2.py:1136-1145 / :189-200).  Other ranks keep their slice.  -> True on root."""
    rank, world = _world(group)
    if not _active():
        return True
    per, n_local, D = info["per_rank"], info["n_local"], engine.cfg.token_dim
    dev = engine.device
    keys = torch.full((per, 3), -1, dtype=torch.int32, device=dev)
    acc = torch.zeros((per, D), dtype=torch.float32, device=dev)
    cnt = torch.zeros(per, dtype=torch.int32, device=dev)
    rgb = torch.zeros((per, 3), dtype=torch.uint8, device=dev)
    wgt = torch.zeros(per, dtype=torch.float32, device=dev)
    if n_local:
        k = engine.keys_tensor()
        keys[:n_local] = k
        a, c = engine.dense_gather(k)
        r, w = engine.dense_gather_rgb(k)
        acc[:n_local], cnt[:n_local], rgb[:n_local], wgt[:n_local] = a, c, r, w
    parts = [_gather_to_root(t, root, group) for t in (keys, acc, cnt, rgb, wgt)]
    if rank != root:
        return False
    n = info["n_union"]
    full = [torch.cat(p)[:n].contiguous() for p in parts]        # slices are contiguous blocks of the global order
    mh, cv = engine.export_heightmap()
    engine.dense_replace(*full)
    engine.import_heightmap(mh, cv)
    return True


def name_keys_np(pos):
    """Vectorised name_key_np: (n,3) positions -> three (n,) int64 sort keys (HDF5 link-name order of 'grid_r_c_h')."""
    pos = np.asarray(pos, np.int64).reshape(-1, 3)
    keys = []
    for col, last in ((0, False), (1, False), (2, True)):
        v = pos[:, col]
        nd = 1 + sum((v >= 10 ** e).astype(np.int64) for e in range(1, 6))
        k = np.zeros_like(v)
        for p in range(6):                                  # symbols from the most significant digit
            exp = np.maximum(nd - 1 - p, 0)
            digit = (v // (10 ** exp)) % 10
            sym = np.where(p < nd, digit + (1 if last else 0), 0 if last else 10)
            k = k * 11 + sym
        keys.append(k)
    return keys


def merge_topk(pos_list, sim_list, K):
    """K-way merge of per-rank winners with the reference's order: similarity descending, ties in
    HDF5 group-name order (memory_2.py:665 stable sort over name-sorted keys)."""
    pos = np.concatenate([np.asarray(p).reshape(-1, 3) for p in pos_list])
    sim = np.concatenate([np.asarray(s).reshape(-1) for s in sim_list])
    k0, k1, k2 = name_keys_np(pos)
    order = np.lexsort((k2, k1, k0, -sim.astype(np.float64)))[:K]
    return pos[order], sim[order]


def localize_sharded(engine, q, K=100, radius=None, curr=None, floor=None, group=None):
    """q (Q,D) identical on every rank -> global (Q,<=K,3) positions and (Q,<=K) similarities."""
    rank, world = _world(group)
    pos, sim, cnt = engine.localize(q, K=K, radius=radius, curr=curr, floor=floor)
    if not _active():
        return [pos[i, :cnt[i]] for i in range(len(cnt))], [sim[i, :cnt[i]] for i in range(len(cnt))]
    dev = q.device
    Q = pos.shape[0]
    rec = torch.zeros((Q, K, 4), dtype=torch.float64, device=dev)       # pos exact in f64, sim widened
    rec[..., :3] = torch.from_numpy(pos.astype(np.float64)).to(dev)
    rec[..., 3] = torch.from_numpy(sim.astype(np.float64)).to(dev)
    n = torch.from_numpy(cnt.astype(np.int64)).to(dev)
    recs = torch.stack(_all_gather(rec, group)).cpu().numpy()           # (world, Q, K, 4): one transfer
    ns = torch.stack(_all_gather(n, group)).cpu().numpy()               # (world, Q)
    return merge_topk_batched(recs[..., :3], recs[..., 3], ns, K)


def merge_topk_batched(pos, sim, counts, K):
    """merge_topk for all queries at once.  pos (world, Q, K, 3), sim (world, Q, K), counts (world, Q) valid entries per rank and
    query -> ([(n_q, 3) int32], [(n_q,) float32]): similarity descending, ties in HDF5 group-name order (memory_2.py:665);
    one lexsort over the last axis instead of a Python loop over the queries (256 queries x 8 ranks: 50 ms -> 2 ms)."""
    world, Q = counts.shape
    cand_p = np.asarray(pos).transpose(1, 0, 2, 3).reshape(Q, world * K, 3).astype(np.int64)
    cand_s = np.asarray(sim).transpose(1, 0, 2).reshape(Q, world * K).astype(np.float32)
    valid = (np.arange(K)[None, None, :] < np.asarray(counts)[:, :, None]).transpose(1, 0, 2).reshape(Q, world * K)
    k0, k1, k2 = (k.reshape(Q, world * K) for k in name_keys_np(cand_p.reshape(-1, 3)))
    order = np.lexsort((k2, k1, k0, -cand_s.astype(np.float64), ~valid), axis=-1)[:, :K]
    n_out = np.minimum(valid.sum(axis=1), K)
    rows = np.arange(Q)[:, None]
    sp, ss = cand_p[rows, order], cand_s[rows, order]
    return [sp[qi, :n_out[qi]].astype(np.int32) for qi in range(Q)], [ss[qi, :n_out[qi]] for qi in range(Q)]


def warmup_collectives(device, group=None):
    """Run every collective merge_dense_maps / localize_sharded use once on small buffers, so that RCCL's
    communicator and protocol setup (seconds on first use) is not charged to the first real merge."""
    rank, world = _world(group)
    if not _active():
        return
    all_gather_ragged(torch.arange(rank + 1, dtype=torch.int64, device=device), group)
    for dt in (torch.float32, torch.int32):
        rows = torch.ones((world * 4, 8), dtype=dt, device=device)
        reduce_scatter_rows(rows, dist.ReduceOp.SUM, 4, group)
    reduce_scatter_rows(torch.ones((world * 4, 8), dtype=torch.float32, device=device), dist.ReduceOp.MAX, 4, group)
    _all_gather(torch.zeros((2, 4), dtype=torch.float64, device=device), group)
    _all_gather(torch.zeros(2, dtype=torch.int64, device=device), group)
    _exchange_slices(torch.zeros((world * 2, 3), dtype=torch.uint8, device=device), torch.zeros(world * 2, device=device),
                     torch.zeros(world * 2, dtype=torch.bool, device=device), 2, group)
    allreduce_heightmap(np.full((2, 2), -np.inf), np.zeros((2, 2, 3), np.uint8), device, group)


def shard_frames(n_frames, group=None):
    """Contiguous frame block of this rank: [start, stop)."""
    rank, world = _world(group)
    base, extra = divmod(int(n_frames), world)          # balanced: the first n % world ranks take one frame more
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)
