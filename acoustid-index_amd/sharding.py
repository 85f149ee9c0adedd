"""Segment sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL/xGMI).

Segments are the natural shard unit: FileSegment.search reads only its own blocks and the per-hash caps are per
(hash, segment) (src/FileSegment.zig:153-174); with supersession resolved per posting every document's score comes
from exactly one segment, hence from exactly one rank.  Per batch every rank
  1. probes ITS segments for the whole query batch -> per-query table of (id, score) with score >= min_score,
     already ordered (score desc, id asc) and truncated to `limit` (fpx_search_resident_partial),
  2. all-gathers the fixed-shape tables [B][limit]{u32 id, u32 score} + [B] counts (RCCL, ~0.3 MB per rank),
  3. merges them: k-way merge, relative cut-off anchored on the GLOBAL best score, truncate (fpx_merge_partials).
No dense per-doc table ever crosses a link."""
import numpy as np


def assign_segments(weights, world):
    """Greedy balance of segments over ranks by weight (blocks or bytes).  Returns rank per segment.
    Equal weights degrade to round-robin (segment s -> rank s % world)."""
    load = [0] * world
    owner = [0] * len(weights)
    for s in sorted(range(len(weights)), key=lambda i: (-weights[i], i)):
        r = min(range(world), key=lambda k: (load[k], k))
        owner[s] = r
        load[r] += weights[s]
    return owner


def gather_tables(dist, table, counts, world, group=None):
    """all-gather of the per-rank tables.  `table`: [B, cap, 2] int32 tensor, `counts`: [B] int32 tensor, on the
    device the process group works on (cuda for nccl/RCCL, cpu for gloo).  Returns ([world,B,cap,2], [world,B])."""
    import torch
    # concatenation along dim 0 is the layout every backend (RCCL and gloo) accepts; viewed rank-major afterwards
    tables = torch.empty((world * table.shape[0],) + tuple(table.shape[1:]), dtype=table.dtype, device=table.device)
    cnts = torch.empty((world * counts.shape[0],), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(tables, table.contiguous(), group=group)
    dist.all_gather_into_tensor(cnts, counts.contiguous(), group=group)
    tables = tables.view((world,) + tuple(table.shape))
    cnts = cnts.view((world,) + tuple(counts.shape))
    return tables, cnts


class ShardedReader:
    """IndexReader over a snapshot whose file segments are split across the ranks of a process group."""

    def __init__(self, fpx, ctx, reader, dist, world, host_staged=False):
        self.fpx, self.ctx, self.reader, self.dist, self.world = fpx, ctx, reader, dist, world
        self.host_staged = host_staged      # debugging aid: exchange through host memory with a CPU backend (gloo)
        self._bufs = {}
        import torch
        # worker threads start with HIP device 0 current: every tensor names the context's device explicitly
        self.device = torch.device("cuda", ctx.device)

    def partial(self, qb):
        """stage 1: probe the local segments; the per-query tables stay in HBM"""
        import torch
        key = (qb.B, qb.cap)
        if key not in self._bufs:
            self._bufs[key] = (torch.zeros((qb.B, qb.cap, 2), dtype=torch.int32, device=self.device),
                               torch.zeros((qb.B,), dtype=torch.int32, device=self.device))
            torch.cuda.current_stream(self.device).synchronize()          # (torch fills on ITS stream; libfpx writes on a stream of its own)
        d_part, d_cnt = self._bufs[key]
        return self.fpx.search_resident_partial(self.reader, qb, d_part.data_ptr(), d_cnt.data_ptr())

    def gather_merge(self, qb, out=None, out_n=None):
        """stages 2 + 3: all-gather of the tables, merge.  Every rank must call this in the same order."""
        import torch
        d_part, d_cnt = self._bufs[(qb.B, qb.cap)]
        if self.host_staged:
            tables, cnts = gather_tables(self.dist, d_part.cpu(), d_cnt.cpu(), self.world)
            tables, cnts = tables.to(self.device), cnts.to(self.device)
        else:
            tables, cnts = gather_tables(self.dist, d_part, d_cnt, self.world)      # RCCL all-gather over xGMI
        # The collective is ordered on torch's current stream; wait for THAT stream only -- a device-wide synchronize would
        # also wait for the next batch's probe kernels, which run on libfpx's own streams from another host thread.
        torch.cuda.current_stream(self.device).synchronize()
        return self.fpx.merge_partials(self.ctx, qb, tables.data_ptr(), cnts.data_ptr(), self.world, out, out_n)

    def search_resident(self, qb, out=None, out_n=None):
        st = self.partial(qb)
        out, out_n = self.gather_merge(qb, out, out_n)
        return out, out_n, st


# ---- hash-range sharding of single segments (second mode) -------------------------------------------------
HALO_BLOCKS = 3        # MAX_BLOCKS_PER_HASH - 1 (src/FileSegment.zig:25): a run starting in the last owned block stays local


def split_by_hash(blocks, block_size, block_index, parts):
    """Cut one segment into `parts` hash-range slices at block boundaries.  Returns a list of
    (blocks_slice, index_slice, lo_excl, hi_incl): the owned blocks plus the halo, and the hash window (None = open)
    that fpx_segment_create_file_slice takes."""
    blocks = np.asarray(blocks, dtype=np.uint8)
    block_index = np.asarray(block_index, dtype=np.uint32)
    nb = len(block_index)
    cuts = [nb * i // parts for i in range(parts + 1)]
    out = []
    for i in range(parts):
        b0, b1 = cuts[i], cuts[i + 1]
        e = min(nb, b1 + HALO_BLOCKS)
        lo = None if b0 == 0 else int(block_index[b0 - 1])
        hi = None if b1 == nb else (int(block_index[b1 - 1]) if b1 > 0 else 0)
        if b1 == b0:                     # more parts than blocks: an empty window
            lo, hi = (0, 0) if lo is None else (lo, lo)
        out.append((blocks[b0 * block_size:e * block_size], block_index[b0:e], lo, hi))
    return out


def exchange_records(dist, records, counts, world, group=None):
    """all-to-all of the hit records: `records` is an int64 tensor grouped by destination rank, counts[r] records for
    rank r (as fpx_probe_resident returns them).  Returns the records this rank received.  Works on cuda tensors over
    RCCL and on cpu tensors over gloo."""
    import torch
    send = torch.tensor([int(c) for c in counts], dtype=torch.int64, device=records.device)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)                     # how much will arrive from every rank
    in_splits = [int(c) for c in counts]
    out_splits = [int(c) for c in recv.cpu().tolist()]
    got = torch.empty((sum(out_splits),) + tuple(records.shape[1:]), dtype=records.dtype, device=records.device)
    dist.all_to_all_single(got, records[:sum(in_splits)].contiguous(), out_splits, in_splits, group=group)
    return got


def exchange_bins(dist, send, send_counts, group=None):
    """all-to-all of the bins of fpx_shard_probe: `send` [world, bpr, cell_cap] int64 and `send_counts` [world, bpr] int32 --
    row r (the bins whose queries rank r finishes) travels to rank r.  Fixed shapes: one collective each, nothing to agree on
    first.  Returns (recv, recv_counts) of the same shapes, row s = what rank s sent."""
    import torch
    recv, recv_counts = torch.empty_like(send), torch.empty_like(send_counts)
    dist.all_to_all_single(recv.view(-1), send.view(-1), group=group)
    dist.all_to_all_single(recv_counts.view(-1), send_counts.view(-1), group=group)
    return recv, recv_counts


class HashShardedReader:
    """IndexReader over a snapshot of hash-range SLICES -- this rank's window of the hash space of every segment.
    A rank makes, sorts and probes only the query hashes of its window.

    Two protocols with the same results.  The BIN protocol (fpx_shard_probe / fpx_shard_score), for snapshots made of groups of
    direct-addressed slices: the records are dropped into the batch's bins of 8 queries as they are produced, the bins are
    dealt to the ranks in contiguous runs and travel with ONE all-to-all of fixed shape, and every rank FINISHES the queries
    of its bins -- no table exchange, no merge; the rank holds the final results of its share of the batch.  The RECORD
    protocol (fpx_probe_resident / fpx_score_partial: records by doc & (world - 1), a size exchange, the records, partial tables,
    all-gather, merge) for everything else."""

    def __init__(self, fpx, ctx, reader, dist, world, host_staged=False, bins=True, group_world=None, rank=None):
        self.fpx, self.ctx, self.reader, self.dist, self.world = fpx, ctx, reader, dist, world
        # group_world: size of the process group the collectives run in, when it is not `world` -- one GPU playing rank 0 of
        # `world` (bench.py's FPX_BENCH_EMULATE_WORLD): the bins are cut for `world` ranks, the 1-rank all-to-all hands them
        # straight back (same volume and shape as a rank's real receive buffer; the results are not a search's)
        self.group_world = world if group_world is None else group_world
        self.rank = (dist.get_rank() if self.group_world == world and world > 1 else 0) if rank is None else rank
        self.host_staged = host_staged      # debugging aid: exchange through host memory with a CPU backend (gloo)
        self.bins = bins
        self._rec = None
        self._bufs = {}
        self._binbufs = {}
        self.cell_cap = 0
        import torch
        self.device = torch.device("cuda", ctx.device)

    def _tables(self, qb):
        import torch
        key = (qb.B, qb.cap)
        if key not in self._bufs:
            self._bufs[key] = (torch.zeros((qb.B, qb.cap, 2), dtype=torch.int32, device=self.device),
                               torch.zeros((qb.B,), dtype=torch.int32, device=self.device))
            torch.cuda.current_stream(self.device).synchronize()          # (torch fills on ITS stream; libfpx writes on a stream of its own)
        return self._bufs[key]

    # ---- bin protocol
    def partial(self, qb):
        """stage 1 (any thread): this rank's records into the send bins.  Returns the stats, or None when the snapshot
        does not qualify for the bin protocol (search_resident then takes the record protocol)."""
        import torch
        fpx = self.fpx
        if not self.bins:
            return None
        bpr = fpx.shard_bins_per_rank(qb.B, self.world)
        if self.cell_cap == 0:
            self.cell_cap = 2048
        key = (qb.B, self.cell_cap)
        if key not in self._binbufs:
            self._binbufs = {key: (torch.empty((self.world, bpr, self.cell_cap), dtype=torch.int64, device=self.device),
                                   torch.zeros((self.world, bpr), dtype=torch.int32, device=self.device))}
            torch.cuda.current_stream(self.device).synchronize()      # (torch fills on ITS stream; libfpx writes on a stream of its own)
        send, send_counts = self._binbufs[key]
        try:
            st, need = fpx.shard_probe(self.reader, qb, self.world, send.data_ptr(), self.cell_cap, send_counts.data_ptr())
        except fpx.FpxError as e:
            if e.status == -4:                  # FPX_E_INVAL: not a snapshot of groups alone, or a batch with a legacy floor
                if "score floor" not in str(e):
                    self.bins = False
                return None
            raise
        if st is None:
            # A bin outgrew the buffer.  The ranks must run the exchange with ONE bin size and each only knows its own need: this
            # rank still takes part in the step's exchange, with every count it sends marked "needs `need` cells" -- every rank
            # receives a piece from every sender, so all of them see the same largest mark (fpx_shard_score: ShardCellsTooSmall)
            # and redo the step with it (gather_merge).  No extra collective, no rank left waiting in one.
            send_counts.fill_(fpx.SHARD_NEED_MARK | int(need))
            torch.cuda.current_stream(self.device).synchronize()
            st = fpx.Stats()
        self._cap_of_step = self.cell_cap
        return st

    def gather_merge(self, qb, out=None, out_n=None):
        """stages 2 + 3 (every rank in the same order): all-to-all of the bins, then this rank's queries are finished.  Fills
        the rank's rows of `out` / `out_n` (the batch's arrays); self.last_range = (q_lo, q_hi) says which."""
        import torch
        fpx = self.fpx
        for attempt in range(4):
            send, send_counts = self._binbufs[(qb.B, self.cell_cap)]
            if self.host_staged:
                recv, recv_counts = exchange_bins(self.dist, send.cpu(), send_counts.cpu())
                recv, recv_counts = recv.to(self.device), recv_counts.to(self.device)
            else:
                recv, recv_counts = exchange_bins(self.dist, send, send_counts)      # RCCL all-to-all over xGMI
            torch.cuda.current_stream(self.device).synchronize()
            try:
                out, out_n, q_lo, q_hi = fpx.shard_score(self.ctx, qb, self.world, self.rank, recv.data_ptr(), self.cell_cap, recv_counts.data_ptr(), out, out_n)
            except fpx.ShardCellsTooSmall as e:
                # some rank's bins outgrew the step's size: EVERY rank is here now, with the same number; probe again, exchange again
                self.cell_cap = max(int(e.need), self.cell_cap + 1)
                self._last_stats = self.partial(qb)
                continue
            self.last_range = (q_lo, q_hi)
            return out, out_n
        raise RuntimeError("the bins' size did not settle in four exchanges")

    # ---- record protocol (any snapshot)
    def _search_records(self, qb, out, out_n):
        import torch
        fpx = self.fpx
        if self._rec is None:
            self._rec = torch.empty((1 << 20,), dtype=torch.int64, device=self.device)
        while True:
            try:
                counts, st = fpx.probe_resident(self.reader, qb, self.world, self._rec.data_ptr(), self._rec.numel())
                break
            except fpx.FpxError:
                need = 0 if self._rec.numel() >= (1 << 34) else self._rec.numel() * 4
                if not need:
                    raise
                self._rec = torch.empty((need,), dtype=torch.int64, device=self.device)
        if self.host_staged:
            got = exchange_records(self.dist, self._rec[:int(sum(int(c) for c in counts))].cpu(), counts, self.world).to(self.device)
        else:
            got = exchange_records(self.dist, self._rec, counts, self.world)
        torch.cuda.current_stream(self.device).synchronize()
        d_part, d_cnt = self._tables(qb)
        fpx.score_partial(self.ctx, qb, got.data_ptr(), got.numel(), d_part.data_ptr(), d_cnt.data_ptr())
        if self.host_staged:
            tables, cnts = gather_tables(self.dist, d_part.cpu(), d_cnt.cpu(), self.world)
            tables, cnts = tables.to(self.device), cnts.to(self.device)
        else:
            tables, cnts = gather_tables(self.dist, d_part, d_cnt, self.world)
        torch.cuda.current_stream(self.device).synchronize()
        out, out_n = fpx.merge_partials(self.ctx, qb, tables.data_ptr(), cnts.data_ptr(), self.world, out, out_n)
        return out, out_n, st

    def search_resident(self, qb, out=None, out_n=None):
        """bin protocol: this rank's queries (self.last_range) are final in out / out_n; record protocol: all of them"""
        st = self.partial(qb)
        if st is None:
            self.last_range = (0, qb.B)
            return self._search_records(qb, out, out_n)
        out, out_n = self.gather_merge(qb, out, out_n)
        return out, out_n, st


class RoutedShardedReader:
    """The protocol that scales (fpx_shard_keys / fpx_shard_probe_keys / fpx_shard_score_share): this rank holds its window of
    the hash space of every segment AND only its share of the batch -- the queries it will finish.  Five stages per step:
      keys(share)      any thread    the share's keys, dealt to the ranks' windows (slot w of the send buffer)
      exchange_keys()  IN STEP ORDER all-to-all #1: slot w and its count to rank w
      probe()          any thread    the received slots -> the batch's bins
      exchange_bins()  IN STEP ORDER all-to-all #2: the bins to the rank whose queries they hold
      score(out, n)    any thread    the final results of the share's queries
    The two exchanges are collectives: every rank must issue them in the same order (bench.py does so from one thread).  Slot and
    bin sizes are agreed without a collective of their own: keys -- an all-reduce(MAX) rides in exchange_keys only when a rank flagged
    an overflow through its counts; bins -- the marked counts of fpx_shard_score (ShardCellsTooSmall)."""

    def __init__(self, fpx, ctx, reader, dist, world, rank=None, group_world=None, host_staged=False):
        import torch
        self.fpx, self.ctx, self.reader, self.dist, self.world = fpx, ctx, reader, dist, world
        self.group_world = world if group_world is None else group_world      # 1: ONE GPU plays `rank` of `world` (bench.py's emulation)
        self.rank = (dist.get_rank() if self.group_world == world and world > 1 else 0) if rank is None else rank
        self.host_staged = host_staged
        self.device = torch.device("cuda", ctx.device)
        self.key_cap = 0
        self._key_cap_agreed = False      # the ranks' first guesses differ (each only knows its own share): agreed once, in exchange_keys
        self.cell_cap = 0
        self._k = self._b = None
        self.last_stats = None

    def _key_bufs(self):
        import torch
        if self._k is None or self._k[0].shape[1] != self.key_cap:
            self._k = (torch.zeros((self.world, self.key_cap), dtype=torch.int64, device=self.device),
                       torch.zeros((self.world,), dtype=torch.int64, device=self.device))
            torch.cuda.current_stream(self.device).synchronize()
        return self._k

    def _bin_bufs(self, bpr):
        import torch
        if self._b is None or self._b[0].shape[1:] != (bpr, self.cell_cap):
            self._b = (torch.empty((self.world, bpr, self.cell_cap), dtype=torch.int64, device=self.device),
                       torch.zeros((self.world, bpr), dtype=torch.int32, device=self.device))
            torch.cuda.current_stream(self.device).synchronize()
        return self._b

    def keys(self, share, B_global):
        fpx = self.fpx
        self.share, self.B = share, B_global
        self.bpr = fpx.shard_bins_per_rank(B_global, self.world)
        if self.key_cap == 0:
            per = int(share.offsets[-1]) // max(1, self.world)
            self.key_cap = per + per // 16 + 1024
        self.key_need = 0
        keys, kcnt = self._key_bufs()
        need = fpx.shard_keys(self.ctx, share, self.world, self.rank, B_global, keys.data_ptr(), self.key_cap, kcnt.data_ptr())
        if need:
            # a slot outgrew the buffer: the counts this rank sends say so (FPX_SHARD_NEED_MARK | need), every receiver sees the mark in
            # exchange_keys, and all ranks redo the keys with the largest need
            self.key_need = int(need)
            kcnt.fill_(fpx.SHARD_NEED_MARK | int(need))
            import torch
            torch.cuda.current_stream(self.device).synchronize()

    def _a2a(self, send):
        import torch
        if self.host_staged:
            s = send.cpu()
            r = torch.empty_like(s)
            self.dist.all_to_all_single(r.view(-1), s.view(-1))
            return r.to(self.device)
        r = torch.empty_like(send)
        self.dist.all_to_all_single(r.view(-1), send.view(-1))
        return r

    def _agree_key_cap(self):
        """The all-to-all of the key slots has ONE shape on every rank, and a rank's first guess comes from its own share (shares
        differ: variable query lengths, a last rank with fewer queries).  One all-reduce(MAX) of (guess, need) the first time --
        afterwards the size only changes through marked counts, which every rank sees alike."""
        import torch
        t = torch.tensor([int(self.key_cap), int(self.key_need)], dtype=torch.int64, device="cpu" if self.host_staged else self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        cap = int(t.max().item())
        if cap != self.key_cap or self.key_need:
            self.key_cap = cap
            self.keys(self.share, self.B)
        self._key_cap_agreed = True

    def exchange_keys(self):
        import torch
        fpx = self.fpx
        if not self._key_cap_agreed:
            self._agree_key_cap()
        for attempt in range(4):
            keys, kcnt = self._key_bufs()
            self.recv_keys, self.recv_kcnt = self._a2a(keys), self._a2a(kcnt)
            torch.cuda.current_stream(self.device).synchronize()
            marks = self.recv_kcnt.cpu().numpy()
            flagged = marks[marks >= fpx.SHARD_NEED_MARK]
            if len(flagged) == 0:
                return
            # every rank received a count from every sender: all of them see the same marks and take the same new size
            self.key_cap = max(self.key_cap + 1, int((flagged & (fpx.SHARD_NEED_MARK - 1)).max()))
            self.keys(self.share, self.B)
        raise RuntimeError("the key slots' size did not settle in four exchanges")

    def emulate_received_keys(self):
        """ONE GPU playing a rank of `world` (group_world == 1): what came back from the 1-rank exchange are this rank's own slots --
        slot w holds the keys of window w.  A real rank receives world slots of ITS window, one per source: the stand-in moves every
        slot's keys into this rank's window (the hash's top bits) and into source w's query range, so that the probe kernel does a real
        rank's work on them (the results are not a search's)."""
        import torch
        wb = int(self.world).bit_length() - 1
        qbits = max(0, int(self.B - 1).bit_length())
        if wb == 0:
            return
        hi_shift = qbits + 32 - wb
        keep = ~(((1 << wb) - 1) << hi_shift)                     # (as a signed 64-bit mask: bit 63 -- a flagged duplicate -- stays)
        share_n = self.share.B
        # two in-place passes over the keys (the window's bits cleared, then rank bits + the source's query offset added: the cleared
        # bits are zero, so the sum is the OR); the stand-in should cost the emulated step as little as it can -- a real rank has no such step
        add = (torch.arange(self.world, device=self.recv_keys.device, dtype=torch.int64) * share_n - self.rank * share_n + (int(self.rank) << hi_shift))[:, None]
        self.recv_keys.bitwise_and_(keep).add_(add)
        torch.cuda.current_stream(self.device).synchronize()

    def probe(self):
        fpx = self.fpx
        if self.cell_cap == 0:
            self.cell_cap = 2048
        send, counts = self._bin_bufs(self.bpr)
        st, need = fpx.shard_probe_keys(self.reader, self.recv_keys.data_ptr(), self.key_cap, self.recv_kcnt.data_ptr(), self.world, self.B,
                                        send.data_ptr(), self.cell_cap, counts.data_ptr())
        if st is None:
            import torch
            counts.fill_(fpx.SHARD_NEED_MARK | int(need))           # (see HashShardedReader.partial)
            torch.cuda.current_stream(self.device).synchronize()
            st = fpx.Stats()
        self.last_stats = st
        return st

    def exchange_bins(self):
        import torch
        send, counts = self._bin_bufs(self.bpr)
        self.recv_bins, self.recv_counts = self._a2a(send), self._a2a(counts)
        torch.cuda.current_stream(self.device).synchronize()

    def score(self, out=None, out_n=None):
        """the share's final results; on ShardCellsTooSmall every rank raises it for the same step: redo probe / exchange_bins / score"""
        fpx = self.fpx
        out, out_n, q_lo, q_hi = fpx.shard_score_share(self.ctx, self.share, self.world, self.rank, self.B, self.recv_bins.data_ptr(), self.cell_cap,
                                                       self.recv_counts.data_ptr(), out, out_n)
        self.last_range = (q_lo, q_hi)
        return out, out_n

    def search(self, share, B_global, out=None, out_n=None, emulate=False):
        """one step, all five stages in turn (tests; bench.py pipelines them)"""
        self.keys(share, B_global)
        self.exchange_keys()
        if emulate:
            self.emulate_received_keys()
        for attempt in range(4):
            st = self.probe()
            self.exchange_bins()
            try:
                out, out_n = self.score(out, out_n)
                return out, out_n, st
            except self.fpx.ShardCellsTooSmall as e:
                self.cell_cap = max(int(e.need), self.cell_cap + 1)
        raise RuntimeError("the bins' size did not settle in four exchanges")
