"""Packet sharding across the GPUs of one node.

Packets are independent, so the path shards with no collective inside the computation.  For one shared arrival-ordered
stream (BASELINE configs[3]: one 64k-packet stream over 8 GPUs) the flow per step is

    StreamScatter    the ingest rank hands every rank its contiguous block of packets (descriptors + IQ + chan arenas) in
                     SUB-BATCHES: exact extents, no padding copies, one grouped send/recv (RCCL over xGMI with backend "nccl")
                     per sub-batch index, posted up front; a rank starts c8p_rx on sub-batch k while k+1 is still in flight, and
                     the ingest rank works on a zero-copy view of its own block meanwhile.  Contiguous blocks keep the PDU
                     emission order a plain concatenation of the ranks.  (scatter_stream: the one-shot form, three dist.scatter)
    c8p_rx           every rank on its own block, on its own GPU, sub-batch by sub-batch as they arrive
    gather_results   the fixed-stride per-packet result arena (c8p_pkt_out) and the PDU arena of every rank, DEVICE-RESIDENT,
                     with all_gather_into_tensor; nothing is unpacked on the host unless someone asks for the PDUs

Everything here works on torch tensors of whatever device the process group's backend wants (cuda for nccl, cpu for gloo
in the tests).  The only host-side objects are the 48-byte packet descriptors, which c8p_rx plans its arenas from.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import abi


def partition(n_pkt, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; block sizes differ by at most one."""
    base, rem = divmod(n_pkt, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def slice_batch(pk, lo, hi):
    """Descriptors of one shard.  Offsets keep pointing into the shared arenas, so nothing is copied."""
    return pk[lo:hi].copy()


class _DevBytes:
    """zero-copy torch view of engine-owned device memory (c8p_buffer pointers) through __cuda_array_interface__"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def device_bytes(ptr, nbytes, device):
    return torch.as_tensor(_DevBytes(ptr, nbytes), device=device)


def _backend_is_gloo(group):
    return dist.get_backend(group) == "gloo"


def _scatter(recv, chunks, src, group):
    dist.scatter(recv, scatter_list=chunks if dist.get_rank(group) == src else None, src=src, group=group)


def block_extents(pk, lo, hi):
    """item ranges [first, last) of a contiguous block of packets inside the iq and chan arenas (antenna-major arenas of a
    two-antenna stream are not contiguous per block: this is the single-arena layout signal / c8p_signal produce)"""
    if hi <= lo:
        return (0, 0), (0, 0)
    b = pk[lo:hi]
    i0, i1 = None, 0
    for a in (0, 1):
        m = b["n_ant"] > a
        if m.any():
            lo_a = int(b["iq_off"][m, a].min())
            i0 = lo_a if i0 is None else min(i0, lo_a)
            i1 = max(i1, int((b["iq_off"][m, a] + b["n_items"][m]).max()))
    return (i0 or 0, i1), (int(b["chan_off"].min()), int(b["chan_off"].max()) + 64)


def scatter_stream(pk_all, iq_all, chan_all, n_total, device, src=0, group=None):
    """Rank `src` holds the arrival-ordered stream (descriptors `pk_all` on the host, `iq_all` / `chan_all` float32 tensors
    viewing complex64 arenas on `device`); every rank gets its contiguous block.  Three scatters of equal-sized chunks (padded
    to the largest block).  Returns (pk_local [host, offsets rebased to the local arenas], iq_local, chan_local)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    hdr = torch.zeros(3, dtype=torch.int64, device=device)
    ext = None
    if rank == src:
        ext = [block_extents(pk_all, *partition(n_total, world, r)) for r in range(world)]
        hdr[0] = max(e[0][1] - e[0][0] for e in ext)
        hdr[1] = max(e[1][1] - e[1][0] for e in ext)
        hdr[2] = max(partition(n_total, world, r)[1] - partition(n_total, world, r)[0] for r in range(world))
    dist.broadcast(hdr, src=src, group=group)
    m_iq, m_ch, m_pk = (int(x) for x in hdr.tolist())
    lo, hi = partition(n_total, world, rank)
    iq_local = torch.empty(2 * m_iq, dtype=torch.float32, device=device)
    ch_local = torch.empty(2 * m_ch, dtype=torch.float32, device=device)
    pk_local = torch.empty(m_pk * abi.PKT_IN_DT.itemsize, dtype=torch.uint8, device=device)
    iq_chunks = ch_chunks = pk_chunks = None
    if rank == src:
        def chunk(t, a, b, m):
            c = t[2 * a:2 * b]
            if b - a == m:
                return c
            out = torch.zeros(2 * m, dtype=t.dtype, device=t.device)
            out[:2 * (b - a)] = c
            return out
        iq_chunks = [chunk(iq_all, e[0][0], e[0][1], m_iq) for e in ext]
        ch_chunks = [chunk(chan_all, e[1][0], e[1][1], m_ch) for e in ext]
        pk_chunks = []
        for r in range(world):
            l, h = partition(n_total, world, r)
            d = pk_all[l:h].copy()
            if h > l:                                   # rebase into the block's own arenas
                for a in (0, 1):
                    m = d["n_ant"] > a
                    d["iq_off"][m, a] -= ext[r][0][0]
                d["chan_off"] -= ext[r][1][0]
            raw = np.zeros(m_pk * abi.PKT_IN_DT.itemsize, np.uint8)
            raw[:d.nbytes] = d.view(np.uint8)
            pk_chunks.append(torch.from_numpy(raw).to(device))
    _scatter(iq_local, iq_chunks, src, group)
    _scatter(ch_local, ch_chunks, src, group)
    _scatter(pk_local, pk_chunks, src, group)
    pk = pk_local.cpu().numpy().view(abi.PKT_IN_DT)[:hi - lo].copy()
    return pk, iq_local, ch_local


def sub_batches(n_local, sub):
    """[a, b) packet ranges of a block of n_local packets cut into sub-batches of `sub` (the last one ragged)"""
    return [(a, min(a + sub, n_local)) for a in range(0, n_local, sub)]


class StreamScatter:
    """The scatter of one arrival-ordered stream as a pipeline (BASELINE configs[3]).

    Per step, on the ingest rank `src`: `begin()` rebases every rank's descriptors to that rank's own arenas, moves them with ONE
    pinned host -> device copy and posts, in this order, the descriptor sends and then — sub-batch index by sub-batch index — one
    grouped batch of sends holding every rank's k-th sub-batch (IQ extent + chan extent, exact sizes, straight out of the stream's
    arenas).  Grouped calls run one after the other on the communicator's stream, so sub-batch 0 of every rank lands first.  On the
    other ranks `begin()` receives the descriptors (device -> pinned host, one copy), derives the extents from them and posts all
    receives into the rank's local arenas; `arrived(k)` then blocks the HOST until sub-batch k is complete (the engine reads its
    inputs on its own streams), which is the moment c8p_rx(k) can be submitted — while k+1 .. are still arriving.  The ingest rank's
    own block is a zero-copy view of the stream: `arrived` returns at once there.  `finish()` joins the sends.
    Works with backend "nccl" (device tensors) and "gloo" (CPU tensors; the CPU tests)."""

    def __init__(self, device, src=0, group=None):
        self.device, self.src, self.group = torch.device(device), src, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self._iq = self._ch = None
        self._works, self._send_works = [], []
        self.subs, self.pk, self.iq, self.chan = [], None, None, None

    def _dev_sync(self):
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()

    def _pin(self, t):
        return t.pin_memory() if self.device.type == "cuda" else t

    @staticmethod
    def _extents(pk):
        """per packet: [first, last) item of its antennas inside the iq arena and of its 64 chan items (vectorised: this runs over the
        whole stream on the ingest rank's critical path)"""
        off, n = pk["iq_off"].astype(np.int64), pk["n_items"].astype(np.int64)
        two = pk["n_ant"] > 1
        first = np.where(two, np.minimum(off[:, 0], off[:, 1]), off[:, 0])
        last = np.where(two, np.maximum(off[:, 0], off[:, 1]), off[:, 0]) + n
        c = pk["chan_off"].astype(np.int64)
        return first, last, c, c + 64

    @staticmethod
    def _rebase(pk, i0, c0):
        """descriptors of a block, offsets moved from the stream's arenas to the block's own (a copy; a view if nothing moves)"""
        if i0 == 0 and c0 == 0:
            return pk
        d = pk.copy()
        off = d["iq_off"]
        off[:, 0] -= np.uint64(i0)
        two = d["n_ant"] > 1
        off[two, 1] -= np.uint64(i0)
        d["chan_off"] -= np.uint32(c0)
        return d

    def begin(self, pk_all, iq_all, chan_all, n_total, sub):
        """pk_all / iq_all / chan_all: the stream on rank src (None elsewhere).  Returns this rank's descriptors (host, offsets into
        self.iq / self.chan, float32 views of complex64 arenas); self.subs = its sub-batches [(a, b)]."""
        world, rank, src = self.world, self.rank, self.src
        isz = abi.PKT_IN_DT.itemsize
        lo, hi = partition(n_total, world, rank)
        n_local = hi - lo
        self.subs = sub_batches(n_local, sub)
        self._works, self._send_works = [], []
        if rank == src:
            first, last, cf, cl = self._extents(pk_all)
            if n_local:
                i0, i1, c0, c1 = int(first[lo:hi].min()), int(last[lo:hi].max()), int(cf[lo:hi].min()), int(cl[lo:hi].max())
            else:
                i0 = i1 = c0 = c1 = 0
            self.pk = self._rebase(pk_all[lo:hi], i0, c0)
            self.iq, self.chan = iq_all[2 * i0:2 * i1], chan_all[2 * c0:2 * c1]          # views: nothing is copied for the own block
            peers = [r for r in range(world) if r != src and partition(n_total, world, r)[1] > partition(n_total, world, r)[0]]
            if peers:
                # the descriptors leave as they are (the receivers rebase their own blocks, all at once, while their items are in
                # flight): ONE pinned host -> device copy of the stream's descriptors, then slices of it
                raw = self._pin(torch.from_numpy(np.ascontiguousarray(pk_all).view(np.uint8).reshape(-1).copy()))
                raw_d = raw.to(self.device, non_blocking=True)
                ops = []
                for r in peers:
                    l_r, h_r = partition(n_total, world, r)
                    ops.append(dist.P2POp(dist.isend, raw_d[l_r * isz:h_r * isz], r, self.group))
                self._send_works += dist.batch_isend_irecv(ops)
                parts = {r: (partition(n_total, world, r)[0], sub_batches(partition(n_total, world, r)[1] - partition(n_total, world, r)[0], sub)) for r in peers}
                for k in range(max(len(sb) for _, sb in parts.values())):
                    ops = []
                    for r in peers:
                        l_r, sb = parts[r]
                        if k >= len(sb):
                            continue
                        a, b = l_r + sb[k][0], l_r + sb[k][1]
                        ops.append(dist.P2POp(dist.isend, iq_all[2 * int(first[a:b].min()):2 * int(last[a:b].max())], r, self.group))
                        ops.append(dist.P2POp(dist.isend, chan_all[2 * int(cf[a:b].min()):2 * int(cl[a:b].max())], r, self.group))
                    self._send_works += dist.batch_isend_irecv(ops)
                self._keep = (raw, raw_d)
            return self.pk
        # ---- a receiving rank
        if n_local == 0:
            self.pk = np.zeros(0, abi.PKT_IN_DT)
            self.iq = self.chan = torch.zeros(0, dtype=torch.float32, device=self.device)
            return self.pk
        raw_d = torch.empty(n_local * isz, dtype=torch.uint8, device=self.device)
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, raw_d, src, self.group)]):
            w.wait()
        raw = self._pin(torch.empty(n_local * isz, dtype=torch.uint8))
        raw.copy_(raw_d, non_blocking=True)                                              # device -> pinned host, one copy
        self._dev_sync()
        pk = raw.numpy().view(abi.PKT_IN_DT)
        first, last, cf, cl = self._extents(pk)
        i0, i1, c0, c1 = int(first.min()), int(last.max()), int(cf.min()), int(cl.max())
        ni, nc = i1 - i0, c1 - c0
        if self._iq is None or self._iq.numel() < 2 * ni:
            self._iq = torch.empty(2 * ni, dtype=torch.float32, device=self.device)
        if self._ch is None or self._ch.numel() < 2 * nc:
            self._ch = torch.empty(2 * nc, dtype=torch.float32, device=self.device)
        self.iq, self.chan = self._iq[:2 * ni], self._ch[:2 * nc]
        for a, b in self.subs:                                                           # every receive is posted before anything else
            j0, j1, d0, d1 = int(first[a:b].min()) - i0, int(last[a:b].max()) - i0, int(cf[a:b].min()) - c0, int(cl[a:b].max()) - c0
            self._works.append(dist.batch_isend_irecv([dist.P2POp(dist.irecv, self.iq[2 * j0:2 * j1], src, self.group),
                                                       dist.P2POp(dist.irecv, self.chan[2 * d0:2 * d1], src, self.group)]))
        self.pk = self._rebase(pk, i0, c0).copy()                                        # (while the items are on their way)
        return self.pk

    def arrived(self, k):
        """blocks the host until the items of sub-batch k are in self.iq / self.chan"""
        if self.rank == self.src or not self._works:
            return
        for w in self._works[k]:
            w.wait()
        self._dev_sync()

    def finish(self):
        """ingest rank: every send has left its arenas (they may be overwritten by the next step's producer)"""
        for w in self._send_works:
            w.wait()
        self._dev_sync()
        self._send_works = []


_SIZES = {}     # (group, n_local, outs bytes, pdu bytes) -> exchanged size table: a stream's steps repeat the same sizes


def gather_results(outs_t, pdu_t, n_local, group=None, reuse_sizes=False, works=None):
    """All-gather the fixed-stride result arenas, device-resident: `outs_t` = this rank's c8p_pkt_out records as bytes,
    `pdu_t` = its PDU arena (uint8 tensors on the backend's device).  Returns (counts [world] int64, outs_all [world, max
    bytes], pdu_all [world, max bytes]); rank r's records are outs_all[r, :counts[r] * 32] and its pdu_off values index
    pdu_all[r].  Two collectives besides the size exchange, no host copies, no per-PDU work.  reuse_sizes: skip the size exchange
    when this rank's sizes are the ones of an earlier call (every rank must pass the same flag: the steps of one stream).
    works: a list — the two collectives are then issued with async_op=True and their Work handles appended: the caller polls
    `is_completed()` before it rewrites the inputs or reads the outputs on the host (no event of its own is needed for that)."""
    world = dist.get_world_size(group)
    dev = outs_t.device
    key = (id(group), world, n_local, outs_t.numel(), pdu_t.numel(), str(dev))
    all_sizes = _SIZES.get(key) if reuse_sizes else None
    if all_sizes is None:
        sizes = torch.tensor([n_local, outs_t.numel(), pdu_t.numel()], dtype=torch.int64, device=dev)
        all_sizes = torch.zeros(world * 3, dtype=torch.int64, device=dev)
        if _backend_is_gloo(group):
            lst = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in range(world)]
            dist.all_gather(lst, sizes, group=group)
            all_sizes = torch.cat(lst)
        else:
            dist.all_gather_into_tensor(all_sizes, sizes, group=group)
        all_sizes = all_sizes.view(world, 3).cpu()
        if reuse_sizes:
            # (valid only if EVERY rank's sizes repeat: a rank whose sizes changed must not find the others skipping the exchange —
            # the flag is for a stream whose steps are identical by construction, bench.py's)
            _SIZES[key] = all_sizes
    m_o, m_p = int(all_sizes[:, 1].max()), int(all_sizes[:, 2].max())

    def padded(t, m):
        if t.numel() == m:
            return t
        out = torch.zeros(max(m, 1), dtype=torch.uint8, device=dev)
        out[:t.numel()] = t
        return out

    def gather(t, m):
        t = padded(t, max(m, 1))
        if _backend_is_gloo(group):
            lst = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(lst, t, group=group)
            return torch.stack(lst)
        out = torch.empty(world * t.numel(), dtype=torch.uint8, device=dev)
        if works is not None:
            works.append(dist.all_gather_into_tensor(out, t, group=group, async_op=True))
        else:
            dist.all_gather_into_tensor(out, t, group=group)
        return out.view(world, t.numel())

    return all_sizes[:, 0].clone(), gather(outs_t, m_o), gather(pdu_t, m_p)      # (counts: a host tensor)


def pdus_of(counts, outs_all, pdu_all, n_total):
    """Unpack gathered arenas into [(global packet index, PDU bytes)] in arrival order — host side, for checks and for a
    consumer that wants Python objects; the gathered tensors themselves are the product."""
    world = len(counts)
    res = []
    for r in range(world):
        n = int(counts[r])
        lo, _ = partition(n_total, world, r)
        outs = outs_all[r, :n * abi.PKT_OUT_DT.itemsize].cpu().numpy().view(abi.PKT_OUT_DT)
        pdu = pdu_all[r].cpu().numpy()
        for p in range(n):
            o = outs[p]
            area = pdu[int(o["pdu_off"]):int(o["pdu_off"]) + int(o["pdu_bytes"])].tobytes()
            for b in abi.split_pdus(area, int(o["n_pdu"])):
                res.append((lo + p, b))
    return res


def count_pdus(counts, outs_all):
    """number of published PDUs / FCS failures over all ranks, computed on the gathered tensor's device"""
    world = len(counts)
    n_pdu = n_fail = 0
    for r in range(world):
        n = int(counts[r])
        rec = outs_all[r, :n * abi.PKT_OUT_DT.itemsize].view(torch.int32).view(n, -1)
        n_pdu += int(rec[:, 0].sum())
        n_fail += int(rec[:, 1].sum())
    return n_pdu, n_fail


# ------------------------------------------------------------------------------------------------ live streams
class LiveShard:
    """K independent live streams over the GPUs of one node: stream g belongs to ONE rank (contiguous blocks, `partition(K, world, rank)`),
    which runs it through its own host/c8p_rawchain handle on its own GPU — detection state, commit point and scan position of a stream
    never leave its rank, so there is no exchange step and no collective in the data path (weak scaling in the number of radios).
    `summary()` all-reduces the counts for a report; the PDUs stay where their radio's consumer is."""

    def __init__(self, n_streams, stream_round_items, device=0, group=None, on_round=None, **rawchain_kw):
        from . import blocks
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lo, self.hi = partition(int(n_streams), self.world, self.rank)
        self.n_streams = int(n_streams)
        self.h = blocks.RawChain(device=device, streams=max(self.hi - self.lo, 1), stream_round_items=stream_round_items, **rawchain_kw) \
            if self.hi > self.lo else None
        self._on_round = on_round
        if self.h is not None:
            self.h.live(self._deliver if on_round else None)
        self.stats = None

    def owner(self, g):
        """rank that serves global stream g"""
        for r in range(self.world):
            lo, hi = partition(self.n_streams, self.world, r)
            if lo <= g < hi:
                return r
        raise IndexError(g)

    def mine(self):
        return range(self.lo, self.hi)

    def _deliver(self, g):
        g["pkt_stream_global"] = g["pkt_stream"].astype(np.int64) + self.lo
        self._on_round(g)

    def push(self, g, items, **kw):
        """items of GLOBAL stream g (must be one of mine())"""
        if not self.lo <= g < self.hi:
            raise ValueError("stream %d belongs to rank %d, not %d" % (g, self.owner(g), self.rank))
        self.h.push(g - self.lo, items, **kw)

    def round(self, wait=True):
        if self.h is not None:
            self.h.round(wait=wait)

    def drain(self):
        self.stats = self.h.drain() if self.h is not None else {"pdus": 0, "packets": 0, "events": 0, "items": 0, "segments": 0}
        return self.stats

    def summary(self, device="cpu"):
        """whole-job counts (all ranks): pdus, packets, events, items"""
        st = self.stats if self.stats is not None else self.drain()
        t = torch.tensor([st["pdus"], st["packets"], st["events"], st["items"]], dtype=torch.int64, device=device)
        if dist.is_initialized() and self.world > 1:
            dist.all_reduce(t, group=self.group)
        return dict(zip(("pdus", "packets", "events", "items"), (int(v) for v in t.cpu())))

    def close(self):
        if self.h is not None:
            self.h.close()
            self.h = None
