"""Packet sharding across the GPUs of one node.

Packets are independent, so the path shards with no collective inside the computation.  For one shared arrival-ordered
stream (BASELINE configs[3]: one 64k-packet stream over 8 GPUs) the flow per step is

    scatter_stream   the ingest rank hands every rank its contiguous block of packets (descriptors + IQ + chan arenas):
                     one dist.scatter per arena = grouped send/recv over xGMI with backend "nccl" (RCCL); contiguous blocks
                     keep the PDU emission order a plain concatenation of the ranks
    c8p_rx           every rank on its own block, on its own GPU
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


def gather_results(outs_t, pdu_t, n_local, group=None):
    """All-gather the fixed-stride result arenas, device-resident: `outs_t` = this rank's c8p_pkt_out records as bytes,
    `pdu_t` = its PDU arena (uint8 tensors on the backend's device).  Returns (counts [world] int64, outs_all [world, max
    bytes], pdu_all [world, max bytes]); rank r's records are outs_all[r, :counts[r] * 32] and its pdu_off values index
    pdu_all[r].  Two collectives besides the size exchange, no host copies, no per-PDU work."""
    world = dist.get_world_size(group)
    dev = outs_t.device
    sizes = torch.tensor([n_local, outs_t.numel(), pdu_t.numel()], dtype=torch.int64, device=dev)
    all_sizes = torch.zeros(world * 3, dtype=torch.int64, device=dev)
    if _backend_is_gloo(group):
        lst = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(lst, sizes, group=group)
        all_sizes = torch.cat(lst)
    else:
        dist.all_gather_into_tensor(all_sizes, sizes, group=group)
    all_sizes = all_sizes.view(world, 3)
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
        dist.all_gather_into_tensor(out, t, group=group)
        return out.view(world, t.numel())

    return all_sizes[:, 0].clone(), gather(outs_t, m_o), gather(pdu_t, m_p)


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
