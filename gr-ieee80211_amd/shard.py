"""Packet sharding across the GPUs of one node.

Packets are independent, so the path shards with no collective inside the computation: rank r takes the r-th
contiguous block of the arrival-ordered batch (keeps PDU emission order a plain concatenation), runs c8p_rx on its
own GPU, and the PDU records are gathered to rank 0.  The only collectives are that gather (RCCL over xGMI with
backend "nccl", gloo on CPU in the tests) plus whatever barrier the caller wants for timing.
"""
import numpy as np
import torch
import torch.distributed as dist


def partition(n_pkt, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; block sizes differ by at most one."""
    base, rem = divmod(n_pkt, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def slice_batch(pk, lo, hi):
    """Descriptors of one shard.  Offsets keep pointing into the shared arenas, so nothing is copied."""
    return pk[lo:hi].copy()


def pack_pdus(pdus_per_packet):
    """[[bytes, ...] per packet] -> (uint8 payload, int32 table of (packet, length))"""
    tab, chunks = [], []
    for p, lst in enumerate(pdus_per_packet):
        for b in lst:
            tab.append((p, len(b)))
            chunks.append(np.frombuffer(b, np.uint8))
    payload = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    return payload, np.array(tab, np.int32).reshape(-1, 2)


def gather_pdus(pdus_per_packet, lo, device=None, group=None):
    """Gather every rank's PDUs on rank 0, in global arrival order.  Returns [(global packet index, bytes)] on rank 0,
    None elsewhere.  Two fixed-shape all_gathers (sizes, then padded payload + table): variable-length data without
    object pickling, usable with both nccl (device tensors) and gloo."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    payload, tab = pack_pdus(pdus_per_packet)
    tab = tab.copy()
    if len(tab):
        tab[:, 0] += lo
    sizes = torch.tensor([len(payload), len(tab)], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    max_b = max(int(s[0]) for s in all_sizes)
    max_t = max(int(s[1]) for s in all_sizes)
    pb = torch.zeros(max(max_b, 1), dtype=torch.uint8, device=dev)
    pt = torch.zeros((max(max_t, 1), 2), dtype=torch.int32, device=dev)
    if len(payload):
        pb[:len(payload)] = torch.from_numpy(payload).to(dev)
    if len(tab):
        pt[:len(tab)] = torch.from_numpy(tab).to(dev)
    all_b = [torch.zeros_like(pb) for _ in range(world)]
    all_t = [torch.zeros_like(pt) for _ in range(world)]
    dist.all_gather(all_b, pb, group=group)
    dist.all_gather(all_t, pt, group=group)
    if rank != 0:
        return None
    out = []
    for r in range(world):
        nb, nt = int(all_sizes[r][0]), int(all_sizes[r][1])
        b = all_b[r][:nb].cpu().numpy().tobytes()
        t = all_t[r][:nt].cpu().numpy()
        o = 0
        for p, n in t:
            out.append((int(p), b[o:o + int(n)]))
            o += int(n)
    return out
