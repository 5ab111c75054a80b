"""Stream sharding across GPUs (one process per GPU, `torch.distributed` over RCCL/xGMI for the
control plane only).  Streams are independent, so the data path needs no collective: each rank owns
a contiguous block of streams and its own LPCNetBatch; the only communication is the timing
barrier / max-reduction of the benchmark and an optional gather of results."""
from __future__ import annotations


def partition(n_streams: int, rank: int, world: int):
    """Contiguous block partition: returns (first, count).  Sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world) or n_streams < 0:
        raise ValueError("bad partition arguments")
    base, extra = divmod(n_streams, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def owner(stream: int, n_streams: int, world: int) -> int:
    """Rank that owns `stream` under partition()."""
    base, extra = divmod(n_streams, world)
    edge = extra * (base + 1)
    if stream < edge:
        return stream // (base + 1)
    return extra + (stream - edge) // max(base, 1)
