"""Multi-GPU sharding: one process per GPU, photons split by contiguous global id range,
geometry replicated, ONE reduction of the per-channel hit arrays per batch.

The reference has no multi-GPU mode at all.  Photons never interact, so
there is no data-path collective; the only exchange is the final PMT-hit reduction:
``hit_count`` (sum) and ``earliest_time`` (min over non-negative float bit patterns, the
ordering chroma/cuda/daq.cu:5-20 relies on).  torch.distributed supplies the transport
(backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).  Because a photon's
random stream is keyed by its GLOBAL id, results do not depend on the number of ranks.
"""
import numpy as np


def shard_range(nphotons, rank, world_size):
    """Contiguous global photon-id range [begin, end) of ``rank``."""
    chunk = -(-int(nphotons) // int(world_size))
    begin = min(int(nphotons), rank * chunk)
    return begin, min(int(nphotons), begin + chunk)


def allreduce_channel_hits(hit_count, earliest_time_bits, device=None):
    """All-reduce the per-channel arrays over the default process group.

    ``hit_count``: uint32 (n,) summed; ``earliest_time_bits``: uint32 (n,) bit patterns of
    non-negative float32 times (0x7f800000 = no hit), reduced with MIN.  Returns NumPy arrays.
    The payload is a few hundred KB, i.e. latency-bound: one collective per array, no bucketing.
    """
    import torch
    import torch.distributed as dist
    counts = torch.from_numpy(np.ascontiguousarray(hit_count).astype(np.int64))
    times = torch.from_numpy(np.ascontiguousarray(earliest_time_bits).astype(np.int64))
    if device is not None:
        counts, times = counts.to(device), times.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(times, op=dist.ReduceOp.MIN)
    return counts.cpu().numpy().astype(np.uint64), times.cpu().numpy().astype(np.uint32)
