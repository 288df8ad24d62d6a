"""Multi-GPU sharding: one process per GPU, photons split by contiguous global id range,
geometry replicated, ONE reduction of the per-channel hit arrays per batch.

The reference has no multi-GPU mode at all.  Photons never interact, so
there is no data-path collective; the only exchange is the final PMT-hit reduction:
``hit_count`` (sum) and ``earliest_time`` (min over non-negative float bit patterns, the
ordering chroma/cuda/daq.cu:5-20 relies on); for a DAQ acquisition over sharded photons also the
integer charge (sum) and the channel histories (bitwise OR), ``allreduce_daq_channels``.  torch.distributed supplies the transport
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


def allreduce_daq_channels(earliest_time_bits, charge_int, histories, device=None):
    """All-reduce the three integer arrays a DAQ acquisition accumulates (chroma/cuda/daq.cu:73-75) when
    its photons were sharded over the ranks: ``earliest_time_bits`` uint32 with MIN (bit patterns of
    non-negative times), ``charge_int`` uint32 with SUM, ``histories`` uint32 with bitwise OR.  OR is not
    a reduction RCCL offers for this use, so the histories are all-gathered (a few hundred KB per rank)
    and OR-ed locally.  Returns NumPy uint32 arrays; the identity without a process group."""
    import torch
    import torch.distributed as dist
    times = torch.from_numpy(np.ascontiguousarray(earliest_time_bits).astype(np.int64))
    charge = torch.from_numpy(np.ascontiguousarray(charge_int).astype(np.int64))
    hist = torch.from_numpy(np.ascontiguousarray(histories).astype(np.int64))
    if device is not None:
        times, charge, hist = times.to(device), charge.to(device), hist.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(times, op=dist.ReduceOp.MIN)
        dist.all_reduce(charge, op=dist.ReduceOp.SUM)
        parts = [torch.empty_like(hist) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, hist)
        for part in parts:
            hist = torch.bitwise_or(hist, part)
    return (times.cpu().numpy().astype(np.uint32), (charge.cpu().numpy() & 0xFFFFFFFF).astype(np.uint32),
            hist.cpu().numpy().astype(np.uint32))
