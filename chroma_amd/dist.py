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


# ---- the reduction inside the library: RCCL on device arrays --------------------------------------------
def init_comm(ctx, rank=None, world_size=None):
    """Give ``ctx`` (a chroma_amd.gpu context) an RCCL communicator over the ranks of the default
    torch.distributed process group: rank 0 asks the library for an id (chroma_comm_unique_id) and
    torch.distributed carries its 128 bytes to the others -- rendezvous is all torch does here; the
    reductions themselves (chroma_allreduce_hits / chroma_allreduce_daq) run inside the library on the
    device arrays, with no copy through the host.  Without a process group (one GPU) nothing happens."""
    import ctypes
    import torch.distributed as dist
    from chroma_amd import _lib
    if not (dist.is_available() and dist.is_initialized()):
        return False
    rank = dist.get_rank() if rank is None else rank
    world_size = dist.get_world_size() if world_size is None else world_size
    ident = (ctypes.c_uint8 * 128)()
    if rank == 0:
        _lib.check(ctx._lib.chroma_comm_unique_id(ident))
    box = [bytes(ident)]
    dist.broadcast_object_list(box, src=0)
    ident = (ctypes.c_uint8 * 128).from_buffer_copy(box[0])
    _lib.check(ctx._lib.chroma_comm_init(ctx.handle, int(world_size), int(rank), ident))
    return True


def allreduce_channel_hits_device(ctx, counts, earliest):
    """In-place all-reduce of the per-channel device arrays GPUPhotons.channel_hits returned
    (``counts``: sum, ``earliest``: min of non-negative float bit patterns) over the context's
    communicator (init_comm).  The identity on a context without one."""
    from chroma_amd import _lib
    _lib.check(ctx._lib.chroma_allreduce_hits(ctx.handle, counts.ptr, None if earliest is None else earliest.ptr, len(counts)))
    return counts, earliest


def publish_packed_geometry(build, key, local_rank, barrier, shm_dir='/dev/shm'):
    """One geometry per NODE instead of one per process.  The process with ``local_rank`` 0 calls
    ``build()`` -> PackedGeometry (mesh, BVH, optics tables and the derived wide tree: the 30 s of
    all-core work of a 170 M-triangle detector), saves it under ``shm_dir`` as plain .npy files and
    passes ``barrier()``; the other processes pass the barrier and memory-map the same files, so the
    ~16 GB of host arrays exist once per node.  Returns (packed, directory); the caller removes the
    directory (remove_published) after every rank has uploaded."""
    import os
    from chroma_amd.gpu.geometry import PackedGeometry
    path = os.path.join(shm_dir, 'chroma_amd_%s_%d' % (key, os.getuid()))
    if local_rank == 0:
        import shutil
        shutil.rmtree(path, ignore_errors=True)
        packed = build()
        if 'wide_nodes' not in packed.arrays:
            packed.attach_wide_tree()
        packed.save(path)
        barrier()
        return packed, path
    barrier()
    return PackedGeometry.load(path, mmap=True), path


def remove_published(path, local_rank, barrier):
    """After every rank has uploaded (first barrier) local rank 0 deletes the files; nobody returns before
    they are gone (second barrier)."""
    import shutil
    barrier()
    if local_rank == 0:
        shutil.rmtree(path, ignore_errors=True)
    barrier()
