"""Multi-GPU sharding: one process per GPU, photons split by contiguous global id range,
geometry replicated, ONE reduction of the per-channel hit arrays per batch.

The reference has no multi-GPU mode at all.  Photons never interact, so
there is no data-path collective; the only exchange is the final PMT-hit reduction:
``hit_count`` (sum) and ``earliest_time`` (min over non-negative float bit patterns, the
ordering chroma/cuda/daq.cu:5-20 relies on); for a DAQ acquisition over sharded photons also the
integer charge (sum) and the channel histories (bitwise OR), ``allreduce_daq_channels``.  torch.distributed supplies the transport
(backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).  Because a photon's
random stream is keyed by its GLOBAL id, its history does not depend on the number of ranks (its last bits can, for the
stragglers of a batch: the reference stops re-normalising dir / pol once fewer than 8192 photons of a batch are alive,
chroma/gpu/photon.py:227-230 -- a rank's result is the reference's for that rank's batch).
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


# ---- start-up helpers: a failure on one rank must be a fast failure of ALL ranks, never a hang -----------------
START_TIMEOUT_S = 240          # rendezvous / collectives of the start-up: well under the driver's 600 s


class StartupError(RuntimeError):
    """Raised on EVERY rank when the start-up of a multi-rank run failed on one of them."""


def init_process_group(backend='nccl', timeout_s=START_TIMEOUT_S, **kw):
    """torch.distributed.init_process_group with a timeout well under the driver's limit: a rank that never
    arrives (it died in its imports, its GPU is gone) fails the others after ``timeout_s`` instead of leaving
    them at the rendezvous until something kills the job."""
    import datetime
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=timeout_s), **kw)
    return dist


def _node_leader(local_rank):
    """Global rank of this node's local rank 0 (ranks of a node are contiguous, as torch.distributed.run
    numbers them)."""
    import torch.distributed as dist
    return dist.get_rank() - int(local_rank)


_NODE_GROUP = {}


def _node_group():
    """The process group of THIS node's ranks (None = the default group: one node, the usual case).  With several
    nodes every rank creates every node's group once (new_group is collective) and keeps its own."""
    import os
    import torch.distributed as dist
    world = dist.get_world_size()
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
    if local_world >= world or world % local_world:
        return None
    if 'group' not in _NODE_GROUP:
        for node in range(world // local_world):
            ranks = list(range(node * local_world, (node + 1) * local_world))
            g = dist.new_group(ranks)
            if dist.get_rank() in ranks:
                _NODE_GROUP['group'] = g
    return _NODE_GROUP['group']


def _broadcast_obj(obj, src, group=None):
    """``obj`` of rank ``src`` on every rank of ``group`` (the payload is tiny)."""
    import torch.distributed as dist
    box = [obj]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


def all_agree(ok, group=None):
    """Every rank reports whether its own part worked; returns True on all ranks only if it worked on all.
    (An all-gather of one small object per rank: works on gloo and on nccl alike.)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(ok)
    votes = [None] * dist.get_world_size(group)
    dist.all_gather_object(votes, bool(ok), group=group)
    return all(votes)


# ---- the reduction inside the library: RCCL on device arrays --------------------------------------------
def init_comm(ctx, rank=None, world_size=None):
    """Give ``ctx`` (a chroma_amd.gpu context) an RCCL communicator over the ranks of the default
    torch.distributed process group: rank 0 asks the library for an id (chroma_comm_unique_id) and
    torch.distributed carries its 128 bytes to the others -- rendezvous is all torch does here; the
    reductions themselves (chroma_allreduce_hits / chroma_allreduce_daq) run inside the library on the
    device arrays, with no copy through the host.  Without a process group (one GPU) nothing happens.

    Failure is collective: rank 0 broadcasts (ok, id-or-error) whatever happened to it, so that an RCCL that
    cannot be loaded or an id that cannot be made raises StartupError on EVERY rank instead of leaving the
    others in the broadcast; and every rank learns whether ALL communicators came up (``all_agree``) before
    anyone uses them -- a rank whose chroma_comm_init failed destroys nothing half-made on the others."""
    import ctypes
    import torch.distributed as dist
    from chroma_amd import _lib
    if not (dist.is_available() and dist.is_initialized()):
        return False
    rank = dist.get_rank() if rank is None else rank
    world_size = dist.get_world_size() if world_size is None else world_size
    msg = None
    if rank == 0:
        try:
            ident = (ctypes.c_uint8 * 128)()
            _lib.check(ctx._lib.chroma_comm_unique_id(ident))
            msg = (True, bytes(ident))
        except Exception as exc:
            msg = (False, '%s: %s' % (type(exc).__name__, exc))
    ok, payload = _broadcast_obj(msg, 0)
    if not ok:
        raise StartupError('rank 0 could not create the RCCL unique id: %s' % payload)
    ident = (ctypes.c_uint8 * 128).from_buffer_copy(payload)
    err = None
    try:
        _lib.check(ctx._lib.chroma_comm_init(ctx.handle, int(world_size), int(rank), ident))
    except Exception as exc:
        err = exc
    if not all_agree(err is None):
        if err is None:
            ctx._lib.chroma_comm_destroy(ctx.handle)
        raise StartupError('the library communicator did not come up on every rank%s' % ('' if err is None else ' (here: %s)' % err))
    return True


def allreduce_channel_hits_device(ctx, counts, earliest):
    """In-place all-reduce of the per-channel device arrays GPUPhotons.channel_hits returned
    (``counts``: sum, ``earliest``: min of non-negative float bit patterns) over the context's
    communicator (init_comm).  The identity on a context without one."""
    from chroma_amd import _lib
    _lib.check(ctx._lib.chroma_allreduce_hits(ctx.handle, counts.ptr, None if earliest is None else earliest.ptr, len(counts)))
    return counts, earliest


def _packed_nbytes(packed):
    return int(sum(a.nbytes for a in packed.arrays.values()))


def _pick_publish_dir(nbytes, candidates):
    """First directory of ``candidates`` that exists, is writable and has room for ``nbytes`` (+5 % + 64 MB)."""
    import os
    import shutil
    for cand in candidates:
        try:
            if cand and os.path.isdir(cand) and os.access(cand, os.W_OK) and shutil.disk_usage(cand).free > 1.05 * nbytes + (64 << 20):
                return cand
        except OSError:
            continue
    return None


def publish_packed_geometry(build, key, local_rank, barrier=None, shm_dir=None):
    """One geometry per NODE instead of one per process.  The process with ``local_rank`` 0 calls
    ``build()`` -> PackedGeometry (mesh, BVH, optics tables and the derived wide tree: the 30 s of
    all-core work of a 170 M-triangle detector), saves it as plain .npy files in a fresh directory
    (tempfile.mkdtemp: no predictable name) under the first of ``shm_dir`` (default /dev/shm), $TMPDIR and the
    system's temporary directory that has ROOM for it (checked: an ENOSPC half-way, or a SIGBUS on a later page
    of a mapping, would otherwise be this rank's death and the others' endless wait), and tells the others where;
    they memory-map the same files, so the ~16 GB of host arrays exist once per node.  If no directory has room the
    ranks fall back to building each their own copy.  Returns (packed, directory or None); the caller removes the
    directory (remove_published) after every rank has uploaded.

    Nothing here can leave a rank waiting for one that failed: whatever happens on local rank 0 -- ``build()``
    raising, the save failing -- is caught and BROADCAST, the directory is removed, and every rank raises
    StartupError; a rank that cannot map the files reports it in the vote that follows and all ranks raise.
    (``barrier`` is accepted for compatibility: the broadcast is the synchronisation.)"""
    import os
    import shutil
    import tempfile
    from chroma_amd.gpu.geometry import PackedGeometry
    leader, group = _node_leader(local_rank), _node_group()
    packed, msg = None, None
    if local_rank == 0:
        path = None
        try:
            packed = build()
            if 'wide_nodes' not in packed.arrays:
                packed.attach_wide_tree()
            where = _pick_publish_dir(_packed_nbytes(packed), [shm_dir or '/dev/shm', os.environ.get('TMPDIR'), tempfile.gettempdir()])
            if where is None:
                msg = ('rebuild', 'no directory with %.1f GB free: every rank builds its own geometry' % (_packed_nbytes(packed) / 1e9))
            else:
                path = tempfile.mkdtemp(prefix='chroma_amd_%s_' % key, dir=where)
                packed.save(path)
                msg = ('ok', path)
        except BaseException as exc:          # (also MemoryError / KeyboardInterrupt: the others must hear of it)
            if path:
                shutil.rmtree(path, ignore_errors=True)
            msg = ('fail', '%s: %s' % (type(exc).__name__, exc))
    status, payload = _broadcast_obj(msg, leader, group)
    if status == 'fail':
        raise StartupError('local rank 0 could not build / publish the geometry: %s' % payload)
    if status == 'rebuild':
        if packed is None:
            packed = build()
            if 'wide_nodes' not in packed.arrays:
                packed.attach_wide_tree()
        return packed, None
    err = None
    if packed is None:
        try:
            packed = PackedGeometry.load(payload, mmap=True)
        except Exception as exc:
            err = exc
    if not all_agree(err is None, group):
        if local_rank == 0:
            shutil.rmtree(payload, ignore_errors=True)
        raise StartupError('the published geometry could not be mapped on every rank%s' % ('' if err is None else ' (here: %s)' % err))
    return packed, payload


def remove_published(path, local_rank, barrier=None):
    """After every rank has uploaded (first barrier) local rank 0 deletes the files; nobody returns before
    they are gone (second barrier).  ``path`` None (the per-rank fallback): nothing to remove."""
    import shutil
    import torch.distributed as dist
    if barrier is None:
        barrier = dist.barrier
    barrier()
    if local_rank == 0 and path:
        shutil.rmtree(path, ignore_errors=True)
    barrier()
