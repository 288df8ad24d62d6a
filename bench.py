#!/usr/bin/env python
"""bench.py -- photons/s of the propagate_hit path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic photons already resident in HBM:
GPUPhotons.propagate to completion (max_steps) + per-channel hit reduction (count + earliest time) +
flat-hit count and compaction (get_flat_hits' two kernels) + -- with more than one GPU -- the RCCL
all-reduce of the per-channel arrays inside the library, + the read-back of those two small arrays.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|detector|lite|tiny|c5]

N > 1: one rank per GPU (the driver starts them through torch.distributed.run; started by hand with
--gpus N and no WORLD_SIZE, this script starts that launcher itself as a child process).  Local rank 0
builds the geometry, its BVH and the derived wide tree ONCE and publishes them under /dev/shm; the
other ranks map the same files.  Every rank generates its own photon shard on the device (Philox
stream keyed by the GLOBAL photon id) and the per-GPU work is fixed (weak scaling); at N = 8 the c3
shard is 1.25e8 photons (BASELINE.json configs[3]: 1e9 photons over 8 GPUs).  Rank 0 prints ONE JSON line.
"""
import argparse
import gc
from ctypes import c_uint32 as ctypes_uint32
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (builder, photons per GPU per step, description)
    'c3': ('detector29k', 100_000_000, '29k-PMT sphere (29 007 PMTs, ~170 M triangles), %s-photon 400 nm bomb per GPU per step'),
    'detector': ('detector', 10_000_000, 'demo.detector() (10 055 PMTs, 59 M triangles), %s-photon 400 nm bomb'),
    'lite': ('detector_lite', 10_000_000, 'C2-lite (501 PMTs, ~3 M triangles), %s-photon 400 nm bomb'),
    'tiny': ('tiny', 1_000_000, 'demo.tiny() (53 PMTs, 390 k triangles), %s-photon 400 nm bomb'),
    'c5': ('scintillator_stress', 10_000_000, 'scintillator cube with thin-film / WLS / dichroic / detecting faces, %s-photon 350 nm bomb'),
}
C4_PHOTONS_PER_GPU = 125_000_000      # configs[3]: 1e9 photons sharded over 8 GPUs
ENGINE_SEED = 12345
# `value` is measured on photons in GENERATION order (the Morton pre-sort of directions is "reported
# separately").  CHROMA_BENCH_SORT=1 puts the headline batches themselves in tools.argsort_direction order before the clock
# starts, as chroma/benchmark.py:80-82 does (round 3's headline input); either way the other input is timed over the same
# number of batches after the headline and reported beside it (config.presorted / config.generation_order).
SORT_DIRECTIONS = os.environ.get('CHROMA_BENCH_SORT', '0') != '0'
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
RAYCAST_KERNEL = {'reference': 'k_raycast_persistent', 'wide': 'k_raycast_wide', 'coop': 'k_raycast_coop'}.get(os.environ.get('CHROMA_WALK', ''), 'k_raycast_quad')


def log(*a):
    if int(os.environ.get('RANK', '0')) == 0:
        print('[bench]', *a, file=sys.stderr, flush=True)


def kernel_source_hash():
    """Identifies the kernels a profile was taken with: profiles/pmc_traffic.json entries carry it, and a
    counter figure is only quoted when it belongs to the sources this run was built from."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'chroma_amd', 'csrc')
    names = ['chroma_hip.hip', 'propagate_device.h', 'device_common.h'] + sorted(n for n in os.listdir(csrc) if n.startswith(('kernel_', 'kernels_')) and n.endswith('.h'))
    for name in names:
        with open(os.path.join(csrc, name), 'rb') as f:
            h.update(f.read())
    with open(os.path.join(ROOT, 'include', 'chroma_math.h'), 'rb') as f:
        h.update(f.read())
    return h.hexdigest()[:12]


def geometry_source_hash():
    """Identifies what a cached PackedGeometry was built from (CHROMA_BENCH_GEOMETRY_CACHE): the tree topology switch
    and the sources of the host builders and of the demo geometries."""
    h = hashlib.sha256((os.environ.get('CHROMA_TREE', 'levels') + '/' + os.environ.get('CHROMA_TIGHT_LEAVES', '0')).encode())
    names = [os.path.join('chroma_amd', 'csrc', n) for n in ('wide_build.cpp', 'wide_build.h', 'wide_device.hip', 'bvh_build.cpp', 'mesh_utils.cpp', 'bvh_device.hip')]
    for sub in ('demo', 'bvh'):
        names += sorted(os.path.join('chroma_amd', sub, n) for n in os.listdir(os.path.join(ROOT, 'chroma_amd', sub)) if n.endswith('.py'))
    names += [os.path.join('chroma_amd', n) for n in ('geometry.py', 'detector.py', 'make.py', 'pmt.py', 'transform.py', 'loader.py', os.path.join('gpu', 'geometry.py'))]
    for name in names:
        try:
            with open(os.path.join(ROOT, name), 'rb') as f:
                h.update(name.encode() + f.read())
        except OSError:
            pass
    return h.hexdigest()[:12]


def gg_plain(gg):
    """True when the geometry's optics are plain (the k_physics<false> build runs)."""
    try:
        d = gg.packed.desc
        import numpy as np
        models = np.ctypeslib.as_array((ctypes_uint32 * int(d.nsurfaces)).from_address(d.surf_model)) if d.nsurfaces and d.surf_model else []
        comps = np.ctypeslib.as_array((ctypes_uint32 * int(d.nmaterials)).from_address(d.mat_num_comp)) if d.nmaterials and d.mat_num_comp else []
        return not (any(int(m) != 0 for m in models) or any(int(c) != 0 for c in comps)) and not os.environ.get('CHROMA_FULL_PHYSICS')
    except Exception:
        return True


def effective_cores():
    """Host threads this process may really use: the affinity mask, capped by the cgroup's CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=2)        # (the first ~0.25 s of sustained load run 1.6 % slower: two batches at C3)
    ap.add_argument('--config', default=os.environ.get('CHROMA_BENCH_CONFIG', 'c3'), choices=sorted(CONFIGS))
    ap.add_argument('--photons', type=int, default=0, help='photons per GPU per step (default: the config\'s)')
    ap.add_argument('--max-steps', type=int, default=100)
    ap.add_argument('--wavelength-hi', type=float, default=0.0, help='> 400: wavelengths uniform in [400, hi] nm')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=0)
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started by hand: become the launcher's parent BEFORE anything touches a GPU (a child process, never an exec)
        import socket
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (start one rank per GPU: python -m torch.distributed.run '
                         '--nproc-per-node N bench.py --gpus N ...)' % (args.gpus, world))

    # torch first: it carries its own libamdhip64 (and RCCL) under the same SONAMEs, so loading it before
    # libchroma_hip.so makes both share one HIP runtime in this process.
    torch = None
    try:
        import torch
        import torch.distributed as dist
    except Exception as exc:        # pragma: no cover
        log('torch not importable (%s): running without it' % exc)
    if world > 1:
        if torch is None:
            raise SystemExit('multi-GPU runs need torch.distributed')
        # (CHROMA_BENCH_BACKEND=gloo is a REHEARSAL of the multi-rank flow on a box with fewer GPUs than ranks: ranks
        #  share devices, rendezvous and reductions go through gloo on host tensors, and the library's RCCL communicator
        #  -- which refuses two ranks on one device -- gives way to the torch.distributed fallback below.  Its number
        #  means nothing.)
        backend = os.environ.get('CHROMA_BENCH_BACKEND', 'nccl')
        ndev = max(1, torch.cuda.device_count())
        device_index = local_rank % ndev
        torch.cuda.set_device(device_index)
        # (a timeout well under the driver's limit on the rendezvous and on every start-up collective: a rank that
        #  never arrives fails the others instead of leaving them waiting)
        from chroma_amd.dist import init_process_group
        if backend == 'nccl':
            init_process_group('nccl', device_id=torch.device('cuda', device_index))
            reduce_device = torch.device('cuda', device_index)
        else:
            init_process_group(backend)
            reduce_device = torch.device('cpu')
    else:
        device_index = local_rank

    import ctypes
    import numpy as np
    from chroma_amd import demo, gpu, event, _lib
    from chroma_amd.loader import create_geometry_from_obj
    from chroma_amd.gpu.geometry import pack_geometry
    from chroma_amd.gpu.photon import _structure, _alloc_fields, GPUPhotonsSlice
    from chroma_amd.gpu.tools import GPUArray, zeros, empty
    from chroma_amd.dist import init_comm, publish_packed_geometry, remove_published, all_agree

    builder, nphotons, desc = CONFIGS[args.config]
    if args.photons:
        nphotons = args.photons
    elif args.config == 'c3' and world == 8:
        nphotons = C4_PHOTONS_PER_GPU
    desc = desc % ('%.3g' % nphotons)

    # ---- geometry: built once per node (its BVH on the GPU, as in the reference), uploaded by every rank ----------
    ctx = gpu.create_cuda_context(device_index)
    t0 = time.time()

    def build_packed():
        geo = create_geometry_from_obj(getattr(demo, builder)())
        return pack_geometry(geo).attach_wide_tree()
    shm_path = None
    geometry_cached = False
    if world > 1:
        packed, shm_path = publish_packed_geometry(build_packed, 'bench_' + args.config, local_rank, dist.barrier)
        # the uploads (validation, staging) run in every rank at once: share the cores from here on
        os.environ.setdefault('CHROMA_HOST_THREADS', str(max(4, effective_cores() // max(1, local_world))))
    elif os.environ.get('CHROMA_BENCH_GEOMETRY_CACHE'):
        # repeated runs on one box (A/B series, counter passes): the packed geometry is kept between them
        # (keyed by everything the saved arrays depend on: the tree topology switch, the builders' sources, the demo
        #  geometry's sources -- a change to any of them builds afresh instead of silently reusing the old tree)
        from chroma_amd.gpu.geometry import PackedGeometry
        cache = os.path.join(os.environ['CHROMA_BENCH_GEOMETRY_CACHE'], '%s_%s' % (args.config, geometry_source_hash()))
        if os.path.exists(os.path.join(cache, 'desc.json')):
            packed = PackedGeometry.load(cache, mmap=True)
            geometry_cached = True
        else:
            packed = build_packed()
            packed.save(cache)
    else:
        packed = build_packed()
    t_build = time.time() - t0
    d = packed.desc
    log('%s: %d triangles, %d nodes, %d wide nodes, %d channels; %s in %.1f s' % (
        args.config, d.ntriangles, d.nnodes, d.nwide, d.nchannels,
        'loaded from the geometry cache' if geometry_cached else
        'built + wide tree' if (world == 1 or local_rank == 0) else 'mapped from %s' % shm_path, t_build))

    t0 = time.time()
    gg = gpu.GPUDetector.from_packed(packed)
    lib_comm = True
    if world > 1:
        # the library's own RCCL communicator (reductions in place on the device arrays).  Should it not come up on
        # some rank, ALL ranks agree to reduce through torch.distributed instead (same RCCL, staged through tensors)
        # (init_comm fails on ALL ranks together or on none: dist.StartupError)
        lib_comm = False
        if backend == 'nccl':
            try:
                init_comm(ctx)
                lib_comm = True
            except Exception as exc:        # pragma: no cover
                print('[bench] rank %d: library communicator unavailable (%s)' % (rank, exc), file=sys.stderr, flush=True)
        lib_comm = all_agree(lib_comm)
        if not lib_comm:
            ctx._lib.chroma_comm_destroy(ctx.handle)
            log('reducing the per-channel arrays through torch.distributed')
        remove_published(shm_path, local_rank, dist.barrier)
    elif os.environ.get('CHROMA_BENCH_COMM'):         # one-rank communicator: exercises the RCCL path on a 1-GPU box
        ident = (ctypes.c_uint8 * 128)()
        _lib.check(ctx._lib.chroma_comm_unique_id(ident))
        _lib.check(ctx._lib.chroma_comm_init(ctx.handle, 1, 0, ident))
    t_upload = time.time() - t0
    log('device: %s; geometry upload %.1f s; traversal stack need %d' % (ctx.device_name(), t_upload, gg.stack_need()))
    run_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    if not run_cpu:
        packed = None                 # (the host arrays are only needed again by the CPU baseline)

    lib = ctx._lib
    nbatches = args.warmup + args.steps
    wl_lo, wl_hi = (350.0 if args.config == 'c5' else 400.0), args.wavelength_hi
    nch = gg.nchannels

    # batches resident in HBM before the timed region -- unless that many do not fit: then a small ring of
    # buffers is refilled by the device bomb generator INSIDE the timed region (counted against the result)
    free, total = ctx.mem_get_info()
    work_bytes = nphotons * (2 * 4 + 2 * 4 + 4 + 2 * 64 + 2 * 64 + 64) + (1 << 30)        # queues, hit hand-off, ray records, working sets, final records
    batch_bytes = nphotons * 64
    sort_scratch = nphotons * 28                               # chroma_photons_sort_direction: codes, indices, radix-sort double buffers, one gathered array
    resident = (nbatches + 1) * batch_bytes + work_bytes + sort_scratch + int(0.12 * nphotons) * 72 < 0.92 * free
    nbuffers = nbatches if resident else 2
    log('HBM: %.1f GB free of %.1f; %d batches of %.1f GB %s' % (free / 1e9, total / 1e9, nbatches, batch_bytes / 1e9,
        'resident' if resident else 'do not fit: %d buffers refilled inside the timed region' % nbuffers))

    class Batch(object):
        """Device photon arrays filled by the on-device bomb generator."""
        def __init__(self):
            self.ph = GPUPhotonsSlice(rng_counters=empty(nphotons, np.uint32, ctx), **_alloc_fields(nphotons, ctx))
            self.struct = _structure(self.ph)
            self.id_base = None

        def fill(self, index, sort=None):
            # global photon ids: batch-major, then rank, then index -> independent of world size
            self.id_base = (index * world + rank) * nphotons
            pos = (ctypes.c_float * 3)(0.0, 0.0, 0.0)
            _lib.check(lib.chroma_generate_bomb(ctx.handle, ctypes.byref(self.struct), nphotons, ENGINE_SEED, self.id_base, pos, wl_lo, wl_hi))
            if SORT_DIRECTIONS if sort is None else sort:
                self.sort()
            return self

        def sort(self):
            # chroma/benchmark.py:80-82: the reference's propagate benchmark puts its photons in the order of
            # tools.argsort_direction before it starts the clock; same here, on the device
            _lib.check(lib.chroma_photons_sort_direction(ctx.handle, ctypes.byref(self.struct), nphotons))
            return self

    # per-step outputs, allocated once: per-channel arrays and the flat-hit buffers (get_flat_hits' destination)
    counts = zeros(nch, np.uint32, ctx)
    earliest = GPUArray(nch, np.uint32, ctx)
    hit_capacity = [0]
    hits = [None, None, None]

    def ensure_hit_buffers(n):
        if n > hit_capacity[0]:
            cap = int(1.25 * n) + 1024
            hits[0] = GPUPhotonsSlice(**_alloc_fields(cap, ctx))
            hits[1] = _structure(hits[0])
            hits[2] = empty(cap, np.int32, ctx)
            hit_capacity[0] = cap

    separate_hits = bool(os.environ.get('CHROMA_BENCH_SEPARATE_HITS'))      # (A/B: round 3's four calls instead of the fused one)

    def run_step(batch, time_kernels, stats):
        rng = _lib.Rng(ENGINE_SEED, batch.id_base)
        aborted = ctypes.c_int32(0)
        st = _lib.PropagateStats()
        counts.fill(np.uint32(0))
        earliest.fill(np.uint32(0x7f800000))
        nhits = ctypes.c_uint32()
        ncopied = ctypes.c_uint32()
        if not separate_hits:
            # propagate_hit as ONE library call (chroma_propagate_hits): the pass that finishes the propagation also makes the
            # per-channel arrays, counts the hits and compacts them with their channels (get_flat_hits, chroma/gpu/photon.py:96-175)
            req = _lib.HitsRequest()
            req.detection_state = event.SURFACE_DETECT
            req.capacity = hit_capacity[0]
            req.dst = ctypes.pointer(hits[1])
            req.d_channels = hits[2].ptr
            req.d_hit_count = counts.ptr
            req.d_earliest_time_bits = earliest.ptr
            _lib.check(lib.chroma_propagate_hits(ctx.handle, gg.handle, ctypes.byref(batch.struct), nphotons, 1, rng,
                                                 args.max_steps, 0, 0, int(time_kernels), ctypes.byref(st), ctypes.byref(aborted), ctypes.byref(req)))
            nhits.value = req.nhits
            ncopied.value = min(req.nhits, req.capacity)
            if req.nhits > req.capacity:          # (the buffers were sized from the counting pass: not expected)
                ensure_hit_buffers(req.nhits)
                _lib.check(lib.chroma_copy_photon_hits(ctx.handle, gg.handle, 0, nphotons, event.SURFACE_DETECT, ctypes.byref(batch.struct),
                                                       ctypes.byref(hits[1]), hits[2].ptr, ctypes.byref(ncopied)))
        else:
            _lib.check(lib.chroma_propagate(ctx.handle, gg.handle, ctypes.byref(batch.struct), nphotons, 1, rng,
                                            args.max_steps, 0, 0, int(time_kernels), ctypes.byref(st), ctypes.byref(aborted)))
            _lib.check(lib.chroma_channel_hits(ctx.handle, gg.handle, nphotons, event.SURFACE_DETECT, ctypes.byref(batch.struct),
                                               counts.ptr, earliest.ptr))
            # get_flat_hits (chroma/gpu/photon.py:96-175): count, then compact the detected photons + their channels
            _lib.check(lib.chroma_count_photon_hits(ctx.handle, gg.handle, 0, nphotons, event.SURFACE_DETECT,
                                                    ctypes.byref(batch.struct), ctypes.byref(nhits)))
            ensure_hit_buffers(nhits.value)
            _lib.check(lib.chroma_copy_photon_hits(ctx.handle, gg.handle, 0, nphotons, event.SURFACE_DETECT, ctypes.byref(batch.struct),
                                                   ctypes.byref(hits[1]), hits[2].ptr, ctypes.byref(ncopied)))
        # the one exchange: per-channel arrays over RCCL, in place on the device (identity on one GPU)
        if lib_comm:
            _lib.check(lib.chroma_allreduce_hits(ctx.handle, counts.ptr, earliest.ptr, nch))
            c, e = counts.get(), earliest.get()
        else:                           # pragma: no cover
            from chroma_amd.dist import allreduce_channel_hits
            c, e = allreduce_channel_hits(counts.get(), earliest.get(), device=reduce_device)
        for k, v in st.as_dict().items():
            stats[k] = stats.get(k, 0) + v
        stats['hits'] = stats.get('hits', 0) + int(nhits.value)
        stats['copied'] = stats.get('copied', 0) + int(ncopied.value)
        stats['channel_sum'] = stats.get('channel_sum', 0) + int(np.asarray(c, dtype=np.uint64).sum())
        return c, e

    def sync_all():
        if world > 1:
            dist.barrier()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()
        ctx.synchronize()

    # ---- algorithmic work per photon: one counting pass on a batch of its own (untimed) -------
    t0 = time.time()
    count_n_stats = {}
    ctx.set_counting(True)
    probe = Batch().fill(10_000)
    ensure_hit_buffers(int(0.12 * nphotons) + 65536)
    run_step(probe, False, count_n_stats)
    ctx.set_counting(False)
    steps_pp = count_n_stats['photon_steps'] / nphotons
    nodes_ps = count_n_stats['nodes_visited'] / max(1, count_n_stats['photon_steps'])
    tris_ps = count_n_stats['triangles_tested'] / max(1, count_n_stats['photon_steps'])
    bytes_per_step = 16.0 * nodes_ps + 48.0 * tris_ps + 40 + 120 + 8
    bytes_per_photon = bytes_per_step * steps_pp
    log('counting pass %.1f s: %.3f steps/photon, %.1f nodes/step, %.2f triangle tests/step -> %.0f B/step, %.0f B/photon; '
        'hit fraction %.4f; %d stack entries spilled' % (time.time() - t0, steps_pp, nodes_ps, tris_ps, bytes_per_step, bytes_per_photon,
                                                         count_n_stats['hits'] / nphotons, count_n_stats.get('stack_spills', 0)))
    if count_n_stats.get('packet_nodes_visited'):
        log('  diagnostic counters: %r' % {k: v for k, v in count_n_stats.items() if k.startswith('packet_')})
    if world > 1:
        assert count_n_stats['channel_sum'] >= count_n_stats['hits'], 'reduced channel counts smaller than this rank\'s own'
    else:
        assert count_n_stats['channel_sum'] == count_n_stats['hits'] == count_n_stats['copied']

    if resident:
        buffers = [probe.fill(0)] + [Batch().fill(i) for i in range(1, nbatches)]
    else:
        buffers = [probe, Batch()]
    ctx.synchronize()

    def batch_for(i):
        return buffers[i] if resident else buffers[i % nbuffers].fill(i)

    sync_all()          # (the first torch.cuda.synchronize() of the process initialises torch's side of the runtime: before the warm-up, not after it)
    for i in range(args.warmup):
        t_step = time.perf_counter()
        # (kernels timed as in the timed steps: the library creates its HIP events at the first timed call -- 6 per step of
        #  max_steps, ~2 ms -- and that set-up belongs to the warm-up, not to a step)
        run_step(batch_for(i), True, {})
        log('  warmup %d: %.1f ms wall' % (i, 1e3 * (time.perf_counter() - t_step)))
    # (the garbage of the set-up -- the packed geometry's 6-17 GB of mapped host arrays above all -- is collected HERE: a
    #  collection that starts inside a step costs it ~40 ms of munmap, measured as a one-step outlier whenever the interpreter
    #  happened to schedule one there; nothing the timed loop allocates needs the cycle collector)
    gc.collect()
    gc.disable()
    sync_all()
    stats = {}
    step_walls = []
    t_start = time.perf_counter()
    for i in range(args.warmup, nbatches):
        t_step = time.perf_counter()
        k0 = stats.get('kernel_ms', 0.0)
        run_step(batch_for(i), True, stats)
        step_walls.append(time.perf_counter() - t_step)
        log('  step %d: %.1f ms wall, %.1f ms in propagate kernels' % (i - args.warmup, 1e3 * step_walls[-1], stats['kernel_ms'] - k0))
    sync_all()
    elapsed = time.perf_counter() - t_start
    gc.enable()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    total_photons = nphotons * args.steps * world
    value = total_photons / elapsed
    kernel_s = stats['kernel_ms'] / 1e3
    launches = max(1, stats['launches'])
    path_achieved = (bytes_per_photon * nphotons * args.steps) / kernel_s / 1e9 if kernel_s > 0 else 0.0
    # dominant kernel: the ray cast (k_raycast_quad; the retry pass after it is normally empty).
    # Algorithmic bytes per photon step: 16 B per child entry fetched (a 128-B wide node = 8 entries)
    # + 48 B per triangle tested + 64 B ray record read + 8 B hit written
    ray_s = stats['raycast_ms'] / 1e3
    ray_launches = max(1, stats['raycast_launches'])
    ray_bytes_per_step = 16.0 * nodes_ps + 48.0 * tris_ps + 64 + 8
    ray_bytes_total = ray_bytes_per_step * steps_pp * nphotons * args.steps
    achieved = ray_bytes_total / ray_s / 1e9 if ray_s > 0 else 0.0
    avg_launch_ms = 1e3 * ray_s / ray_launches
    log('timed: %.3f s for %d steps; propagate kernels %.3f s in %d launches, of which ray cast %.3f s in %d launches '
        '(avg %.3f ms); hits/photon %.4f' % (elapsed, args.steps, kernel_s, launches, ray_s, ray_launches,
                                             avg_launch_ms, stats['hits'] / (nphotons * args.steps)))
    # HBM traffic of the same command from rocprofv3 PMC passes (tools/pmc_traffic.py -> profiles/pmc_traffic.json):
    # quoted only when that profile was taken with the kernels this run was built from
    traffic, traffic_source, measured = None, 'none: no PMC profile of this command under profiles/', None
    src_hash = kernel_source_hash()
    sq, physics_traffic, physics_sq = None, None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
        key = '%s:%d:%d' % (args.config, nphotons, args.max_steps)
        # issue-side counters and the second kernel's traffic: quoted under the same rule (this build's hash only)
        if pmc.get(key + ':sq', {}).get('source_hash') == src_hash:
            sq = pmc[key + ':sq']
        if pmc.get(key + ':physics', {}).get('source_hash') == src_hash:
            physics_traffic = pmc[key + ':physics']
        if pmc.get(key + ':physics:sq', {}).get('source_hash') == src_hash:
            physics_sq = pmc[key + ':physics:sq']
        if key in pmc:
            rec = pmc[key]
            if rec.get('source_hash') == src_hash:
                traffic = rec['hbm_bytes_per_launch']
                traffic_source = 'profiles/pmc_traffic.json[%s]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, %s, %d launches, kernels %s (= this build); %s' % (
                    key, rec.get('date', '?'), rec.get('launches', 0), src_hash, rec.get('correction', ''))
                measured = traffic / (avg_launch_ms * 1e-3) / 1e9
            else:
                traffic_source = 'stale: profiles/pmc_traffic.json[%s] was taken with kernels %s, this build is %s' % (key, rec.get('source_hash', '?'), src_hash)
    except Exception:
        pass

    def rate_stats(walls):
        """mean and standard deviation of the per-batch rates, and of the per-batch times, over one leg's batches"""
        r = [nphotons / w for w in walls]
        return {'value': float(np.mean(r)), 'std': float(np.std(r)), 'batches': len(r),
                'ms_per_batch': 1e3 * float(np.mean(walls)), 'ms_per_batch_std': 1e3 * float(np.std(walls))}

    def timed_leg(prepare, nleg):
        """`nleg` batches through run_step, each between two device synchronisations (the same step definition as the
        headline, kernels timed by HIP events on the library's stream); `prepare(i)` returns batch i with its input ready
        BEFORE the clock starts.  Returns the walls and the summed statistics."""
        walls, leg_stats = [], {}
        gc.collect()
        gc.disable()                  # (as in the headline loop: no cycle collection inside a batch)
        try:
            for i in range(nleg):
                b = prepare(i)
                ctx.synchronize()
                t0 = time.perf_counter()
                run_step(b, True, leg_stats)
                ctx.synchronize()
                walls.append(time.perf_counter() - t0)
        finally:
            gc.enable()
        return walls, leg_stats

    def leg_roofline(leg, leg_stats, nodes_per_step, tris_per_step, steps_per_photon, nleg):
        """the ray cast of one leg against the HBM roof: ALGORITHMIC bytes (with the counts given) over the HIP-event
        time of the leg's ray-cast launches"""
        ray_s = leg_stats.get('raycast_ms', 0.0) / 1e3
        if ray_s <= 0:
            return
        total = (16.0 * nodes_per_step + 48.0 * tris_per_step + 64 + 8) * steps_per_photon * nphotons * nleg
        leg['raycast_ms_per_batch'] = 1e3 * ray_s / nleg
        leg['raycast_avg_launch_ms'] = 1e3 * ray_s / max(1, leg_stats.get('raycast_launches', 0))
        leg['raycast_algorithmic_GBps'] = total / ray_s / 1e9
        leg['raycast_frac'] = total / ray_s / 1e9 / HBM_PEAK_GBS
        leg['physics_ms_per_batch'] = leg_stats.get('physics_ms', 0.0) / nleg

    headline_order = rate_stats(step_walls)
    # (the extra legs below run on one GPU only: run_step holds the all-reduce of the per-channel arrays, a collective EVERY
    #  rank would have to enter)
    extra_legs = world == 1 and not os.environ.get('CHROMA_BENCH_NO_EXACT') and os.environ.get('CHROMA_WALK', 'quad') == 'quad'
    nleg = int(os.environ.get('CHROMA_BENCH_LEG_BATCHES', args.steps))

    def leg_batch(i, index_base, sort):
        # (resident: buffer i of the headline's own set, refilled for the leg; else the two-buffer ring)
        return (buffers[i % len(buffers)]).fill(index_base + i, sort=sort)

    # ---- the OTHER input order over the same number of batches: photons put in tools.argsort_direction order before the clock
    # starts, as the reference's own propagate benchmark does (chroma/benchmark.py:80-82) -- or, with CHROMA_BENCH_SORT=1,
    # generation order -- and what the ordering itself costs on the device (it is NOT inside any timed region)
    other_order = sort_cost = None
    if extra_legs:
        walls, leg_stats = timed_leg(lambda i: leg_batch(i, 30_000, not SORT_DIRECTIONS), nleg)
        other_order = rate_stats(walls)
        leg_roofline(other_order, leg_stats, nodes_ps, tris_ps, steps_pp, nleg)
        log('%s: %.4g +- %.2g photons/s over %d batches (%.1f +- %.1f ms)' % (
            'photons pre-sorted by direction' if not SORT_DIRECTIONS else 'photons in generation order', other_order['value'], other_order['std'],
            nleg, other_order['ms_per_batch'], other_order['ms_per_batch_std']))
        sorts = []
        for i in range(min(nleg, 5)):
            b = leg_batch(i, 40_000, False)
            ctx.synchronize()
            t0 = time.perf_counter()
            b.sort()
            ctx.synchronize()
            sorts.append(time.perf_counter() - t0)
        sort_cost = {'ms_per_batch': 1e3 * float(np.mean(sorts)), 'std': 1e3 * float(np.std(sorts)), 'batches': len(sorts),
                     'what': 'chroma_photons_sort_direction on the device (Morton code of theta and phi, radix sort, gather of the ten arrays); outside every timed region'}
        log('direction sort of one batch on the device: %.1f +- %.1f ms' % (sort_cost['ms_per_batch'], sort_cost['std']))
    presorted, generation = (headline_order, other_order) if SORT_DIRECTIONS else (other_order, headline_order)

    # ---- the exact walk (chroma_set_walk LITERAL, GPUPhotons.propagate(exact=True)): chroma/cuda/mesh.h:42-118 for every ray,
    # same step definition, same number of batches as the headline, photons in the headline's order
    exact_walk = None
    if extra_legs:
        ctx.set_walk('literal')
        try:
            # (the exact walk makes the REFERENCE's tests: its own counts for its algorithmic bytes, from one counting batch)
            lit_counts = {}
            ctx.set_counting(True)
            run_step(leg_batch(0, 19_000, SORT_DIRECTIONS), False, lit_counts)
            ctx.set_counting(False)
            walls, leg_stats = timed_leg(lambda i: leg_batch(i, 20_000, SORT_DIRECTIONS), nleg)
        finally:
            ctx.set_counting(False)
            ctx.set_walk('quad')
        exact_walk = rate_stats(walls)
        exact_walk['kernel'] = 'k_raycast_literal'
        lit_steps = max(1, lit_counts.get('photon_steps', 0))
        exact_walk['nodes_per_step'] = lit_counts.get('nodes_visited', 0) / lit_steps
        exact_walk['triangle_tests_per_step'] = lit_counts.get('triangles_tested', 0) / lit_steps
        leg_roofline(exact_walk, leg_stats, exact_walk['nodes_per_step'], exact_walk['triangle_tests_per_step'], lit_steps / nphotons, nleg)
        exact_walk['slower_than_default'] = headline_order['value'] / exact_walk['value']
        log('exact (literal reference) walk: %.4g +- %.2g photons/s over %d batches (%.1fx slower than the default walk)' % (
            exact_walk['value'], exact_walk['std'], nleg, exact_walk['slower_than_default']))

    # the second kernel of a step, k_physics (main pass): per photon step it reads the hit entry (8 B), the photon's
    # record (64 B) and the winning triangle's record (48 B), writes a survivor's record, next ray and queue slot
    # (64 + 64 + 4 B) or a finished photon's ten array entries (64 B); every photon ends exactly once
    phys_s = stats.get('physics_ms', 0.0) / 1e3
    phys_bytes_per_step = 120.0 + 132.0 * (1.0 - 1.0 / steps_pp) + 64.0 / steps_pp
    phys_bytes_total = phys_bytes_per_step * steps_pp * nphotons * args.steps
    phys_launches = max(1, stats.get('physics_launches', 0))
    physics = {'kernel': 'k_physics<%s>' % ('false' if gg_plain(gg) else 'true'), 'kernel_s': phys_s, 'launches': int(stats.get('physics_launches', 0)),
               'ms_per_batch': 1e3 * phys_s / args.steps,
               'algorithmic_bytes_per_photon_step': phys_bytes_per_step,
               'algorithmic_GBps': phys_bytes_total / phys_s / 1e9 if phys_s > 0 else None,
               'algorithmic_frac': phys_bytes_total / phys_s / 1e9 / HBM_PEAK_GBS if phys_s > 0 else None,
               'hbm_measured_GBps': None, 'traffic_over_algorithmic': None,
               'valu_lane_utilisation': physics_sq['valu_lane_utilisation'] if physics_sq else None,
               'wait_frac': physics_sq['wait_frac'] if physics_sq else None}
    if physics_traffic and phys_s > 0:
        # (the PMC figure is per launch over main AND fix-up launches, which come in pairs: two launches per step)
        per_step = 2.0 * physics_traffic['hbm_bytes_per_launch']
        physics['hbm_measured_GBps'] = per_step * stats.get('physics_launches', 0) / phys_s / 1e9
        physics['traffic_over_algorithmic'] = per_step * stats.get('physics_launches', 0) / phys_bytes_total

    cpu_baseline = None
    if run_cpu:
        import oracle
        cores = effective_cores()
        geo_s2c = packed.arrays['solid_id_to_channel_index']
        geo_solid = packed.arrays['solid_id_map']

        def cpu_pass(ph):
            t0 = time.perf_counter()
            end, _, _ = oracle.propagate(packed, ph, seed=ENGINE_SEED, photon_id_base=0, max_steps=args.max_steps, nthreads=cores)
            det = (end.flags & event.SURFACE_DETECT) != 0
            tri = end.last_hit_triangles
            ok = det & (tri > -1)
            chan = geo_s2c[geo_solid[tri[ok]]]
            np.bincount(chan[chan >= 0], minlength=nch)
            return time.perf_counter() - t0
        # warm-up (also the calibration): then 3 repetitions of a sample sized for ~6 s each, 3e6..3e7 photons
        calib = 1_000_000 if args.config in ('c3', 'detector') else 400_000
        ph = oracle.generate_bomb(calib, seed=ENGINE_SEED, id_base=0, wavelength_lo=wl_lo, wavelength_hi=wl_hi)
        rate0 = calib / cpu_pass(ph)
        sample = args.cpu_sample or int(min(30_000_000, max(3_000_000 if args.config in ('c3', 'detector') else 400_000, 6.0 * rate0)))
        ph = oracle.generate_bomb(sample, seed=ENGINE_SEED, id_base=0, wavelength_lo=wl_lo, wavelength_hi=wl_hi)
        times = [cpu_pass(ph) for _ in range(3)]
        rates = [sample / t for t in times]
        cpu_baseline = {'value': float(np.mean(rates)), 'std': float(np.std(rates)), 'unit': 'photons/s', 'cores': cores,
                        'per_thread': float(np.mean(rates)) / cores, 'kind': 'port', 'repetitions': 3,
                        'sample': '%d photons of the same bomb on %s, oracle/chroma_oracle.c on %d host threads (of %d logical CPUs), '
                                  'propagate + hit histogram; one warm-up pass of %d photons (%.3g/s), then 3 passes of %.1f / %.1f / %.1f s' % (
                                      sample, args.config, cores, os.cpu_count() or 0, calib, rate0, times[0], times[1], times[2])}
        log('cpu baseline: %.3g +- %.2g photons/s on %d threads (%.3g per thread)' % (cpu_baseline['value'], cpu_baseline['std'], cores, cpu_baseline['per_thread']))
        if args.config == 'tiny':
            # the pure-NumPy restatement, 1e4 photons, one core (numpy is single-threaded here)
            from oracle import numpy_propagate as npp
            tab = npp.Tables(packed)
            ph1 = oracle.generate_bomb(10_000, seed=ENGINE_SEED, id_base=0, wavelength_lo=wl_lo, wavelength_hi=wl_hi)
            npp.propagate(packed, ph1, seed=1, max_steps=args.max_steps, tables=tab)          # warm-up
            t0 = time.perf_counter()
            reps = 3
            for r in range(reps):
                npp.propagate(packed, ph1, seed=2 + r, max_steps=args.max_steps, tables=tab)
            dtn = (time.perf_counter() - t0) / reps
            cpu_baseline['numpy_c1'] = {'value': 10_000 / dtn, 'unit': 'photons/s', 'cores': 1,
                                        'sample': '1e4 photons of the same bomb, oracle/numpy_propagate.py, mean of %d runs, %.2f s each' % (reps, dtn)}
            log('numpy C1 baseline: %.3g photons/s on 1 core' % (10_000 / dtn))

    if rank == 0:
        result = {
            'metric': 'photons/sec (propagate_hit)', 'value': value, 'unit': 'photons/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,          # BASELINE.md holds no published number for this metric (BASELINE.json "published": {})
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': desc, 'geometry': args.config, 'triangles': int(d.ntriangles),
                       'bvh_nodes': int(d.nnodes), 'channels': int(d.nchannels),
                       'photons_per_gpu_per_step': nphotons, 'max_steps': args.max_steps,
                       'wavelength_nm': [wl_lo, wl_hi] if wl_hi > wl_lo else wl_lo,
                       'engine_seed': ENGINE_SEED, 'parallelism': 'photon shards x%d, geometry replicated, per-channel arrays all-reduced (%s)' % (
                           world, 'RCCL inside the library' if lib_comm else ('torch.distributed, %s' % (backend if world > 1 else '-'))),
                       'step': 'propagate(max_steps) + channel hit arrays + flat-hit count and compaction' + (' (one library call: chroma_propagate_hits)' if not separate_hits else ' (four library calls)') + (' + all-reduce' if world > 1 else ''),
                       'inputs': ('resident in HBM' if resident else 'bomb regenerated on the device inside the timed region (memory)') +
                                 ('; photons in the order of tools.argsort_direction (Morton code of theta, phi), as the reference\'s own '
                                  'propagate benchmark prepares them before its clock starts (chroma/benchmark.py:80-82)' if SORT_DIRECTIONS else
                                  '; photons in generation order (SURVEY.md 8d: the direction pre-sort of chroma/benchmark.py:80-82 is reported separately: config.presorted, config.sort)'),
                       'target_photons_per_s_per_gpu': 2.5e6, 'vs_target': value / world / 2.5e6,
                       'steps_per_photon': steps_pp, 'nodes_per_step': nodes_ps, 'triangle_tests_per_step': tris_ps,
                       # the three legs, each over the same number of batches with its spread: `value`'s own input order, the other
                       # order, and the exact walk; what putting a batch in direction order costs (never inside a timed region)
                       'generation_order': generation, 'presorted': presorted, 'sort': sort_cost, 'exact_walk': exact_walk,
                       'exact_walk_photons_per_s': exact_walk['value'] if exact_walk else None,
                       'value_is': 'presorted' if SORT_DIRECTIONS else 'generation_order',
                       'geometry_build_s': t_build, 'geometry_cached': geometry_cached, 'geometry_upload_s': t_upload,
                       'reduction': ('none: one GPU' if world == 1 and not os.environ.get('CHROMA_BENCH_COMM') else
                                     'library RCCL (chroma_allreduce_hits, in place on the device arrays)' if lib_comm else
                                     'torch.distributed fallback (%s), staged through host tensors' % backend)},
            # achieved/frac: ALGORITHMIC bytes over the HIP-event time of the ray-cast launches (cache hits included, so
            # this is effective bandwidth); traffic: fabric-side bytes per launch from PMC counters of this very build,
            # hbm_measured_*: that traffic over the same launch time.  The kernel is latency-bound.
            'roofline': {'bound': 'hbm', 'kernel': RAYCAST_KERNEL, 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
                         'hbm_measured_GBps': measured, 'hbm_measured_frac': (measured / HBM_PEAK_GBS) if measured else None,
                         'limiter': 'VALU issue, not bandwidth: SQ_ACTIVE_INST_VALU x 4 = %s of the SIMD cycles of the launches' % (
                             ('%.0f %% (PMC pass of this build, profiles/pmc_traffic.json)' % (100 * sq['valu_issue_frac'])) if sq else
                             '98 % in the last counter pass (profiles/r02/pmc_quad_final.txt; no SQ pass of this build yet)'),
                         'valu_issue_frac': sq['valu_issue_frac'] if sq else None,
                         'valu_lane_utilisation': sq['valu_lane_utilisation'] if sq else None,
                         'salu_per_valu': sq['salu_per_valu'] if sq else None,
                         'physics': physics,
                         'algorithmic_bytes_per_launch': ray_bytes_total / ray_launches,
                         'algorithmic_bytes_per_photon_step': ray_bytes_per_step,
                         'launches': int(stats['raycast_launches']), 'avg_launch_ms': avg_launch_ms,
                         'kernel_s': ray_s, 'kernel_source_hash': src_hash,
                         'path': {'achieved': path_achieved, 'frac': path_achieved / HBM_PEAK_GBS,
                                  'algorithmic_bytes_per_photon': bytes_per_photon, 'launches': int(stats['launches']),
                                  'kernel_s': kernel_s}},
            'cpu_baseline': cpu_baseline,
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        ctx._lib.chroma_comm_destroy(ctx.handle)       # (before torch tears its own communicator down)
        dist.destroy_process_group()


if __name__ == '__main__':
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        # a rank that fails must END (non-zero), at once: the launcher then stops the others.  No clean-up of process
        # groups here -- destroy_process_group on a broken group is itself a place to hang
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
