#!/usr/bin/env python
"""bench.py -- photons/s of the propagate_hit path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic photons already resident in
HBM: GPUPhotons.propagate to completion (max_steps) + per-channel hit reduction + flat-hit
count (+ the RCCL all-reduce of the per-channel arrays when more than one GPU runs).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|detector|lite|tiny]

For N > 1 the driver launches one rank per GPU through torch.distributed.run; every rank builds
the (replicated) geometry, generates its own photon shard on the device (Philox stream keyed by
the global photon id) and the per-GPU work is fixed (weak scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (builder, photons per GPU per step, description)
    'c3': ('detector29k', 100_000_000, '29k-PMT sphere (29 007 PMTs, ~170 M triangles), 1e8-photon 400 nm bomb per GPU per step'),
    'detector': ('detector', 10_000_000, 'demo.detector() (10 055 PMTs, 59 M triangles), 1e7-photon 400 nm bomb'),
    'lite': ('detector_lite', 10_000_000, 'C2-lite (501 PMTs, ~3 M triangles), 1e7-photon 400 nm bomb'),
    'tiny': ('tiny', 1_000_000, 'demo.tiny() (53 PMTs, 390 k triangles), 1e6-photon 400 nm bomb'),
    'c5': ('scintillator_stress', 10_000_000, 'scintillator cube with thin-film / WLS / dichroic / detecting faces, 1e7-photon 350 nm bomb'),
}
ENGINE_SEED = 12345
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def log(*a):
    if int(os.environ.get('RANK', '0')) == 0:
        print('[bench]', *a, file=sys.stderr, flush=True)


RAYCAST_KERNEL = {'reference': 'k_raycast_persistent', 'wide': 'k_raycast_wide', 'coop': 'k_raycast_coop'}.get(os.environ.get('CHROMA_WALK', ''), 'k_raycast_quad')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', default=os.environ.get('CHROMA_BENCH_CONFIG', 'c3'), choices=sorted(CONFIGS))
    ap.add_argument('--photons', type=int, default=0, help='photons per GPU per step (default: the config\'s)')
    ap.add_argument('--max-steps', type=int, default=100)
    ap.add_argument('--wavelength-hi', type=float, default=0.0, help='> 400: wavelengths uniform in [400, hi] nm')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=0)
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1 and 'CHROMA_HOST_THREADS' not in os.environ:
        # every rank builds the same geometry on the same host at the same time: share the cores
        os.environ['CHROMA_HOST_THREADS'] = str(max(4, (os.cpu_count() or 8) // int(os.environ.get('LOCAL_WORLD_SIZE', world))))

    # torch first: it carries its own libamdhip64 under the same SONAME, so loading it before
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
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))

    import numpy as np
    from chroma_amd import demo, gpu, event
    from chroma_amd.loader import create_geometry_from_obj
    from chroma_amd.gpu.geometry import pack_geometry
    from chroma_amd.gpu.photon import _structure
    from chroma_amd.gpu.tools import GPUArray, zeros
    from chroma_amd import _lib
    import ctypes

    builder, nphotons, desc = CONFIGS[args.config]
    if args.photons:
        nphotons = args.photons

    t0 = time.time()
    geo = create_geometry_from_obj(getattr(demo, builder)())
    t_build = time.time() - t0
    log('%s: %d triangles, %d nodes, %d channels; built in %.1f s' % (
        args.config, len(geo.mesh.triangles), len(geo.bvh.nodes), geo.num_channels(), t_build))

    ctx = gpu.create_cuda_context(local_rank)
    t0 = time.time()
    packed_for_cpu = pack_geometry(geo) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    gg = gpu.GPUDetector(geo)
    log('device: %s; geometry upload %.1f s; traversal stack need %d' % (ctx.device_name(), time.time() - t0, gg.stack_need()))

    lib = ctx._lib
    nbatches = args.warmup + args.steps
    wl_lo, wl_hi = (350.0 if args.config == 'c5' else 400.0), args.wavelength_hi

    class Batch(object):
        """Device photon arrays filled by the on-device bomb generator."""
        def __init__(self, index):
            from chroma_amd.gpu.tools import vec, empty
            n = nphotons
            self.pos, self.dir, self.pol = empty(n, vec.float3, ctx), empty(n, vec.float3, ctx), empty(n, vec.float3, ctx)
            self.wavelengths, self.t, self.weights = empty(n, np.float32, ctx), empty(n, np.float32, ctx), empty(n, np.float32, ctx)
            self.flags, self.evidx, self.rng_counters = empty(n, np.uint32, ctx), empty(n, np.uint32, ctx), empty(n, np.uint32, ctx)
            self.last_hit_triangles = empty(n, np.int32, ctx)
            self.struct = _structure(self)
            # global photon ids: batch-major, then rank, then index -> independent of world size
            self.id_base = (index * world + rank) * n
            pos = (ctypes.c_float * 3)(0.0, 0.0, 0.0)
            _lib.check(lib.chroma_generate_bomb(ctx.handle, ctypes.byref(self.struct), n, ENGINE_SEED, self.id_base, pos, wl_lo, wl_hi))

    def run_step(batch, time_kernels, stats):
        rng = _lib.Rng(ENGINE_SEED, batch.id_base)
        aborted = ctypes.c_int32(0)
        st = _lib.PropagateStats()
        _lib.check(lib.chroma_propagate(ctx.handle, gg.handle, ctypes.byref(batch.struct), nphotons, 1, rng,
                                        args.max_steps, 0, 0, int(time_kernels), ctypes.byref(st), ctypes.byref(aborted)))
        counts = zeros(gg.nchannels, np.uint32, ctx)
        earliest = GPUArray(gg.nchannels, np.uint32, ctx).fill(np.uint32(0x7f800000))
        _lib.check(lib.chroma_channel_hits(ctx.handle, gg.handle, nphotons, event.SURFACE_DETECT, ctypes.byref(batch.struct),
                                           counts.ptr, earliest.ptr))
        nhits = ctypes.c_uint32()
        _lib.check(lib.chroma_count_photon_hits(ctx.handle, gg.handle, 0, nphotons, event.SURFACE_DETECT,
                                                ctypes.byref(batch.struct), ctypes.byref(nhits)))
        c, e = counts.get(), earliest.get()
        if world > 1:
            from chroma_amd.dist import allreduce_channel_hits
            c, e = allreduce_channel_hits(c, e, device=torch.device('cuda', local_rank))
        for k, v in st.as_dict().items():
            stats[k] = stats.get(k, 0) + v
        stats['hits'] = stats.get('hits', 0) + int(nhits.value)
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
    probe = Batch(10_000 + 0)
    run_step(probe, False, count_n_stats)
    ctx.set_counting(False)
    del probe
    steps_pp = count_n_stats['photon_steps'] / nphotons
    nodes_ps = count_n_stats['nodes_visited'] / max(1, count_n_stats['photon_steps'])
    tris_ps = count_n_stats['triangles_tested'] / max(1, count_n_stats['photon_steps'])
    bytes_per_step = 16.0 * nodes_ps + 48.0 * tris_ps + 40 + 120 + 8
    bytes_per_photon = bytes_per_step * steps_pp
    log('counting pass %.1f s: %.3f steps/photon, %.1f nodes/step, %.2f triangle tests/step -> %.0f B/step, %.0f B/photon; '
        'hit fraction %.4f' % (time.time() - t0, steps_pp, nodes_ps, tris_ps, bytes_per_step, bytes_per_photon,
                               count_n_stats['hits'] / nphotons))

    batches = [Batch(i) for i in range(nbatches)]
    ctx.synchronize()

    for i in range(args.warmup):
        t_step = time.perf_counter()
        run_step(batches[i], False, {})
        log('  warmup %d: %.1f ms wall' % (i, 1e3 * (time.perf_counter() - t_step)))
    sync_all()
    stats = {}
    t_start = time.perf_counter()
    for i in range(args.warmup, nbatches):
        t_step = time.perf_counter()
        k0 = stats.get('kernel_ms', 0.0)
        run_step(batches[i], True, stats)
        log('  step %d: %.1f ms wall, %.1f ms in propagate kernels' % (i - args.warmup, 1e3 * (time.perf_counter() - t_step), stats['kernel_ms'] - k0))
    sync_all()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=torch.device('cuda', local_rank))
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
    log('timed: %.3f s for %d steps; propagate kernels %.3f s in %d launches, of which ray cast %.3f s in %d launches '
        '(avg %.3f ms); hits/photon %.4f' % (elapsed, args.steps, kernel_s, launches, ray_s, ray_launches,
                                             1e3 * ray_s / ray_launches, stats['hits'] / (nphotons * args.steps)))
    # HBM traffic of the same command from rocprofv3 PMC passes (profiles/, corrected as calibrated there)
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
        key = '%s:%d:%d' % (args.config, nphotons, args.max_steps)
        if key in pmc:
            traffic = pmc[key]['hbm_bytes_per_launch']
    except Exception:
        pass

    cpu_baseline = None
    if packed_for_cpu is not None:
        import oracle
        cores = os.cpu_count() or 1
        sample = args.cpu_sample or {'tiny': 400_000, 'lite': 1_000_000}.get(args.config, 3_000_000)
        ph = oracle.generate_bomb(sample, seed=ENGINE_SEED, id_base=0, wavelength_lo=wl_lo, wavelength_hi=wl_hi)
        t0 = time.perf_counter()
        end, _, ost = oracle.propagate(packed_for_cpu, ph, seed=ENGINE_SEED, photon_id_base=0, max_steps=args.max_steps, nthreads=cores)
        det = (end.flags & event.SURFACE_DETECT) != 0
        tri = end.last_hit_triangles
        ok = det & (tri > -1)
        chan = geo.solid_id_to_channel_index[geo.solid_id[tri[ok]]]
        np.bincount(chan[chan >= 0], minlength=gg.nchannels)
        dt = time.perf_counter() - t0
        cpu_baseline = {'value': sample / dt, 'unit': 'photons/s', 'cores': cores, 'kind': 'port',
                        'sample': '%d photons of the same bomb on %s, oracle/chroma_oracle.c on %d host threads, '
                                  'propagate + hit histogram, %.1f s wall' % (sample, args.config, cores, dt)}
        log('cpu baseline: %.3g photons/s on %d threads' % (sample / dt, cores))
        if args.config == 'tiny':
            # the pure-NumPy restatement, 1e4 photons, one core (numpy is single-threaded here)
            from oracle import numpy_propagate as npp
            tab = npp.Tables(packed_for_cpu)
            ph1 = oracle.generate_bomb(10_000, seed=ENGINE_SEED, id_base=0, wavelength_lo=wl_lo, wavelength_hi=wl_hi)
            npp.propagate(packed_for_cpu, ph1, seed=1, max_steps=args.max_steps, tables=tab)          # warm-up
            t0 = time.perf_counter()
            reps = 3
            for r in range(reps):
                npp.propagate(packed_for_cpu, ph1, seed=2 + r, max_steps=args.max_steps, tables=tab)
            dtn = (time.perf_counter() - t0) / reps
            cpu_baseline['numpy_c1'] = {'value': 10_000 / dtn, 'unit': 'photons/s', 'cores': 1,
                                        'sample': '1e4 photons of the same bomb, oracle/numpy_propagate.py, mean of %d runs, %.2f s each' % (reps, dtn)}
            log('numpy C1 baseline: %.3g photons/s on 1 core' % (10_000 / dtn))

    if rank == 0:
        result = {
            'metric': 'photons/sec (propagate_hit)', 'value': value, 'unit': 'photons/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': value / 2.5e6 if args.config == 'c3' else None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': desc, 'geometry': args.config, 'triangles': int(len(geo.mesh.triangles)),
                       'bvh_nodes': int(len(geo.bvh.nodes)), 'channels': int(geo.num_channels()),
                       'photons_per_gpu_per_step': nphotons, 'max_steps': args.max_steps,
                       'wavelength_nm': [wl_lo, wl_hi] if wl_hi > wl_lo else wl_lo,
                       'engine_seed': ENGINE_SEED, 'parallelism': 'photon shards x%d, geometry replicated' % world,
                       'steps_per_photon': steps_pp, 'nodes_per_step': nodes_ps, 'triangle_tests_per_step': tris_ps,
                       'geometry_build_s': t_build},
            'roofline': {'bound': 'hbm', 'kernel': RAYCAST_KERNEL, 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'algorithmic_bytes_per_launch': ray_bytes_total / ray_launches,
                         'algorithmic_bytes_per_photon_step': ray_bytes_per_step,
                         'launches': int(stats['raycast_launches']), 'avg_launch_ms': 1e3 * ray_s / ray_launches,
                         'kernel_s': ray_s,
                         'path': {'achieved': path_achieved, 'frac': path_achieved / HBM_PEAK_GBS,
                                  'algorithmic_bytes_per_photon': bytes_per_photon, 'launches': int(stats['launches']),
                                  'kernel_s': kernel_s}},
            'cpu_baseline': cpu_baseline,
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
