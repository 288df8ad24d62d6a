"""``chroma-sim``: run photon-bomb events through a detector and write the hits.

The reference driver (bin/chroma-sim:32-112) takes a detector string, generates events with
GEANT4 and writes a ROOT file; neither GEANT4 nor ROOT is part of this engine, so this driver
keeps the command-line shape (detector string, -n/--nevents, -o/--output, -s/--seed, -j device)
but the particle source is the isotropic photon bomb of chroma/benchmark.py:77-83 and the output is
an ``.npz``: per event the flat hits (channel, t, wavelength, pos, flags), with --run-daq the channel
times and charges, with --save-photons-beg/--save-photons-end the photons themselves and with --track
every photon's state after each step.
"""
import argparse
import sys
import time

import numpy as np


def bomb_event(nphotons, wavelength, pos, rng):
    from chroma_amd.event import Photons
    theta = rng.uniform(0, 2 * np.pi, nphotons)
    u = rng.uniform(-1, 1, nphotons)
    c = np.sqrt(1 - u * u)
    d = np.column_stack([c * np.cos(theta), c * np.sin(theta), u])
    theta = rng.uniform(0, 2 * np.pi, nphotons)
    u = rng.uniform(-1, 1, nphotons)
    c = np.sqrt(1 - u * u)
    a = np.column_stack([c * np.cos(theta), c * np.sin(theta), u])
    pol = np.cross(a, d)
    pol /= np.linalg.norm(pol, axis=1)[:, None]
    if isinstance(wavelength, tuple):
        wl = rng.uniform(wavelength[0], wavelength[1], nphotons)
    else:
        wl = np.full(nphotons, wavelength)
    return Photons(np.tile(np.asarray(pos, dtype=float), (nphotons, 1)), d, pol, wl)


def main(argv=None):
    ap = argparse.ArgumentParser(prog='chroma-sim', description=__doc__.split('\n\n')[0])
    ap.add_argument('detector', help='"@module.function" returning a Detector/Geometry, e.g. @chroma_amd.demo.tiny')
    ap.add_argument('-o', '--output', default='out.npz')
    ap.add_argument('-n', '--nevents', type=int, default=10)
    ap.add_argument('-s', '--seed', type=int, default=None)
    ap.add_argument('-j', '--device', type=int, default=None, help='GPU index')
    ap.add_argument('--nphotons', type=int, default=100000, help='photons per event')
    ap.add_argument('--wavelength', default='400', help='nm, or "lo:hi" for a uniform range')
    ap.add_argument('--pos', default='0,0,0')
    ap.add_argument('--max-steps', type=int, default=100)
    ap.add_argument('--run-daq', action='store_true')
    ap.add_argument('--save-photons-beg', action='store_true', help='also write the initial photons of every event (bin/chroma-sim:51-53)')
    ap.add_argument('--save-photons-end', action='store_true', help='also write the final photons of every event (bin/chroma-sim:54-56)')
    ap.add_argument('--track', action='store_true', help='also write every photon\'s state after each step (Simulation(photon_tracking=True))')
    ap.add_argument('--exact', action='store_true', help='the reference\'s own traversal loop for every ray: its hit triangle on EVERY ray, several times slower (Simulation(exact=True))')
    ap.add_argument('--cache-dir', default=None, help='directory of the BVH cache (off by default)')
    args = ap.parse_args(argv)

    from chroma_amd.loader import load_geometry_from_string
    from chroma_amd.sim import Simulation

    wl = tuple(float(x) for x in args.wavelength.split(':')) if ':' in args.wavelength else float(args.wavelength)
    pos = [float(x) for x in args.pos.split(',')]
    t0 = time.time()
    detector = load_geometry_from_string(args.detector, cache_dir=args.cache_dir)
    print('geometry: %d triangles, BVH %d nodes (%.1f s)' % (len(detector.mesh.triangles), len(detector.bvh.nodes), time.time() - t0))
    sim = Simulation(detector, seed=args.seed, cuda_device=args.device, geant4_processes=0, photon_tracking=args.track, exact=args.exact)
    rng = np.random.default_rng(sim.seed)
    events = (bomb_event(args.nphotons, wl, pos, rng) for _ in range(args.nevents))
    out = {'nevents': np.array(args.nevents), 'nphotons': np.array(args.nphotons), 'seed': np.array(sim.seed)}
    t0 = time.time()
    nhits = 0
    for ev in sim.simulate(events, keep_photons_beg=args.save_photons_beg, keep_photons_end=args.save_photons_end,
                           keep_hits=False, keep_flat_hits=hasattr(detector, 'num_channels'),
                           run_daq=args.run_daq, max_steps=args.max_steps):
        key = 'ev%d' % ev.id
        for tag, ph in (('photons_beg', ev.photons_beg), ('photons_end', ev.photons_end)):
            if ph is not None:
                for name in ('pos', 'dir', 'pol', 'wavelengths', 't', 'flags', 'last_hit_triangles', 'weights'):
                    out['%s/%s/%s' % (key, tag, name)] = getattr(ph, name)
        if args.track and getattr(ev, 'photon_tracks', None) is not None:
            # one row per (photon, step): ragged tracks flattened, with the photon index beside them
            tracks = ev.photon_tracks
            out[key + '/track/photon'] = np.concatenate([np.full(len(tr), i, dtype=np.uint32) for i, tr in enumerate(tracks)]
                                                         ) if tracks else np.empty(0, dtype=np.uint32)
            for name in ('pos', 'dir', 't', 'flags'):
                parts = [getattr(tr, name) for tr in tracks if len(tr)]
                out['%s/track/%s' % (key, name)] = np.concatenate(parts) if parts else np.empty(0)
        if ev.flat_hits is not None:
            h = ev.flat_hits
            nhits += len(h)
            out[key + '/channel'] = h.channel
            out[key + '/t'] = h.t
            out[key + '/wavelength'] = h.wavelengths
            out[key + '/pos'] = h.pos
            out[key + '/flags'] = h.flags
        if ev.channels is not None:
            out[key + '/daq_hit'] = ev.channels.hit
            out[key + '/daq_t'] = ev.channels.t
            out[key + '/daq_q'] = ev.channels.q
    dt = time.time() - t0
    np.savez_compressed(args.output, **out)
    print('%d events, %d photons, %d hits in %.2f s (%.1f events/s, %.3g photons/s) -> %s' % (
        args.nevents, args.nevents * args.nphotons, nhits, dt, args.nevents / dt, args.nevents * args.nphotons / dt, args.output))
    return 0


if __name__ == '__main__':
    sys.exit(main())
