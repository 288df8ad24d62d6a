"""The two request/reply servers of the reference, around ``Simulation.simulate``.

``ChromaServer`` (bin/chroma-server:11-40) exchanges pickled ``Photons``; ``RatServer``
(bin/chroma-server-rat:28-71) speaks the packed binary format RAT-PAC's chroma client sends:

request   u32 nphotons, u32 eventid,
          f64[n] x, y, z, dx, dy, dz, polx, poly, polz, wavelength, t   (eleven arrays, in this order),
          u32[n] track id
reply     u32 nhits, u32 eventid,
          f32[nhits] x, y, z, dx, dy, dz, polx, poly, polz, wavelength, t  -- the detected photons, grouped
          by channel (channels ascending, the order of ``ev.hits``),
          u32[nhits] channel index, twice (the first copy stands in for the track id the reference never
          filled in, chroma-server-rat:68-69)

The codec is plain functions on ``bytes`` so that it can be used and tested without a socket; the
serving loops take any object with the ZeroMQ REP socket's methods.  ZeroMQ itself is imported only by
``bind`` (it is not part of this image).
"""
import numpy as np

from .event import Photons

_RAT_FIELDS = 11


def decode_rat_request(msg):
    """bytes -> (Photons, eventid, track ids).

    The reference slices the track ids from byte ``88 n`` -- eight bytes early, the header not counted
    (chroma-server-rat:36) -- and never uses them; here they are read from where the client puts them."""
    msg = bytes(msg)
    if len(msg) < 8:
        raise ValueError('RAT request shorter than its 8-byte header')
    n, eventid = (int(v) for v in np.frombuffer(msg, dtype=np.uint32, count=2))
    body = 8 * _RAT_FIELDS * n
    if len(msg) < 8 + body:
        raise ValueError('RAT request announces %d photons but holds %d bytes' % (n, len(msg)))
    cols = np.frombuffer(msg, dtype=np.float64, count=_RAT_FIELDS * n, offset=8).reshape(_RAT_FIELDS, n)
    rest = len(msg) - 8 - body
    tracks = np.frombuffer(msg, dtype=np.uint32, count=min(n, rest // 4), offset=8 + body).copy()
    photons = Photons(cols[0:3].T, cols[3:6].T, cols[6:9].T, cols[9], cols[10])
    return photons, eventid, tracks


def encode_rat_request(photons, eventid, track_ids=None):
    """The client's side of the format (tests, and a Python client)."""
    n = len(photons)
    tracks = np.zeros(n, dtype=np.uint32) if track_ids is None else np.asarray(track_ids, dtype=np.uint32)
    cols = [photons.pos[:, 0], photons.pos[:, 1], photons.pos[:, 2], photons.dir[:, 0], photons.dir[:, 1],
            photons.dir[:, 2], photons.pol[:, 0], photons.pol[:, 1], photons.pol[:, 2], photons.wavelengths, photons.t]
    return (np.asarray([n, eventid], dtype=np.uint32).tobytes() +
            b''.join(np.ascontiguousarray(c, dtype=np.float64).tobytes() for c in cols) + tracks.tobytes())


def encode_rat_reply(hits, eventid):
    """``ev.hits`` (channel -> Photons) -> bytes (chroma-server-rat:45-69)."""
    chans = list(hits.keys())
    parts = [hits[c] for c in chans]
    hit = Photons.join(parts) if parts else Photons()
    chanidx = (np.concatenate([np.full(len(hits[c]), c, dtype=np.uint32) for c in chans])
               if chans else np.empty(0, dtype=np.uint32))
    cols = [hit.pos[:, 0], hit.pos[:, 1], hit.pos[:, 2], hit.dir[:, 0], hit.dir[:, 1], hit.dir[:, 2],
            hit.pol[:, 0], hit.pol[:, 1], hit.pol[:, 2], hit.wavelengths, hit.t]
    return (np.asarray([len(hit), eventid], dtype=np.uint32).tobytes() +
            b''.join(np.ascontiguousarray(c, dtype=np.float32).tobytes() for c in cols) +
            chanidx.tobytes() + chanidx.tobytes())


def decode_rat_reply(msg):
    """bytes -> (Photons with ``channel`` set, eventid): what the RAT side reads."""
    msg = bytes(msg)
    n, eventid = (int(v) for v in np.frombuffer(msg, dtype=np.uint32, count=2))
    cols = np.frombuffer(msg, dtype=np.float32, count=_RAT_FIELDS * n, offset=8).reshape(_RAT_FIELDS, n)
    chan = np.frombuffer(msg, dtype=np.uint32, count=n, offset=8 + 4 * _RAT_FIELDS * n + 4 * n)
    return Photons(cols[0:3].T, cols[3:6].T, cols[6:9].T, cols[9], cols[10], channel=chan.copy()), eventid


def bind(address):
    """A ZeroMQ REP socket bound to ``address``."""
    try:
        import zmq
    except ImportError as exc:
        raise ImportError('the chroma servers need pyzmq, which is not installed: %s' % exc)
    socket = zmq.Context.instance().socket(zmq.REP)
    socket.bind(address)
    return socket


class ChromaServer(object):
    """Listens for pickled ``Photons`` and replies with their final states (bin/chroma-server:11-40)."""

    def __init__(self, address, detector, socket=None, sim=None):
        self.address = address
        self.socket = socket if socket is not None else bind(address)
        self.detector = detector
        if sim is None:
            from .sim import Simulation
            sim = Simulation(detector)
        self.sim = sim

    def handle_one(self):
        photons_in = self.socket.recv_pyobj()
        ev = next(self.sim.simulate(photons_in, keep_photons_end=True))
        self.socket.send_pyobj(ev.photons_end)

    def serve_forever(self):
        while True:
            self.handle_one()


class RatServer(object):
    """The packed-binary server RAT-PAC talks to (bin/chroma-server-rat:19-71): every request is one
    event; the reply holds every photon detected on a channel (the DAQ runs on the RAT side)."""

    def __init__(self, address, detector, socket=None, sim=None, max_steps=1000):
        self.address = address
        self.socket = socket if socket is not None else bind(address)
        if sim is None:
            from .sim import Simulation
            sim = Simulation(detector)
        self.sim = sim
        self.max_steps = max_steps

    def reply_to(self, msg):
        photons, eventid, _ = decode_rat_request(msg)
        ev = next(self.sim.simulate(photons, keep_photons_beg=False, keep_photons_end=False, keep_hits=True,
                                    run_daq=False, max_steps=self.max_steps))
        return encode_rat_reply(ev.hits if ev.hits is not None else {}, eventid)

    def handle_one(self):
        self.socket.send(self.reply_to(self.socket.recv()))

    def serve_forever(self):
        while True:
            self.handle_one()


def main_server(argv=None):
    """bin/chroma-server <detector> [--address tcp://127.0.0.1:5024]

    The request/reply payloads are PICKLES (recv_pyobj / send_pyobj, as bin/chroma-server:24-38): whoever can reach
    the socket can run code in this process.  The default therefore listens on the loopback interface only (the
    reference binds tcp://*:5024); pass --address to serve a trusted network, or use chroma-server-rat, whose
    packed binary format carries numbers only."""
    import argparse
    from .loader import load_geometry_from_string
    ap = argparse.ArgumentParser(description='Serves a chroma geometry on a ZeroMQ socket: pickled Photons in, final Photons out')
    ap.add_argument('detector', help='a chroma geometry identifier string')
    ap.add_argument('--address', default='tcp://127.0.0.1:5024')
    args = ap.parse_args(argv)
    ChromaServer(args.address, load_geometry_from_string(args.detector)).serve_forever()


def main_server_rat(argv=None):
    """bin/chroma-server-rat <detector> [--address ipc:///tmp/ipc_chroma]"""
    import argparse
    from .loader import load_geometry_from_string
    ap = argparse.ArgumentParser(description='Serves a chroma geometry on a ZeroMQ socket that speaks a language RAT understands')
    ap.add_argument('detector', help='a chroma geometry identifier string')
    ap.add_argument('--address', '-a', default='ipc:///tmp/ipc_chroma')
    args = ap.parse_args(argv)
    RatServer(args.address, load_geometry_from_string(args.detector)).serve_forever()
