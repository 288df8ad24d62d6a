"""The request/reply servers (chroma_amd/server.py): the packed wire format of bin/chroma-server-rat and
the serving loops, driven through an in-memory socket and a stand-in simulation -- no ZeroMQ, no GPU."""
import struct

import numpy as np
import pytest

from chroma_amd import server
from chroma_amd.event import Photons


def _photons(n, seed=3):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    return Photons(rng.uniform(-100, 100, (n, 3)), d, np.roll(d, 1, axis=1), rng.uniform(300, 700, n), rng.uniform(0, 50, n))


def test_request_layout_is_the_reference_clients():
    """Header, eleven f64 columns in the order x y z dx dy dz polx poly polz wavelength t, u32 track ids
    (bin/chroma-server-rat:33-40), built here by hand."""
    n, eventid = 3, 77
    cols = [np.arange(n, dtype=np.float64) + 10 * k for k in range(11)]
    tracks = np.array([5, 6, 7], dtype=np.uint32)
    msg = struct.pack('<II', n, eventid) + b''.join(c.tobytes() for c in cols) + tracks.tobytes()
    ph, ev, tr = server.decode_rat_request(msg)
    assert ev == eventid and np.array_equal(tr, tracks)
    assert np.array_equal(ph.pos, np.stack(cols[0:3], axis=1).astype(np.float32))
    assert np.array_equal(ph.dir, np.stack(cols[3:6], axis=1).astype(np.float32))
    assert np.array_equal(ph.pol, np.stack(cols[6:9], axis=1).astype(np.float32))
    assert np.array_equal(ph.wavelengths, cols[9].astype(np.float32)) and np.array_equal(ph.t, cols[10].astype(np.float32))
    assert server.encode_rat_request(ph, eventid, tracks) == msg


def test_requests_that_lie_about_their_length_are_refused():
    with pytest.raises(ValueError):
        server.decode_rat_request(b'\x01\x00')
    with pytest.raises(ValueError):
        server.decode_rat_request(struct.pack('<II', 4, 0) + b'\x00' * 100)
    ph, ev, tr = server.decode_rat_request(struct.pack('<II', 0, 9))          # an empty event is fine
    assert len(ph) == 0 and ev == 9 and len(tr) == 0


def test_reply_groups_hits_by_channel_with_the_index_twice():
    a, b = _photons(4, 1), _photons(2, 2)
    hits = {3: a, 11: b}
    msg = server.encode_rat_reply(hits, 42)
    n = 6
    assert len(msg) == 8 + 4 * 11 * n + 2 * 4 * n
    assert struct.unpack('<II', msg[:8]) == (n, 42)
    x = np.frombuffer(msg, dtype=np.float32, count=n, offset=8)
    assert np.array_equal(x, np.concatenate([a.pos[:, 0], b.pos[:, 0]]))
    t = np.frombuffer(msg, dtype=np.float32, count=n, offset=8 + 4 * 10 * n)
    assert np.array_equal(t, np.concatenate([a.t, b.t]))
    chan = np.frombuffer(msg, dtype=np.uint32, count=2 * n, offset=8 + 4 * 11 * n)
    assert np.array_equal(chan, np.tile(np.array([3, 3, 3, 3, 11, 11], dtype=np.uint32), 2))
    back, ev = server.decode_rat_reply(msg)
    assert ev == 42 and np.array_equal(back.channel, chan[:n]) and np.array_equal(back.dir, np.concatenate([a.dir, b.dir]))
    empty = server.encode_rat_reply({}, 1)
    assert empty == struct.pack('<II', 0, 1)


class _Event(object):
    def __init__(self, photons):
        self.photons_end = photons
        ch = (np.arange(len(photons)) % 3).astype(np.uint32)
        self.hits = {int(c): photons[ch == c] for c in np.unique(ch)}


class _Sim(object):
    """Stands in for Simulation: 'propagates' by shifting the time, remembers how it was called."""
    def __init__(self):
        self.calls = []

    def simulate(self, photons, **kwargs):
        self.calls.append(kwargs)
        out = photons[np.arange(len(photons))]
        out.t = out.t + np.float32(1.0)
        yield _Event(out)


class _Socket(object):
    def __init__(self, inbox):
        self.inbox, self.outbox = list(inbox), []

    def recv(self):
        return self.inbox.pop(0)

    def send(self, msg):
        self.outbox.append(msg)

    recv_pyobj, send_pyobj = recv, send


def test_rat_server_answers_a_request_with_the_hits_of_that_event():
    ph = _photons(9)
    sock, sim = _Socket([server.encode_rat_request(ph, 5)]), _Sim()
    srv = server.RatServer('inproc://x', None, socket=sock, sim=sim)
    srv.handle_one()
    assert sim.calls == [dict(keep_photons_beg=False, keep_photons_end=False, keep_hits=True, run_daq=False, max_steps=1000)]
    hits, ev = server.decode_rat_reply(sock.outbox[0])
    assert ev == 5 and len(hits) == 9
    assert np.array_equal(hits.channel, np.repeat(np.arange(3, dtype=np.uint32), 3))          # grouped by channel
    order = np.concatenate([np.arange(9)[np.arange(9) % 3 == c] for c in range(3)])
    assert np.array_equal(hits.t, ph.t[order] + np.float32(1.0)) and np.array_equal(hits.pos, ph.pos[order])


def test_chroma_server_returns_the_final_photons():
    ph = _photons(5)
    sock = _Socket([ph])
    server.ChromaServer('inproc://y', None, socket=sock, sim=_Sim()).handle_one()
    assert np.array_equal(sock.outbox[0].t, ph.t + np.float32(1.0))


def test_binding_without_zeromq_says_so():
    try:
        import zmq      # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match='pyzmq'):
            server.bind('inproc://z')
