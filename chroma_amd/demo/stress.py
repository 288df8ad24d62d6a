"""Divergent-path stress geometry: every physics branch of chroma/cuda/photon.h is reachable.

Built with the public Material / Surface / DichroicProps API only.
"""
import numpy as np


def scintillator_stress():
    """Small geometry that reaches every physics branch (BASELINE.md C5 in miniature): a cube
    of 2-component re-emitting scintillator whose faces carry a thin-film (complex) surface,
    a WLS surface, a dichroic surface and a default detecting surface, inside a vacuum world
    box with a black surface."""
    from chroma_amd.geometry import Solid, Material, Surface, DichroicProps, vacuum, standard_wavelengths
    from chroma_amd.detector import Detector
    from chroma_amd.make import box
    wl = standard_wavelengths.astype(float)
    scint = Material('scint')
    scint.set('refractive_index', 1.5)
    scint.set('absorption_length', 50.0)
    scint.set('scattering_length', 200.0)
    cdf = np.clip((wl - 400.0) / 100.0, 0.0, 1.0)
    tgrid = np.arange(0, 1000, 0.05)
    tcdf = 1.0 - np.exp(-tgrid / 5.0)
    tcdf /= tcdf[-1]
    for prob, share in ((0.8, 75.0), (0.3, 150.0)):
        p = Material('tmp'); p.set('x', prob)
        scint.comp_reemission_prob.append(p.x)
        c = Material('tmp'); c.set('x', cdf)
        scint.comp_reemission_wvl_cdf.append(c.x)
        scint.comp_reemission_time_cdf.append(np.column_stack([tgrid, tcdf]).astype(np.float32))
        a = Material('tmp'); a.set('x', share)
        scint.comp_absorption_length.append(a.x)

    film = Surface('film', model=1)
    film.set('detect', 0.3); film.set('eta', 2.0); film.set('k', 1.5)
    film.set('reflect_diffuse', 0.2)
    film.thickness = 20e-6      # mm
    film.transmissive = 1
    wls = Surface('wls', model=2)
    wls.set('absorb', 0.5); wls.set('reemit', 0.7); wls.set('reflect_specular', 0.1); wls.set('reflect_diffuse', 0.1)
    wls.set('reemission_cdf', cdf)
    dich = Surface('dichroic', model=3)
    angles = np.array([0.0, 0.4, 0.8, 1.2, np.pi / 2])
    refl = [np.column_stack([wl, np.clip(0.2 + 0.1 * k + (wl - 300) / 2000.0, 0, 0.9)]) for k in range(5)]
    tran = [np.column_stack([wl, np.clip(0.6 - 0.1 * k - (wl - 300) / 4000.0, 0, 0.9) * 0.9]) for k in range(5)]
    dich.dichroic_props = DichroicProps(angles, refl, tran)
    pmt = Surface('pmt')
    pmt.set('detect', 0.4); pmt.set('absorb', 0.2); pmt.set('reflect_diffuse', 0.2); pmt.set('reflect_specular', 0.1)
    black = Surface('black'); black.set('absorb', 1.0)

    mesh = box(200.0, 200.0, 200.0)
    ntri = len(mesh.triangles)
    surfaces = np.empty(ntri, dtype=object)
    for i, s in enumerate(np.array_split(np.arange(ntri), 4)):
        surfaces[s] = [film, wls, dich, pmt][i]
    det = Detector(vacuum)
    det.add_pmt(Solid(mesh, scint, vacuum, surface=surfaces))
    det.add_solid(Solid(box(2000.0, 2000.0, 2000.0), vacuum, vacuum, surface=black))
    return det


