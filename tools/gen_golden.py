#!/opt/conda/bin/python3.9
"""Generate tests/golden/ref_host_model.npz by IMPORTING the reference's pure-NumPy modules.

Run in the build container only:   /opt/conda/bin/python3.9 tools/gen_golden.py
(the reference's Python needs NumPy 1.x; /root/reference is read, never copied).

What is captured (inputs + the reference's outputs, no reference code):
  * the optics tables of chroma/demo/optics.py resampled on the 188-point grid exactly as
    chroma/gpu/geometry.py:41-45 does (np.interp -> float32);
  * meshes produced by chroma/make.py for fixed arguments (vertices, triangles);
  * chroma/tools.py offset() and chroma/pmt.py build_pmt()/build_light_collector_from_file()
    on THIS repo's analytic PMT outline, written to a temporary CSV;
  * chroma/demo spherical_spiral() points; chroma/tools.py argsort_direction();
  * Geometry.flatten()/Detector.flatten() of a small detector, with materials/surfaces
    recorded by NAME per triangle (their numeric order is nondeterministic in the reference);
  * chroma/transform.py rotate / make_rotation_matrix; chroma/sample.py uniform_sphere.
chroma/__init__.py is bypassed (it imports pygame); pycuda/pytools/particle are absent and
are only referenced at import time by modules that are not exercised here.
"""
import os
import sys
import tempfile
import types

import numpy as np

REF = '/root/reference'
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

pkg = types.ModuleType('chroma')
pkg.__path__ = [REF + '/chroma']
sys.modules['chroma'] = pkg
for name in ['pycuda', 'pycuda.gpuarray', 'pycuda.driver', 'pycuda.tools', 'pycuda.compiler',
             'pycuda.characterize', 'pytools', 'particle']:
    sys.modules[name] = types.ModuleType(name)

import chroma.geometry as RG          # noqa: E402
import chroma.make as RM              # noqa: E402
import chroma.tools as RT             # noqa: E402
import chroma.pmt as RP               # noqa: E402
import chroma.detector as RD          # noqa: E402
import chroma.transform as RX         # noqa: E402
import chroma.sample as RS            # noqa: E402
import chroma.demo.optics as RO       # noqa: E402

from chroma_amd.demo.pmt import pmt_outline, cone_outline   # this repo's analytic outline (inputs)

out = {}

# ---- optics tables -----------------------------------------------------------------------
wl = RG.standard_wavelengths


def resample(prop):
    return np.interp(wl, prop[:, 0], prop[:, 1]).astype(np.float32)


for mname in ('water', 'glass', 'vacuum'):
    m = getattr(RO, mname)
    for prop in ('refractive_index', 'absorption_length', 'scattering_length'):
        out['optics/%s/%s' % (mname, prop)] = resample(getattr(m, prop))
for sname in ('black_surface', 'shiny_surface', 'lambertian_surface', 'glossy_surface', 'red_absorb_surface',
              'r7081hqe_photocathode'):
    s = getattr(RO, sname)
    for prop in ('detect', 'absorb', 'reemit', 'reflect_diffuse', 'reflect_specular', 'eta', 'k', 'reemission_cdf'):
        out['optics/%s/%s' % (sname, prop)] = resample(getattr(s, prop))
out['optics/wavelengths'] = wl

# ---- make.py meshes ------------------------------------------------------------------------
meshes = {
    'box': RM.box(100.0, 200.0, 300.0, center=(1.0, 2.0, 3.0)),
    'cube': RM.cube(1000.0),
    'sphere': RM.sphere(50.0, nsteps=16),
    'cylinder': RM.cylinder(10.0, 40.0, radius2=5.0, nsteps=12),
    'segmented_cylinder': RM.segmented_cylinder(10.0, 30.0, nsteps=8, nsegments=20),
    'torus': RM.torus(5.0, 20.0, nsteps=8, circle_steps=6),
    'cylinder_along_z': RM.cylinder_along_z(7.0, 3.0, points=10),
    'bipyramid': RM.rotate_extrude([0, 1, 0], [-1, 0, 1], nsteps=4),
    'polygon': RM.convex_polygon(np.array([0.0, 1.0, 1.0, 0.0]), np.array([0.0, 0.0, 1.0, 1.0])),
}
for k, m in meshes.items():
    out['mesh/%s/vertices' % k] = m.vertices
    out['mesh/%s/triangles' % k] = m.triangles

# ---- PMT pipeline on this repo's outline ------------------------------------------------------
tmp = tempfile.mkdtemp()
pmt_csv = os.path.join(tmp, 'pmt.txt')
cone_csv = os.path.join(tmp, 'cone.txt')
outline = pmt_outline()
full = np.vstack([outline, outline * np.array([-1.0, 1.0])])      # both x signs, as a drawing has
np.savetxt(pmt_csv, full, delimiter=',')
np.savetxt(cone_csv, cone_outline(), delimiter=',')
out['pmt/outline_full'] = full
out['pmt/cone_outline'] = cone_outline()
half = RT.read_csv(pmt_csv)
half = half[half[:, 0] < 0]
half[:, 0] = -half[:, 0]
half = half[np.argsort(half[:, 1])]
half[0, 0] = 0.0
half[-1, 0] = 0.0
out['pmt/offset_in'] = half
out['pmt/offset_out'] = RT.offset(half, -3.0)
pmt = RP.build_pmt(pmt_csv, 3.0, outer_material=RO.water, glass=RO.glass, vacuum=RO.vacuum,
                   photocathode_surface=RO.r7081hqe_photocathode, back_surface=RO.shiny_surface, nsteps=24)
out['pmt/vertices'] = pmt.mesh.vertices
out['pmt/triangles'] = pmt.mesh.triangles
out['pmt/surface_names'] = np.array(['' if s is None else s.name for s in pmt.surface])
out['pmt/inner_names'] = np.array([m.name for m in pmt.inner_material])
out['pmt/color'] = pmt.color
lc = RP.build_light_collector_from_file(cone_csv, outer_material=RO.water, surface=RO.shiny_surface, nsteps=24)
out['lc/vertices'] = lc.mesh.vertices
out['lc/triangles'] = lc.mesh.triangles

# ---- spiral, direction sort, transforms ----------------------------------------------------------
import importlib.util                                                            # noqa: E402
# chroma/demo/__init__.py imports stl/checkerboard helpers; take spherical_spiral's text-free
# behaviour by calling it through a module object created from its source file.
spec = importlib.util.spec_from_file_location('chroma.demo', REF + '/chroma/demo/__init__.py',
                                              submodule_search_locations=[REF + '/chroma/demo'])
try:
    demo = importlib.util.module_from_spec(spec)
    sys.modules['chroma.demo'] = demo
    spec.loader.exec_module(demo)
    out['spiral/points'] = np.array(list(demo.spherical_spiral(2000.0, 700.0)))
    out['spiral/count_14000_350'] = np.array(sum(1 for _ in demo.spherical_spiral(14000.0, 350.0)))
    out['spiral/count_23780_350'] = np.array(sum(1 for _ in demo.spherical_spiral(23780.0, 350.0)))
except Exception as exc:      # pragma: no cover
    print('spherical_spiral not captured:', exc)

rng = np.random.RandomState(7)
dirs = RX.normalize(rng.normal(size=(500, 3)))
out['dirsort/dirs'] = dirs
out['dirsort/order'] = RT.argsort_direction(dirs)
pts = rng.normal(size=(20, 3))
out['transform/points'] = pts
out['transform/rotated'] = RX.rotate(pts, 0.7, (1.0, 2.0, -0.5))
out['transform/matrix'] = RX.make_rotation_matrix(0.7, (1.0, 2.0, -0.5))
np.random.seed(11)
out['sample/uniform_sphere'] = RS.uniform_sphere(100)

# ---- flatten of a small detector ---------------------------------------------------------------------
det = RD.Detector(RO.water)
det.add_solid(RG.Solid(RM.box(1000.0, 1000.0, 1000.0), RO.water, RO.vacuum, surface=RO.black_surface, color=0x11))
sph = RM.sphere(30.0, nsteps=8)
surf = np.where(np.mean(sph.assemble(), axis=1)[:, 1] > 0, RO.r7081hqe_photocathode, RO.shiny_surface)
ball = RG.Solid(sph, RO.glass, RO.water, surface=surf, color=0x22)
for k, (angle, disp) in enumerate([(0.0, (100.0, 0.0, 0.0)), (0.5, (-100.0, 50.0, 0.0)), (1.1, (0.0, -120.0, 60.0))]):
    det.add_pmt(ball, RX.make_rotation_matrix(angle, (0.0, 0.0, 1.0)), disp)
det.add_solid(RG.Solid(RM.cube(20.0), RO.glass, RO.water), displacement=(200.0, 200.0, 200.0))
det.flatten()
out['flatten/vertices'] = det.mesh.vertices
out['flatten/triangles'] = det.mesh.triangles
out['flatten/solid_id'] = det.solid_id
out['flatten/colors'] = det.colors
out['flatten/inner_names'] = np.array([det.unique_materials[i].name for i in det.inner_material_index])
out['flatten/outer_names'] = np.array([det.unique_materials[i].name for i in det.outer_material_index])
out['flatten/surface_names'] = np.array(['' if i == -1 else det.unique_surfaces[i].name for i in det.surface_index])
out['flatten/solid_id_to_channel_index'] = det.solid_id_to_channel_index
out['flatten/channel_index_to_solid_id'] = det.channel_index_to_solid_id
out['flatten/channel_index_to_position'] = det.channel_index_to_position
det.set_time_dist_gaussian(1.5, -7.5, 7.5)
det.set_charge_dist_gaussian(1.0, 0.1, 0.0, 1.5)
out['flatten/time_cdf_x'], out['flatten/time_cdf_y'] = det.time_cdf
out['flatten/charge_cdf_x'], out['flatten/charge_cdf_y'] = det.charge_cdf

dst = os.path.join(REPO, 'tests', 'golden', 'ref_host_model.npz')
np.savez_compressed(dst, **out)
print('wrote %s: %d arrays, %.1f KB' % (dst, len(out), os.path.getsize(dst) / 1e3))
