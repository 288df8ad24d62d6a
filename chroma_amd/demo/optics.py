"""Demo optical properties: water, glass, vacuum and a handful of surfaces.

These are physical data tables, the same ones chroma's demo geometry uses
(chroma/demo/optics.py): the water tables come from WCSim, the glass from the SNO+
optics database ('glass_sno'), the photocathode efficiency from the Hamamatsu
R7081HQE data sheet.  They are kept here as one table per source so that the
benchmark geometries have the reference's optics.
"""
import numpy as np

from chroma_amd.geometry import Material, Surface

# --- trivial materials / surfaces -------------------------------------------------
vacuum = Material('vacuum')
vacuum.set('refractive_index', 1.0)
vacuum.set('absorption_length', 1e6)
vacuum.set('scattering_length', 1e6)

lambertian_surface = Surface('lambertian_surface')
lambertian_surface.set('reflect_diffuse', 1)

black_surface = Surface('black_surface')
black_surface.set('absorb', 1)

shiny_surface = Surface('shiny_surface')
shiny_surface.set('reflect_specular', 1)

glossy_surface = Surface('glossy_surface')
glossy_surface.set('reflect_diffuse', 0.5)
glossy_surface.set('reflect_specular', 0.5)

red_absorb_surface = Surface('red_absorb')
red_absorb_surface.set('absorb', [0.0, 0.0, 1.0], [465, 545, 685])
red_absorb_surface.set('reflect_diffuse', [1.0, 1.0, 0.0], [465, 545, 685])

# --- R7081HQE photocathode ----------------------------------------------------------
# quantum efficiency in percent at 260, 270, ... 710 nm (data sheet, serial ZD0062)
_R7081HQE_QE_PERCENT = [
    0.00, 0.04, 0.07, 0.77, 4.57, 11.80, 17.70, 23.50, 27.54, 30.52,
    31.60, 31.90, 32.20, 32.00, 31.80, 30.80, 30.16, 29.24, 28.31, 27.41,
    26.25, 24.90, 23.05, 21.58, 19.94, 18.48, 17.01, 15.34, 12.93, 10.17,
    7.86, 6.23, 5.07, 4.03, 3.18, 2.38, 1.72, 0.95, 0.71, 0.44,
    0.25, 0.14, 0.07, 0.03, 0.02, 0.00]
_qe_nm = np.arange(260.0, 711.0, 10.0)

r7081hqe_photocathode = Surface('r7081hqe_photocathode')
_detect = np.column_stack([_qe_nm, np.array(_R7081HQE_QE_PERCENT)])   # float64 table
_detect[:, 1] /= 100.0                                                # percent -> fraction
r7081hqe_photocathode.detect = _detect
# about as many photons are absorbed without being detected as are detected ...
r7081hqe_photocathode.absorb = _detect
# ... and the rest is reflected diffusely
r7081hqe_photocathode.set('reflect_diffuse', 1.0 - _detect[:, 1] - _detect[:, 1], wavelengths=_detect[:, 0])

# --- glass ---------------------------------------------------------------------------
glass = Material('glass')
glass.set('refractive_index', 1.49)
glass.absorption_length = np.array(
    [(200, 0.1e-6), (300, 0.1e-6), (330, 1000.0), (500, 2000.0),
     (600, 1000.0), (770, 500.0), (800, 0.1e-6), (1000, 0.1e-6)])
glass.set('scattering_length', 1e6)

# --- water (WCSim) ---------------------------------------------------------------------
# rows: photon energy [GeV], refractive index, absorption length [cm], Rayleigh length [cm]
_WCSIM_WATER = np.array([
    (1.56962e-09, 1.32885, 22.8154, 167024.4),
    (1.58974e-09, 1.32906, 28.6144, 158726.7),
    (1.61039e-09, 1.32927, 35.9923, 150742),
    (1.63157e-09, 1.32948, 45.4086, 143062.5),
    (1.65333e-09, 1.3297, 57.4650, 135680.2),
    (1.67567e-09, 1.32992, 72.9526, 128587.4),
    (1.69863e-09, 1.33014, 75, 121776.3),
    (1.72222e-09, 1.33037, 81.2317, 115239.5),
    (1.74647e-09, 1.3306, 120.901, 108969.5),
    (1.77142e-09, 1.33084, 160.243, 102958.8),
    (1.7971e-09, 1.33109, 193.797, 97200.35),
    (1.82352e-09, 1.33134, 215.045, 91686.86),
    (1.85074e-09, 1.3316, 227.786, 86411.33),
    (1.87878e-09, 1.33186, 243.893, 81366.79),
    (1.90769e-09, 1.33213, 294.113, 76546.42),
    (1.93749e-09, 1.33241, 321.735, 71943.46),
    (1.96825e-09, 1.3327, 342.931, 67551.29),
    (1.99999e-09, 1.33299, 362.967, 63363.36),
    (2.03278e-09, 1.33329, 378.212, 59373.25),
    (2.06666e-09, 1.33361, 449.602, 55574.61),
    (2.10169e-09, 1.33393, 740.143, 51961.24),
    (2.13793e-09, 1.33427, 1116.06, 48527.00),
    (2.17543e-09, 1.33462, 1438.78, 45265.87),
    (2.21428e-09, 1.33498, 1615.48, 42171.94),
    (2.25454e-09, 1.33536, 1769.86, 39239.39),
    (2.29629e-09, 1.33576, 2109.67, 36462.50),
    (2.33962e-09, 1.33617, 2304.13, 33835.68),
    (2.38461e-09, 1.3366, 2444.97, 31353.41),
    (2.43137e-09, 1.33705, 3076.83, 29010.30),
    (2.47999e-09, 1.33753, 4901.5, 26801.03),
    (2.53061e-09, 1.33803, 6666.57, 24720.42),
    (2.58333e-09, 1.33855, 7873.95, 22763.36),
    (2.63829e-09, 1.33911, 9433.81, 20924.88),
    (2.69565e-09, 1.3397, 10214.5, 19200.07),
    (2.75555e-09, 1.34033, 10845.8, 17584.16),
    (2.81817e-09, 1.341, 15746.9, 16072.45),
    (2.88371e-09, 1.34172, 20201.8, 14660.38),
    (2.95237e-09, 1.34248, 22025.8, 13343.46),
    (3.02438e-09, 1.34331, 21142.2, 12117.33),
    (3.09999e-09, 1.34419, 15083.9, 10977.70),
    (3.17948e-09, 1.34515, 11751, 9920.416),
    (3.26315e-09, 1.3462, 8795.34, 8941.407),
    (3.35134e-09, 1.34733, 8741.23, 8036.711),
    (3.44444e-09, 1.34858, 7102.37, 7202.470),
    (3.54285e-09, 1.34994, 6060.68, 6434.927),
    (3.64705e-09, 1.35145, 4498.56, 5730.429),
    (3.75757e-09, 1.35312, 3039.56, 5085.425),
    (3.87499e-09, 1.35498, 2232.2, 4496.467),
    (3.99999e-09, 1.35707, 1938, 3960.210),
    (4.13332e-09, 1.35943, 1811.58, 3473.413),
    (4.27585e-09, 1.36211, 1610.32, 3032.937),
    (4.42856e-09, 1.36518, 1338.7, 2635.746),
    (4.59258e-09, 1.36872, 1095.3, 2278.907),
    (4.76922e-09, 1.37287, 977.525, 1959.588),
    (4.95999e-09, 1.37776, 965.258, 1675.064),
    (5.16665e-09, 1.38362, 1082.86, 1422.710),
    (5.39129e-09, 1.39074, 876.434, 1200.004),
    (5.63635e-09, 1.39956, 633.723, 1004.528),
    (5.90475e-09, 1.41075, 389.87, 833.9666),
    (6.19998e-09, 1.42535, 142.011, 686.1063),
])
hc_over_GeV = 1.2398424468024265e-06   # h*c in GeV*nm
# tables are listed by rising energy; flip them so the wavelength rises
_water = _WCSIM_WATER[::-1]
wcsim_wavelengths = hc_over_GeV / _water[:, 0]

water = Material('water')
water.density = 1.0                                   # g/cm^3
water.composition = {'H': 0.1119, 'O': 0.8881}        # fraction by mass
water.set('refractive_index', wavelengths=wcsim_wavelengths, value=_water[:, 1])
water.set('absorption_length', wavelengths=wcsim_wavelengths, value=_water[:, 2] * 10.0)            # cm -> mm
water.set('scattering_length', wavelengths=wcsim_wavelengths, value=_water[:, 3] * 10.0 * 0.625)    # cm -> mm, tuned
