"""Numerical constants of the path: what the reference computes from astropy's
CODATA-2018 set at import time (radiative.py:11,34-40; values measured by importing
the reference)."""
from . import units as u

E_GAUSS = 4.803204712570263e-10
C_CGS = 29979245800.0
HBAR_CGS = 1.0545718176461565e-27
M_E_G = 9.1093837015e-28
ALPHA_FS = 0.0072973525693
MEC2_ERG = 8.187105776823886e-07
MEC2_EV = 510998.9499961643
AR_CGS = 7.565733250280007e-15
R0_CM = 2.817940324670788e-13
ERG_PER_EV = 1.602176634e-12
ERG_TO_EV = 624150907446.0764
M_P_GEV = 0.9382720881604903
T_TH_GEV = 0.27966184

# Quantity spellings used by model functions and tests (astropy.constants names)
c = C_CGS * u.cm / u.s
m_e = M_E_G * u.g
mec2 = MEC2_ERG * u.erg
mec2_unit = u.Unit(mec2)
sigma_sb = (AR_CGS * C_CGS / 4.0) * u.Unit("erg/(cm2 s K4)")

# astropy's own rounding of (energy unit)/erg and (energy unit)->GeV, measured with
# astropy 4.3.1.  Only the particle-grid sizing needs them: the point count is
# int(nEed * (log10(Emax/mec2) - log10(Emin/mec2))) (radiative.py:152-154), an int()
# truncation of a float, so the last bit of the unit ratio can change n by one.
ASTROPY_TO_ERG = {"eV": 1.602176634e-12, "keV": 1.6021766339999998e-09,
                  "MeV": 1.6021766339999998e-06, "GeV": 0.0016021766339999997,
                  "TeV": 1.6021766339999999, "PeV": 1602.176634, "erg": 1.0, "J": 10000000.0}
ASTROPY_TO_GEV = {"eV": 1e-09, "keV": 1e-06, "MeV": 0.001, "GeV": 1.0, "TeV": 1000.0,
                  "PeV": 1000000.0, "erg": 624.1509074460764, "J": 6241509074.460763}


def energy_ratio_to(q, table, fallback_unit):
    """q (energy Quantity) expressed through astropy's factor when its unit is a
    named one, else through this package's own conversion"""
    f = table.get(q.unit.name)
    if f is None:
        return q.to(fallback_unit).value
    return q.value * f
