#!/opt/conda/bin/python3.9
"""Golden vectors for the model surface around the hot path: TableModel
as a particle distribution, EblAbsorptionModel, PionDecayKelner06.  Produced by running
THE REFERENCE in the build container (same loader as gen_golden.py):

    /opt/conda/bin/python3.9 tests/golden/gen_golden_extra.py

Writes tests/golden/extra.npz (inputs + expected outputs; data only)."""
import importlib
import os
import sys
import types
import warnings

import numpy as np

for n, f in (("asscalar", lambda a: a.item()), ("alen", len), ("rank", np.ndim)):
    if not hasattr(np, n):
        setattr(np, n, f)
SRC = "/root/reference/src/naima"
pkg = types.ModuleType("naima")
pkg.__path__ = [SRC]
pkg.__file__ = SRC + "/__init__.py"
pkg.__package__ = "naima"
sys.modules["naima"] = pkg
sys.modules.setdefault("emcee", types.ModuleType("emcee"))
for m in ("extern", "extern.validator", "utils", "model_utils", "radiative", "models", "core"):
    importlib.import_module("naima." + m)
warnings.simplefilter("ignore")
import astropy.units as u  # noqa: E402

import naima.models as nmodels  # noqa: E402
import naima.radiative as nrad  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}

# ---- TableModel (models.py:425-467) ------------------------------------------------
te = np.logspace(9, 15, 40)  # eV
tv = 3e33 * (te / 1e13) ** -2.3 * np.exp(-((te / 8e13) ** 1.2))  # 1/eV
out["tm_energy_eV"], out["tm_values_per_eV"] = te, tv
tm = nmodels.TableModel(te * u.eV, tv / u.eV, amplitude=2.5)
ecall = np.concatenate([[5e8], np.logspace(9, 15, 17), [2e15]])
out["tm_call_e_eV"] = ecall
out["tm_call"] = tm(ecall * u.eV).to("1/eV").value
Eg = np.logspace(8, 14.5, 21)
Ex = np.logspace(1, 5, 13)
out["tm_Egamma_eV"], out["tm_Ex_eV"] = Eg, Ex
ic = nrad.InverseCompton(tm, seed_photon_fields=["CMB", "FIR"])
out["tm_ic_flux"] = ic.flux(Eg * u.eV, 1.5 * u.kpc).to("1/(s cm2 eV)").value
out["tm_We_gt_1TeV_erg"] = ic.compute_We(Eemin=1 * u.TeV).to("erg").value
syn = nrad.Synchrotron(tm, B=15 * u.uG)
out["tm_syn_flux"] = syn.flux(Ex * u.eV, 1.5 * u.kpc).to("1/(s cm2 eV)").value
br = nrad.Bremsstrahlung(tm, n0=2 / u.cm ** 3)
out["tm_brems_flux"] = br.flux(Eg * u.eV, 1.5 * u.kpc).to("1/(s cm2 eV)").value
tp = nmodels.TableModel(te * u.eV, tv / u.eV)
pp = nrad.PionDecay(tp, nh=3 / u.cm ** 3, useLUT=False)
out["tm_pp_flux"] = pp.flux(Eg * u.eV, 1.5 * u.kpc).to("1/(s cm2 eV)").value
out["tm_Wp_erg"] = pp.Wp.to("erg").value

# ---- EblAbsorptionModel (models.py:470-552) ----------------------------------------
ee = np.concatenate([[2e8, 9.99e8], np.logspace(9, 14, 31), [1.0001e14, 3e14]])  # eV
out["ebl_e_eV"] = ee
for z in (0.005, 0.5, 1.234, 3.99):
    ebl = nmodels.EblAbsorptionModel(z)
    out["ebl_transmission_z%s" % z] = ebl.transmission(ee * u.eV)
    inside = (ee >= 1e9) & (ee <= 1e14)
    out["ebl_call_z%s" % z] = np.asarray(ebl(ee[inside] * u.eV).value, dtype=float)

# ---- PionDecayKelner06 (radiative.py:1543-1767) ------------------------------------
Ek = np.logspace(9.5, 14.5, 14)  # eV: both sides of Etrans = 0.1 TeV
out["k06_E_eV"] = Ek
pl = nmodels.PowerLaw(4e35 / u.eV, 1 * u.TeV, 2.2)
ecpl = nmodels.ExponentialCutoffPowerLaw(4e35 / u.eV, 1 * u.TeV, 2.0, 100 * u.TeV)
for tag, pdist in (("pl", pl), ("ecpl", ecpl)):
    k = nrad.PionDecayKelner06(pdist, nh=2 / u.cm ** 3)
    out["k06_%s_flux" % tag] = k.flux(Ek * u.eV, 1 * u.kpc).to("1/(s cm2 eV)").value
    out["k06_%s_nhat" % tag] = float(k.nhat)
khi = nrad.PionDecayKelner06(pl, nh=2 / u.cm ** 3)
out["k06_pl_flux_hi_only"] = khi.flux(Ek[Ek >= 1e11] * u.eV, 1 * u.kpc).to("1/(s cm2 eV)").value

np.savez_compressed(os.path.join(HERE, "extra.npz"), **out)
for k_, v in sorted(out.items()):
    print(k_, np.shape(v))
