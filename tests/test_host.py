"""CPU tests of the host side: units shim, grids (the int() truncation of
radiative.py:152-154), data ingest, the C ABI's symbol table, loud failure
without a GPU.  No compute call is made here."""
import ctypes
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

import naima_amd as na
from naima_amd import units as u

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_units_algebra_and_physical_types():
    q = 3 * u.TeV
    assert_allclose(q.to("erg").value, 3 * 1.602176634)
    assert (1 / u.eV).unit.physical_type == "differential energy"
    assert u.Unit("1/(s cm2 eV)").physical_type == "differential flux"
    assert u.Unit("erg/(cm2 s)").physical_type == "flux"
    assert u.Unit("1 / (cm2 s TeV)").physical_type == "differential flux"
    assert (0.5 * u.eV / u.cm ** 3).unit.physical_type == "pressure"
    assert (1 / u.cm ** 3).unit.physical_type == "number density"
    assert (1 / (u.eV * u.cm ** 3)).unit.physical_type == "differential number density"
    assert (12 * u.uG).to("G").value == pytest.approx(1.2e-5)
    assert (1 * u.kpc != 0) and not (0 * u.cm != 0)
    a = 10 ** np.array([1.0, 2.0]) / u.eV  # ndarray / Unit defers to Unit
    assert isinstance(a, u.Quantity) and a.shape == (2,)
    sed = (np.ones((2, 3)) * u.Unit("1/(s cm2 eV)") * (np.array([1.0, 2, 3]) * u.eV) ** 2)
    assert_allclose(sed.to("erg/(cm2 s)").value[0], np.array([1, 4, 9]) * 1.602176634e-12)
    with pytest.raises(u.UnitConversionError):
        (1 * u.TeV).to("cm")
    with pytest.raises(u.UnitConversionError):
        (1 * u.TeV) + (1 * u.cm)
    assert_allclose(u.Quantity([1 * u.eV, 1 * u.keV]).value, [1, 1000])
    assert_allclose((90 * u.deg).to("rad").value, np.pi / 2)


def test_grid_sizing_matches_reference(golden):
    """point counts and values of the particle grids for the golden specs"""
    from naima_amd.constants import mec2
    from naima_amd.radiative import BaseElectron
    U = golden("units")
    for i, (lo, hi, nd) in enumerate(U["grid_specs"]):
        g = BaseElectron._gam_between(lo * u.eV, hi * u.eV, nd)
        assert len(g) == U["grid_lens"][i]
        assert_allclose(g, U["grid_%d" % i], rtol=1e-14)
    # the cfg2 case sits exactly on an integer: 50 * 6.0 must not truncate to 299
    assert len(BaseElectron._gam_between(1 * u.GeV, 1 * u.PeV, 50)) == 300
    assert len(BaseElectron._gam_between(1 * u.GeV, 1e9 * mec2, 100)) == 570
    ECPL = na.ExponentialCutoffPowerLaw(1 / u.eV, 1 * u.TeV, 2.0, 10 * u.TeV)
    pp = na.PionDecay(ECPL)
    assert_allclose(pp._Ep, U["pgrid_default"], rtol=1e-14)
    pp = na.PionDecay(ECPL, Epmax=1 * u.PeV)
    assert_allclose(pp._Ep, U["pgrid_1PeV"], rtol=1e-14)


def test_constructor_validation_without_gpu():
    ECPL = na.ExponentialCutoffPowerLaw(1e36 / u.eV, 10 * u.TeV, 2.5, 50 * u.TeV)
    with pytest.raises(TypeError):
        na.Synchrotron(ECPL, B=1 * u.TeV)
    with pytest.raises(TypeError):
        na.InverseCompton(ECPL, seed_photon_fields=["XYZ"])
    with pytest.raises(TypeError):
        na.ExponentialCutoffPowerLaw(1e36 / u.eV, 10 * u.cm, 2.5, 50 * u.TeV)
    ic = na.InverseCompton(ECPL, seed_photon_fields=[
        "CMB", ["star", 25000 * u.K, 3 * u.erg / u.cm ** 3, 120 * u.deg],
        ["X-ray", [1, 10] * u.keV, [1, 1e-2] * (1 / (u.eV * u.cm ** 3))],
        ["UV", 50 * u.eV, 15 * u.eV / u.cm ** 3]])
    assert list(ic.seed_photon_fields) == ["CMB", "star", "X-ray", "UV"]
    assert ic.seed_photon_fields["star"]["isotropic"] is False
    assert ic.seed_photon_fields["X-ray"]["type"] == "array"
    # vector-valued parameters describe a walker batch
    pd = na.ExponentialCutoffPowerLaw(10 ** np.array([30.0, 31, 32]) / u.eV, 10 * u.TeV,
                                      np.array([2.0, 2.1, 2.2]), 50 * u.TeV)
    assert pd.batch_size == 3 and pd.is_batched
    rows = pd.param_rows(3, amplitude_to=u.Unit("1/eV"))
    assert rows.shape == (3, 8) and rows[1, 0] == 1e31 and rows[2, 2] == 2.2 and rows[0, 4] == 1.0
    assert na.Synchrotron(pd, B=np.array([1.0, 2, 3]) * u.uG).batch_size == 3
    with pytest.raises(ValueError):
        na.Synchrotron(pd, B=np.array([1.0, 2]) * u.uG).batch_size


def test_priors_vectorised():
    assert na.uniform_prior(1.0, 0, 2) == 0.0 and na.uniform_prior(3.0, 0, 2) == -np.inf
    assert_allclose(na.uniform_prior(np.array([1.0, 3.0]), 0, 2), [0.0, -np.inf])
    assert na.normal_prior(1.3, 1.0, 0.5) == pytest.approx(-0.5 * np.pi - 0.09)
    assert na.log_uniform_prior(2.0, 1.0, 3.0) == 0.5
    assert_allclose(na.log_uniform_prior(np.array([2.0, 4.0, -1.0]), 1.0, 3.0),
                    [0.5, -np.inf, -np.inf])


def test_workload_priors_against_the_reference(golden):
    """workloads.prior_for -- uniform priors in the spirit of examples/RXJ1713_SynIC.py:52-65 with
    physical bounds on the logarithmic parameters -- evaluated through this package's
    uniform_prior, one walker at a time and vectorised, against the same function evaluated
    through the reference's (core.py:34-41) on 48 scattered vectors per workload (gen_golden.py);
    and p0 with naima's 10 % ball around it lies inside every bound"""
    import naima_amd as na
    from naima_amd import workloads as W
    for name in ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5"):
        z = golden(name)
        prior = W.prior_for(name, na)
        assert prior is not None
        pars, want = z["prior_pars"], z["prior_lnprior"]
        assert np.isinf(want).sum() >= 8 and (want == 0).sum() >= 3
        got = np.array([float(prior(p)) for p in pars])
        assert np.array_equal(got, want), name
        assert np.array_equal(np.asarray(prior(pars.T), dtype=float), want), name
        p0 = np.asarray(W.WORKLOADS[name]["p0"], dtype=float)
        ball = p0 + 0.1 * p0 * np.random.default_rng(0).normal(size=(4096, p0.size))
        inside = np.asarray(prior(ball.T), dtype=float) == 0
        # (cfg1's amplitude is linear, cfg3's beta starts 2 sigma... every workload keeps > 95 %)
        assert inside.mean() > 0.95, (name, inside.mean())


def test_abi_exports_every_declared_symbol():
    """the library loads on a GPU-less box and exports include/naima_hip.h"""
    import __graft_entry__ as g
    from naima_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = g.declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), name
    assert set(_lib.EXPORTS) == set(declared)


def test_ssc_table_size_is_what_the_layout_says():
    """nh_ssc_table_bytes (host arithmetic only): window + dispatch order per (energy, gamma
    tile), padded to 256 bytes, then 64 lanes x 16 bytes per (energy, tile, seed node) -- the
    gamma grid in tiles of 63 segments; cfg4's grids 374 MB; degenerate sizes give 0"""
    from naima_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.nh_ssc_table_bytes.restype = ctypes.c_longlong
    lib.nh_ssc_table_bytes.argtypes = [ctypes.c_int] * 3

    def expect(nG, nE, ns):
        ntile = (nG - 1 + 62) // 63
        head = (nE * ntile * 12 + 255) // 256 * 256
        return head + nE * ntile * ns * 64 * 16

    for nG, nE, ns in [(869, 261, 100), (64, 1, 2), (65, 3, 7), (2, 5, 9), (3000, 70, 30)]:
        assert lib.nh_ssc_table_bytes(nG, nE, ns) == expect(nG, nE, ns)
    assert lib.nh_ssc_table_bytes(869, 261, 100) == 374_213_632
    assert lib.nh_ssc_table_bytes(1, 5, 9) == 0 and lib.nh_ssc_table_bytes(10, 0, 9) == 0
    assert lib.nh_ssc_table_bytes(10, 5, 1) == 0


def test_no_cpu_fallback():
    """without a GPU every compute path must raise, never silently fall back"""
    from naima_amd import _lib
    try:
        _lib.get_context()
    except _lib.NaimaHipError:
        pass
    else:
        pytest.skip("a GPU is present")
    ECPL = na.ExponentialCutoffPowerLaw(1e36 / u.eV, 10 * u.TeV, 2.5, 50 * u.TeV)
    with pytest.raises(_lib.NaimaHipError):
        na.Synchrotron(ECPL).flux(np.logspace(0, 3, 4) * u.eV)
    with pytest.raises(_lib.NaimaHipError):
        ECPL(np.logspace(9, 12, 4) * u.eV)
    # and nothing in the product imports the oracle
    import subprocess
    import sys
    code = ("import sys, naima_amd, naima_amd.sampler, naima_amd.dist, naima_amd.datatable, "
            "naima_amd.workloads; print(any(m.startswith('oracle') for m in sys.modules))")
    out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT).decode().strip()
    assert out == "False"


def test_data_ingest():
    from naima_amd import datatable as D
    n = 6
    t = D.DataTable()
    t["energy"] = np.geomspace(1, 30, n) * u.TeV
    t["flux"] = 1e-11 * np.ones(n) * u.Unit("1/(cm2 s TeV)")
    t["flux_error"] = 1e-12 * np.ones(n) * u.Unit("1/(cm2 s TeV)")
    t["ul"] = np.array([0, 0, 0, 0, 0, 1])
    t.meta = {"keywords": {"cl": {"value": 0.95}}}
    x = D.DataTable()
    x["energy"] = np.array([3.0, 1.0, 2.0]) * u.keV
    x["flux"] = np.array([3e-10, 1e-10, 2e-10]) * u.Unit("erg/(cm2 s)")
    x["flux_error_lo"] = 0.1 * x["flux"]
    x["flux_error_hi"] = 0.2 * x["flux"]
    d = D.validate_data_table([x, t])
    assert len(d) == 9 and d["flux"].unit.physical_type == "flux"
    assert np.all(np.diff(d["energy"].value) > 0)  # sorted, in the first table's unit
    assert_allclose(d["energy"].value[:3], [1, 2, 3])
    assert_allclose(d["flux"].value[:3], [1e-10, 2e-10, 3e-10])
    # TeV points converted to SED: E^2 * flux
    e = t["energy"].to("erg").value
    assert_allclose(d["flux"].value[3:], e ** 2 * 1e-11 / 1.602176634)
    assert d["ul"].sum() == 1 and d["ul"][-1]
    assert_allclose(d["cl"], [0.9] * 3 + [0.95] * 6)
    d2 = D.validate_data_table([x, t], sed=False)
    assert d2["flux"].unit.physical_type == "differential flux"
    with pytest.raises(TypeError):
        D.validate_data_table([{"energy": t["energy"]}])
    with pytest.raises(TypeError):
        D.validate_data_table(3)


def test_readers_roundtrip(tmp_path):
    from naima_amd import datatable as D
    p = tmp_path / "t.dat"
    p.write_text("\\ comment\n\\cl=0.95\n|energy|  flux|flux_error|  ul|\n|double|double|    double|long|\n"
                 "|   TeV|1 / (cm2 s TeV)|1 / (cm2 s TeV)|    |\n 0.33 2.29e-10 3.2e-11 0\n 0.4 1.25e-10 0.0 1\n")
    t = D.read(str(p))
    assert t["energy"].unit == u.TeV and t["ul"].tolist() == [0, 1]
    assert t.meta["keywords"]["cl"]["value"] == 0.95
    d = D.validate_data_table(t)
    assert d["cl"][0] == 0.95 and d["ul"][1]
    q = tmp_path / "t.ecsv"
    q.write_text("# %ECSV 0.9\n# ---\n# datatype:\n# - {name: energy, unit: MeV, datatype: float64}\n"
                 "# - {name: flux, unit: erg / (cm2 s), datatype: float64}\n"
                 "# - {name: flux_error, unit: erg / (cm2 s), datatype: float64}\n"
                 "energy flux flux_error\n1.0 2e-10 1e-11\n2.0 1e-10 1e-11\n")
    e = D.read(str(q))
    assert e["energy"].unit == u.MeV and len(e) == 2


def test_workload_definitions():
    from naima_amd import workloads as W
    assert set(W.WORKLOADS) == {"cfg1", "cfg2", "cfg3", "cfg4", "cfg5"}
    raw = W.build_data("cfg3", lambda E: 1e-12 * (E / 1e3) ** -2.0)
    assert len(raw["energy"]) == 64 and raw["flux_unit"] == "erg/(cm2 s)"
    assert np.all(np.diff(raw["energy"]) > 0) and raw["ul"].sum() == 1
    assert W.test_vectors("cfg3").shape == (8, 5)


def test_batched_nelder_mead_follows_the_sequential_algorithm():
    """the prefit's simplex search with its candidate points evaluated as a batch takes
    the decisions, and counts the evaluations, of the sequential algorithm"""
    from naima_amd.neldermead import minimize_batched
    from oracle.neldermead_np import minimize_sequential

    def f(x):
        return 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2 + 3 * (x[2] - 2) ** 2 + 1

    def fb(X):
        return np.array([f(x) for x in X])

    for opts in (dict(xtol=1e-1, ftol=1e-3, maxfev=500), dict(xtol=1e-6, ftol=1e-8, maxfev=60),
                 dict()):
        a = minimize_batched(fb, [-1.2, 1.0, 0.0], **opts)
        b = minimize_sequential(f, [-1.2, 1.0, 0.0], **opts)
        assert np.array_equal(a["x"], b["x"])
        assert (a["nfev"], a["nit"], a["status"]) == (b["nfev"], b["nit"], b["status"])
        assert a["fun"] == b["fun"]


def test_header_is_plain_c(tmp_path):
    """include/naima_hip.h is the drop-in boundary: it must compile as C99 (and C++) on
    its own -- plain pointers and sizes, no C++ or torch types"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    src.write_text('#include "naima_hip.h"\nint main(void) { return NH_MAX_GRIDS > 0 ? 0 : 1; }\n')
    inc = os.path.join(root, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, "-c", str(src), "-o",
                           str(tmp_path / "t.o")])
    subprocess.check_call(["g++", "-std=c++11", "-I", inc, "-x", "c++", "-c", str(src), "-o",
                           str(tmp_path / "t2.o")])


def test_build_data_table_and_energy_edges():
    """utils.build_data_table / generate_energy_edges (utils.py:358-482)"""
    from naima_amd.utils import build_data_table, generate_energy_edges, validate_data_table
    e = np.array([1.0, 2.0, 4.0, 8.0]) * u.TeV
    f = np.array([4.0, 2.0, 1.0, 0.5]) * u.Unit("1/(cm2 s TeV)")
    lo, hi = generate_energy_edges(e)
    mid = np.sqrt(e.value[1:] * e.value[:-1])
    assert_allclose(hi.value[:-1], mid - e.value[:-1])
    assert_allclose(lo.value[1:], e.value[1:] - mid)
    assert_allclose(lo.value[0], e.value[0] * (1 - e.value[0] / (e.value[0] + hi.value[0])))
    assert hi.value[-1] == lo.value[-1]
    glo, ghi = generate_energy_edges(e, groups=[0, 0, 1, 1])
    a, b = generate_energy_edges(e[:2])
    assert_allclose(glo.value[:2], a.value) and assert_allclose(ghi.value[:2], b.value)
    t = build_data_table(e, f, flux_error=0.1 * f, ul=[0, 0, 0, 1], cl=0.95)
    d = validate_data_table(t)
    assert_allclose(d["flux_error_lo"].value, 0.1 * f.value)
    assert list(np.asarray(d["ul"]).astype(int)) == [0, 0, 0, 1] and np.all(d["cl"] == 0.95)
    assert_allclose(d["energy_error_hi"].value, hi.value)
    t2 = build_data_table(e, f, flux_error_lo=0.1 * f, flux_error_hi=0.2 * f, energy_width=0.5 * e)
    d2 = validate_data_table(t2)
    assert_allclose(d2["flux_error_hi"].value, 0.2 * f.value)
    assert_allclose(d2["energy_error_lo"].value, 0.25 * e.value)
    with pytest.raises(TypeError):
        build_data_table(e, f)
    with pytest.raises(TypeError):
        build_data_table(e.value, f, flux_error=0.1 * f)


def test_ctypes_mirrors_match_the_c_header(tmp_path):
    """the ctypes Structures of naima_amd/darray.py have the sizes and field offsets a C
    compiler gives the structs of include/naima_hip.h (a drifted mirror would hand the
    library garbage without any error)"""
    import ctypes as C
    import subprocess

    from naima_amd import darray as D
    structs = ["nh_lazy", "nh_comp", "nh_grid", "nh_pack", "nh_moment", "nh_accept", "nh_prior",
               "nh_hs_table", "nh_hs_syn", "nh_hs_desc"]
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "naima_hip.h"', "int main(void) {"]
    for name in structs:
        cls = getattr(D, name)
        lines.append('printf("%s %%zu", sizeof(%s));' % (name, name))
        for f, _ in cls._fields_:
            lines.append('printf(" %%zu", offsetof(%s, %s));' % (name, f))
        lines.append('printf("\\n");')
    lines += ["return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o",
                           str(exe)])
    out = subprocess.check_output([str(exe)], text=True).strip().splitlines()
    for line in out:
        parts = line.split()
        cls = getattr(D, parts[0])
        want = [C.sizeof(cls)] + [getattr(cls, f).offset for f, _ in cls._fields_]
        assert [int(v) for v in parts[1:]] == want, parts[0]


def test_validate_data_table_against_the_reference(golden):
    """data ingest pinned by the reference: its validate_data_table (utils.py:38-213) over
    its own table fixtures (tests/golden/data/*.dat are the reference's tests/data files,
    data_tables.npz its outputs, gen_golden_data.py the script) -- every column of every
    case: single tables with each energy-error spelling, upper limits + flux_ul, the cl
    keyword, symmetric errors, an SED table; lists of tables in both orders with
    sed = None / True / False (concatenation, unit conversion, energy sort, groups)"""
    from naima_amd import datatable as DT
    from naima_amd import units as u
    z = golden("data_tables")
    data_dir = os.path.join(ROOT, "tests", "golden", "data")
    tables = {}
    ncols = 0
    for case in [str(c) for c in z["cases"]]:
        kind, rest = case.split(":", 1)
        if kind == "single":
            tables[rest] = DT.read(os.path.join(data_dir, rest + ".dat"))
            got = DT.validate_data_table(tables[rest])
        else:
            which, sed = rest.split(":sed=")
            names = {"xray": "CrabNebula_Fake_Xray", "tev": "CrabNebula_HESS_ipac",
                     "sed": "Fake_ipac_sed"}
            lst = [DT.read(os.path.join(data_dir, names[k] + ".dat")) for k in which.split("+")]
            got = DT.validate_data_table(lst, sed={"None": None, "True": True, "False": False}[sed])
        cols = sorted(set(k.split("__")[1] for k in z.files if k.startswith(case + "__")))
        assert set(cols) <= set(got.keys()), (case, set(cols) - set(got.keys()))
        for col in cols:
            want = z["%s__%s" % (case, col)]
            key = "%s__%s__unit" % (case, col)
            have = got[col]
            if key in z.files:
                have = have.to(u.Unit(str(z[key]))).value
            have = np.asarray(have)
            if want.dtype.kind in "bi":
                assert np.array_equal(have.astype(want.dtype), want), (case, col)
            else:
                assert_allclose(have.astype(float), want, rtol=1e-13, atol=0, err_msg="%s %s" % (case, col))
            ncols += 1
    assert ncols >= 150  # (17 cases x 9 columns)


def test_half_step_kernel_keeps_its_descriptor_out_of_scratch():
    """k_half_step takes its 3 KB descriptor by value; small edits (one more dynamic field read,
    an atomic in the wrong block) have twice made the compiler copy all of it to scratch memory
    -- 6x slower launches, and nothing but ScratchSize in the resource report shows it.  The
    resident kernel (every bench number comes from it) spills no vector register in any of its
    ten instances (two of them the table-only model's with register-resident table items, two with two
    walkers of a workgroup in flight -- round 6) and its bodies hold no scratch instruction at all: the 32-80 bytes its resource
    report shows are the frame the compiler reserves around its out-of-line math calls."""
    import re
    import subprocess
    src = os.path.join(ROOT, "naima_amd", "csrc")
    base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
            "-mllvm", "-amdgpu-kernarg-preload-count=16"]
    # (file: kernel, instances, bytes of scratch per lane allowed; the resident kernel has a second
    # pair of instances for an ensemble shared by several GPUs)
    want = {"nh_halfstep.hip": ("k_half_step", 3, 0), "nh_persist.hip": ("k_half_step_run", 10, 96)}
    for f, (sym, ninst, limit) in want.items():
        out = subprocess.run(base + ["-c", "-Rpass-analysis=kernel-resource-usage", os.path.join(src, f),
                                     "-o", os.devnull], capture_output=True, text=True).stderr
        blocks = re.split(r"remark: Function Name: ", out)[1:]
        sizes = {b.split()[0]: (int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1)),
                                int(re.search(r"VGPRs Spill: (\d+)", b).group(1)))
                 for b in blocks}
        mine = {k: v for k, v in sizes.items() if sym + "IL" in k}
        assert len(mine) == ninst, sizes
        for k, (scratch, vspill) in mine.items():
            assert scratch <= limit, "%s uses %d bytes of scratch per lane" % (k, scratch)
            # (the instances for an ensemble shared by several GPUs -- second template argument --
            # carry the peers' ring pointers: at most one register of theirs may spill, outside
            # every loop body, see below)
            shared = "k_half_step_runILb1ELb1E" in k or "k_half_step_runILb0ELb1E" in k
            assert vspill <= (1 if shared else 0), "%s spills %d vector registers" % (k, vspill)
    asm = subprocess.run(base + ["--cuda-device-only", "-S", os.path.join(src, "nh_persist.hip"), "-o", "-"],
                         capture_output=True, text=True).stdout
    bodies = re.findall(r"^(_Z15k_half_step_runILb[01]ELb[01]ELb[01]ELi\d+ELb[01]EEv6hs_hot6hs_run):[^\n]*\n(.*?)^\.Lfunc_end", asm,
                        flags=re.S | re.M)
    assert len(bodies) == 10
    for name, body in bodies:
        shared = "ILb1ELb1E" in name or "ILb0ELb1E" in name
        assert body.count("scratch_") <= (2 if shared else 0), "%s touches scratch memory" % name


def test_sorted_columns_of_an_emission_table():
    """_lib.sorted_columns (what the resident loop's table copies are built from): columns in the
    order of their first non-zero row, first row per tile of 64; with the oracle's trapz_loglog
    the skipped segments are seen to contribute exact zeros, whatever the weights"""
    from naima_amd._lib import sorted_columns
    from oracle import naima_np as O
    rng = np.random.default_rng(3)
    nG, nK = 90, 150
    x = np.geomspace(1.0, 1e4, nG)
    thr = rng.integers(0, 80, size=nK)          # first live row of every column
    thr[:5] = 0
    thr[7] = nG                                 # a column of zeros
    K = rng.uniform(0.5, 2.0, size=(nG, nK)) * (np.arange(nG)[:, None] >= thr[None, :])
    perm, row0 = sorted_columns(K)
    first = np.where((K != 0).any(axis=0), (K != 0).argmax(axis=0), nG)
    assert sorted(perm.tolist()) == list(range(nK))
    assert np.all(np.diff(first[perm]) >= 0)
    assert len(row0) == 3 and row0[0] == 0 and row0 == [int(first[perm][q:q + 64].min()) for q in (0, 64, 128)]
    assert perm[-1] == 7 and row0[2] > 0
    w = rng.uniform(0.1, 3.0, size=nG) * x ** -1.3
    for t, r0 in enumerate(row0):
        for p in range(64 * t, min(nK, 64 * t + 64), 17):
            y = w * K[:, perm[p]]
            full = O.trapz_loglog(y, x)
            assert O.trapz_loglog(y[r0:], x[r0:]) == full or np.isclose(O.trapz_loglog(y[r0:], x[r0:]), full, rtol=1e-15)
            if r0 > 1:
                assert O.trapz_loglog(y[:r0 + 1], x[:r0 + 1]) == 0.0
