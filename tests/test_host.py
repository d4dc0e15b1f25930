"""Host logic (CPU): dispersion stream, motor perturbation, wind synthesis and the per-sample
flattening must reproduce the reference's inputs BIT FOR BIT (golden fixtures captured from
inside the reference's simulate_flight call, oracle/gen_golden.py); the C-ABI library must load
and export every symbol include/erpl_mc.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from erpl_monte_carlo_sim_amd import _abi, flatten, models

import helpers as H

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def params():
    return H.load_json("params.json")


@pytest.mark.parametrize("stream", ["seed_i", "seed_42"])
def test_parameter_stream_bit_exact(params, stream):
    got = flatten.generate_parameter_samples(H.UNCERTAINTY, 64, stream=stream)
    for g, e in zip(got, params[stream]):
        for k, v in e.items():
            assert np.array_equal(np.asarray(g[k], dtype=np.float64), np.asarray(v, dtype=np.float64)), (stream, k)


def test_motor_perturbation_bit_exact(params):
    for i, e in enumerate(params["liquid_perturbed"]):
        m = models.LiquidMotor().perturb_for_monte_carlo(np.random.RandomState(i))
        for k, v in e.items():
            assert getattr(m, k) == v, (i, k)
    for i, e in enumerate(params["solid_perturbed"]):
        m = models.SolidMotor().perturb_for_monte_carlo(np.random.RandomState(i))
        for k, v in e.items():
            assert np.array_equal(np.asarray(getattr(m, k), dtype=np.float64), np.asarray(v)), (i, k)
        # the kernel applies the multiplier to the shared unscaled curve: same bits
        assert np.array_equal(models.SolidMotor().thrust_curve_thrust * m._thrust_multiplier,
                              m.thrust_curve_thrust)


def test_wind_profiles_bit_exact(params):
    wm = models.WindModel()
    for i, e in enumerate(params["csv_perturbed"]):
        got = wm.perturb_wind_profile(H.CSV_ALT, H.CSV_WIND, np.random.RandomState(i))
        assert np.array_equal(got, np.array(e)), i
    grid = np.linspace(0, 25000, 100)
    for i, e in enumerate(params["synthetic_profiles"]):
        s = params["seed_i"][i]
        got = wm.generate_stochastic_profile(grid, s["wind_speed"], s["wind_direction"],
                                             random_state=np.random.RandomState(i))
        assert np.array_equal(got, np.array(e)), i


def test_csv_loader(tmp_path):
    p = tmp_path / "w.csv"
    p.write_text("altitude,u,v,w\n0,2,0,0\n5000,5,1,0\n10000,8,2,0\n15000,10,2,0\n20000,12,3,0\n25000,15,3,0\n")
    alt, w = models.WindModel().load_wind_profile_from_csv(str(p))
    assert np.array_equal(alt, H.CSV_ALT) and np.array_equal(w, H.CSV_WIND)
    p.write_text("altitude,u,v\n0,2,0\n100,3,1\n")
    alt, w = models.WindModel().load_wind_profile_from_csv(str(p))
    assert w.shape == (2, 3) and np.all(w[:, 2] == 0)


def test_legacy_random_streams_match_numpy():
    """erpl_mc_legacy_random_streams == np.random.RandomState(seed) draw for draw (normal / uniform mixed,
    gaussian cache included), for small, large and 32-bit-limit seeds."""
    ops = "ggugguugggu" * 30
    seeds = np.array([0, 1, 2, 41, 42, 999, 123456789, 2**31 - 1, 2**31, 2**32 - 1] + list(range(100, 190)), dtype=np.uint32)
    got = flatten.legacy_streams(seeds, ops, threads=3)
    for s, row in zip(seeds, got):
        rs = np.random.RandomState(int(s))
        ref = np.array([rs.normal() if o == "g" else rs.random_sample() for o in ops])
        assert np.array_equal(ref, row), int(s)
    assert np.array_equal(flatten.legacy_streams(seeds, ops, by_output=True), got.T)
    assert flatten.legacy_streams(np.zeros(0, dtype=np.uint32), "gg").shape == (0, 2)


@pytest.mark.parametrize("kind,base,planar", [("liquid", "csv", False), ("solid", "csv", True),
                                              ("liquid", "none", True), ("solid", "none", False)])
def test_vectorised_batch_equals_per_sample_loop(kind, base, planar):
    """The all-samples-at-once host preparation is bit-identical to the per-sample restatement of
    monte_carlo.py:228-288 (which the tests below pin to the reference's own captured inputs)."""
    n = 150
    assert flatten.generate_parameter_samples(H.UNCERTAINTY, 40)[7]["wind_speed"] == \
        flatten.generate_parameter_samples_loop(H.UNCERTAINTY, 40)[7]["wind_speed"]
    pl_fast = flatten.generate_parameter_samples(H.UNCERTAINTY, n)
    pl_loop = flatten.generate_parameter_samples_loop(H.UNCERTAINTY, n)
    for a, b in zip(pl_fast, pl_loop):
        assert a.keys() == b.keys()
        for k in a:
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    kw = dict(base_altitude_profile=H.CSV_ALT, base_wind_profile=H.CSV_WIND) if base == "csv" else {}
    args = (models.Rocket(), H.make_motor(kind), models.WindModel(), H.EXAMPLE_IC)
    fast = flatten.dispersed_batch(*args, pl_fast, planar=planar, **kw)
    arr = flatten.dispersed_batch(*args, flatten.generate_parameter_arrays(H.UNCERTAINTY, n), planar=planar, **kw)
    loop = flatten.dispersed_batch_loop(*args, pl_loop, planar=planar, **kw)
    for name in ("ic", "rocket", "motor", "alt_grid", "wind"):
        assert np.array_equal(getattr(fast, name), getattr(loop, name)), name
        assert np.array_equal(getattr(arr, name), getattr(loop, name)), name
    sub = flatten.dispersed_batch(*args, [pl_fast[i] for i in (5, 3, 77)], planar=planar, **kw)   # any subset / order
    assert np.array_equal(sub.wind, loop.wind[:, :, [5, 3, 77]]) and np.array_equal(sub.motor, loop.motor[:, [5, 3, 77]])
    assert flatten.dispersed_batch(*args, [], **kw).n == 0
    # the C AR(1) synthesis against the same recursion written as NumPy array expressions
    wm, seeds = models.WindModel(), np.arange(n, dtype=np.uint32)
    P = flatten.generate_parameter_arrays(H.UNCERTAINTY, n)
    if base == "csv":
        g = flatten.legacy_streams(seeds, "g" * (3 * len(H.CSV_ALT)), by_output=True)
        assert np.array_equal(flatten.legacy_wind_profiles(wm, H.CSV_ALT, seeds, base=H.CSV_WIND),
                              flatten._ar1_profiles(wm, H.CSV_ALT, g, base=np.asarray(H.CSV_WIND, dtype=np.float64)))
    else:
        alt = np.linspace(0, 25000, 100)
        g = flatten.legacy_streams(seeds, "g" * 300, by_output=True)
        cd, sd = np.cos(P["wind_direction"]), np.sin(P["wind_direction"])
        mean = [P["wind_speed"] * ((np.float64(a) / 10.0) ** wm.power_law_exponent) for a in alt]
        assert np.array_equal(flatten.legacy_wind_profiles(wm, alt, seeds, speed=P["wind_speed"], cdir=cd, sdir=sd, threads=2),
                              flatten._ar1_profiles(wm, alt, g, mean_u=[m * cd for m in mean], mean_v=[m * sd for m in mean]))


def _mc_batch(kind, base, stream, ids):
    pl = flatten.generate_parameter_samples(H.UNCERTAINTY, max(ids) + 1, stream=stream)
    pl = [pl[i] for i in ids]
    kw = dict(base_altitude_profile=H.CSV_ALT, base_wind_profile=H.CSV_WIND) if base == "csv" else {}
    return flatten.dispersed_batch(models.Rocket(), H.make_motor(kind), models.WindModel(), H.EXAMPLE_IC, pl, **kw)


def test_dispersed_batch_matches_reference_inputs():
    """Flattened per-sample rows == what the reference handed to simulate_flight."""
    idx, arr = H.load_flights("flights_mc")
    for g, entries in H.group_flights(idx).items():
        kind, base, stream = g
        ids = [e["key"][3] for e in entries]
        got = _mc_batch(kind, base, stream, ids)
        exp = H.batch_from_golden(entries, arr)
        assert np.array_equal(got.ic, exp.ic), g
        assert np.array_equal(got.rocket, exp.rocket), g
        assert np.array_equal(got.motor, exp.motor), g
        assert np.array_equal(got.alt_grid, exp.alt_grid), g
        assert np.array_equal(got.wind, exp.wind), g


def test_planar_batch_matches_reference_inputs():
    idx, arr = H.load_flights("flights_planar")
    for g, entries in H.group_flights(idx).items():
        ids = [e["key"][1] for e in entries]
        pl = flatten.generate_parameter_samples(H.UNCERTAINTY, max(ids) + 1)
        got = flatten.dispersed_batch(models.Rocket(), H.make_motor(g[0]), models.WindModel(), H.EXAMPLE_IC,
                                      [pl[i] for i in ids], base_altitude_profile=H.CSV_ALT,
                                      base_wind_profile=H.CSV_WIND, planar=True)
        exp = H.batch_from_golden(entries, arr)
        for name in ("ic", "rocket", "motor", "wind"):
            assert np.array_equal(getattr(got, name), getattr(exp, name)), (g, name)
        assert np.all(got.wind[:, 1, :] == 0)


def test_single_flight_batch():
    idx, arr = H.load_flights("flights_named")
    e = [x for x in idx if x["key"] == "liquid_csv_nominal"][0]
    b = flatten.single_flight_batch(models.Rocket(), models.LiquidMotor(), H.EXAMPLE_IC, H.CSV_WIND, H.CSV_ALT)
    exp = H.batch_from_golden([e], arr)
    for name in ("ic", "rocket", "motor", "wind", "alt_grid"):
        assert np.array_equal(getattr(b, name), getattr(exp, name)), name
    b0 = flatten.single_flight_batch(models.Rocket(), models.SolidMotor(), {}, None, None)
    assert b0.k_wind == 0 and np.array_equal(b0.ic[6:10, 0], [1.0, 0, 0, 0])


def test_config_rejects_bad_models():
    r = models.Rocket()
    r.Cd_data = {"mach": [0.0, 1.0], "cd0": [0.4], "cda": [1.0, 1.0]}
    with pytest.raises(flatten.UnsupportedModel):
        flatten.config_from_objects(r, models.LiquidMotor(), models.StandardAtmosphere())
    r = models.Rocket()
    r.Cd_data["mach"] = list(np.linspace(0, 3, 20))
    with pytest.raises(flatten.UnsupportedModel):
        flatten.config_from_objects(r, models.LiquidMotor(), models.StandardAtmosphere())
    with pytest.raises(flatten.UnsupportedModel):
        flatten.config_from_objects(models.Rocket(), object(), models.StandardAtmosphere())


# ---------------------------------------------------------------------------------- C ABI
def test_header_and_ctypes_agree():
    """Every erpl_mc_* prototype in include/erpl_mc.h is in _abi.EXPORTS and vice versa, and the
    numeric limits agree."""
    hdr = open(os.path.join(REPO, "include", "erpl_mc.h")).read()
    declared = set(re.findall(r"\b(erpl_mc_[a-z_]+)\s*\(", hdr))
    assert declared == set(_abi.EXPORTS)
    for name, val in (("ERPL_SUMMARY_DIM", _abi.SUMMARY_DIM), ("ERPL_TRAJ_DIM", _abi.TRAJ_DIM),
                      ("ERPL_MAX_MACH_KNOTS", _abi.MAX_MACH_KNOTS), ("ERPL_MAX_CURVE_KNOTS", _abi.MAX_CURVE_KNOTS),
                      ("ERPL_MAX_WIND_KNOTS", _abi.MAX_WIND_KNOTS), ("ERPL_MC_ABI_VERSION", _abi.ABI_VERSION)):
        assert re.search(rf"#define {name} {val}\b", hdr), name


def test_struct_layout_matches_c_compiler(tmp_path):
    """sizeof/offsetof of the ctypes mirrors == what gcc sees in include/erpl_mc.h."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "erpl_mc.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(erpl_config), sizeof(erpl_batch),'
                   'sizeof(erpl_out), offsetof(erpl_config, motor_kind), offsetof(erpl_config, dt_initial),'
                   'offsetof(erpl_batch, ic), offsetof(erpl_out, traj));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    got = [C.sizeof(_abi.ErplConfig), C.sizeof(_abi.ErplBatch), C.sizeof(_abi.ErplOut),
           _abi.ErplConfig.motor_kind.offset, _abi.ErplConfig.dt_initial.offset,
           _abi.ErplBatch.ic.offset, _abi.ErplOut.traj.offset]
    assert [int(x) for x in out] == got


def test_library_loads_and_exports_all_symbols():
    if not os.path.exists(_abi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _abi.load_library()
    for name in _abi.EXPORTS:
        assert hasattr(lib, name), name
    assert lib.erpl_mc_abi_version() == _abi.ABI_VERSION


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product path must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _abi.load_library()
    ctx = C.c_void_p()
    rc = lib.erpl_mc_create(0, C.byref(ctx))
    assert rc == -3 and b"no HIP device" in lib.erpl_mc_last_error()
    from erpl_monte_carlo_sim_amd.engine import TrajectoryEngine
    with pytest.raises(_abi.ErplError):
        TrajectoryEngine()


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_abi.ErplError):
        _abi.load_library(str(tmp_path / "nope.so"))


def test_product_never_touches_oracle():
    """The product package must not import, link or load anything under oracle/."""
    root = os.path.join(REPO, "erpl_monte_carlo_sim_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".inc", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"(import|from)\s+oracle|liberpl_oracle|oracle/", txt), f


def test_device_batch_validation_rejects_bad_inputs():
    """Boundary validation happens on the host before anything is uploaded (no GPU needed to fail)."""
    from erpl_monte_carlo_sim_amd.engine import DeviceBatch
    good = flatten.single_flight_batch(models.Rocket(), models.LiquidMotor(), H.EXAMPLE_IC, H.CSV_WIND, H.CSV_ALT)
    cases = []
    b = good.take([0]); b.rocket[0, 0] = -1.0; cases.append(b)
    b = good.take([0]); b.motor[2, 0] = 0.0; cases.append(b)
    b = good.take([0]); b.motor[3, 0] = float("inf"); cases.append(b)
    b = good.take([0]); b.ic[2, 0] = float("nan"); cases.append(b)
    b = good.take([0]); b.wind[1, 0, 0] = float("nan"); cases.append(b)
    b = good.take([0]); b.alt_grid[2] = b.alt_grid[1]; cases.append(b)
    for bad in cases:
        with pytest.raises(_abi.ErplError):
            DeviceBatch.from_host(bad, "cpu")


def test_overridden_model_methods_are_refused():
    """a user object that redefines a method the kernels hard-code cannot be flattened."""
    from erpl_monte_carlo_sim_amd.flatten import UnsupportedModel, reject_overrides

    class SubRocket(models.Rocket):
        def get_aerodynamic_coefficients(self, *a, **k):
            return {}

    class Deeper(SubRocket):
        pass

    class Plain(models.Rocket):          # adds attributes only: fine
        extra = 1

    class Reference:                     # stands for the reference's own class of the same name
        pass
    Reference.__name__ = "StandardAtmosphere"
    Reference.get_properties = lambda self, h: None

    class Duck:                          # unrelated class with its own physics
        def get_gravity(self, h):
            return 9.81

    for obj, role, ok in ((SubRocket(), "rocket", False), (Deeper(), "rocket", False), (Plain(), "rocket", True),
                          (models.Rocket(), "rocket", True), (Reference(), "atmosphere", True), (Duck(), "atmosphere", False),
                          (models.WindModel(), "wind_model", True)):
        if ok:
            reject_overrides(obj, role)
        else:
            with pytest.raises(UnsupportedModel):
                reject_overrides(obj, role)
    patched = models.LiquidMotor()
    patched.get_thrust = lambda t, p: 0.0
    with pytest.raises(UnsupportedModel, match="instance"):
        flatten.config_from_objects(models.Rocket(), patched, models.StandardAtmosphere())
    with pytest.raises(UnsupportedModel, match="SubRocket"):
        flatten.config_from_objects(SubRocket(), models.LiquidMotor(), models.StandardAtmosphere())


def test_device_batch_refuses_tables_of_the_wrong_shape():
    """The kernels index the wind table by (row, sample) and the altitude grid by row; a tensor of another shape
    would be an out-of-bounds read on the GPU, so the host object refuses it before anything is launched."""
    import torch
    from erpl_monte_carlo_sim_amd import _abi
    from erpl_monte_carlo_sim_amd.engine import DeviceBatch
    n, k = 8, 5
    ic, rk, mt = torch.zeros(13, n, dtype=torch.float64), torch.zeros(2, n, dtype=torch.float64), torch.zeros(4, n, dtype=torch.float64)
    alt = torch.linspace(0, 1000, k, dtype=torch.float64)
    ok = DeviceBatch(ic, rk, mt, alt, torch.zeros(k, 3, n, dtype=torch.float64), _abi.PREC_F64)
    assert ok.k_wind == k and ok.n == n
    assert DeviceBatch(ic, rk, mt, None, None, _abi.PREC_F32).k_wind == 0
    for bad_alt, bad_wind, prec in (
            (alt, torch.zeros(k * 3, n, dtype=torch.float64), _abi.PREC_F64),            # flattened rows
            (alt, torch.zeros(k, 3, n + 1, dtype=torch.float64), _abi.PREC_F64),         # another sample count
            (alt[:-1], torch.zeros(k, 3, n, dtype=torch.float64), _abi.PREC_F64),        # fewer altitudes than knots
            (alt, torch.zeros(k, 3, n, dtype=torch.float64), _abi.PREC_F32),             # fp64 table for the fp32 build
            (alt.float(), torch.zeros(k, 3, n, dtype=torch.float64), _abi.PREC_F64),     # altitudes must be fp64
            (None, torch.zeros(k, 3, n, dtype=torch.float64), _abi.PREC_F64)):
        with pytest.raises(ValueError):
            DeviceBatch(ic, rk, mt, bad_alt, bad_wind, prec)


def test_parameter_arrays_in_parallel_blocks_equal_one_block(monkeypatch):
    """Round 4: at 10^5+ samples the per-sample dispersion streams are drawn and scaled in blocks by a few Python
    workers (flatten.generate_parameter_arrays); the values must be those of the single-block path, bit for bit."""
    from erpl_monte_carlo_sim_amd import sampling
    n = 5 * 4096 + 17
    monkeypatch.setattr(flatten, "_PARAM_CHUNK", 4096)          # 6 blocks, the last one ragged
    monkeypatch.setattr(flatten, "host_workers", lambda: 3)
    a = flatten.generate_parameter_arrays(sampling.DEFAULT_UNCERTAINTY, n)
    monkeypatch.setattr(flatten, "_PARAM_CHUNK", 10 ** 9)
    b = flatten.generate_parameter_arrays(sampling.DEFAULT_UNCERTAINTY, n)
    assert a.keys() == b.keys()
    for k in a:
        assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k
