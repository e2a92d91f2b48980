"""Host-side mirror of the reference's user API for the hot path, over the C ABI (include/petlion_hip.h):

    p   = petlion(LCO; N_p=10, ..., temperature=false)         reference src/external.jl:2-18, src/params.jl:119-174
    sol = simulate(p, tf; I=-1, SOC=1, V_min=..., reltol=...)    reference src/model_evaluation.jl:11-85
    simulate!(sol, p, tf; V=:hold, I_min=1/20)                  reference src/model_evaluation.jl:87-97   (here: simulate_b)
    ens = simulate_ensemble(p, Theta, protocol; SOC=...)         new: the ensemble axis the MI355X path exists for

Julia spellings map as  simulate! -> simulate_b,  :hold -> "hold",  :rest -> "rest",  p.θ[:D_sp] -> p.θ["D_sp"].
Everything numerical happens in the HIP library; this file only marshals arguments (no CPU compute path).
"""
from __future__ import annotations

import ctypes as C
import math
import os

import time
import numpy as np

from . import _capi as cap
from . import closures, grids
from .params import EXIT_REASONS, Bounds, Opts, bounds_LCO, bounds_LGM50, bounds_NMC, calc_I1C, theta_LCO, theta_LGM50, theta_NMC

LCO = "LCO"
NMC = "NMC"
NMC_LGM50 = "LGM50"


class _N:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class Model:
    """The reference's `model` struct, reduced to the data contracts of the hot path (src/structures.jl:336-345)."""

    def __init__(self, cathode, N, temperature, aging, lib_path=None, precision="f64", device=-1, solid_diffusion="Fickian", thermodynamic_factor="linear", rxn="BV", waves_per_cell=1, grid_lib=None):
        if cathode not in (LCO, NMC, NMC_LGM50):
            raise NotImplementedError("chemistry %r is outside the hot-path scope (LCO, NMC and NMC_LGM50 are built)" % (cathode,))
        self.cathode = cathode
        self.N = N
        self.temperature = bool(temperature)
        if aging not in (False, None, True, "SEI"):
            raise NotImplementedError("aging=%r: the reference knows false and :SEI (src/params.jl:119-174)" % (aging,))
        aging = "SEI" if aging in (True, "SEI") else False
        self.aging = aging
        self.θ = {LCO: theta_LCO, NMC: theta_NMC, NMC_LGM50: theta_LGM50}[cathode]()
        self.θ["I1C"] = calc_I1C(self.θ)
        self.bounds = {LCO: bounds_LCO, NMC: bounds_NMC, NMC_LGM50: bounds_LGM50}[cathode]()
        self.opts = Opts()
        self._lib = cap.load(lib_path)
        self._lib_path = lib_path
        self._builtin_lib = lib_path is None          # the product library (not the tests' emulator build / an experiment build)
        if precision not in ("f64", "mixed", "f64_reforder"):
            raise ValueError("precision: 'f64', 'mixed' (fp32 storage of the Newton-matrix factors, everything else fp64) or 'f64_reforder' (fp64 with the finite-volume rows in the "
                             "reference's operation order: PLH_PREC_F64_REFORDER)")
        self.precision = precision
        desc = cap.ModelDesc({LCO: cap.CHEM_LCO, NMC: cap.CHEM_NMC, NMC_LGM50: cap.CHEM_LGM50}[cathode], N.p, N.s, N.n, N.a, N.z, N.r_p, N.r_n, int(self.temperature), int(bool(aging)), 8,
                             {"f64": cap.PREC_F64, "mixed": cap.PREC_MIXED, "f64_reforder": cap.PREC_F64_REFORDER}[precision], int(device),
                             {"Fickian": 0, "quadratic": 1, "polynomial": 2}[solid_diffusion], {"linear": 0, "nonlinear": 1}[thermodynamic_factor], {"BV": 0, "MHC": 1}[rxn], int(waves_per_cell))
        self.waves_per_cell = int(waves_per_cell)
        self.solid_diffusion, self.thermodynamic_factor, self.rxn = solid_diffusion, thermodynamic_factor, rxn
        # another discretisation than the built-in 10/10/10/10: its kernels are a library of their own, compiled on first use and cached (grids.py), registered before the
        # handle is created (`grid_lib`: a library built elsewhere -- the tests' emulator build)
        # (N_r_p != N_r_n: a 7-entry grid, the anode's N_r last -- the particle phases then run on the larger of the two as lane stride, dfn_cell.h)
        g = grids.grid_tuple(N.p, N.s, N.n, N.r_p if solid_diffusion == "Fickian" else 10, N.a if self.temperature else 10, N.z if self.temperature else 10,
                             N.r_n if solid_diffusion == "Fickian" else None)
        grids.check(g, thermal=self.temperature, sei=bool(aging))
        if grid_lib is None and g != grids.DEFAULT:
            if lib_path is not None:
                raise ValueError("a non-default discretisation with an explicit library path needs grid_lib= as well")
            vid = grids.variant_id({LCO: "LCO_LIC6", NMC: "NMC_LIC6", NMC_LGM50: "LGM50"}[cathode], bool(aging), self.temperature, {"f64": 0, "mixed": 1, "f64_reforder": 2}[precision],
                                   solid_diffusion.upper(), thermodynamic_factor.upper(), rxn, waves_per_cell == 2)
            if vid is None:
                raise NotImplementedError("this combination of model options is not instantiated on the device")
            # (a variant that failed the kernel self-test on this machine with the built-in flags -- under THIS flag table and these sources: the marker carries the build
            #  identity -- goes to its fall-back build directly, and grids.library keeps it out of every union library of the built-in flags: ADVICE r05)
            grid_lib = grids.library(g, [vid], machine_licm=True) if grids.needs_fallback(g, vid) else grids.library(g, [vid])
            self._grid_lib_built, self._grid_key = grid_lib, (g, vid)
        if grid_lib:                                     # (False: register nothing -- the tests' way to reach the C ABI's own refusal)
            cap.check(self._lib, self._lib.plh_register_grid_library(os.fsencode(grid_lib)), "plh_register_grid_library")
        h = C.c_void_p()
        cap.check(self._lib, self._lib.plh_model_create(C.byref(desc), C.byref(h)), "plh_model_create")
        self._h = h
        self.N.tot = self._lib.plh_n_states(h)
        self.N.diff = self._lib.plh_n_diff(h)
        self.N.alg = self.N.tot - self.N.diff
        self.lds_bytes = self._lib.plh_lds_bytes(h)            # LDS per cell of this variant
        self.save_start_dict = {}                              # p.cache.save_start_dict (src/structures.jl:305): opts.save_start
        self.θ_keys = [self._lib.plh_theta_key(h, i).decode("utf-8") for i in range(self._lib.plh_n_theta(h))]
        self.ind = {}                       # p.ind: state name -> slice into Y (reference state_indices, src/external.jl:275-365)
        for i in range(self._lib.plh_n_sections(h)):
            nm, a, ln = C.c_char_p(), C.c_int(), C.c_int()
            cap.check(self._lib, self._lib.plh_section(h, i, C.byref(nm), C.byref(a), C.byref(ln)), "plh_section")
            self.ind[nm.value.decode("utf-8")] = slice(a.value, a.value + ln.value)
        self.variant = "%s_%s%s%s%s%s" % (cathode.lower(), "thermal" if self.temperature else "iso", "_sei" if aging else "",          # matching oracle variant (tests)
                                          {"Fickian": "", "quadratic": "_quad", "polynomial": "_poly"}[solid_diffusion], "_nu" if thermodynamic_factor == "nonlinear" else "",
                                          "_mhc" if rxn == "MHC" else "")
        if g != grids.DEFAULT:
            self.variant += "_g%d_%d_%d_%d" % g[:4] + ("_%d_%d" % g[4:6] if self.temperature else "") + ("_rn%d" % g[6] if len(g) == 7 else "")

    theta = property(lambda self: self.θ)

    def compile_closures(self, protocol, n_cells=None, force=False, _emu_include=None, verify=True):
        """compile the input closures of `protocol` into device code and attach them to this model (closure_lib.py; hipcc, once per closure set, cached): later
        simulate_ensemble / simulate calls with the same closures run them compiled instead of interpreted; every other call is unaffected.  Returns the library path
        (None: the protocol has no closure input)."""
        from . import closure_lib
        runs, _ = make_protocol(self, protocol, n_cells)
        #  a closure library is compiled on the user's machine with the aggressive flag set of the built-in kernels, which this toolchain is known to miscompile now
        # and then (buildflags.py): it is VERIFIED before use -- compiled against interpreted on a short run of the same protocol, bit for bit (the contract of a compiled
        # closure) --, rebuilt ONCE with the conservative set when that fails, and dropped (interpreter) when that fails too.  A model whose grid library already needed the
        # conservative set starts there.
        licm = "_licm" in os.path.basename(getattr(self, "_grid_lib_built", None) or "")
        lib = closure_lib.library(self, runs, force=force, emu_include=_emu_include, machine_licm=licm)
        if not lib:
            return None
        for attempt in range(2):
            cap.check(self._lib, self._lib.plh_model_attach_closure_library(self._h, os.fsencode(lib)), "plh_model_attach_closure_library")
            self._closure_digest = closure_lib.digest(closure_lib.expr_runs(runs))
            why = self._verify_closure_library(protocol, lib) if verify else None
            if why is None:
                return lib
            if attempt == 0 and not licm:
                lib = closure_lib.library(self, runs, force=force, emu_include=_emu_include, machine_licm=True)
                licm = True
                continue
            break
        self._lib.plh_model_attach_closure_library(self._h, None)          # detach: the interpreter serves these closures
        self._closure_digest = None
        import warnings
        warnings.warn("compiled closures of this protocol do not reproduce the interpreter (%s): the library %s is not used" % (why, os.path.basename(lib)))
        return None

    def _verify_closure_library(self, protocol, lib):
        """None if a short run of `protocol` (every run cut to 30 s, 2 cells at SOC 0.5) through the attached library equals the interpreter's bit for bit, else what differs.  Verified
        libraries are remembered next to the file (<lib>.verified holds the build identity)."""
        mark = lib + ".verified"
        ident = build_info(self)
        if os.path.exists(mark) and open(mark).read() == ident:
            return None
        short = [dict(r, tf=min(float(r.get("tf", 30.0)), 30.0)) for r in protocol]
        Th = np.tile(self.theta_vector(), (2, 1))
        try:
            a = simulate_ensemble(self, Th, short, SOC=0.5)
            if not self._lib.plh_last_integrate_compiled(self._h):
                return "the attached library was not selected for its own protocol"
            self._lib.plh_model_attach_closure_library(self._h, None)
            try:
                b = simulate_ensemble(self, Th, short, SOC=0.5)
            finally:
                cap.check(self._lib, self._lib.plh_model_attach_closure_library(self._h, os.fsencode(lib)), "plh_model_attach_closure_library")
        except Exception as e:              # (a protocol that cannot run per-cell arrays cut to two cells, ...: nothing verified, nothing refused)
            return None if isinstance(e, (ValueError, TypeError)) else repr(e)
        if not np.array_equal(a.run_info["flag"], b.run_info["flag"]):
            return "exit flags %r, interpreted %r" % (a.run_info["flag"].tolist(), b.run_info["flag"].tolist())
        if not (np.array_equal(np.asarray(a.Y), np.asarray(b.Y)) and np.array_equal(a.run_info["t_end"], b.run_info["t_end"])):
            return "end states differ by %.1e" % float(np.abs(np.asarray(a.Y) - np.asarray(b.Y)).max())
        try:
            open(mark, "w").write(ident)
        except OSError:
            pass
        return None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.plh_model_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def theta_vector(self, overrides=None):
        """dense θ_tot in θ_keys order (update_θ!, src/generate_functions.jl:364-372)."""
        th = dict(self.θ)
        if overrides:
            th.update(overrides)
        return np.array([th[k] for k in self.θ_keys], dtype=np.float64)

    def jac_pattern(self, mode=cap.MODE_I):
        nnz = C.c_int(0)
        cap.check(self._lib, self._lib.plh_jac_pattern(self._h, mode, C.byref(nnz), None, None), "plh_jac_pattern")
        cp = np.zeros(self.N.tot + 1, dtype=np.int32)
        ri = np.zeros(nnz.value, dtype=np.int32)
        cap.check(self._lib, self._lib.plh_jac_pattern(self._h, mode, C.byref(nnz), cp.ctypes.data, ri.ctypes.data), "plh_jac_pattern")
        return cp, ri


def petlion(cathode=LCO, *, N_p=10, N_s=10, N_n=10, N_a=10, N_z=10, N_r_p=10, N_r_n=10, temperature=None,
            solid_diffusion="Fickian", Fickian_method="finite_difference", aging=False, jacobian="symbolic", SOC=1.0,
            thermodynamic_factor="linear", rxn_p="BV", rxn_n="BV", precision="f64", device=-1, waves_per_cell=1, _lib_path=None, _grid_lib=None):
    """petlion(cathode; kwargs...) -- reference src/params.jl:119-174.  `jacobian` is accepted and ignored (the device
    Jacobian is hand-derived); unsupported structural options raise."""
    if temperature is None:
        temperature = cathode == NMC_LGM50       # the reference's defaults: system_LGM50_NMC_LiC6 has temperature = true (src/params.jl:695), LCO / NMC false (139, 389)
    if solid_diffusion not in ("Fickian", "quadratic", "polynomial") or (solid_diffusion == "Fickian" and Fickian_method != "finite_difference"):
        raise NotImplementedError("solid diffusion: Fickian (finite_difference), quadratic and polynomial are built; the BETA spectral method is not (SURVEY.md 8f)")
    if thermodynamic_factor not in ("linear", "nonlinear") or rxn_p not in ("BV", "MHC") or rxn_n != rxn_p:
        raise NotImplementedError("thermodynamic_factor: linear / nonlinear; rxn_p = rxn_n in (BV, MHC)")
    mk = lambda: Model(cathode, _N(p=N_p, s=N_s, n=N_n, a=N_a, z=N_z, r_p=N_r_p, r_n=N_r_n), temperature, aging, _lib_path, precision, device, solid_diffusion, thermodynamic_factor, rxn_p, waves_per_cell, _grid_lib)
    p = mk()
    p.opts.SOC = SOC
    try:
        _selftest_new_grid_library(p)
    except RuntimeError as err:
        # a grid library compiled here with the built-in flags (buildflags.py) failed the kernel self-test: ONE rebuild with MachineLICM on -- the flag set every grid library of
        # r04 passed under -- is registered after it (the latest registration wins in plh_model_create) and checked the same way; the marker makes later processes go there directly
        import warnings
        lib = getattr(p, "_grid_lib_built", None)
        if not lib or lib.endswith("_licm.so"):
            raise
        warnings.warn("petlion.jl_amd: %s -- rebuilding %s with MachineLICM on (fall-back flags)" % (err, os.path.basename(lib)), RuntimeWarning)
        grids.mark_fallback(*p._grid_key, err)
        p = mk()
        p.opts.SOC = SOC
        _selftest_new_grid_library(p)
    _selftest_unvalidated_build(p)
    return p


def selftest(p, n_cells=2, tf=100.0):
    """Power-on check of the kernels of ONE model variant: every k_integrate instantiation (plain / stop times / table / closure / general control row / refinement,
    csrc/dfn_integrate.h GenFlag) runs the same 1C discharge and must reproduce the plain kernel -- flags and end times equal, SOC (exact for a constant current) to 1e-11,
    the first saved SOC equal to SOC0, voltage to 1e-9 where the input is the same number.  Raises RuntimeError naming the instantiation.  It exists because a kernel of this size
    sits at the register allocator's limits: one build was seen to start a run with a garbage SOC accumulator in one instantiation only; the built-in variants
    are checked by the GPU test suite, a grid library compiled on the user's machine at first use is checked here."""
    Th = np.tile(p.theta_vector(), (n_cells, 1))
    ps = p.ind["Φ_s"]
    base = simulate_ensemble(p, Th, [{"I": -1.0, "tf": tf}], SOC=1.0)
    o_stop, o_ref = Opts(), Opts()
    o_stop.tstops = [1e7]; o_ref.refine = 1
    cases = [("stop times", [{"I": -1.0, "tf": tf}], o_stop, 1e-12), ("table input", [{"I": ([0.0, 1e7], [-1.0, -1.0]), "tf": tf}], None, 1e-9),
             ("closure input", [{"I": lambda t: -1.0 + 0.0 * t, "tf": tf}], None, 1e-9),
             ("general control row", [{"I": lambda t, Y, q: -1.0 + 1e-12 * (Y[ps.start] - Y[ps.stop - 1]), "tf": tf}], None, 2e-3), ("refinement", [{"I": -1.0, "tf": tf}], o_ref, 2e-3)]
    bad = None
    if not ((base.run_info["flag"][:, 0] == 0).all() and np.abs(base.run_info["SOC"][:, 0] - (1.0 - tf / 3600.0)).max() < 1e-11 and np.abs(base.SOC[:, 0] - 1.0).max() == 0.0):
        bad = "plain"
    if not bad:
        # every comparison below is against the PLAIN kernel of the same build: a wrong plain kernel would pass them all.  Known answer first.
        why = known_answer_check(p)
        if why:
            raise RuntimeError("kernel self-test failed: the plain kernel of %s does not reproduce the known answer of the validated binary (%s) -- a miscompiled or modified build "
                               "(DESIGN.md 5a)" % (p.variant, why))
    for name, proto, o, vtol in cases:
        if bad:
            break
        e = simulate_ensemble(p, Th, proto, SOC=1.0, opts=o)
        if not (np.array_equal(e.run_info["flag"], base.run_info["flag"]) and np.abs(e.run_info["t_end"] - base.run_info["t_end"]).max() == 0.0
                and np.abs(e.run_info["SOC"] - base.run_info["SOC"]).max() < 1e-11 and np.abs(e.run_info["V"] - base.run_info["V"]).max() <= vtol and np.abs(e.SOC[:, 0] - 1.0).max() == 0.0):
            bad = name
    if not bad:
        # the continuation path (a second run inherits SOC, time, V / I from the first: the path of the miscompile that was seen) in the plain and the table instantiation,
        # and the sensitivity instantiation, whose states must be those of the plain kernel bit for bit
        two = [{"I": -1.0, "tf": tf / 2}, {"I": -1.0, "tf": tf / 2}]
        two_tab = [{"I": -1.0, "tf": tf / 2}, {"I": ([0.0, 1e7], [-1.0, -1.0]), "tf": tf / 2}]
        for name, proto in (("plain, two runs", two), ("table input, two runs", two_tab)):
            e = simulate_ensemble(p, Th, proto, SOC=1.0)
            if not ((e.run_info["flag"] == 0).all() and np.abs(e.run_info["SOC"][:, 1] - (1.0 - tf / 3600.0)).max() < 1e-11 and np.abs(e.run_info["t_end"][:, 1] - tf).max() < 1e-9
                    and np.abs(e.run_info["V"][:, 1] - base.run_info["V"][:, 0]).max() <= 2e-3):
                bad = name
                break
    if not bad and p.waves_per_cell != 2:
        # (compared like the other instantiations -- flags and end times equal, SOC to 1e-11, voltage and states within the tolerance the run was made at: the sensitivity kernel
        #  is another compilation of the step loop, and on a build other than the validated one its sums may be contracted differently; bit-identity is what the validated
        #  binary shows, tests/test_sensitivities.py, not what a correct build must show)
        e = simulate_ensemble(p, Th, [{"I": -1.0, "tf": tf}], SOC=1.0, sens=[p.θ_keys[0]])
        scale = np.abs(base.Y).max(axis=0) + 1e-300
        if not (np.array_equal(e.run_info["flag"], base.run_info["flag"]) and np.abs(e.run_info["t_end"] - base.run_info["t_end"]).max() == 0.0
                and np.array_equal(e.counters["n_steps"], base.counters["n_steps"]) and np.array_equal(e.counters["n_newton"], base.counters["n_newton"])
                and np.abs(e.run_info["SOC"] - base.run_info["SOC"]).max() < 1e-11 and np.abs(e.run_info["V"] - base.run_info["V"]).max() <= 1e-8
                and (np.abs(e.Y - base.Y) / scale).max() <= 1e-8 and (np.asarray(e.sens_stat)[:, 1] == 0).all()):          # (the header's guarantee -- same steps, states to ~1e-9 -- held to 1e-8, not to 10 x reltol)
            bad = "sensitivity"
    if bad:
        raise RuntimeError("kernel self-test failed: the %s instantiation of %s does not reproduce the plain kernel on a 1C discharge -- a miscompiled build "
                           "(rebuild; see DESIGN.md 5a)" % (bad, p.variant))


_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "selftest_golden.json")
KA_TOL = 2e-6        # 200 x the reltol of the known-answer protocol: two correct builds (other compiler, other contraction of a sum) agree to ~1e-7 there (tests/test_gpu_tight.py)
KA_ATOL = 1e-9       # 10 x the abstol of the known-answer protocol: the absolute floor of a state-section comparison (known_answer_check)


def known_answer_protocol(p):
    """the fixed protocol of the kernel self-test's known answer: 2 cells (default theta; D_sp x 1.3, k_n x 0.8), 3 runs (1C discharge 100 s, V hold 50 s, rest 50 s) at reltol
    1e-8 / abstol 1e-10 -- tight, so that the result does not depend on the last bits of the step selection (at the default tolerances two correct builds differ by 1e-6)"""
    Th = np.tile(p.theta_vector(), (2, 1))
    Th[1, p.θ_keys.index("D_sp")] *= 1.3
    Th[1, p.θ_keys.index("k_n")] *= 0.8
    o = Opts()
    o.reltol, o.abstol, o.reltol_init, o.abstol_init, o.maxiters = 1e-8, 1e-10, 1e-8, 1e-10, 100000
    return Th, [{"I": -1.0, "tf": 100.0}, {"V": "hold", "tf": 50.0}, {"I": "rest", "tf": 50.0}], o


def known_answer_digest(p):
    """what is compared: per run (t_end, V, I, SOC), per state section of the end state (max |Y|, sum Y)"""
    Th, proto, o = known_answer_protocol(p)
    e = simulate_ensemble(p, Th, proto, SOC=1.0, opts=o, max_points=20000)
    d = {"flags": e.run_info["flag"].tolist(), "t_end": e.run_info["t_end"].tolist(), "V": e.run_info["V"].tolist(), "I": e.run_info["I"].tolist(), "SOC": e.run_info["SOC"].tolist(),
         "sections": {nm: [[float(np.abs(e.Y[c, sl]).max()), float(e.Y[c, sl].sum())] for c in range(2)] for nm, sl in p.ind.items()}}
    return d


def _golden_key(p):
    return "%s|%s" % (p.variant, p.precision)


def known_answer_check(p):
    """None if the plain kernel reproduces the committed known answer of this variant / grid (selftest_golden.json: produced by tools/make_selftest_golden.py on the binary that
    profiles/validated_build.json names), else a description of the first disagreement.  A discretisation without a committed answer (a grid library for a grid of the user's
    own) is checked against the built-in default-grid kernel of the same model instead -- two discretisations of the same cell agree to their truncation error, which catches a
    broken equation, not a last-digit one."""
    import json
    try:
        gold = json.load(open(_GOLDEN))["digests"]
    except (OSError, ValueError, KeyError):
        gold = {}
    g = gold.get(_golden_key(p))
    d = known_answer_digest(p)
    if g is None:
        return _cross_grid_check(p, d)
    if d["flags"] != g["flags"]:
        return "exit flags %r, known %r" % (d["flags"], g["flags"])
    for k in ("t_end", "V", "I", "SOC"):
        a, b = np.asarray(d[k]), np.asarray(g[k])
        if not np.all(np.abs(a - b) <= KA_TOL * np.maximum(1.0, np.abs(b))):
            return "%s of the runs %r, known %r" % (k, a.tolist(), b.tolist())
    for nm, rows in g["sections"].items():
        for c in range(2):
            mx, sm = d["sections"][nm][c]
            gmx, gsm = rows[c]
            n = p.ind[nm].stop - p.ind[nm].start
            # (absolute floor: ten times the ABSOLUTE tolerance the protocol is integrated with -- a field that has relaxed to ~1e-5 of its scale, Phi_e at the end of the rest, is
            #  only determined to abstol, and two correct builds whose step sequences differ in one step differ there by that much: r06, 1.1e-10 in max |Phi_e| = 2.08e-5 V)
            if abs(mx - gmx) > KA_TOL * max(gmx, 1e-300) + KA_ATOL or abs(sm - gsm) > KA_TOL * n * max(gmx, 1e-300) + n * KA_ATOL:
                return "state section %s of cell %d: max |Y| %.12g (known %.12g), sum %.12g (known %.12g)" % (nm, c, mx, gmx, sm, gsm)
    return None


def _cross_grid_check(p, d):
    """no committed answer for this discretisation: the same model on the built-in default grid (whose own answer IS committed) must give the same cell voltage, current and
    SOC at the run ends to the truncation error of a coarse grid"""
    if getattr(p, "_lib_path", None) is not None or not getattr(p, "_grid_lib_built", None):
        return None
    try:
        ref = Model(p.cathode, _N(p=10, s=10, n=10, a=10, z=10, r_p=10, r_n=10), p.temperature, p.aging, None, p.precision, -1, p.solid_diffusion, p.thermodynamic_factor, p.rxn, p.waves_per_cell, None)
    except Exception:
        return None
    g = known_answer_digest(ref)
    if d["flags"] != g["flags"]:
        return "exit flags %r, default grid %r" % (d["flags"], g["flags"])
    # (a coarse grid -- 2 control volumes per section -- is tens of mV from the default one: this catches a broken equation, not a discretisation)
    if np.abs(np.asarray(d["V"]) - np.asarray(g["V"])).max() > 0.15 or np.abs(np.asarray(d["SOC"]) - np.asarray(g["SOC"])).max() > 5e-3 or np.abs(np.asarray(d["t_end"]) - np.asarray(g["t_end"])).max() > 1e-6:
        return "run-end V / SOC / t_end %r / %r / %r, default grid %r / %r / %r" % (d["V"], d["SOC"], d["t_end"], g["V"], g["SOC"], g["t_end"])
    return None


def _gpu_visible(p):
    """the library's own device count (no second GPU runtime in the process, no dependence on torch)"""
    try:
        return p._lib.plh_device_count() > 0
    except AttributeError:
        return False


def _guarded_selftest(p, marker, identity, what):
    """run selftest(p) once per (marker file, identity string); warn -- never skip silently -- when no GPU is visible"""
    import warnings
    try:
        if os.path.exists(marker) and open(marker).read().strip() == identity:
            return
    except OSError:
        pass
    if not _gpu_visible(p):
        warnings.warn("petlion.jl_amd: kernel self-test of %s skipped (no GPU visible to the library); it runs on the first machine with a GPU" % what, RuntimeWarning)
        return
    selftest(p)
    try:
        with open(marker, "w") as f:
            f.write(identity + "\n")
    except OSError:
        pass


def _selftest_new_grid_library(p):
    """a grid library compiled at first use (petlion.jl_amd/grids.py) is checked once per (library, variant, compiler / host build identity) on the first machine with a GPU
    that loads it (a cache copied from another machine carries the identity it was checked under; a different compiler or host library checks again)"""
    lib = getattr(p, "_grid_lib_built", None)
    if not lib:
        return
    _guarded_selftest(p, "%s.%s.selftest" % (lib, p.variant), build_info(p), "the grid library %s" % os.path.basename(lib))


def build_info(p=None):
    """plh_build_info(): compiler versions, flag hash and source hash of the loaded library"""
    lib = p._lib if p is not None else cap.load()
    try:
        return lib.plh_build_info().decode("utf-8", "replace")
    except AttributeError:
        return "unknown"


_VALIDATED = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "validated_build.json")


def _selftest_unvalidated_build(p):
    """the built-in kernels are validated by the GPU test suite, and profiles/validated_build.json records the identity of the binary that run validated.  A
    library whose plh_build_info() differs -- another hipcc, other flags, edited sources -- has not been through that run: the first petlion() of each variant then runs the
    kernel self-test (every k_integrate instantiation against the plain one, including a two-run continuation) once per build and machine."""
    if not getattr(p, "_builtin_lib", False) or getattr(p, "_grid_lib_built", None):
        return
    info = build_info(p)
    try:
        import json
        if json.load(open(_VALIDATED)).get("build_info") == info:
            return
    except (OSError, ValueError):
        pass
    _guarded_selftest(p, "%s.%s.selftest" % (cap.LIB_PATH, p.variant), info, "this build of libpetlion_hip.so (%s: not the validated binary)" % info)


# ---------------------------------------------------------------------------------------------------------------------
_INPUTS = ("I", "V", "dT", "P", "η_p", "res") + tuple(cap.DSTATE)
_MODE = {"I": cap.MODE_I, "V": cap.MODE_V, "dT": cap.MODE_DT, "P": cap.MODE_P, "η_p": cap.MODE_ETA_P, "res": cap.MODE_RES}
_MODE.update({k: cap.MODE_DSTATE for k in cap.DSTATE})       # dc_s_p_max, ..., dc_e_min (input_methods.jl:190-247): the rate of one differential state, chosen per cell at the start of the run
_BOUND_KW = Bounds.FIELDS


def _make_run(p, name, inp, tf, bounds):
    r = cap.Run()
    r.mode = _MODE[name]
    res_x = 0.0
    if name in cap.DSTATE:
        r.dstate = cap.DSTATE[name]
        if callable(inp) or isinstance(inp, (tuple, list)):
            raise ValueError("%s takes a number or :hold" % name)
    if name == "res":
        # user-defined control residual (reference input_methods.jl:155-175, custom_res!, model_evaluation.jl:155-172): res = f  ->  0 - f = 0;  res = (x, f)  ->  x - f = 0
        if isinstance(inp, (tuple, list)) and len(inp) == 2 and callable(inp[1]) and not callable(inp[0]):
            res_x, inp = float(inp[0]), inp[1]
        if not callable(inp):
            raise ValueError("res = f or res = (x, f) with a closure f(t, Y, p)")
    if isinstance(inp, str):
        if inp == "hold":
            r.value_kind, r.value = cap.VAL_HOLD, 0.0
        elif inp == "rest":
            if name not in ("I", "P"):
                raise ValueError("Unsupported input symbol.")       # input_methods.jl:24,97
            r.value_kind, r.value = cap.VAL_REST, 0.0
        else:
            raise ValueError("Unsupported input symbol.")
    elif callable(inp):
        # a closure cannot cross the C ABI, its expression can: traced once into a postfix program (closures.py; PLH_VAL_EXPR).  A closure that branches on t or Y in Python
        # cannot be traced (closures.TraceError says how to rewrite it); a piecewise-linear table (t, values) + opts.tdiscon remains the other way in.
        if name == "dT":
            raise ValueError("function inputs for dT are not defined by the reference")
        (ops, args), tree = closures.trace(inp, p, with_tree=True)
        r.value_kind, r.value = cap.VAL_EXPR, 0.0
        r.n_tab = ops.size
        # a closure of Y: the entries d f / d Y[c] of the control row, as the reference's symbolic differentiation builds them (scalar_residual.jl:276-416); the programs ride
        # behind the main one in the same arrays (plh_run.dcol / dofs)
        der = closures.row_derivatives(tree, p.N.tot, p.N.diff)
        keep = []
        if der is not None:
            dcol = np.ascontiguousarray(der[0], dtype=np.int32)
            dofs = np.ascontiguousarray(np.cumsum([ops.size] + [q[0].size for q in der[1]]), dtype=np.int32)
            ops = np.concatenate([ops] + [q[0] for q in der[1]]); args = np.concatenate([args] + [q[1] for q in der[1]])
            r.n_dcol = dcol.size
            r.dcol = dcol.ctypes.data_as(C.POINTER(C.c_int)); r.dofs = dofs.ctypes.data_as(C.POINTER(C.c_int))
            keep = [dcol, dofs]
        r.tab_t = ops.ctypes.data_as(C.POINTER(C.c_double)); r.tab_v = args.ctypes.data_as(C.POINTER(C.c_double))
        r._keep = (ops, args, *keep)
        if name == "res":
            if der is None:
                raise ValueError("res: the closure must read the state (Y, or YP of differential states)")
            r.value = res_x
    elif isinstance(inp, (tuple, list)) and len(inp) == 2 and np.ndim(inp[0]) == 1:
        if name == "dT":
            raise ValueError("time-dependent dT inputs are not defined by the reference")
        tt = np.ascontiguousarray(inp[0], dtype=np.float64); vv = np.ascontiguousarray(inp[1], dtype=np.float64)
        if tt.shape != vv.shape or tt.size < 1 or (np.diff(tt) < 0).any():
            raise ValueError("table input: (t, v) arrays of equal length with non-decreasing times")
        r.value_kind, r.value = cap.VAL_TABLE, float(vv[0])
        r.n_tab = tt.size
        r.tab_t = tt.ctypes.data_as(C.POINTER(C.c_double)); r.tab_v = vv.ctypes.data_as(C.POINTER(C.c_double))
        r._keep = (tt, vv)                              # keep the arrays alive as long as the run descriptor
    elif isinstance(inp, np.ndarray) and inp.ndim == 1:                       # one constant input per cell (ensemble axis of the protocol)
        vv = np.ascontiguousarray(inp, dtype=np.float64)
        r.value_kind, r.value = cap.VAL_CONST, float(vv[0])
        r.value_cell = vv.ctypes.data_as(C.POINTER(C.c_double))
        r._keep = (vv,)
    else:
        r.value_kind, r.value = cap.VAL_CONST, float(inp)
    if isinstance(tf, np.ndarray):                                            # one run length per cell
        tt_ = np.ascontiguousarray(tf, dtype=np.float64)
        r.tf = float(tt_[0]); r.tf_cell = tt_.ctypes.data_as(C.POINTER(C.c_double)); r._keep_tf = tt_
    else:
        r.tf = float(tf)
    for f in cap.BOUND_FIELDS:
        setattr(r.bounds, f, getattr(bounds, "η_plating_min" if f == "eta_plating_min" else f))
    return r


def _opts_struct(o, p=None):
    s = cap.Opts(o.abstol, o.reltol, o.abstol if o.abstol_init is None else o.abstol_init,
                 o.reltol if o.reltol_init is None else o.reltol_init, int(o.maxiters), int(bool(o.check_bounds)),
                 int(bool(o.interp_final)), int(o.max_order), int(bool(o.jac_every_step)), float(o.init_step))
    td = np.ascontiguousarray(list(getattr(o, "tdiscon", []) or []), dtype=np.float64)
    s.n_tdiscon = td.size
    s.tdiscon = td.ctypes.data_as(C.POINTER(C.c_double)) if td.size else None
    s.refine = int(getattr(o, "refine", 0))
    ts = np.ascontiguousarray(list(getattr(o, "tstops", []) or []), dtype=np.float64)      # opts.tstops (run-local times; model_evaluation.jl:292-294)
    s.n_tstops = ts.size
    s.tstops = ts.ctypes.data_as(C.POINTER(C.c_double)) if ts.size else None
    s._keep = (td, ts)                               # the arrays must outlive the struct
    s.yp_alg_zero = int(bool(getattr(o, "yp_alg_zero", False)))
    # opts.stop_function (src/structures.jl:283, src/checks.jl:26): a closure g(t, Y, YP, p) traced into a postfix program like an input closure (closures.py); the run ends
    # when g > 0, exit flag 12, with the same linear back-interpolation as the built-in bounds (plh_opts.stop_ops)
    sf = getattr(o, "stop_function", None)
    s.n_stop, s.stop_ops, s.stop_args = 0, None, None
    if sf is not None:
        if p is None:
            raise ValueError("opts.stop_function needs the model it is traced for")
        ops, args = sf if isinstance(sf, tuple) else closures.trace(sf, p)
        ops, args = np.ascontiguousarray(ops, dtype=np.float64), np.ascontiguousarray(args, dtype=np.float64)
        s.n_stop = ops.size
        s.stop_ops, s.stop_args = ops.ctypes.data_as(C.POINTER(C.c_double)), args.ctypes.data_as(C.POINTER(C.c_double))
        s._keep = s._keep + (ops, args)
    return s


class RunResult:
    def __init__(self, name, tspan, flag, iterations, info):
        self.name = name
        self.tspan = tspan
        self.flag = int(flag)
        self.exit_reason = EXIT_REASONS.get(int(flag), "error %d" % flag)
        self.iterations = int(iterations)
        self.info = info


class Solution:
    """reference `solution` (src/outputs.jl:78-105), for the outputs the device records per step."""

    def __init__(self):
        self.t = np.zeros(0)
        self.V = np.zeros(0)
        self.I = np.zeros(0)
        self.SOC = np.zeros(0)
        self.P = np.zeros(0)
        self.Y = None          # last state vector (sol.Y[end])
        self.YP = None
        self.T_avg = None      # [points] with temperature = true
        self.Y_all = None      # [points, N] when the run was made with outputs = "all" / a state name (sol.Y of the reference)
        self._ind = None
        self.results = []
        self.counters = None

    def __len__(self):
        return len(self.t)

    def __getattr__(self, name):
        # sol.c_e, sol.T, sol.j, sol.Φ_e ... : per-step state sections, available when the states were kept (outputs = "all")
        ind = self.__dict__.get("_ind")
        if ind and name in ind:
            if self.__dict__.get("Y_all") is None:
                raise AttributeError("%s was not saved: run with outputs='all' (or outputs=(%r,))" % (name, name))
            return self.Y_all[:, ind[name]]
        raise AttributeError(name)

    def isempty(self):
        return len(self.results) == 0

    def __call__(self, t, interp_bc="interpolate", k=3):
        """sol(t): the reference's post-interpolation of a solution (src/save_outputs.jl:74-133).  Every requested time is assigned to the run whose
        tspan contains it (first match; before the first run -> run 1, otherwise the last run), each run's saved points get an interpolating spline
        of degree min(k, points) -- Dierckx `Spline1D(t, x; k, bc)` with s = 0, i.e. FITPACK curfit/splev, which is exactly what
        scipy.interpolate.splrep/splev call -- evaluated with bc = "nearest" (interp_bc = "interpolate") or "extrapolate"."""
        from scipy.interpolate import splev, splrep
        if interp_bc not in ("interpolate", "extrapolate"):
            raise ValueError("Invalid interp_bc method.")
        tq = np.atleast_1d(np.asarray(t, dtype=np.float64))
        spans = [r.tspan for r in self.results]
        which = np.full(tq.shape, len(spans) - 1)
        for q, tv in enumerate(tq):
            if tv < spans[0][0]:
                which[q] = 0
                continue
            for i, (a, b) in enumerate(spans):
                if a <= tv <= b:
                    which[q] = i
                    break
        out = Solution()
        out.t = tq
        out.results = [self.results[i] for i in sorted(set(which.tolist()))]
        out.Y, out.YP, out.counters = self.Y, self.YP, self.counters
        out._ind = self._ind
        start = np.concatenate([[0], np.cumsum([r.iterations for r in self.results])])
        names = ["V", "I", "SOC", "P"] + (["T_avg"] if self.T_avg is not None else []) + (["Y_all"] if self.Y_all is not None else [])
        for name in names:
            x = getattr(self, name)
            y = np.zeros(tq.shape + x.shape[1:])
            for i in set(which.tolist()):
                pts = slice(int(start[i]), int(start[i + 1]))
                n = int(start[i + 1] - start[i])
                kk = min(k, n)
                if kk >= n:
                    kk = max(1, n - 1)          # FITPACK needs more points than the degree
                sel = which == i
                if n < 2:
                    y[sel] = x[pts][0]
                    continue
                ext = 3 if interp_bc == "interpolate" else 0
                if x.ndim == 1:
                    y[sel] = splev(tq[sel], splrep(self.t[pts], x[pts], k=kk, s=0), ext=ext)
                else:                           # VectorOfArray fields: one spline per state (save_outputs.jl:104-119)
                    for c in range(x.shape[1]):
                        y[sel, c] = splev(tq[sel], splrep(self.t[pts], x[pts, c], k=kk, s=0), ext=ext)
            setattr(out, name, y)
        return out

    def __repr__(self):
        if self.isempty():
            return "PETLION simulation (empty)"
        r = self.results[-1]
        runs = " → ".join(x.name for x in self.results)
        return ("PETLION simulation\n  --------\n  Runs:    %s\n  Time:    %.2f s\n  Current: %.4gC\n  Voltage: %.4f V\n"
                "  Power:   %.4f W/m²\n  SOC:     %.4f\n  Exit:    %s" % (runs, self.t[-1], self.I[-1], self.V[-1], self.P[-1], self.SOC[-1], r.exit_reason))


def exit_reasons(sol):
    return [r.exit_reason for r in sol.results]


def final_exit_reason(sol):
    return sol.results[-1].exit_reason


def _split_kwargs(p, kw):
    inputs = {k: kw.pop(k) for k in list(kw) if k in _INPUTS}
    if len(inputs) != 1:
        raise ValueError("exactly one input (I, V or dT) must be selected")     # check_input_arguments, checks.jl:270-282
    bounds = p.bounds.copy(**{k: kw.pop(k) for k in list(kw) if k in _BOUND_KW})
    return inputs, bounds, kw


_STATE_OUTPUTS = ("Y", "c_e", "c_s_avg", "T", "film", "SOH", "j", "j_s", "Φ_e", "Φ_s")


def _wants_states(p, outputs):
    """outputs = :all or any per-node state asks for the state vector of every step (solution_states_logic, src/outputs.jl:107-131)."""
    outs = (outputs,) if isinstance(outputs, str) else tuple(outputs or ())
    return "all" in outs or any(x in _STATE_OUTPUTS for x in outs)


def calc_SOC(p, Y):
    """calc_SOC(Y, p) (reference src/physics_equations/scalar_residual.jl:95-102): the anode's mean c_s_avg as a fraction of its stoichiometry window"""
    Y = np.asarray(Y, dtype=np.float64)
    cs = p.ind["c_s_avg"]
    n_p = p.N.p * p.N.r_p if p.solid_diffusion == "Fickian" else p.N.p
    return (Y[..., cs.start + n_p:cs.stop].mean(axis=-1) / p.θ["c_max_n"] - p.θ["θ_min_n"]) / (p.θ["θ_max_n"] - p.θ["θ_min_n"])


def simulate(p, tf=1e6, *, sol=None, SOC=None, initial_states=None, **kw):
    """simulate(p, tf; I=..|V=..|dT=.., SOC, abstol, reltol, ..., V_max, V_min, ...) for ONE cell (n_cells = 1 ensemble).
    initial_states = Y (reference src/model_evaluation.jl:15, 102-110, 193-199): a new solution that starts from this state vector instead of initial_guess!; its SOC is
    calc_SOC(Y)."""
    inputs, bounds, rest = _split_kwargs(p, kw)
    if initial_states is not None:
        if sol is not None and not sol.isempty():
            raise ValueError("Cannot set `initial_states` and continue a previous run.")         # model_evaluation.jl:105-108
        initial_states = np.ascontiguousarray(initial_states, dtype=np.float64)
        if initial_states.shape != (p.N.tot,):
            raise ValueError("initial_states must have length N.tot")
    o = Opts()
    o.__dict__.update(p.opts.__dict__)
    for k, v in rest.items():
        if not hasattr(o, k):
            raise TypeError("unknown keyword %r" % k)
        setattr(o, k, v)
    (name, inp), = inputs.items()
    # simulate(p, tf::Vector): run to tf[end], then post-interpolate onto tf (model_evaluation.jl:79-80, 148-149)
    tf_interp = None
    if isinstance(tf, (list, tuple, np.ndarray)):
        tf_interp = np.asarray(tf, dtype=np.float64)
        if tf_interp.ndim != 1 or tf_interp.size == 0:
            raise ValueError("tf must be a number or a non-empty vector of times")
        tf = float(tf_interp[-1])
    new = sol is None or sol.isempty()
    sol = Solution() if sol is None else sol
    keep_Y = _wants_states(p, o.outputs) or sol.Y_all is not None
    soc0 = (p.opts.SOC if SOC is None else SOC) if new else sol.SOC[-1]
    if initial_states is not None:
        soc0 = float(calc_SOC(p, initial_states))                      # (starting_from_initial_state: the estimated SOC, model_evaluation.jl:197-199)
    # opts.save_start (src/structures.jl:281, model_evaluation.jl:384-411): the algebraic states a consistent initialisation ended with, kept per (method, SOC rounded to 4
    # digits, input value rounded to 4 digits) in the model's cache and used as the first guess of the next initialisation with the same key.  The reference looks the key up for
    # every run; here it is the warm start of a NEW solution with a numeric input (a continued run already starts from the previous run's states).
    ss_key, ss_hit = None, False
    if getattr(o, "save_start", False) and new and initial_states is None and isinstance(inp, (int, float)) and not isinstance(inp, bool):
        ss_key = (name, round(float(soc0), 4), round(float(inp), 4))
        ss_hit = ss_key in p.save_start_dict
        if ss_hit:
            th1 = p.theta_vector()[None, :]
            Y0 = np.zeros((1, p.N.tot)); s1 = np.array([float(soc0)])
            cap.check(p._lib, p._lib.plh_initial_guess(p._h, 1, th1.ctypes.data, s1.ctypes.data, Y0.ctypes.data, cap.PLH_HOST, None), "plh_initial_guess")
            Y0[0, p.N.diff:] = p.save_start_dict[ss_key]
            initial_states = Y0[0]
    want_Y0 = ss_key is not None and not ss_hit                        # (a miss: the initialised algebraic states are the first saved state vector)
    ens = _integrate(p, p.theta_vector()[None, :], np.array([soc0]), [_make_run(p, name, inp, tf, bounds)], o,
                     Y_init=(None if initial_states is None else initial_states[None, :]) if new else sol.Y[None, :], t_init=None if new else np.array([sol.t[-1]]), keep_Y=keep_Y or want_Y0)
    if want_Y0 and int(ens["run_info"][0, 0]["flag"]) >= 0:
        p.save_start_dict[ss_key] = ens["Y_all"][0, 0, p.N.diff:].copy()
    n = int(ens["n_pts"][0])
    ri = ens["run_info"][0, 0]
    if ri["flag"] < 0:                                                # the reference's error() paths: `sol` is left untouched
        raise RuntimeError(EXIT_REASONS.get(int(ri["flag"]), "error"))
    I1C = calc_I1C(p.θ)
    for fld in ("t", "V", "I", "SOC"):
        setattr(sol, fld, np.concatenate([getattr(sol, fld), ens[fld][0, :n]]))
    sol.P = sol.I * I1C * sol.V                                   # calc_P, scalar_residual.jl:87
    sol._ind = p.ind
    if "T_avg" in ens:
        sol.T_avg = np.concatenate([sol.T_avg if sol.T_avg is not None else np.zeros(0), ens["T_avg"][0, :n]])
    if keep_Y:
        prev = sol.Y_all if sol.Y_all is not None else np.zeros((0, p.N.tot))
        if prev.shape[0] != len(sol.t) - n:
            raise ValueError("the solution being continued was not saved with outputs='all'")
        sol.Y_all = np.concatenate([prev, ens["Y_all"][0, :n]])
    sol.Y, sol.YP = ens["Y"][0].copy(), ens["YP"][0].copy()
    t_start = ens["t"][0, 0]
    sol.results.append(RunResult(name, (t_start, ri["t_end"]), ri["flag"], ri["iterations"], ri))
    sol.counters = ens["counters"][0]
    if tf_interp is not None:
        return sol(tf_interp, interp_bc=o.interp_bc)
    return sol


def simulate_b(sol, p, tf=1e6, **kw):
    """simulate!(sol, p, tf; kw...) -- continue `sol` (reference src/model_evaluation.jl:87-97)."""
    return simulate(p, tf, sol=sol, **kw)


def _integrate(p, theta, SOC0, runs, o, Y_init=None, t_init=None, device=False, stream=None, max_points=None, keep_Y=False, keep_YP=True, sens=None):
    """one plh_integrate call; numpy in / numpy out (host pointers) or torch device tensors (device=True).
    keep_YP = False: YP_final is not requested (the reference keeps YP only with var_keep.YP; the kernel then does not store the previous point's YP per step).
    sens: list of theta keys -> plh_integrate_sens, bufs["dY_dtheta"][cell, k, state], bufs["dV_dtheta"][cell, k, point], bufs["sens_stat"][cell, 3]."""
    lib, h = p._lib, p._h
    n = theta.shape[0]
    N = p.N.tot
    arr = (cap.Run * len(runs))(*runs)
    os_ = _opts_struct(o, p)
    mp = int(max_points or o.max_points)
    out = cap.Outputs()
    out.max_pts = mp
    if device:
        import torch
        dev = theta.device
        mk = lambda *shape, dt=torch.float64: torch.empty(*shape, dtype=dt, device=dev)
        bufs = dict(t=mk(n, mp), V=mk(n, mp), I=mk(n, mp), SOC=mk(n, mp), n_pts=mk(n, dt=torch.int32), Y=mk(n, N), YP=mk(n, N),
                    run_info=torch.empty(n * len(runs) * cap.RUN_INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev),
                    counters=torch.empty(n * cap.COUNTERS_DTYPE.itemsize, dtype=torch.uint8, device=dev))
        kind = cap.PLH_DEVICE
    else:
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        SOC0 = np.ascontiguousarray(SOC0, dtype=np.float64)
        # (np.empty: the library overwrites every array in full; only the first n_pts[i] entries of a per-point row are meaningful)
        bufs = dict(t=np.empty((n, mp)), V=np.empty((n, mp)), I=np.empty((n, mp)), SOC=np.empty((n, mp)), n_pts=np.zeros(n, np.int32),
                    Y=np.empty((n, N)), YP=np.empty((n, N)), run_info=np.zeros((n, len(runs)), cap.RUN_INFO_DTYPE),
                    counters=np.zeros(n, cap.COUNTERS_DTYPE))
        if Y_init is not None:
            Y_init = np.ascontiguousarray(Y_init, dtype=np.float64)
            t_init = None if t_init is None else np.ascontiguousarray(t_init, dtype=np.float64)      # (None: initial_states -- a new solution from these states)
        kind = cap.PLH_HOST
    out.t, out.V, out.I, out.SOC = cap.ptr(bufs["t"]), cap.ptr(bufs["V"]), cap.ptr(bufs["I"]), cap.ptr(bufs["SOC"])
    out.T_avg = None
    out.Y_all = None
    if p.temperature:                   # per-step average temperature (calc_T_avg) only exists with temperature = true
        bufs["T_avg"] = mk(n, mp) if device else np.empty((n, mp))
        out.T_avg = cap.ptr(bufs["T_avg"])
    if keep_Y:                          # outputs = :all : every saved state vector
        bufs["Y_all"] = mk(n, mp, N) if device else np.empty((n, mp, N))
        out.Y_all = cap.ptr(bufs["Y_all"])
    out.n_pts, out.Y_final, out.YP_final = cap.ptr(bufs["n_pts"]), cap.ptr(bufs["Y"]), cap.ptr(bufs["YP"]) if keep_YP else None
    out.run_info, out.counters = cap.ptr(bufs["run_info"]), cap.ptr(bufs["counters"])
    if sens:
        cols = np.ascontiguousarray([p.θ_keys.index(k) for k in sens], dtype=np.int32)
        ns = len(cols)
        if device:
            bufs["dY_dtheta"], bufs["dV_dtheta"], bufs["sens_stat"] = mk(n, ns, N), mk(n, ns, mp), mk(n, 3, dt=torch.int32)
        else:
            bufs["dY_dtheta"], bufs["dV_dtheta"], bufs["sens_stat"] = np.empty((n, ns, N)), np.empty((n, ns, mp)), np.zeros((n, 3), np.int32)
        if Y_init is not None:
            raise ValueError("sensitivities are integrated for new solutions only")
        cap.check(lib, lib.plh_integrate_sens(h, n, cap.ptr(theta), cap.ptr(SOC0), len(runs), arr, C.byref(os_), C.byref(out), ns, cols.ctypes.data,
                                              cap.ptr(bufs["dY_dtheta"]), cap.ptr(bufs["dV_dtheta"]), cap.ptr(bufs["sens_stat"]), kind, stream), "plh_integrate_sens")
    else:
        t_call = time.perf_counter()
        rc = lib.plh_integrate(h, n, cap.ptr(theta), cap.ptr(SOC0), cap.ptr(Y_init), cap.ptr(t_init), len(runs), arr, C.byref(os_), C.byref(out), kind, stream)
        bufs["call_ms"] = 1e3 * (time.perf_counter() - t_call)          # wall time inside plh_integrate (a blocking PLH_HOST call: staging, kernel, the way back)
        cap.check(lib, rc, "plh_integrate")
    if device:
        if stream is not None and int(stream) != torch.cuda.current_stream(dev).cuda_stream:
            # the buffers were allocated on torch's current stream but the kernel runs on `stream`: tell the caching allocator, or it may hand the
            # memory out again while the kernel still writes it
            ext = torch.cuda.ExternalStream(int(stream), device=dev)
            for v in list(bufs.values()) + [theta, SOC0]:
                if hasattr(v, "record_stream"):
                    v.record_stream(ext)
        bufs["kernel_ms"] = lambda: lib.plh_last_kernel_ms(h)     # evaluated on access: the launch is asynchronous on `stream`
    else:
        bufs["kernel_ms"] = lib.plh_last_kernel_ms(h)
    return bufs


class EnsembleSolution:
    """Per-cell results of an ensemble run (arrays indexed [cell, point]); sol[i] gives a single-cell Solution."""

    def __init__(self, p, bufs, run_names):
        self.p = p
        self.t, self.V, self.I, self.SOC = bufs["t"], bufs["V"], bufs["I"], bufs["SOC"]
        self.n_pts = bufs["n_pts"]
        self.Y, self.YP = bufs["Y"], bufs["YP"]
        self.T_avg = bufs.get("T_avg")          # [cell, point] with temperature = true
        self.Y_all = bufs.get("Y_all")          # [cell, point, state] with outputs = "all"
        self._run_info = bufs["run_info"]
        self._counters = bufs["counters"]
        self.run_names = run_names
        self._kernel_ms = bufs.get("kernel_ms", -1.0)
        self.call_ms = bufs.get("call_ms")      # wall time of the plh_integrate call itself (host pointers: includes the copies back)
        # forward parameter sensitivities (simulate_ensemble(..., sens=[keys])): [cell, k, state] at the end of the protocol, [cell, k, point] for the voltage
        self.dY_dtheta, self.dV_dtheta, self.sens_stat = bufs.get("dY_dtheta"), bufs.get("dV_dtheta"), bufs.get("sens_stat")

    # With device=True the launch is asynchronous: the per-cell summaries stay in HBM until they are looked at (the first access synchronises),
    # so a host loop can enqueue launches back to back.
    @property
    def run_info(self):
        if not isinstance(self._run_info, np.ndarray):
            self._run_info = self._run_info.cpu().numpy().view(cap.RUN_INFO_DTYPE).reshape(self.t.shape[0], len(self.run_names))
        return self._run_info

    @property
    def counters(self):
        if not isinstance(self._counters, np.ndarray):
            self._counters = self._counters.cpu().numpy().view(cap.COUNTERS_DTYPE).reshape(self.t.shape[0])
        return self._counters

    @property
    def kernel_ms(self):
        """duration of the integrate kernel of the handle's LAST launch (HIP events on the launch stream)"""
        if callable(self._kernel_ms):
            self._kernel_ms = self._kernel_ms()
        return self._kernel_ms

    @property
    def n_cells(self):
        return self.t.shape[0]

    def flags(self):
        return self.run_info["flag"]

    def __getitem__(self, i):
        host = lambda x: (x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)).copy()      # device=True results are torch tensors in HBM
        s = Solution()
        n = int(self.n_pts[i])
        s.t, s.V, s.I, s.SOC = host(self.t[i, :n]), host(self.V[i, :n]), host(self.I[i, :n]), host(self.SOC[i, :n])
        s.P = s.I * calc_I1C(self.p.θ) * s.V
        s.Y, s.YP = host(self.Y[i]), host(self.YP[i])
        s._ind = self.p.ind
        if self.T_avg is not None:
            s.T_avg = host(self.T_avg[i, :n])
        if self.Y_all is not None:
            s.Y_all = host(self.Y_all[i, :n])
        t0 = 0.0
        for k, nm in enumerate(self.run_names):
            ri = self.run_info[i, k]
            s.results.append(RunResult(nm, (t0, ri["t_end"]), ri["flag"], ri["iterations"], ri))
            t0 = ri["t_end"]
        s.counters = self.counters[i]
        return s


def make_protocol(p, protocol, n_cells=None):
    """protocol: list of dicts like {"I": 2, "tf": 1800, "V_max": 4.1} / {"V": "hold", "I_min": 1/20} -- each dict is the
    keyword set of one simulate()/simulate!() call.  An input value or `tf` given as a 1-D numpy array of length n_cells is applied cell by cell
    (a C-rate sweep is `{"I": -np.linspace(0.5, 3, n)}`)."""
    runs, names = [], []
    for step in protocol:
        step = dict(step)
        tf = step.pop("tf", 1e6)
        inputs, bounds, rest = _split_kwargs(p, step)
        if rest:
            raise TypeError("unknown protocol keys %r" % list(rest))
        (name, inp), = inputs.items()
        for what, arr in (("input", inp), ("tf", tf)):
            if isinstance(arr, np.ndarray) and (n_cells is None or arr.shape != (n_cells,)):
                raise ValueError("per-cell %s: expected a 1-D array of length n_cells" % what)
        runs.append(_make_run(p, name, inp, tf, bounds))
        names.append(name)
    return runs, names


def simulate_ensemble(p, Theta, protocol, *, SOC=None, opts=None, device=False, stream=None, max_points=None, outputs=None, YP=True, sens=None, initial_states=None):
    """Integrate an ensemble of independent cells on this process's GPU.

    Theta: [n_cells, n_theta] array in `p.θ_keys` order (numpy = host memory; torch CUDA tensor with device=True = already in HBM).
    protocol: list of run dicts (see make_protocol) shared by all cells.  SOC: scalar or [n_cells] initial SOC.
    outputs: "all" (or any state name) also returns every saved state vector as ens.Y_all [cell, point, state] -- 8 N bytes per point.
    YP = False: do not return YP of the final point (the reference's default: var_keep.YP is off unless :YP is among the outputs).
    sens = ["D_sp", "k_n", ...]: forward sensitivities with respect to these entries of θ next to the states (plh_integrate_sens): ens.dY_dtheta[cell, k, state] at the
    end of the protocol, ens.dV_dtheta[cell, k, point] at every saved point; the states and saved points are those of the call without `sens`.
    """
    n = Theta.shape[0]
    runs, names = make_protocol(p, protocol, n)
    o = opts or p.opts
    soc = p.opts.SOC if SOC is None else SOC
    if device:
        import torch
        soc0 = soc if hasattr(soc, "device") else torch.full((n,), float(soc), dtype=torch.float64, device=Theta.device)
    else:
        soc0 = np.full(n, float(soc)) if np.isscalar(soc) else np.asarray(soc, dtype=np.float64)
    Y0 = None
    if initial_states is not None:          # [n_cells, N.tot] (host array): every cell starts a NEW solution from its own state vector; SOC = calc_SOC of it unless given
        if device:
            raise ValueError("initial_states: host arrays (device = False)")
        Y0 = np.ascontiguousarray(initial_states, dtype=np.float64)
        if Y0.shape != (n, p.N.tot):
            raise ValueError("initial_states must be [n_cells, N.tot]")
        if SOC is None:
            # (with the theta the run is made with -- a sweep over c_max_n / the anode's stoichiometry window changes every cell's SOC of the same state vector)
            th = np.asarray(Theta if not hasattr(Theta, "cpu") else Theta.cpu(), dtype=np.float64)
            col = lambda k: th[:, p.θ_keys.index(k)] if k in p.θ_keys else np.full(n, p.θ[k])
            cs = p.ind["c_s_avg"]
            n_p = p.N.p * p.N.r_p if p.solid_diffusion == "Fickian" else p.N.p
            soc0 = np.ascontiguousarray((Y0[:, cs.start + n_p:cs.stop].mean(axis=1) / col("c_max_n") - col("θ_min_n")) / (col("θ_max_n") - col("θ_min_n")))
    bufs = _integrate(p, Theta, soc0, runs, o, Y_init=Y0, device=device, stream=stream, max_points=max_points,
                      keep_Y=_wants_states(p, o.outputs if outputs is None else outputs), keep_YP=YP, sens=sens)
    return EnsembleSolution(p, bufs, names)


def theta_matrix(p, n_cells, overrides=None):
    """[n_cells, n_theta] matrix of the model's current θ, with per-cell overrides {key: array[n_cells]}."""
    Th = np.tile(p.theta_vector(), (n_cells, 1))
    if overrides:
        for k, v in overrides.items():
            Th[:, p.θ_keys.index(k)] = v
    return Th


class HostPipeline:
    """Host-inclusive ensemble calls at kernel rate: `depth` in-flight calls over pinned host buffers (plh_host_alloc) and one HIP stream each
    (PLH_HOST_ASYNC): call k+1's parameter upload and kernel overlap call k's device-to-host copies.  This is SURVEY.md 8(d)'s measurement shape --
    parameters start in host memory, per-cell summaries and sampled outputs end in host memory -- without paying the copies serially.

        pipe = HostPipeline(p, n_cells, protocol, max_points=256)
        for k, Theta in enumerate(batches):
            slot = k % pipe.depth
            if k >= pipe.depth: consume(pipe.wait(slot))          # results of call k - depth
            pipe.submit(slot, Theta)
    """

    def __init__(self, p, n_cells, protocol, SOC=1.0, opts=None, max_points=256, depth=2, streams=None):
        import torch                                                   # streams only (plumbing)
        self.p, self.n, self.depth = p, int(n_cells), int(depth)
        self.runs, self.names = make_protocol(p, protocol, n_cells)
        self.arr = (cap.Run * len(self.runs))(*self.runs)
        self.opts = _opts_struct(opts or p.opts, p)
        self.mp = int(max_points)
        self._torch_streams = streams or [torch.cuda.Stream() for _ in range(self.depth)]
        self.streams = [s.cuda_stream for s in self._torch_streams]
        lib = p._lib
        N, P, nr = p.N.tot, len(p.θ_keys), len(self.runs)
        self._blocks = []

        def pinned(shape, dtype):
            dt = np.dtype(dtype); nbytes = int(np.prod(shape)) * dt.itemsize
            ptr = C.c_void_p()
            cap.check(lib, lib.plh_host_alloc(C.byref(ptr), nbytes), "plh_host_alloc")
            self._blocks.append(ptr)
            return np.frombuffer((C.c_char * nbytes).from_address(ptr.value), dtype=dt).reshape(shape)
        self.slots = []
        for _ in range(self.depth):
            b = dict(theta=pinned((self.n, P), np.float64), soc=pinned((self.n,), np.float64), t=pinned((self.n, self.mp), np.float64), V=pinned((self.n, self.mp), np.float64),
                     n_pts=pinned((self.n,), np.int32), run_info=pinned((self.n, nr), cap.RUN_INFO_DTYPE), counters=pinned((self.n,), cap.COUNTERS_DTYPE))
            b["soc"][:] = float(SOC)
            out = cap.Outputs()
            out.max_pts = self.mp
            out.t, out.V, out.n_pts = cap.ptr(b["t"]), cap.ptr(b["V"]), cap.ptr(b["n_pts"])
            out.run_info, out.counters = cap.ptr(b["run_info"]), cap.ptr(b["counters"])
            b["out"] = out
            self.slots.append(b)

    def submit(self, slot, Theta):
        b = self.slots[slot]
        np.copyto(b["theta"], Theta)                                   # the caller's parameters into the pinned block (what a Julia host would own directly)
        lib, h = self.p._lib, self.p._h
        cap.check(lib, lib.plh_integrate(h, self.n, cap.ptr(b["theta"]), cap.ptr(b["soc"]), None, None, len(self.runs), self.arr, C.byref(self.opts),
                                         C.byref(b["out"]), cap.PLH_HOST_ASYNC, self.streams[slot]), "plh_integrate")

    def wait(self, slot):
        cap.check(self.p._lib, self.p._lib.plh_synchronize(self.p._h, self.streams[slot]), "plh_synchronize")
        return self.slots[slot]

    def close(self):
        for s in range(self.depth):
            self.p._lib.plh_synchronize(self.p._h, self.streams[s])
        for ptr in self._blocks:
            self.p._lib.plh_host_free(ptr)
        self._blocks = []
        self.slots = []
