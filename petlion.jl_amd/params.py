"""Chemistry parameter tables, default bounds and default options -- the host-side data contract of petlion().

Restated from the reference's parameter files (values only): LCO src/params.jl:5-56, LiC6 58-117, system_LCO_LiC6 176-289.
Key names are the reference's Symbols (UTF-8), so `p.θ["ϵ_p"] = 0.485` reads like the reference's `p.θ[:ϵ_p] = 0.485`.
"""
from __future__ import annotations

import math
from collections import OrderedDict

F = 96485.3321233          # src/structures.jl:10
R = 8.31446261815324       # src/structures.jl:11


def theta_LCO():
    th = OrderedDict()
    # LCO cathode, src/params.jl:10-45
    th.update({"D_sp": 1e-14, "D_p": 7.5e-10, "k_p": 2.334e-11, "λ_MHC_p": 6.26e-20, "θ_min_p": 0.99174, "θ_max_p": 0.49550,
               "l_p": 80e-6, "σ_p": 100.0, "ϵ_p": 0.385, "ϵ_fp": 0.025, "brugg_p": 4.0, "c_max_p": 51554.0, "Rp_p": 2e-6,
               "λ_p": 2.1, "ρ_p": 2500.0, "Cp_p": 700.0, "Ea_D_sp": 5000.0, "Ea_k_p": 5000.0})
    # LiC6 anode, src/params.jl:61-110
    th.update({"D_sn": 3.9e-14, "D_n": 7.5e-10, "k_n": 5.0310e-11, "λ_MHC_n": 6.26e-20, "θ_max_n": 0.85510, "θ_min_n": 0.01429,
               "l_n": 88e-6, "σ_n": 100.0, "ϵ_n": 0.485, "ϵ_fn": 0.0326, "brugg_n": 4.0, "c_max_n": 30555.0, "Rp_n": 2e-6,
               "λ_n": 1.7, "ρ_n": 2500.0, "Cp_n": 700.0, "Ea_D_sn": 5000.0, "Ea_k_n": 5000.0,
               "R_SEI": 0.01, "M_n": 7.3e-4, "k_n_aging": 1.0, "i_0_jside": 1.5e-6, "Uref_s": 0.4, "w": 2.0})
    # system_LCO_LiC6, src/params.jl:179-226
    th.update({"D_s": 7.5e-10, "l_s": 25e-6, "l_a": 10e-6, "l_z": 10e-6, "σ_a": 3.55e7, "σ_z": 5.96e7, "ϵ_s": 0.724,
               "brugg_s": 4.0, "t₊": 0.364, "c_e₀": 1000.0, "T₀": 25 + 273.15, "T_amb": 25 + 273.15,
               "λ_s": 0.16, "λ_a": 237.0, "λ_z": 401.0, "ρ_s": 1100.0, "ρ_a": 2700.0, "ρ_z": 8940.0,
               "Cp_s": 700.0, "Cp_a": 897.0, "Cp_z": 385.0, "h_cell": 1.0})
    return th


def theta_NMC():
    """NMC cathode + LiC6_NMC anode + system_NMC_LiC6: reference src/params.jl:295-332, 334-367, 436-452.
    The SEI block (R_SEI ... w, rho_n) is NOT defined by the reference for this chemistry; aging=:SEI borrows the LiC6 values of
    src/params.jl:98-110 and rho_n = 2500 (params.jl:90) -- a build decision, stated in DESIGN.md."""
    th = OrderedDict()
    th.update({"D_sp": 2e-14, "k_p": 6.3066e-10, "θ_min_p": 0.955473, "θ_max_p": 0.359749, "l_p": 41.6e-6, "σ_p": 100.0, "ϵ_p": 0.3,
               "ϵ_fp": 0.12, "brugg_p": 1.5, "c_max_p": 51830.0, "Rp_p": 7.5e-6, "Ea_D_sp": 2.5e4, "Ea_k_p": 3e4})
    th.update({"D_sn": 1.5e-14, "k_n": 6.3466e-10, "θ_max_n": 0.790813, "θ_min_n": 0.001, "l_n": 48e-6, "σ_n": 100.0, "ϵ_n": 0.3,
               "ϵ_fn": 0.038, "brugg_n": 1.5, "c_max_n": 31080.0, "Rp_n": 10e-6, "Ea_D_sn": 4e4, "Ea_k_n": 3e4})
    th.update({"l_s": 25e-6, "ϵ_s": 0.4, "brugg_s": 1.5, "t₊": 0.38, "c_e₀": 1200.0, "T₀": 25 + 273.15, "T_amb": 25 + 273.15})
    th.update({"R_SEI": 0.01, "M_n": 7.3e-4, "k_n_aging": 1.0, "i_0_jside": 1.5e-6, "Uref_s": 0.4, "w": 2.0, "ρ_n": 2500.0})
    return th


def calc_I1C(th):
    """1C current density [A/m^2], reference src/physics_equations/auxiliary_states_and_coefficients.jl:632-647."""
    eps_sp = 1.0 - (th["ϵ_fp"] + th["ϵ_p"])
    eps_sn = 1.0 - (th["ϵ_fn"] + th["ϵ_n"])
    return (F / 3600.0) * min(eps_sp * th["l_p"] * th["c_max_p"] * (th["θ_min_p"] - th["θ_max_p"]),
                              eps_sn * th["l_n"] * th["c_max_n"] * (th["θ_max_n"] - th["θ_min_n"]))


class Bounds:
    """reference boundary_stop_conditions (src/structures.jl:237-250); NaN = bound disabled."""
    FIELDS = ("V_max", "V_min", "SOC_max", "SOC_min", "T_max", "c_s_n_max", "I_max", "I_min", "η_plating_min", "c_e_min", "dfilm_max")

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, math.nan)
        for k, v in kw.items():
            if k not in self.FIELDS:
                raise KeyError(k)
            setattr(self, k, float(v))

    def copy(self, **over):
        b = Bounds(**{f: getattr(self, f) for f in self.FIELDS})
        for k, v in over.items():
            if k not in self.FIELDS:
                raise KeyError(k)
            setattr(b, k, float(v))
        return b

    def __repr__(self):
        return "Bounds(" + ", ".join("%s=%g" % (f, getattr(self, f)) for f in self.FIELDS if not math.isnan(getattr(self, f))) + ")"


def bounds_LCO():
    """src/params.jl:233-252"""
    return Bounds(V_min=2.5, V_max=4.3, SOC_min=0.0, SOC_max=1.0, T_max=55 + 273.15)


def theta_LGM50():
    """NMC_LGM50 + LiC6_LGM50 + system_LGM50_NMC_LiC6 (Chen et al. 2020), reference src/params.jl:514-560, 576-625, 776-801"""
    th = OrderedDict()
    th.update({"D_sp": 4e-15, "k_p": 3.5445802224420315e-11, "λ_MHC_p": 0.0, "θ_min_p": 0.8395, "θ_max_p": 17038.0 / 63104.0, "l_p": 75.6e-6, "σ_p": 0.18, "ϵ_p": 0.335,
               "ϵ_fp": 0.0, "brugg_p": 1.5, "c_max_p": 63104.0, "Rp_p": 5.22e-06, "Ea_D_sp": 0.0, "Ea_k_p": 17800.0})
    th.update({"D_sn": 3.3e-14, "k_n": 6.716046737258585e-12, "λ_MHC_n": 0.0, "θ_max_n": 29866.0 / 33133, "θ_min_n": 0.0481727, "l_n": 85.2e-6, "σ_n": 215.0, "ϵ_n": 0.25,
               "ϵ_fn": 0.0, "brugg_n": 1.5, "c_max_n": 33133.0, "Rp_n": 5.86e-6, "Ea_D_sn": 3.03e4, "Ea_k_n": 35000.0})
    th.update({"D_e": 8.794e-11, "l_s": 12e-6, "ϵ_s": 0.47, "brugg_s": 1.5, "t₊": 0.2594, "c_e₀": 1000.0, "T₀": 25 + 273.15, "T_amb": 25 + 273.15})
    # heat equation (temperature = true is this chemistry's default in the reference): src/params.jl:531-533, 593-595, 779-800
    th.update({"λ_p": 2.1, "ρ_p": 3262.0, "Cp_p": 700.0, "λ_n": 1.7, "ρ_n": 1657.0, "Cp_n": 700.0, "l_a": 16e-6, "l_z": 12e-6, "σ_a": 36.914e6, "σ_z": 58.41e6,
               "λ_s": 0.16, "λ_a": 237.0, "λ_z": 401.0, "ρ_s": 397.0, "ρ_a": 2700.0, "ρ_z": 8960.0, "Cp_s": 700.0, "Cp_a": 897.0, "Cp_z": 385.0, "h_cell": 1.0})
    return th


def bounds_LGM50():
    """src/params.jl:803-813"""
    return Bounds(V_min=2.5, V_max=4.2, SOC_min=0.0, SOC_max=1.0, T_max=55 + 273.15)


def bounds_NMC():
    """src/params.jl:456-475"""
    return Bounds(V_min=2.8, V_max=4.2, SOC_min=0.0, SOC_max=1.0)


class Opts:
    """reference options_simulation (src/structures.jl:266-285) with the LCO defaults of src/params.jl:255-283."""
    def __init__(self, SOC=1.0):
        self.SOC = SOC
        self.outputs = ("t", "V")
        self.abstol = 1e-6
        self.reltol = 1e-3
        self.abstol_init = None     # None -> abstol (model_evaluation.jl:21)
        self.reltol_init = None     # None -> reltol (model_evaluation.jl:22)
        self.maxiters = 10_000
        self.check_bounds = True
        self.reinit = True
        self.verbose = False
        self.interp_final = True
        self.tstops = []
        self.tdiscon = []           # known discontinuities of a time-dependent input (run-local times), reference opts.tdiscon
        self.interp_bc = "interpolate"
        self.save_start = False     # warm start of the consistent initialisation from the model's cache (reference opts.save_start, model_evaluation.jl:384-411)
        # build-specific knobs (not in the reference)
        self.max_order = 5
        self.jac_every_step = False
        self.init_step = 0.0        # 0 = IDA's automatic initial step; > 0 = IDASetInitStep
        self.refine = 0             # n > 0: n steps of iterative refinement of every linear solve (parity mode, plh_opts.refine)
        self.yp_alg_zero = False    # True: start the integrator with YP_alg = 0 (the package version of the reference's example notebooks; plh_opts.yp_alg_zero)
        self.max_points = 2048      # capacity of the per-cell output buffers
        # reference opts.stop_function (src/structures.jl:283; src/checks.jl:26: called after the built-in checks at every accepted step).  Here a closure g(t, Y, YP, p)
        # of the same kind as an input closure: the run ends when g > 0 (exit flag 12, "Stop function"), back-interpolated like a built-in bound.  None: no hook.
        self.stop_function = None


# reference exit strings, src/checks.jl:6-217
EXIT_REASONS = {
    0: "Final time reached", 1: "Below min. voltage", 2: "Above max. voltage", 3: "Below min. SOC", 4: "Above max. SOC",
    5: "Above max. temperature", 6: "Above max. c_s_n", 7: "Above max. C-rate", 8: "Below min. C-rate", 9: "Below min. c_e",
    10: "Above max. film growth rate", 11: "Below min. η_plating", 12: "Stop function",
    -1: "(not run)", -11: "Could not initialize DAE", -12: "Model failed to converge", -13: "Reached max iterations",
    -14: "Output buffer full",
}
