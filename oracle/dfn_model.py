"""
ORACLE (test infrastructure, NOT product code) -- equation restatement of the PETLION DFN/P2D model.

This file restates, formula by formula, the residual F(t, Y, YP, theta) of the reference's
`residuals_PET!` (reference src/physics_equations/scalar_residual.jl:27-66) for the configurations of
SURVEY.md section 8: LCO/LiC6 or NMC/LiC6, Fickian finite-difference solid diffusion, optional 1D
temperature, optional SEI aging.  It is written generically over the scalar type, so the same code is
 * evaluated with Python floats (checks against the notebook known-answers),
 * evaluated with complex numbers (complex-step Jacobian to check the symbolic one),
 * traced with sympy symbols (oracle/codegen.py differentiates it and emits straight-line C --
   the same pipeline shape as the reference's Symbolics -> sparsejacobian -> build_function,
   reference src/generate_functions.jl:102-164, 289-307).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything in oracle/.

Parity status: the reference cannot run here (no Julia, no SUNDIALS/KLU); this restatement is pinned against
the reference's notebook outputs (tests/golden/notebook_kats.json): I1C bit-exact, V(t=0) to 1e-10, end-of-run
summaries to the reference's own reltol (1e-3).  At the IDA/KLU boundary parity is UNPINNED.

State layout (0-based here; reference src/external.jl:275-365, src/outputs.jl:78-94):
  Y = [ c_e (Np+Ns+Nn) | c_s_avg (Np*Nrp + Nn*Nrn) | T (Na+Np+Ns+Nn+Nz, if temperature) | film (Nn, SEI) |
        SOH (1, SEI) || j (Np+Nn) | Phi_e (Np+Ns+Nn) | Phi_s (Np+Nn) | j_s (Nn, SEI) | I (1) ]
"""
from __future__ import annotations

import math
from collections import OrderedDict

F_CONST = 96485.3321233          # reference src/structures.jl:10
R_CONST = 8.31446261815324       # reference src/structures.jl:11
T_REF = 298.15                   # 25 + 273.15, reference src/physics_equations/custom_functions.jl:24,125


# ----------------------------------------------------------------------------------------------------------------
# scalar-type-generic math
# ----------------------------------------------------------------------------------------------------------------
class FloatOps:
    """float / complex evaluation (complex: analytic continuation for the complex-step check)."""
    @staticmethod
    def _cm(x):
        import cmath
        return cmath if isinstance(x, complex) else math
    def sqrt(self, x): return self._cm(x).sqrt(x)
    def exp(self, x): return self._cm(x).exp(x)
    def sinh(self, x): return self._cm(x).sinh(x)
    def atan(self, x): return self._cm(x).atan(x)
    def tanh(self, x): return self._cm(x).tanh(x)
    def log(self, x): return self._cm(x).log(x)
    def erf(self, x):
        if isinstance(x, complex):       # first-order continuation (complex-step checks only)
            return math.erf(x.real) + 1j * x.imag * 2.0 / math.sqrt(math.pi) * math.exp(-x.real ** 2)
        return math.erf(x)
    def pow(self, x, y):
        return x ** y
    def relu(self, x, minval=0.0):   # max(minval, x)  (sqrt_ReLU argument, custom_functions.jl:210)
        xr = x.real if isinstance(x, complex) else x
        return x if xr > minval else (minval + 0 * x)
    def abs(self, x):
        xr = x.real if isinstance(x, complex) else x
        return x if xr >= 0 else -x
    def where_eq(self, a, b, x, y):  # ifelse(a == b, x, y)
        ar = a.real if isinstance(a, complex) else a
        return x if ar == b else y
    def where_gt(self, a, b, x, y):  # ifelse(a > b, x, y)
        ar = a.real if isinstance(a, complex) else a
        return x if ar > b else y
    def const(self, v): return v
    def aux(self, name, expr): return expr         # (symbolic tracing names this intermediate; plain evaluation just uses its value)


class SymOps:
    """sympy tracing."""
    def __init__(self):
        import sympy
        self.sp = sympy
    def sqrt(self, x): return self.sp.sqrt(x)
    def exp(self, x): return self.sp.exp(x)
    def sinh(self, x): return self.sp.sinh(x)
    def atan(self, x): return self.sp.atan(x)
    def tanh(self, x): return self.sp.tanh(x)
    def log(self, x): return self.sp.log(x)
    def erf(self, x): return self.sp.erf(x)
    def pow(self, x, y): return self.sp.Pow(x, y)
    def relu(self, x, minval=0.0):
        return self.sp.Piecewise((x, x > minval), (self.sp.Float(minval), True))
    def abs(self, x):
        return self.sp.Piecewise((x, x >= 0), (-x, True))
    def where_eq(self, a, b, x, y):
        return self.sp.Piecewise((x, self.sp.Eq(a, b)), (y, True))
    def where_gt(self, a, b, x, y):
        return self.sp.Piecewise((x, a > b), (y, True))
    def const(self, v): return self.sp.Float(v)
    def aux(self, name, expr):
        """name an intermediate (the surface concentration of the quadratic / polynomial particle models): the residual is traced in terms of the new
        symbol, codegen applies the chain rule -- differentiating the OCV polynomials through a nested expression is what makes sympy crawl"""
        if not hasattr(self, "aux_defs"):
            self.aux_defs = {}
        sym = self.sp.Symbol(name, real=True)
        self.aux_defs[sym] = self.sp.sympify(expr)
        return sym


# ----------------------------------------------------------------------------------------------------------------
# parameters (reference src/params.jl)
# ----------------------------------------------------------------------------------------------------------------
def theta_LCO():
    """LCO cathode + LiC6 anode + system parameters: reference src/params.jl:5-56, 58-117, 176-226."""
    th = OrderedDict()
    # LCO (params.jl:10-45)
    th["D_sp"] = 1e-14; th["D_p"] = 7.5e-10; th["k_p"] = 2.334e-11; th["λ_MHC_p"] = 6.26e-20
    th["θ_min_p"] = 0.99174; th["θ_max_p"] = 0.49550; th["l_p"] = 80e-6; th["σ_p"] = 100.0
    th["ϵ_p"] = 0.385; th["ϵ_fp"] = 0.025; th["brugg_p"] = 4.0; th["c_max_p"] = 51554.0; th["Rp_p"] = 2e-6
    th["λ_p"] = 2.1; th["ρ_p"] = 2500.0; th["Cp_p"] = 700.0; th["Ea_D_sp"] = 5000.0; th["Ea_k_p"] = 5000.0
    # LiC6 (params.jl:61-110)
    th["D_sn"] = 3.9e-14; th["D_n"] = 7.5e-10; th["k_n"] = 5.0310e-11; th["λ_MHC_n"] = 6.26e-20
    th["θ_max_n"] = 0.85510; th["θ_min_n"] = 0.01429; th["l_n"] = 88e-6; th["σ_n"] = 100.0
    th["ϵ_n"] = 0.485; th["ϵ_fn"] = 0.0326; th["brugg_n"] = 4.0; th["c_max_n"] = 30555.0; th["Rp_n"] = 2e-6
    th["λ_n"] = 1.7; th["ρ_n"] = 2500.0; th["Cp_n"] = 700.0; th["Ea_D_sn"] = 5000.0; th["Ea_k_n"] = 5000.0
    th["R_SEI"] = 0.01; th["M_n"] = 7.3e-4; th["k_n_aging"] = 1.0; th["i_0_jside"] = 1.5e-6
    th["Uref_s"] = 0.4; th["w"] = 2.0
    # system (params.jl:179-226)
    th["D_s"] = 7.5e-10; th["l_s"] = 25e-6; th["l_a"] = 10e-6; th["l_z"] = 10e-6
    th["σ_a"] = 3.55e7; th["σ_z"] = 5.96e7; th["ϵ_s"] = 0.724; th["brugg_s"] = 4.0; th["t₊"] = 0.364
    th["c_e₀"] = 1000.0; th["T₀"] = 25 + 273.15; th["T_amb"] = 25 + 273.15
    th["λ_s"] = 0.16; th["λ_a"] = 237.0; th["λ_z"] = 401.0
    th["ρ_s"] = 1100.0; th["ρ_a"] = 2700.0; th["ρ_z"] = 8940.0
    th["Cp_s"] = 700.0; th["Cp_a"] = 897.0; th["Cp_z"] = 385.0; th["h_cell"] = 1.0
    return th


def theta_NMC():
    """NMC cathode + LiC6_NMC anode + system: reference src/params.jl:295-332, 334-367, 436-452.
    The reference NMC chemistry defines no SEI parameters: config C5 borrows the LiC6 values of
    params.jl:98-110 and rho_n = 2500 (params.jl:90) -- a build decision, stated in DESIGN.md."""
    th = OrderedDict()
    th["D_sp"] = 2e-14; th["k_p"] = 6.3066e-10; th["θ_min_p"] = 0.955473; th["θ_max_p"] = 0.359749
    th["l_p"] = 41.6e-6; th["σ_p"] = 100.0; th["ϵ_p"] = 0.3; th["ϵ_fp"] = 0.12; th["brugg_p"] = 1.5
    th["c_max_p"] = 51830.0; th["Rp_p"] = 7.5e-6; th["Ea_D_sp"] = 2.5e4; th["Ea_k_p"] = 3e4
    th["D_sn"] = 1.5e-14; th["k_n"] = 6.3466e-10; th["θ_max_n"] = 0.790813; th["θ_min_n"] = 0.001
    th["l_n"] = 48e-6; th["σ_n"] = 100.0; th["ϵ_n"] = 0.3; th["ϵ_fn"] = 0.038; th["brugg_n"] = 1.5
    th["c_max_n"] = 31080.0; th["Rp_n"] = 10e-6; th["Ea_D_sn"] = 4e4; th["Ea_k_n"] = 3e4
    th["l_s"] = 25e-6; th["ϵ_s"] = 0.4; th["brugg_s"] = 1.5; th["t₊"] = 0.38
    th["c_e₀"] = 1200.0; th["T₀"] = 25 + 273.15; th["T_amb"] = 25 + 273.15
    # borrowed for aging=:SEI (not defined by the reference for NMC)
    th["R_SEI"] = 0.01; th["M_n"] = 7.3e-4; th["k_n_aging"] = 1.0; th["i_0_jside"] = 1.5e-6
    th["Uref_s"] = 0.4; th["w"] = 2.0; th["ρ_n"] = 2500.0
    return th


def theta_LGM50():
    """NMC_LGM50 cathode + LiC6_LGM50 anode + system_LGM50_NMC_LiC6 (Chen et al. 2020): reference src/params.jl:514-560, 576-625, 776-801.  Only the entries the
    isothermal Fickian model reads matter here; the thermal ones are listed for completeness."""
    th = OrderedDict()
    th["D_sp"] = 4e-15; th["k_p"] = 3.5445802224420315e-11; th["λ_MHC_p"] = 0.0; th["θ_min_p"] = 0.8395; th["θ_max_p"] = 17038.0 / 63104.0
    th["l_p"] = 75.6e-6; th["σ_p"] = 0.18; th["ϵ_p"] = 0.335; th["ϵ_fp"] = 0.0; th["brugg_p"] = 1.5
    th["c_max_p"] = 63104.0; th["Rp_p"] = 5.22e-06; th["Ea_D_sp"] = 0.0; th["Ea_k_p"] = 17800.0
    th["D_sn"] = 3.3e-14; th["k_n"] = 6.716046737258585e-12; th["λ_MHC_n"] = 0.0; th["θ_max_n"] = 29866.0 / 33133; th["θ_min_n"] = 0.0481727
    th["l_n"] = 85.2e-6; th["σ_n"] = 215.0; th["ϵ_n"] = 0.25; th["ϵ_fn"] = 0.0; th["brugg_n"] = 1.5
    th["c_max_n"] = 33133.0; th["Rp_n"] = 5.86e-6; th["Ea_D_sn"] = 3.03e4; th["Ea_k_n"] = 35000.0
    th["D_e"] = 8.794e-11; th["l_s"] = 12e-6; th["ϵ_s"] = 0.47; th["brugg_s"] = 1.5; th["t₊"] = 0.2594
    th["c_e₀"] = 1000.0; th["T₀"] = 25 + 273.15; th["T_amb"] = 25 + 273.15
    # heat equation (temperature = true is the reference default of this chemistry): params.jl:531-533 (cathode), 593-595 (anode), 779-800 (system)
    th["λ_p"] = 2.1; th["ρ_p"] = 3262.0; th["Cp_p"] = 700.0; th["λ_n"] = 1.7; th["ρ_n"] = 1657.0; th["Cp_n"] = 700.0
    th["l_a"] = 16e-6; th["l_z"] = 12e-6; th["σ_a"] = 36.914e6; th["σ_z"] = 58.41e6
    th["λ_s"] = 0.16; th["λ_a"] = 237.0; th["λ_z"] = 401.0; th["ρ_s"] = 397.0; th["ρ_a"] = 2700.0; th["ρ_z"] = 8960.0
    th["Cp_s"] = 700.0; th["Cp_a"] = 897.0; th["Cp_z"] = 385.0; th["h_cell"] = 1.0
    return th


BOUNDS_DEFAULT = {
    # reference src/params.jl:233-252 (LCO) and 456-475 (NMC); NaN = disabled
    "LCO": dict(V_min=2.5, V_max=4.3, SOC_min=0.0, SOC_max=1.0, T_max=55 + 273.15, c_s_n_max=math.nan,
                I_max=math.nan, I_min=math.nan, η_plating_min=math.nan, c_e_min=math.nan, dfilm_max=math.nan),
    "NMC": dict(V_min=2.8, V_max=4.2, SOC_min=0.0, SOC_max=1.0, T_max=math.nan, c_s_n_max=math.nan,
                I_max=math.nan, I_min=math.nan, η_plating_min=math.nan, c_e_min=math.nan, dfilm_max=math.nan),
    "LGM50": dict(V_min=2.5, V_max=4.2, SOC_min=0.0, SOC_max=1.0, T_max=55 + 273.15, c_s_n_max=math.nan,      # params.jl:803-813
                  I_max=math.nan, I_min=math.nan, η_plating_min=math.nan, c_e_min=math.nan, dfilm_max=math.nan),
}


def calc_I1C(th, ops=None):
    """reference src/physics_equations/auxiliary_states_and_coefficients.jl:632-647"""
    eps_sp = 1.0 - (th["ϵ_fp"] + th["ϵ_p"])
    eps_sn = 1.0 - (th["ϵ_fn"] + th["ϵ_n"])
    a = eps_sp * th["l_p"] * th["c_max_p"] * (th["θ_min_p"] - th["θ_max_p"])
    b = eps_sn * th["l_n"] * th["c_max_n"] * (th["θ_max_n"] - th["θ_min_n"])
    if ops is not None and hasattr(ops, "sp"):
        return (F_CONST / 3600.0) * ops.sp.Min(a, b)
    return (F_CONST / 3600.0) * (a if a < b else b)


# ----------------------------------------------------------------------------------------------------------------
# finite-difference matrices for the particle (reference src/physics_equations/numerical_tools.jl:8-87)
# ----------------------------------------------------------------------------------------------------------------
def fd_first_order(n):
    dx = 1.0 / (n - 1)
    first = [[-109584.0, 322560, -564480, 752640, -705600, 451584, -188160, 46080, -5040],
             [-5040.0, -64224, 141120, -141120, 117600, -70560, 28224, -6720, 720],
             [720.0, -11520, -38304, 80640, -50400, 26880, -10080, 2304, -240],
             [-240.0, 2880, -20160, -18144, 50400, -20160, 6720, -1440, 144]]
    ith = [144.0, -1536, 8064, -32256, 0, 32256, -8064, 1536, -144]
    last = [[-144.0, 1440, -6720, 20160, -50400, 18144, 20160, -2880, 240],
            [240.0, -2304, 10080, -26880, 50400, -80640, 38304, 11520, -720],
            [-720.0, 6720, -28224, 70560, -117600, 141120, -141120, 64224, 5040],
            [5040.0, -46080, 188160, -451584, 705600, -752640, 564480, -322560, 109584]]
    M = [[0.0] * n for _ in range(n)]
    for r in range(4):
        for c in range(9):
            M[r][c] = float(first[r][c])
            M[n - 4 + r][n - 9 + c] = float(last[r][c])
    for k, i in enumerate(range(4, n - 4)):
        for c in range(9):
            M[i][k + c] = float(ith[c])
    return M, 1.0 / (40320 * dx), dx


def fd_second_order(n):
    dx = 1.0 / (n - 1)
    M = [[0.0] * n for _ in range(n)]
    M[0][:6] = [-415 / 6, 96.0, -36.0, 32 / 3, -3 / 2, 0.0]
    M[1][:6] = [10.0, -15.0, -4.0, 14.0, -6.0, 1.0]
    for k, i in enumerate(range(2, n - 2)):
        M[i][k:k + 5] = [-1.0, 16.0, -30.0, 16.0, -1.0]
    M[n - 2][n - 6:] = [1.0, -6.0, 14.0, -4.0, -15.0, 10.0]
    M[n - 1][n - 6:] = [0.0, -3 / 2, 32 / 3, -36.0, 96.0, -415 / 6]
    return M, 1.0 / (12 * dx * dx), dx


# ----------------------------------------------------------------------------------------------------------------
# model description
# ----------------------------------------------------------------------------------------------------------------
class Layout:
    """0-based index ranges.  reference src/external.jl:275-365."""
    def __init__(self, Np=10, Ns=10, Nn=10, Na=10, Nz=10, Nrp=10, Nrn=10, temperature=False, aging=False, solid_diffusion="Fickian"):
        if solid_diffusion != "Fickian":               # one volume-averaged concentration per particle (c_s_indices, aux...jl:696-703)
            Nrp = Nrn = 1
        self.Np, self.Ns, self.Nn, self.Na, self.Nz, self.Nrp, self.Nrn = Np, Ns, Nn, Na, Nz, Nrp, Nrn
        self.temperature, self.aging = bool(temperature), bool(aging)
        self.solid_diffusion = solid_diffusion
        o = 0
        self.c_e = (o, o + Np + Ns + Nn); o = self.c_e[1]
        self.c_s = (o, o + Np * Nrp + Nn * Nrn); o = self.c_s[1]
        self.T = None
        if temperature:
            self.T = (o, o + Na + Np + Ns + Nn + Nz); o = self.T[1]
        self.film = self.SOH = self.j_s = None
        if aging:
            self.film = (o, o + Nn); o = self.film[1]
            self.SOH = (o, o + 1); o = self.SOH[1]
        self.Q = None
        if solid_diffusion == "polynomial":            # Q is active only for the polynomial approximation (states_definition.jl:60-67)
            self.Q = (o, o + Np + Nn); o = self.Q[1]
        self.N_diff = o
        self.j = (o, o + Np + Nn); o = self.j[1]
        self.Phi_e = (o, o + Np + Ns + Nn); o = self.Phi_e[1]
        self.Phi_s = (o, o + Np + Nn); o = self.Phi_s[1]
        if aging:
            self.j_s = (o, o + Nn); o = self.j_s[1]
        self.I = o; o += 1
        self.N = o
        self.N_alg = self.N - self.N_diff


class Model:
    def __init__(self, cathode="LCO", temperature=False, aging=False, solid_diffusion="Fickian", thermodynamic_factor="linear", rxn="BV", t_conduction="matrix", phi_s_form="matrix", **Nkw):
        """solid_diffusion: "Fickian" (finite difference), "quadratic", "polynomial" (params.jl:140); thermodynamic_factor: "linear" (nu = 1) or
        "nonlinear" (custom_functions.jl:177-203); rxn: "BV" or "MHC" for both electrodes (custom_functions.jl:212-298)"""
        assert solid_diffusion in ("Fickian", "quadratic", "polynomial") and thermodynamic_factor in ("linear", "nonlinear") and rxn in ("BV", "MHC")
        self.cathode = cathode
        # t_conduction: how the heat-conduction stencil of residuals_T! is EVALUATED -- "matrix": coefficients times temperatures, summed (what the reference's A_T * T generates:
        # three terms of 6e6 K/s cancelling to 0.1 K/s, 1e-9 K/s of rounding per row); "difference": the same stencil on the differences of neighbouring temperatures (named
        # intermediates dT_k = T[k+1] - T[k], exact to their own last bit).  Algebraically identical; the second form is what the device evaluates since r03, and the variant
        # lco_thermal_tdiff exists so that the dT = :hold leg -- whose control row sums all fifty rows -- can be compared at tight tolerances.
        assert t_conduction in ("matrix", "difference")
        self.t_conduction = t_conduction
        # phi_s_form: how the [1, -2, 1] Laplacian of residuals_Φ_s! (residuals.jl:656-703: block_tridiag(N) * Φ_s .- f) is EVALUATED.  "matrix": Φ_s[i-1] - 2 Φ_s[i] + Φ_s[i+1] - f_i
        # as ONE sum, which the code generator orders its own way -- the generated C adds the source term f_i ~ 1e-6 V to a potential of ~4 V BEFORE the Laplacian cancels, so the
        # row comes out quantised at ulp(Φ_s) = 8.9e-16 V (cathode).  J^-1 turns that into ~5e-11 V of common-mode noise in Φ_e / Φ_s and 1e-9 in I: 4e-8 ... 4e-7 in the weighted
        # norm, i.e. ABOVE the local error (1e-8) of the first steps of a :hold leg, which restarts at h = 1e-3 s -- IDA's start-up order selection then reads noise (r05,
        # DESIGN.md 5; measured with ORC_TRACE_EE).  "difference": the same stencil on named differences dPs_k = Φ_s[k+1] - Φ_s[k] (exact), like t_conduction = "difference".
        # Algebraically identical; which of the two orders the reference's own generated code has is not knowable here (Symbolics' term order; no Julia).
        assert phi_s_form in ("matrix", "difference")
        self.phi_s_form = phi_s_form
        self.solid_diffusion, self.thermodynamic_factor, self.rxn = solid_diffusion, thermodynamic_factor, rxn
        self.lay = Layout(temperature=temperature, aging=aging, solid_diffusion=solid_diffusion, **Nkw)
        self.theta = {"LCO": theta_LCO, "NMC": theta_NMC, "LGM50": theta_LGM50}[cathode]()
        if cathode == "LGM50" and aging:
            raise ValueError("LGM50 is built without aging (its SEI / stress parameters belong to aging models the reference marks unused)")
        if cathode == "NMC" and temperature:
            raise ValueError("the reference NMC chemistry defines no thermal parameters (params.jl:295-367)")
        self.bounds = dict(BOUNDS_DEFAULT[cathode])

    @property
    def name(self):
        return "%s_%s%s%s%s%s" % (self.cathode.lower(), "thermal" if self.lay.temperature else "iso", "_sei" if self.lay.aging else "",
                                  {"Fickian": "", "quadratic": "_quad", "polynomial": "_poly"}[self.solid_diffusion],
                                  "_nu" if self.thermodynamic_factor == "nonlinear" else "", "_mhc" if self.rxn == "MHC" else "")


def harmonic_mean(beta, x1, x2):
    """reference src/physics_equations/numerical_tools.jl:156"""
    return x1 * x2 / (beta * x2 + (1.0 - beta) * x1)


def interpolate_electrolyte_grid(Kp, Ks, Kn, th, lay):
    """reference src/physics_equations/numerical_tools.jl:106-154 (harmonic means on CV edges; last n edge = 0)."""
    dxp, dxs, dxn = 1.0 / lay.Np, 1.0 / lay.Ns, 1.0 / lay.Nn
    def medio(K):
        return [K[i] * K[i + 1] / (0.5 * K[i + 1] + 0.5 * K[i]) for i in range(len(K) - 1)]
    def beta_ij(dxi, li, dxj, lj):
        return dxi * li / 2 / (dxj * lj / 2 + dxi * li / 2)
    def iface(b, Ki, Kj):
        return Ki[-1] * Kj[0] / (b * Kj[0] + (1 - b) * Ki[-1])
    b_ps = beta_ij(dxp, th["l_p"], dxs, th["l_s"])
    b_sn = beta_ij(dxs, th["l_s"], dxn, th["l_n"])
    Kp_e = medio(Kp) + [iface(b_ps, Kp, Ks)]
    Ks_e = medio(Ks) + [iface(b_sn, Ks, Kn)]
    Kn_e = medio(Kn) + [0.0]
    return Kp_e, Ks_e, Kn_e


def interpolate_edges(c, th, lay):
    """reference numerical_tools.jl:158-189 (used for c_e and for T restricted to p|s|n)."""
    Np, Ns, Nn = lay.Np, lay.Ns, lay.Nn
    dxp, dxs, dxn = 1.0 / Np, 1.0 / Ns, 1.0 / Nn
    cp = [harmonic_mean(0.5, c[i], c[i + 1]) for i in range(Np - 1)]
    b_ps = dxp * th["l_p"] / 2 / (dxp * th["l_p"] / 2 + dxs * th["l_s"] / 2)
    cp.append(harmonic_mean(b_ps, c[Np - 1], c[Np]))
    cs = [harmonic_mean(0.5, c[Np + i], c[Np + i + 1]) for i in range(Ns - 1)]
    b_sn = dxs * th["l_s"] / 2 / (dxn * th["l_n"] / 2 + dxs * th["l_s"] / 2)
    cs.append(harmonic_mean(b_sn, c[Np + Ns - 1], c[Np + Ns]))
    cn = [harmonic_mean(0.5, c[Np + Ns + i], c[Np + Ns + i + 1]) for i in range(Nn - 1)]
    return cp, cs, cn


def edge_fluxes(c, th, lay):
    """reference numerical_tools.jl:193-215"""
    Np, Ns, Nn = lay.Np, lay.Ns, lay.Nn
    dxp, dxs, dxn = 1.0 / Np, 1.0 / Ns, 1.0 / Nn
    fp = [(c[i + 1] - c[i]) / (dxp * th["l_p"]) for i in range(Np - 1)]
    fp.append((c[Np] - c[Np - 1]) / (dxp * th["l_p"] / 2 + dxs * th["l_s"] / 2))
    fs = [(c[Np + i + 1] - c[Np + i]) / (dxs * th["l_s"]) for i in range(Ns - 1)]
    fs.append((c[Np + Ns] - c[Np + Ns - 1]) / (dxn * th["l_n"] / 2 + dxs * th["l_s"] / 2))
    fn = [(c[Np + Ns + i + 1] - c[Np + Ns + i]) / (dxn * th["l_n"]) for i in range(Nn - 1)]
    return fp, fs, fn


# ---- closures (reference src/physics_equations/custom_functions.jl) ----------------------------------------------
def arrhenius(ops, Ea, T, thermal):
    """temperature_switch(T == T_ref, 1, exp(-Ea/R (1/T - 1/T_ref))): custom_functions.jl:1, 16-31, 44-57.
    With temperature=true the switch always takes the exp branch."""
    e = ops.exp(-(Ea / R_CONST) * (1.0 / T - 1.0 / T_REF))
    return e if thermal else ops.where_eq(T, T_REF, 1.0 + 0 * e, e)


def K_eff_fn(c, T):
    """custom_functions.jl:96"""
    return 1e-4 * c * ((-10.5 + 0.668 * 1e-3 * c + 0.494 * 1e-6 * c ** 2)
                       + (0.074 - 1.78 * 1e-5 * c - 8.86 * 1e-10 * c ** 2) * T
                       + (-6.96 * 1e-5 + 2.8 * 1e-8 * c) * T ** 2) ** 2


def D_eff_fn(ops, c, T):
    """custom_functions.jl:83 (NMC system default, params.jl:407)"""
    return 1e-4 * ops.pow(ops.const(10.0), (-4.43 - 54.0 / (T - 229 - 5e-3 * c) - 0.22e-3 * c))


def OCV_LCO(ops, x, T, thermal):
    """custom_functions.jl:123-136"""
    U = ((-4.656 + 88.669 * x ** 2 - 401.119 * x ** 4 + 342.909 * x ** 6 - 462.471 * x ** 8 + 433.434 * x ** 10)
         / (-1 + 18.933 * x ** 2 - 79.532 * x ** 4 + 37.311 * x ** 6 - 73.083 * x ** 8 + 95.96 * x ** 10))
    dUdT = (-0.001 * (0.199521039 - 0.928373822 * x + 1.364550689000003 * x ** 2 - 0.6115448939999998 * x ** 3)
            / (1 - 5.661479886999997 * x + 11.47636191 * x ** 2 - 9.82431213599998 * x ** 3 + 3.048755063 * x ** 4))
    corr = dUdT * (T - T_REF)
    U = U + (corr if thermal else ops.where_eq(T, T_REF, 0 * corr, corr))
    return U, dUdT


def OCV_LiC6(ops, x, T, thermal):
    """custom_functions.jl:139-152"""
    sq0 = ops.sqrt(ops.relu(x, 0.0))
    sq1 = ops.sqrt(ops.relu(x, 1e-4))
    U = (0.7222 + 0.1387 * x + 0.029 * sq0 - 0.0172 / x + 0.0019 / (sq1 * x)
         + 0.2808 * ops.exp(0.9 - 15 * x) - 0.7984 * ops.exp(0.4465 * x - 0.4108))
    num = 0.001 * (0.005269056 + 3.299265709 * x - 91.79325798 * x ** 2 + 1004.911008 * x ** 3 - 5812.278127 * x ** 4
                   + 19329.7549 * x ** 5 - 37147.8947 * x ** 6 + 38379.18127 * x ** 7 - 16515.05308 * x ** 8)
    den = (1 - 48.09287227 * x + 1017.234804 * x ** 2 - 10481.80419 * x ** 3 + 59431.3 * x ** 4 - 195881.6488 * x ** 5
           + 374577.3152 * x ** 6 - 385821.1607 * x ** 7 + 165705.8597 * x ** 8)
    dUdT = num / den
    corr = dUdT * (T - T_REF)
    U = U + (corr if thermal else ops.where_eq(T, T_REF, 0 * corr, corr))
    return U, dUdT


def OCV_NMC(ops, x, T, thermal):
    """custom_functions.jl:154-162"""
    return -10.72 * x ** 4 + 23.88 * x ** 3 - 16.77 * x ** 2 + 2.595 * x + 4.563, 0.0


def OCV_LiC6_with_NMC(ops, x, T, thermal):
    """custom_functions.jl:164-174"""
    U = (0.1493 + 0.8493 * ops.exp(-61.79 * x) + 0.3824 * ops.exp(-665.8 * x) - ops.exp(39.42 * x - 41.92)
         - 0.03131 * ops.atan(25.59 * x - 4.099) - 0.009434 * ops.atan(32.49 * x - 15.74))
    return U, 0.0


def OCV_NMC_LGM50(ops, x, T, thermal):
    """params.jl:563-572 (dU/dT = 0)"""
    return (-0.8090 * x + 4.4875 - 0.0428 * ops.tanh(18.5138 * (x - 0.5542)) - 17.7326 * ops.tanh(15.7890 * (x - 0.3117))
            + 17.5842 * ops.tanh(15.9308 * (x - 0.3120))), 0.0


def OCV_LiC6_LGM50(ops, x, T, thermal):
    """params.jl:627-636 (dU/dT = 0)"""
    return (1.9793 * ops.exp(-39.3631 * x) + 0.15561 - 0.0909 * ops.tanh(29.8538 * (x - 0.1234)) - 0.04478 * ops.tanh(14.9159 * (x - 0.2769))
            - 0.0205 * ops.tanh(30.4444 * (x - 0.6103)) - 0.09259 * ops.tanh(17.08 * (x - 1))), 0.0


def D_eff_LGM50_fn(D_e, c):
    """params.jl:646"""
    return D_e * ((c / 1000) ** 2 - 4.516715942688196 * (c / 1000) + 5.5287696156470325)


def K_eff_LGM50_fn(ops, c):
    """params.jl:660"""
    return 0.1297 * (c / 1000) ** 3 - 2.51 * ops.pow(c / 1000, 1.5) + 3.329 * (c / 1000)


def rxn_BV(ops, c_s_star, c_e, T, eta, k, c_max):
    """custom_functions.jl:212-231 (alpha = 0.5 branch, sqrt = sqrt_ReLU)"""
    return 2.0 * k * ops.sqrt(ops.relu(c_e * c_s_star * (c_max - c_s_star), 0.0)) * ops.sinh(0.5 * F_CONST * eta / (R_CONST * T))


def rxn_MHC(ops, c_s_star, c_e, T, eta, k, lam, c_max, c_e0):
    """custom_functions.jl:241-298, the alpha = 0.5 branch that is taken (Zeng, Smith, Bai, Bazant 2014): Marcus-Hush-Chidsey kinetics in the uniformly
    valid approximation; eta_f = F eta/(R T) + log_ReLU(c_e/c_e0 / (c_s*/c_max); minval = 1e-4)"""
    eta_hat = eta * (F_CONST / (R_CONST * T))
    theta_i = c_s_star / c_max
    ce_hat = c_e / c_e0
    eta_f = eta_hat + ops.log(ops.relu(ce_hat / theta_i, 1e-4))
    a = 1.0 + ops.sqrt(lam)
    k0 = k / ((1.0 - ops.erf((lam - ops.sqrt(a)) / (2.0 * ops.sqrt(lam)))) / 2.0)
    coeff = k0 * (1.0 - ops.erf((lam - ops.sqrt(a + eta_f ** 2)) / (2.0 * ops.sqrt(lam))))
    return coeff * (1.0 / (1.0 + ops.exp(-eta_f)) * c_e0 * c_s_star - 1.0 / (1.0 + ops.exp(eta_f)) * c_e * c_max) * ops.sqrt((1.0 - c_s_star / c_max) / c_e0)


def thermodynamic_factor_nonlinear(ops, c_e, T):
    """custom_functions.jl:191: 0.601 - 0.24 (c_e/1000)^0.5 + 0.982 (1 - 0.0052 (T - 293)) (c_e/1000)^1.5"""
    return 0.601 - 0.24 * ops.pow(c_e / 1000.0, 0.5) + 0.982 * (1.0 - 0.0052 * (T - 293.0)) * ops.pow(c_e / 1000.0, 1.5)


# ----------------------------------------------------------------------------------------------------------------
# the residual
# ----------------------------------------------------------------------------------------------------------------
MODE_I, MODE_V, MODE_DT = 0, 1, 2   # control-row kinds (reference input_methods.jl:9,40,182-189)


def temperature_weights(th, lay):
    """weights w_i / L of temperature_weighting, reference aux...jl:649-676"""
    la, lp, ls, ln, lz = th["l_a"], th["l_p"], th["l_s"], th["l_n"], th["l_z"]
    w = [la / lay.Na] * lay.Na + [lp / lay.Np] * lay.Np + [ls / lay.Ns] * lay.Ns + [ln / lay.Nn] * lay.Nn + [lz / lay.Nz] * lay.Nz
    return w, (la + lp + ls + ln + lz)


def residual(model, ops, Y, YP, th, mode=MODE_I, value=0.0, with_control=True):
    """F(t,Y,YP,theta): rows 0..N-2 = residuals_PET! (scalar_residual.jl:27-66); row N-1 = control row
    (scalar_residual.jl:167-172, input_methods.jl:182-189).  Autonomous: no explicit t."""
    lay = model.lay
    Np, Ns, Nn, Na, Nz, Nrp, Nrn = lay.Np, lay.Ns, lay.Nn, lay.Na, lay.Nz, lay.Nrp, lay.Nrn
    thermal, aging = lay.temperature, lay.aging
    Ne = Np + Ns + Nn
    F, R = F_CONST, R_CONST
    res = [None] * lay.N

    c_e = Y[lay.c_e[0]:lay.c_e[1]]
    c_s = Y[lay.c_s[0]:lay.c_s[1]]
    j = Y[lay.j[0]:lay.j[1]]
    Phi_e = Y[lay.Phi_e[0]:lay.Phi_e[1]]
    Phi_s = Y[lay.Phi_s[0]:lay.Phi_s[1]]
    I_C = Y[lay.I]
    if thermal:
        T = Y[lay.T[0]:lay.T[1]]
    else:
        T = [th["T₀"]] * (Na + Np + Ns + Nn + Nz)       # build_T!, aux...jl:180-190
    T_p = T[Na:Na + Np]; T_s = T[Na + Np:Na + Np + Ns]; T_n = T[Na + Np + Ns:Na + Np + Ns + Nn]
    if aging:
        film = Y[lay.film[0]:lay.film[1]]
        j_s = Y[lay.j_s[0]:lay.j_s[1]]

    # --- build_auxiliary_states! (aux...jl:6-52) ---
    I1C = calc_I1C(th, ops)
    I_dens = I_C * I1C                                   # build_I_V!, aux...jl:54-70
    eps_sp = 1.0 - (th["ϵ_fp"] + th["ϵ_p"])             # active_material, aux...jl:537-545
    eps_sn = 1.0 - (th["ϵ_fn"] + th["ϵ_n"])
    eps_p = 1.0 - (th["ϵ_fp"] + eps_sp)                 # build_ϵ!, aux...jl:92-105
    eps_n = 1.0 - (th["ϵ_fn"] + eps_sn)
    eps_s = th["ϵ_s"]
    sig_p = th["σ_p"] * eps_sp; sig_n = th["σ_n"] * eps_sn   # build_σ_eff_p!, aux...jl:107-122
    a_p = 3 * eps_sp / th["Rp_p"]; a_n = 3 * eps_sn / th["Rp_n"]   # build_a!, aux...jl:124-139
    # j_total (aux...jl:160-178)
    jt = list(j)
    if aging:
        for i in range(Nn):
            jt[Np + i] = jt[Np + i] + j_s[i]
    Dsp = [th["D_sp"] * arrhenius(ops, th["Ea_D_sp"], T_p[i], thermal) for i in range(Np)]
    Dsn = [th["D_sn"] * arrhenius(ops, th["Ea_D_sn"], T_n[i], thermal) for i in range(Nn)]
    sd = model.solid_diffusion
    if sd == "polynomial":
        Qs = Y[lay.Q[0]:lay.Q[1]]
    # surface concentrations (build_c_s_star!, aux...jl:193-248)
    if sd == "Fickian":                                  # last radial node of each particle
        cs_star_p = [c_s[(i + 1) * Nrp - 1] for i in range(Np)]
        cs_star_n = [c_s[Np * Nrp + (i + 1) * Nrn - 1] for i in range(Nn)]
    elif sd == "quadratic":                              # c_avg - Rp/(5 D_s) j   (aux...jl:212-230)
        cs_star_p = [ops.aux("csp%d" % i, c_s[i] - (th["Rp_p"] / (Dsp[i] * 5)) * j[i]) for i in range(Np)]
        cs_star_n = [ops.aux("csn%d" % i, c_s[Np + i] - (th["Rp_n"] / (Dsn[i] * 5)) * j[Np + i]) for i in range(Nn)]
    else:                                                # c_avg + Rp/(35 D_s) (-j + 8 D_s Q)   (aux...jl:231-248)
        cs_star_p = [ops.aux("csp%d" % i, c_s[i] + (th["Rp_p"] / (Dsp[i] * 35)) * (-j[i] + 8 * Dsp[i] * Qs[i])) for i in range(Np)]
        cs_star_n = [ops.aux("csn%d" % i, c_s[Np + i] + (th["Rp_n"] / (Dsn[i] * 35)) * (-j[Np + i] + 8 * Dsn[i] * Qs[Np + i])) for i in range(Nn)]
    # OCV (aux...jl:250-270)
    ocv_p = {"LCO": OCV_LCO, "NMC": OCV_NMC, "LGM50": OCV_NMC_LGM50}[model.cathode]
    ocv_n = {"LCO": OCV_LiC6, "NMC": OCV_LiC6_with_NMC, "LGM50": OCV_LiC6_LGM50}[model.cathode]
    U_p, dU_p, U_n, dU_n = [], [], [], []
    for i in range(Np):
        u, d = ocv_p(ops, cs_star_p[i] / th["c_max_p"], T_p[i], thermal); U_p.append(u); dU_p.append(d)
    for i in range(Nn):
        u, d = ocv_n(ops, cs_star_n[i] / th["c_max_n"], T_n[i], thermal); U_n.append(u); dU_n.append(d)
    # overpotentials (aux...jl:272-300)
    eta_p = [Phi_s[i] - Phi_e[i] - U_p[i] for i in range(Np)]
    eta_n = [Phi_s[Np + i] - Phi_e[Np + Ns + i] - U_n[i] for i in range(Nn)]
    if aging:
        R_film = [th["R_SEI"] + film[i] / th["k_n_aging"] for i in range(Nn)]
        eta_n = [eta_n[i] - F * j[Np + i] * R_film[i] for i in range(Nn)]
    # K_eff, D_eff, D_s_eff (aux...jl:302-342) with the porosity substitution of aux...jl:141-158
    bp, bs, bn = th["brugg_p"], th["brugg_s"], th["brugg_n"]
    if model.cathode == "LGM50":                        # K_eff_LGM50, params.jl:660-672
        Kp = [ops.pow(eps_p, bp) * K_eff_LGM50_fn(ops, c_e[i]) for i in range(Np)]
        Ks = [ops.pow(eps_s, bs) * K_eff_LGM50_fn(ops, c_e[Np + i]) for i in range(Ns)]
        Kn = [ops.pow(eps_n, bn) * K_eff_LGM50_fn(ops, c_e[Np + Ns + i]) for i in range(Nn)]
    else:
        Kp = [ops.pow(eps_p, bp) * K_eff_fn(c_e[i], T_p[i]) for i in range(Np)]
        Ks = [ops.pow(eps_s, bs) * K_eff_fn(c_e[Np + i], T_s[i]) for i in range(Ns)]
        Kn = [ops.pow(eps_n, bn) * K_eff_fn(c_e[Np + Ns + i], T_n[i]) for i in range(Nn)]
    if model.cathode == "LGM50":                        # D_eff_LGM50, params.jl:646-658
        Dp = [ops.pow(eps_p, bp) * D_eff_LGM50_fn(th["D_e"], c_e[i]) for i in range(Np)]
        Ds = [ops.pow(eps_s, bs) * D_eff_LGM50_fn(th["D_e"], c_e[Np + i]) for i in range(Ns)]
        Dn = [ops.pow(eps_n, bn) * D_eff_LGM50_fn(th["D_e"], c_e[Np + Ns + i]) for i in range(Nn)]
    elif model.cathode == "LCO":                        # D_eff_linear, custom_functions.jl:59-69
        Dp = [th["D_p"] * ops.pow(eps_p, bp)] * Np
        Ds = [th["D_s"] * ops.pow(eps_s, bs)] * Ns
        Dn = [th["D_n"] * ops.pow(eps_n, bn)] * Nn
    else:                                               # D_eff, custom_functions.jl:83-94
        Dp = [ops.pow(eps_p, bp) * D_eff_fn(ops, c_e[i], T_p[i]) for i in range(Np)]
        Ds = [ops.pow(eps_s, bs) * D_eff_fn(ops, c_e[Np + i], T_s[i]) for i in range(Ns)]
        Dn = [ops.pow(eps_n, bn) * D_eff_fn(ops, c_e[Np + Ns + i], T_n[i]) for i in range(Nn)]

    hp, hs, hn = th["l_p"] / Np, th["l_s"] / Ns, th["l_n"] / Nn     # Δx*l per section
    if model.thermodynamic_factor == "linear":                        # thermodynamic_factor_linear, custom_functions.jl:177
        nu = [1.0] * Ne
    else:                                                             # thermodynamic_factor, custom_functions.jl:191-203
        nu = [thermodynamic_factor_nonlinear(ops, c_e[i], (T_p + T_s + T_n)[i]) for i in range(Ne)]

    # --- residuals_c_e! (residuals.jl:6-106) ---
    Dp_e, Ds_e, Dn_e = interpolate_electrolyte_grid(Dp, Ds, Dn, th, lay)
    De = Dp_e + Ds_e + Dn_e          # edge i = between CV i and i+1 (section-local "x" of block_matrix_maker)
    hsec = [hp] * Np + [hs] * Ns + [hn] * Nn
    sec_first = {0, Np, Np + Ns}
    rhs = [None] * Ne
    for i in range(Ne):
        # -block_matrix_maker: diag = -(x[i] + x[i-1]) with x[i-1]=0 at a section start; off-diagonals +x
        xl = 0.0 if i in sec_first else De[i - 1]
        xi = De[i]
        acc = -(xi + xl) * c_e[i]
        if i not in sec_first:
            acc = acc + De[i - 1] * c_e[i - 1]
        if (i + 1) not in sec_first and i + 1 < Ne:
            acc = acc + De[i] * c_e[i + 1]
        rhs[i] = acc / (hsec[i] ** 2)
    # interface rows (residuals.jl:38-88)
    den_ps = hp / 2 + hs / 2
    last_p = Dp_e[Np - 2] / hp; first_s = Dp_e[Np - 1] / den_ps
    i = Np - 1
    rhs[i] = (last_p * c_e[i - 1] - (last_p + first_s) * c_e[i] + first_s * c_e[i + 1]) / hp
    second_s = Ds_e[0] / hs
    i = Np
    rhs[i] = (first_s * c_e[i - 1] - (first_s + second_s) * c_e[i] + second_s * c_e[i + 1]) / hs
    den_sn = hs / 2 + hn / 2
    last_s = Ds_e[Ns - 2] / hs; first_n = Ds_e[Ns - 1] / den_sn
    i = Np + Ns - 1
    rhs[i] = (last_s * c_e[i - 1] - (last_s + first_n) * c_e[i] + first_n * c_e[i + 1]) / hs
    second_n = Dn_e[0] / hn
    i = Np + Ns
    rhs[i] = (first_n * c_e[i - 1] - (first_n + second_n) * c_e[i] + second_n * c_e[i + 1]) / hn
    tp = th["t₊"]
    for i in range(Np):
        rhs[i] = rhs[i] + (1 - tp) * nu[i] * a_p * jt[i]
    for i in range(Nn):
        rhs[Np + Ns + i] = rhs[Np + Ns + i] + (1 - tp) * nu[Np + Ns + i] * a_n * jt[Np + i]
    eps_cv = [eps_p] * Np + [eps_s] * Ns + [eps_n] * Nn
    for i in range(Ne):
        res[lay.c_e[0] + i] = rhs[i] / eps_cv[i] - YP[lay.c_e[0] + i]

    # --- residuals_c_s_avg!, Fickian finite difference (residuals.jl:128-180) ---
    def particle_rows(cs, jj, Rp, Dse, Nr):
        M1, c1, dx = fd_first_order(Nr)
        M2, c2, _ = fd_second_order(Nr)
        d1 = [c1 * sum(M1[r][k] * cs[k] for k in range(Nr) if M1[r][k] != 0.0) for r in range(Nr)]
        d1[Nr - 1] = -jj / Dse * Rp                 # BC at r = 1
        d1[0] = 0.0                                 # BC at r = 0
        d2 = [c2 * sum(M2[r][k] * cs[k] for k in range(Nr) if M2[r][k] != 0.0) for r in range(Nr)]
        d2[Nr - 1] = d2[Nr - 1] + 50 * dx * d1[Nr - 1] * c2
        out = [(Dse / Rp ** 2) * (3 * d2[0])]
        for k in range(1, Nr):
            r = k / (Nr - 1)
            out.append((Dse / Rp ** 2) * (d2[k] + 2.0 / r * d1[k]))
        return out
    o = lay.c_s[0]
    if sd == "Fickian":
        for i in range(Np):
            rows = particle_rows(c_s[i * Nrp:(i + 1) * Nrp], j[i], th["Rp_p"], Dsp[i], Nrp)
            for k in range(Nrp):
                res[o + i * Nrp + k] = rows[k] - YP[o + i * Nrp + k]
        o = lay.c_s[0] + Np * Nrp
        for i in range(Nn):
            rows = particle_rows(c_s[Np * Nrp + i * Nrn:Np * Nrp + (i + 1) * Nrn], j[Np + i], th["Rp_n"], Dsn[i], Nrn)
            for k in range(Nrn):
                res[o + i * Nrn + k] = rows[k] - YP[o + i * Nrn + k]
    else:                                                # quadratic / polynomial approximation (residuals.jl:108-127): d c_avg/dt = -3 j / Rp
        for i in range(Np):
            res[o + i] = -3 * j[i] / th["Rp_p"] - YP[o + i]
        for i in range(Nn):
            res[o + Np + i] = -3 * j[Np + i] / th["Rp_n"] - YP[o + Np + i]
    if sd == "polynomial":                               # residuals_Q! (residuals.jl:237-258)
        o = lay.Q[0]
        for i in range(Np):
            res[o + i] = (-Dsp[i] * Qs[i] - 45 / 2 * j[i]) / th["Rp_p"] ** 2 - YP[o + i]
        for i in range(Nn):
            res[o + Np + i] = (-Dsn[i] * Qs[Np + i] - 45 / 2 * j[Np + i]) / th["Rp_n"] ** 2 - YP[o + Np + i]

    # --- residuals_j! (residuals.jl:491-517) ---
    for i in range(Np):
        k = th["k_p"] * arrhenius(ops, th["Ea_k_p"], T_p[i], thermal)
        if model.rxn == "MHC":
            res[lay.j[0] + i] = rxn_MHC(ops, cs_star_p[i], c_e[i], T_p[i], eta_p[i], k, th["λ_MHC_p"], th["c_max_p"], th["c_e₀"]) - j[i]
        else:
            res[lay.j[0] + i] = rxn_BV(ops, cs_star_p[i], c_e[i], T_p[i], eta_p[i], k, th["c_max_p"]) - j[i]
    for i in range(Nn):
        k = th["k_n"] * arrhenius(ops, th["Ea_k_n"], T_n[i], thermal)
        if model.rxn == "MHC":
            res[lay.j[0] + Np + i] = rxn_MHC(ops, cs_star_n[i], c_e[Np + Ns + i], T_n[i], eta_n[i], k, th["λ_MHC_n"], th["c_max_n"], th["c_e₀"]) - j[Np + i]
        else:
            res[lay.j[0] + Np + i] = rxn_BV(ops, cs_star_n[i], c_e[Np + Ns + i], T_n[i], eta_n[i], k, th["c_max_n"]) - j[Np + i]

    # --- residuals_Φ_e! (residuals.jl:554-654) ---
    Kp_e, Ks_e, Kn_e = interpolate_electrolyte_grid(Kp, Ks, Kn, th, lay)
    Ke = Kp_e + Ks_e + Kn_e
    APhi = [None] * Ne
    for i in range(Ne):
        xl = 0.0 if i in sec_first else Ke[i - 1]
        acc = (Ke[i] + xl) * Phi_e[i]
        if i not in sec_first:
            acc = acc - Ke[i - 1] * Phi_e[i - 1]
        if (i + 1) not in sec_first and i + 1 < Ne:
            acc = acc - Ke[i] * Phi_e[i + 1]
        APhi[i] = acc / hsec[i]
    APhi[Ne - 1] = Phi_e[Ne - 1]                                     # Φ_e(x=L) = 0 row (residuals.jl:586)
    den = hp / 2 + hs / 2
    lastp = Kp_e[Np - 2] / hp
    i = Np - 1
    APhi[i] = -lastp * Phi_e[i - 1] + (lastp + Kp_e[Np - 1] / den) * Phi_e[i] - Kp_e[Np - 1] / den * Phi_e[i + 1]
    firsts = Ks_e[0] / hs
    i = Np
    APhi[i] = -Kp_e[Np - 1] / den * Phi_e[i - 1] + (firsts + Kp_e[Np - 1] / den) * Phi_e[i] - firsts * Phi_e[i + 1]
    den = hn / 2 + hs / 2
    lasts = Ks_e[Ns - 2] / hs
    i = Np + Ns - 1
    APhi[i] = -lasts * Phi_e[i - 1] + (lasts + Ks_e[Ns - 1] / den) * Phi_e[i] - Ks_e[Ns - 1] / den * Phi_e[i + 1]
    firstn = Kn_e[0] / hn
    i = Np + Ns
    APhi[i] = -Ks_e[Ns - 1] / den * Phi_e[i - 1] + (firstn + Ks_e[Ns - 1] / den) * Phi_e[i] - firstn * Phi_e[i + 1]
    cbp, cbs, cbn = interpolate_edges(c_e, th, lay)
    Tb_p, Tb_s, Tb_n = interpolate_edges(T[Na:Na + Ne], th, lay)
    fxp, fxs, fxn = edge_fluxes(c_e, th, lay)
    Kfac = [2 * R * (1 - tp) * nu[i] / F for i in range(Ne - 1)]     # nu of the LEFT control volume of each edge (residuals.jl:626-629)
    g = ([Kp_e[i] * Tb_p[i] * fxp[i] / cbp[i] for i in range(Np)]
         + [Ks_e[i] * Tb_s[i] * fxs[i] / cbs[i] for i in range(Ns)]
         + [Kn_e[i] * Tb_n[i] * fxn[i] / cbn[i] for i in range(Nn - 1)])       # Ne-1 edges
    f = [None] * Ne
    for i in range(Ne - 1):
        f[i] = -Kfac[i] * (g[i] - (g[i - 1] if i > 0 else 0.0))
    f[Ne - 1] = 0.0
    for i in range(Np):
        f[i] = f[i] + hp * F * a_p * jt[i]
    for i in range(Nn):
        if Np + Ns + i < Ne - 1:
            f[Np + Ns + i] = f[Np + Ns + i] + hn * F * a_n * jt[Np + i]
    f[Ne - 1] = 0.0
    for i in range(Ne):
        res[lay.Phi_e[0] + i] = APhi[i] - f[i]

    # --- residuals_Φ_s! (residuals.jl:656-703) ---
    Ps_p = Phi_s[:Np]; Ps_n = Phi_s[Np:]
    ps_diff = model.phi_s_form == "difference"
    dPs = {id(v): [ops.aux("dPs_%d" % (off + k), v[k + 1] - v[k]) for k in range(len(v) - 1)] for v, off in ((Ps_p, 0), (Ps_n, Np))} if ps_diff else None

    def lap(v, i, n):
        if ps_diff:
            d = dPs[id(v)]
            return d[0] if i == 0 else (-d[n - 2] if i == n - 1 else d[i] - d[i - 1])
        if i == 0:
            return -v[0] + v[1]
        if i == n - 1:
            return v[n - 2] - v[n - 1]
        return v[i - 1] - 2 * v[i] + v[i + 1]
    for i in range(Np):
        fp_ = hp ** 2 * a_p * F * jt[i]
        if i == 0:
            fp_ = fp_ - I_dens * hp
        res[lay.Phi_s[0] + i] = lap(Ps_p, i, Np) - fp_ / sig_p
    for i in range(Nn):
        fn_ = hn ** 2 * a_n * F * jt[Np + i]
        if i == Nn - 1:
            fn_ = fn_ + I_dens * hn
        res[lay.Phi_s[0] + Np + i] = lap(Ps_n, i, Nn) - fn_ / sig_n

    # --- aging rows ---
    if aging:
        # residuals_film! (residuals.jl:260-276)
        for i in range(Nn):
            res[lay.film[0] + i] = -j_s[i] * th["M_n"] / th["ρ_n"] - YP[lay.film[0] + i]
        # residuals_SOH! (residuals.jl:278-297) with trapz / extrapolate_section (external.jl:469-523)
        xr = [0.0] + [(1 / (2 * Nn)) + k * ((1 - 1 / Nn) / (Nn - 1)) for k in range(Nn)] + [1.0]
        x3 = xr[1:4]
        def extrap0(x, y):
            # extrap_x_0, external.jl:493-495 (second-order polynomial through 3 points, evaluated at 0)
            q = (y[2] - y[0] - ((x[1] - x[0]) ** -1) * (x[2] - x[0]) * (y[1] - y[0])) * ((x[2] ** 2 - x[0] ** 2 - ((x[1] - x[0]) ** -1) * (x[1] ** 2 - x[0] ** 2) * (x[2] - x[0])) ** -1)
            return y[0] - q * x[0] ** 2 - (y[1] - y[0] - q * (x[1] ** 2 - x[0] ** 2)) * ((x[1] - x[0]) ** -1) * x[0]
        yr = [extrap0(x3, j_s[0:3])] + list(j_s) + [extrap0(x3, [j_s[Nn - 1], j_s[Nn - 2], j_s[Nn - 3]])]
        xs = [v * th["l_n"] for v in xr]
        tz = 0.0
        for k in range(1, len(xs)):
            tz = tz + 0.5 * (xs[k] - xs[k - 1]) * (yr[k] + yr[k - 1])
        j_s_int = -tz * F * a_n / (3600 * I1C)
        res[lay.SOH[0]] = -j_s_int - YP[lay.SOH[0]]
        # residuals_j_s! (residuals.jl:519-552)
        for i in range(Nn):
            eta_s = Phi_s[Np + i] - Phi_e[Np + Ns + i] - th["Uref_s"] - F * jt[Np + i] * R_film[i]
            base = (th["i_0_jside"] * ops.pow(I_dens / I1C, th["w"]) / F) * (-ops.exp(-0.5 * F / (R * T_n[i]) * eta_s))
            calc = -ops.abs(base)
            calc = ops.where_gt(I_dens, 0.0, calc, 0 * calc)
            res[lay.j_s[0] + i] = j_s[i] - calc

    # --- residuals_T! (residuals.jl:299-489) + build_heat_generation_rates! (aux...jl:344-518) ---
    if thermal:
        ha, hz = th["l_a"] / Na, th["l_z"] / Nz
        Pe_p = Phi_e[:Np]; Pe_s = Phi_e[Np:Np + Ns]; Pe_n = Phi_e[Np + Ns:]
        ce_p = c_e[:Np]; ce_s = c_e[Np:Np + Ns]; ce_n = c_e[Np + Ns:]
        def fwd_left(x, h): return (-3 * x[0] + 4 * x[1] - x[2]) / (2 * h)
        def fwd_right(x, h):
            xr_ = x[::-1]
            return -((-3 * xr_[0] + 4 * xr_[1] - xr_[2]) / (2 * h))
        def central(x, h): return [(x[k + 2] - x[k]) / (2 * h) for k in range(len(x) - 2)]
        def acd_right(xl, hl, xr_, hr): return 2 * (xr_[0] - xl[-2]) / (3 * hl + hr)
        def acd_left(xl, hl, xr_, hr): return 2 * (xr_[1] - xl[-1]) / (hl + 3 * hr)
        dPs_p = [fwd_left(Ps_p, hp)] + central(Ps_p, hp) + [fwd_right(Ps_p, hp)]
        dPs_n = [fwd_left(Ps_n, hn)] + central(Ps_n, hn) + [fwd_right(Ps_n, hn)]
        dPe_p = [fwd_left(Pe_p, hp)] + central(Pe_p, hp) + [acd_right(Pe_p, hp, Pe_s, hs)]
        dPe_s = [acd_left(Pe_p, hp, Pe_s, hs)] + central(Pe_s, hs) + [acd_right(Pe_s, hs, Pe_n, hn)]
        dPe_n = [acd_left(Pe_s, hs, Pe_n, hn)] + central(Pe_n, hn) + [fwd_right(Pe_n, hn)]
        dce_p = [fwd_left(ce_p, hp)] + central(ce_p, hp) + [acd_right(ce_p, hp, ce_s, hs)]
        dce_s = [acd_left(ce_p, hp, ce_s, hs)] + central(ce_s, hs) + [acd_right(ce_s, hs, ce_n, hn)]
        dce_n = [acd_left(ce_s, hs, ce_n, hn)] + central(ce_n, hn) + [fwd_right(ce_n, hn)]
        Q = [None] * (Na + Ne + Nz)
        for i in range(Na):
            Q[i] = (I_dens ** 2) / th["σ_a"]
        for i in range(Nz):
            Q[Na + Ne + i] = (I_dens ** 2) / th["σ_z"]
        for i in range(Np):
            q_rev = F * a_p * jt[i] * T_p[i] * dU_p[i]
            q_rxn = F * a_p * jt[i] * eta_p[i]
            q_ohm = (Kp[i] * dPe_p[i] ** 2 + 2 * R * Kp[i] * T_p[i] * (1 - tp) * nu[i] / F * (dce_p[i] / ce_p[i]) * dPe_p[i]
                     + sig_p * dPs_p[i] ** 2)
            Q[Na + i] = q_rev + q_rxn + q_ohm
        for i in range(Ns):
            Q[Na + Np + i] = Ks[i] * dPe_s[i] ** 2 + 2 * R * Ks[i] * T_s[i] * (1 - tp) * nu[Np + i] / F * (dce_s[i] / ce_s[i]) * dPe_s[i]
        for i in range(Nn):
            q_rev = F * a_n * jt[Np + i] * T_n[i] * dU_n[i]
            q_rxn = F * a_n * jt[Np + i] * eta_n[i]
            q_ohm = (Kn[i] * dPe_n[i] ** 2 + 2 * R * Kn[i] * T_n[i] * (1 - tp) * nu[Np + Ns + i] / F * (dce_n[i] / ce_n[i]) * dPe_n[i]
                     + sig_n * dPs_n[i] ** 2)
            Q[Na + Np + Ns + i] = q_rev + q_rxn + q_ohm
        NT = Na + Ne + Nz
        lam = [th["λ_a"]] * Na + [th["λ_p"]] * Np + [th["λ_s"]] * Ns + [th["λ_n"]] * Nn + [th["λ_z"]] * Nz
        hT = [ha] * Na + [hp] * Np + [hs] * Ns + [hn] * Nn + [hz] * Nz
        rcp = ([th["ρ_a"] * th["Cp_a"]] * Na + [th["ρ_p"] * th["Cp_p"]] * Np + [th["ρ_s"] * th["Cp_s"]] * Ns
               + [th["ρ_n"] * th["Cp_n"]] * Nn + [th["ρ_z"] * th["Cp_z"]] * Nz)
        starts = [0, Na, Na + Np, Na + Np + Ns, Na + Ne]       # first CV of a,p,s,n,z
        ends = [Na - 1, Na + Np - 1, Na + Np + Ns - 1, Na + Ne - 1, NT - 1]
        AT = [None] * NT
        diff_form = model.t_conduction == "difference"
        dTk = [ops.aux("dT_%d" % k, T[k + 1] - T[k]) for k in range(NT - 1)] if diff_form else None
        for i in range(NT):
            # block_tridiag per section: Neumann corners (-1), interior (-2)
            first = i in starts; last = i in ends
            acc = 0.0
            if not first:
                acc = acc + ((-dTk[i - 1]) if diff_form else (T[i - 1] - T[i]))
            if not last:
                acc = acc + (dTk[i] if diff_form else (T[i + 1] - T[i]))
            AT[i] = lam[i] * acc / hT[i] ** 2
        # interfaces (residuals.jl:354-439): left CV = last of section k, right CV = first of section k+1
        for k in range(4):
            il = ends[k]; ir = starts[k + 1]
            hl, hr = hT[il], hT[ir]
            beta = (hl / 2) / (hl / 2 + hr / 2)
            lam_if = harmonic_mean(beta, lam[il], lam[ir])
            den_ = hr / 2 + hl / 2
            last_l = lam[il] / hl
            first_r = lam_if / den_
            second_r = lam[ir] / hr
            if diff_form:
                AT[il] = (-last_l * dTk[il - 1] + first_r * dTk[il]) / hl
                AT[ir] = (-first_r * dTk[ir - 1] + second_r * dTk[ir]) / hr
            else:
                AT[il] = (last_l * T[il - 1] - (last_l + first_r) * T[il] + first_r * T[il + 1]) / hl
                AT[ir] = (first_r * T[ir - 1] - (second_r + first_r) * T[ir] + second_r * T[ir + 1]) / hr
        BC = [0.0] * NT
        BC[0] = th["h_cell"] * (th["T_amb"] - T[0]) / ha
        BC[NT - 1] = -th["h_cell"] * (T[NT - 1] - th["T_amb"]) / hz
        for i in range(NT):
            res[lay.T[0] + i] = (AT[i] + Q[i] + BC[i]) / rcp[i] - YP[lay.T[0] + i]

    # --- control row ---
    if with_control:
        if mode == MODE_I:
            res[lay.I] = I_C - value                               # method_I, input_methods.jl:9
        elif mode == MODE_V:
            res[lay.I] = Phi_s[0] - Phi_s[Np + Nn - 1] - value      # method_V, input_methods.jl:40; calc_V scalar_residual.jl:86
        elif mode == MODE_DT:
            w, L = temperature_weights(th, lay)                     # constant_temperature, aux...jl:649-679
            s = 0.0
            for i in range(len(w)):
                s = s + YP[lay.T[0] + i] * w[i]
            res[lay.I] = value - s / L                              # scalar_residual.jl:172
        else:
            raise ValueError(mode)
    else:
        res[lay.I] = 0.0
    return res


def initial_guess_generic(model, ops, SOC, th):
    """reference src/states_definition.jl:80-121 (I left at 0; the caller sets it, input_methods.jl:11-30)."""
    lay = model.lay
    Y = [0.0] * lay.N
    for i in range(lay.c_e[0], lay.c_e[1]):
        Y[i] = th["c_e₀"]
    csp = th["c_max_p"] * (SOC * (th["θ_max_p"] - th["θ_min_p"]) + th["θ_min_p"])
    csn = th["c_max_n"] * (SOC * (th["θ_max_n"] - th["θ_min_n"]) + th["θ_min_n"])
    o = lay.c_s[0]
    for i in range(lay.Np * lay.Nrp):
        Y[o + i] = csp
    for i in range(lay.Nn * lay.Nrn):
        Y[o + lay.Np * lay.Nrp + i] = csn
    if lay.temperature:
        for i in range(lay.T[0], lay.T[1]):
            Y[i] = th["T₀"]
    if lay.aging:
        Y[lay.SOH[0]] = 1.0
    ocv_p = {"LCO": OCV_LCO, "NMC": OCV_NMC, "LGM50": OCV_NMC_LGM50}[model.cathode]
    ocv_n = {"LCO": OCV_LiC6, "NMC": OCV_LiC6_with_NMC, "LGM50": OCV_LiC6_LGM50}[model.cathode]
    Up = ocv_p(ops, csp / th["c_max_p"], th["T₀"], lay.temperature)[0]
    Un = ocv_n(ops, csn / th["c_max_n"], th["T₀"], lay.temperature)[0]
    for i in range(lay.Np):
        Y[lay.Phi_s[0] + i] = Up
    for i in range(lay.Nn):
        Y[lay.Phi_s[0] + lay.Np + i] = Un
    return Y


def initial_guess(model, SOC, th=None):
    return initial_guess_generic(model, FloatOps(), SOC, model.theta if th is None else th)


def used_theta_keys(model):
    """Sorted (code-point order == Julia's sort of Symbols) list of theta entries the residual reads:
    the analogue of get_only_θ_used_in_model, reference src/generate_functions.jl:327-363."""
    class Rec(dict):
        def __init__(self, d): super().__init__(d); self.used = set()
        def __getitem__(self, k): self.used.add(k); return super().__getitem__(k)
    th = Rec(model.theta)
    Y = initial_guess(model, 0.5)
    Y[model.lay.I] = 1.0
    residual(model, FloatOps(), Y, [0.0] * model.lay.N, th, MODE_I, 1.0)
    if model.lay.temperature:
        residual(model, FloatOps(), Y, [0.0] * model.lay.N, th, MODE_DT, 0.0)
    return sorted(th.used)
