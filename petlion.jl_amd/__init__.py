"""petlion.jl_amd -- MI355X-native ensemble DFN/P2D time stepping behind PETLION's simulate()/simulate!() vocabulary.

Only the hot path of SURVEY.md section 8: residual / analytic Jacobian / structured linear solve /
consistent initialisation / variable-order BDF + Newton / stop logic, as hand-written HIP for gfx950 behind the C ABI
of include/petlion_hip.h.  There is no CPU compute path in this package.
"""
from . import _capi, buildflags, closure_lib, closures, configs, grids  # noqa: F401
from .api import (LCO, NMC, NMC_LGM50, EnsembleSolution, Model, Solution, exit_reasons, final_exit_reason, make_protocol, petlion,  # noqa: F401
                  selftest, simulate, simulate_b, simulate_ensemble, theta_matrix)
from .params import EXIT_REASONS, Bounds, Opts, calc_I1C  # noqa: F401
