/*
 * petlion_hip.h -- C ABI of the MI355X-native DFN/P2D time-stepping path (libpetlion_hip.so).
 *
 * This is the drop-in boundary a Julia host binds with `ccall` (bindings/julia/PetlionHIP.jl).
 * Every entry point names the reference interface it replaces; citations are relative to the reference repository
 * (MarcBerliner/PETLION.jl @ v1.0.6).  Plain C, caller-owned arrays, no callbacks, no exceptions across the boundary.
 *
 * Conventions
 *   - all reals are IEEE fp64 (the reference is Float64 everywhere, src/structures.jl:337).
 *   - batched arrays are cell-major: X[cell*stride + k]  (one cell's vector is contiguous, like the reference's Vector).
 *   - `ptr_kind`: PLH_HOST  = the arrays are host memory (the library stages them through the device);
 *                 PLH_DEVICE = the arrays already live in this GPU's HBM (no copies; asynchronous on `stream`).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - return value: 0 ok; <0 API misuse / HIP failure (PLH_E_*).  Per-cell outcomes are reported in `status`/`flag` arrays
 *     using the reference's exit flags 0..11 (src/checks.jl:6,40,47,66,73,90,97,116,152,175,194,217) and negative codes
 *     for the reference's error() paths.
 *   - a handle is used by one host thread at a time (the reference model object is not re-entrant either,
 *     src/external.jl:135-139); different handles / GPUs may be used concurrently.  One handle may have launches in flight on several
 *     streams: the per-launch device workspaces (previous-point scratch, protocol copy, staging blocks) are kept per stream.
 *   - a handle is bound to one HIP device (plh_model_desc.device); every call switches to it and restores the caller's device.
 */
#ifndef PETLION_HIP_H
#define PETLION_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define PLH_HOST 0
#define PLH_DEVICE 1
#define PLH_HOST_ASYNC 2   /* plh_integrate only: the arrays are PINNED host memory from plh_host_alloc; the call enqueues the host-to-device copies, the kernel and
                              the device-to-host copies on `stream` and returns -- the outputs are valid after plh_synchronize(m, stream).  Two streams and two
                              buffer sets overlap one call's copies with the next call's kernel (the host-inclusive pipeline of bench.py). */

#define PLH_E_ARG (-1)          /* invalid argument */
#define PLH_E_UNSUPPORTED (-2)  /* model option outside the hot-path scope */
#define PLH_E_HIP (-3)          /* HIP runtime failure (no GPU, out of memory, launch failure) */

/* chemistries: reference src/params.jl:5-289 (LCO + LiC6), 295-507 (NMC + LiC6_NMC), 514-849 (NMC_LGM50 + LiC6_LGM50) */
#define PLH_CHEM_LCO_LIC6 0
#define PLH_CHEM_NMC_LIC6 1
#define PLH_CHEM_LGM50 2      /* NMC_LGM50 + LiC6_LGM50 (Chen et al. 2020), reference src/params.jl:514-849: own OCVs, D_eff(c_e), K_eff(c_e) */

/* operating modes = the control row of the DAE (reference src/physics_equations/input_methods.jl:9,40,182-189,
 * src/physics_equations/scalar_residual.jl:167-172) */
#define PLH_MODE_I 0   /* current, C-rate:      Y[I] - value                      */
#define PLH_MODE_V 1   /* voltage, V:           Phi_s[1] - Phi_s[end] - value     */
#define PLH_MODE_DT 2  /* dT_avg/dt, K/s:       value - sum_i w_i YP[T_i] / L     (temperature models only) */
#define PLH_MODE_P 3   /* power, W/m^2:         Y[I] I1C (Phi_s[1] - Phi_s[end]) - value   (method_P, input_methods.jl:80-111) */
#define PLH_MODE_ETA_P 4 /* plating overpotential, V: Phi_s.n[1] - Phi_e.n[1] - value     (method_η_p, input_methods.jl:113-152) */
#define PLH_N_MODES 5   /* the modes above: the ones with an exported Jacobian pattern (plh_jac_pattern, plh_residual, ...) */
#define PLH_MODE_DSTATE 6 /* rate of change of ONE differential state held at value: value - YP[index] = 0 (reference dc_s_p_max / dc_s_p_min / dc_s_n_max / dc_s_n_min / dc_e_max /
                            dc_e_min, input_methods.jl:190-247: state_deriv_func(ind) as a run_residual).  plh_run.dstate (PLH_DSTATE_*) says which state: chosen per cell from the
                            state at the START of the run (the extreme surface concentration of an electrode / electrolyte concentration at the end of the previous run), so the run
                            must continue a solution (not the first run of a protocol unless Y_init is given).  PLH_VAL_CONST or PLH_VAL_HOLD (= 0).  The consistent initialisation
                            uses the row with YP[index] replaced by its differential equation (scalar_residual.jl:335-362).  plh_integrate / plh_ensemble_run only. */
#define PLH_DSTATE_CS_P_MAX 1
#define PLH_DSTATE_CS_P_MIN 2
#define PLH_DSTATE_CS_N_MAX 3
#define PLH_DSTATE_CS_N_MIN 4
#define PLH_DSTATE_CE_MAX 5
#define PLH_DSTATE_CE_MIN 6
#define PLH_MODE_RES 5 /* user-defined control residual (reference method_res, input_methods.jl:155-175; run_residual, scalar_residual.jl:172):  value - f(t, Y, theta) = 0 with
                          the closure f as PLH_VAL_EXPR and its derivative programs (n_dcol >= 1: the reference always differentiates this row, scalar_residual.jl:262-274);
                          plh_integrate / plh_ensemble_run only.  Closures of YP (the dc_s_*, dc_e_* modes are such) are not supported. */

/* plh_model_desc.solid_diffusion / thermodynamic_factor / rxn: the model options of petlion(...; solid_diffusion, thermodynamic_factor, rxn_p, rxn_n)
 * (reference src/params.jl:140-172; equations: residuals.jl:108-127,237-258, aux...jl:193-248, custom_functions.jl:177-203, 212-298) */
#define PLH_SD_FICKIAN 0     /* :Fickian with Fickian_method = :finite_difference (N_r radial nodes per particle) */
#define PLH_SD_QUADRATIC 1   /* :quadratic  -- one volume-averaged concentration per particle, c_s* = c_avg - Rp/(5 D_s) j */
#define PLH_SD_POLYNOMIAL 2  /* :polynomial -- c_avg and the flux moment Q per particle (Subramanian et al.) */
#define PLH_TF_LINEAR 0      /* thermodynamic_factor_linear: nu = 1 */
#define PLH_TF_NONLINEAR 1   /* thermodynamic_factor: nu(c_e, T) = 0.601 - 0.24 (c_e/1000)^0.5 + 0.982 (1 - 0.0052 (T - 293)) (c_e/1000)^1.5 */
#define PLH_RXN_BV 0         /* rxn_BV in both electrodes */
#define PLH_RXN_MHC 1        /* rxn_MHC in both electrodes (Marcus-Hush-Chidsey; theta gains λ_MHC_p, λ_MHC_n) */

/* plh_model_desc.precision */
#define PLH_PREC_F64 0    /* everything fp64 (default; the parity configuration) */
#define PLH_PREC_MIXED 1  /* block-Thomas factors and particle resolvents stored in fp32 in LDS; states, residuals, Jacobian entries, time, error control fp64 */
#define PLH_PREC_F64_REFORDER 2 /* everything fp64, and the finite-volume rows (c_e, Phi_e, heat conduction) evaluated in the reference's OPERATION ORDER: matrix form A x - f with
                                   every product rounded before the sum (src/physics_equations/residuals.jl:6-106, 554-654, 299-489) instead of the default build's
                                   conservative edge-flux / difference form.  Same equations and Jacobian; the rows carry the reference's evaluation rounding, so the
                                   integrator's step / order decisions in :hold legs follow the reference's more closely.  LCO isothermal and LCO with
                                   temperature. */

/* how `value` is obtained (reference input_methods.jl:11-30,53-63; model_evaluation.jl:165-170) */
#define PLH_VAL_CONST 0
#define PLH_VAL_HOLD 1  /* :hold  -- the value reached at the end of the previous run */
#define PLH_VAL_REST 2  /* :rest  -- I = 0, all bound checks skipped (src/checks.jl:12,388) */
#define PLH_VAL_TABLE 3   /* time-dependent input from plh_run.tab_t / tab_v */
#define PLH_VAL_EXPR 4    /* input = a closure f(t, Y, YP, theta) given as a postfix program (PLH_OP_*) in plh_run.tab_t (opcodes) / tab_v (operands) */
/* Postfix programs for PLH_VAL_EXPR: the form in which a reference input closure `I = (t, Y, YP, p) -> ...` (scalar_residual.jl:169-170, input_methods.jl:159-176) crosses the
   C ABI.  Instruction k is (opcode tab_t[k], operand tab_v[k]); the machine is a stack of at most 16 doubles, the program must leave exactly one value.  Operands: the constant
   for CONST, a 0-based state index for Y / YP, a 0-based position in the model's theta_keys for THETA; unused otherwise.  The closure is evaluated with the current iterate inside
   every residual evaluation of the run (also during the consistent initialisation), like run.func in scalar_residual!.
   Derivatives (plh_run.n_dcol / dcol / dofs): the reference differentiates a closure of the state symbolically and puts the row d(method - f)/dY into the Newton matrix
   (differentiate_residual_func, scalar_residual.jl:276-416).  The caller does the same differentiation (it holds the expression) and passes, for each state column dcol[k]
   the closure reads, a program for d f / d Y[dcol[k]]: instructions [dofs[k], dofs[k+1]) of the SAME tab_t / tab_v arrays (dofs[0] >= n_tab).  A column dcol[k] = N + i
   (N = n_states, i a DIFFERENTIAL state) is d f / d YP[i]: it enters the integration row times cj, and the consistent-initialisation row through the differential equation of
   state i (YP[i] -> rhs_i(Y), as the reference substitutes there, scalar_residual.jl:335-362; the closure itself is then evaluated with YP = rhs(Y)).  The device evaluates the
   programs at every Jacobian refresh and solves with the general control row (one extra structured solve per factorisation, two dot products per solve).  n_dcol = 0 is the
   reference's own fallback for closures it cannot differentiate (scalar_residual.jl:248-274, _get_method_funcs_no_differentiation): same converged states, other Newton steps. */
#define PLH_OP_CONST 0
#define PLH_OP_T 1
#define PLH_OP_Y 2
#define PLH_OP_YP 3
#define PLH_OP_THETA 4
#define PLH_OP_ADD 5
#define PLH_OP_SUB 6
#define PLH_OP_MUL 7
#define PLH_OP_DIV 8
#define PLH_OP_NEG 9
#define PLH_OP_SIN 10
#define PLH_OP_COS 11
#define PLH_OP_EXP 12
#define PLH_OP_LOG 13
#define PLH_OP_SQRT 14
#define PLH_OP_POW 15
#define PLH_OP_ABS 16
#define PLH_OP_MIN 17
#define PLH_OP_MAX 18
#define PLH_OP_LT 19      /* a b -> (a < b) as 1.0 / 0.0 ; LE, GT, GE alike */
#define PLH_OP_LE 20
#define PLH_OP_GT 21
#define PLH_OP_GE 22
#define PLH_OP_SELECT 23  /* c a b -> (c != 0 ? a : b)   (ifelse) */
#define PLH_OP_TANH 24
#define PLH_N_OPS 25
#define PLH_EXPR_STACK 16
#define PLH_MAX_DCOL 60     /* most state columns a closure's derivative programs may name (plh_run.n_dcol) */

/* per-cell status beyond the reference's exit flags */
#define PLH_FLAG_RUNNING (-1)
#define PLH_ERR_INIT (-11)      /* "Could not initialize DAE in 100 iterations", src/model_evaluation.jl:456 */
#define PLH_ERR_STALL (-12)     /* "Model failed to converge at t = ...", src/checks.jl:233,236 */
#define PLH_ERR_MAXITERS (-13)  /* "Reached max iterations", src/checks.jl:239 */
#define PLH_ERR_OUTPUT_FULL (-14)
#define PLH_FLAG_STOP_FUNCTION 12   /* the run ended on the caller's stop function (plh_opts.stop_ops): the reference's opts.stop_function hook, src/checks.jl:26 */

typedef struct plh_model_s* plh_model_t;

/* mirrors the structural keyword arguments of petlion(cathode; N_p, ..., temperature, aging)
 * (reference src/params.jl:119-174, src/external.jl:2-18) */
typedef struct {
  int chemistry;                                  /* PLH_CHEM_* */
  int N_p, N_s, N_n, N_a, N_z, N_r_p, N_r_n;      /* reference defaults: 10 each */
  int temperature;                                /* 0/1 */
  int aging_SEI;                                  /* 0/1 */
  int real_bytes;                                 /* 8: states, residuals, time and tolerances are fp64 (the reference is Float64 everywhere) */
  int precision;                                  /* PLH_PREC_* : storage precision of the LDS-resident Newton-matrix factors (config C5's fp32 leg) */
  int device;                                     /* HIP device ordinal the handle binds to (every call of the handle runs there); -1 = the device that is
                                                     current when plh_model_create is called */
  int solid_diffusion;                            /* PLH_SD_* */
  int thermodynamic_factor;                       /* PLH_TF_* */
  int rxn;                                        /* PLH_RXN_* */
  int waves_per_cell;                             /* 0 or 1: one wavefront integrates one cell (every variant); 2: two wavefronts per cell -- wave 1 owns the particle rows and
                                                     runs next to wave 0's finite-volume work (LCO isothermal Fickian fp64 only).  Same algorithm, same results to rounding. */
} plh_model_desc;

/* reference boundary_stop_conditions (src/structures.jl:237-250); NaN disables a bound */
typedef struct {
  double V_max, V_min, SOC_max, SOC_min, T_max, c_s_n_max, I_max, I_min, eta_plating_min, c_e_min, dfilm_max;
} plh_bounds;

/* one run of a protocol == one simulate()/simulate!() call (src/model_evaluation.jl:11-97) */
typedef struct {
  int mode;        /* PLH_MODE_* */
  int value_kind;  /* PLH_VAL_*  */
  double value;
  double tf;       /* run length in run-local time; reference default 1e6 (model_evaluation.jl:13) */
  plh_bounds bounds;
  /* PLH_VAL_TABLE: the input is a function of the run-local time (reference run_function: I = t -> ..., scalar_residual.jl:169-170) given as a
     piecewise-linear table; a repeated knot time is a jump (right-continuous), the last value holds beyond the last knot.  The arrays are HOST
     memory like the protocol itself (plh_integrate stages them).  List jump times in plh_opts.tdiscon as with the reference's `tdiscon`. */
  /* PLH_VAL_EXPR: n_tab instructions, tab_t[k] = opcode (PLH_OP_*, stored as a double), tab_v[k] = operand (see above); HOST arrays, staged like a table. */
  int n_tab;
  int closure_id;  /* written by the library (the caller's value is ignored): which compiled closure of an attached closure library this run uses, -1 = interpreted
                      (plh_model_attach_closure_library).  Sits in what was padding: no other field moved. */
  const double* tab_t; const double* tab_v;
  /* ensemble axis of the protocol itself: per-cell input value (PLH_VAL_CONST only, e.g. a C-rate sweep) and per-cell run length, [n_cells] HOST
     arrays staged by plh_integrate; NULL = every cell uses `value` / `tf`.  (New: the reference runs one cell per simulate() call.) */
  const double* value_cell; const double* tf_cell;
  /* PLH_VAL_EXPR of the state: derivative programs of the control row (see PLH_VAL_EXPR above); HOST arrays, dcol[n_dcol] 0-based columns in ascending order (Y columns, then
     N + i for YP of differential states), dofs[n_dcol + 1] instruction offsets into tab_t / tab_v.  n_dcol = 0: no differentiation. */
  int n_dcol;
  int dstate;      /* PLH_MODE_DSTATE: PLH_DSTATE_* ; 0 otherwise */
  const int* dcol; const int* dofs;
} plh_run;

/* reference options_simulation (src/structures.jl:266-285), the numerical subset */
typedef struct {
  double abstol, reltol, abstol_init, reltol_init;   /* reference defaults 1e-6, 1e-3, =abstol, =reltol */
  int maxiters;                                      /* 10000 */
  int check_bounds, interp_final;                    /* 1, 1 */
  int max_order;                                     /* BDF order cap, 5 */
  int jac_every_step;                                /* 0: IDA's Jacobian-reuse policy */
  double init_step;                                  /* 0: IDA's automatic h0 = 0.5/||y'||_wrms; >0: IDASetInitStep (src/checks.jl:231) */
  int n_tdiscon; const double* tdiscon;              /* opts.tdiscon (src/structures.jl:279), any length, HOST array like the protocol (staged by
                                                        plh_integrate): tstops at tdiscon - reltol/2 (model_evaluation.jl:295-297) */
  int refine;                                        /* 0 (default): plain structured solves.  n > 0: n steps of iterative refinement of every linear solve
                                                        (init Newton and corrector) against the factored matrix -- the parity mode: the solution no longer
                                                        depends on the elimination order (structured here, KLU's in the reference) beyond ~1e-13 */
  int n_tstops; const double* tstops;                /* opts.tstops (src/structures.jl:278, model_evaluation.jl:292-294): times, in run-local time, the integrator must hit
                                                        exactly (a saved point lands on each); any length, HOST array staged like tdiscon; applies to every run of the protocol */
  int yp_alg_zero;                                   /* 0 (default): newtons_method! hands the integrator its finite-difference estimate of the algebraic derivatives
                                                        (src/model_evaluation.jl:462-477).  1: the integrator starts with YP_alg = 0, as the package version that produced the
                                                        reference's example notebooks did -- with it the printed step history of examples/model_inputs_and_outputs.ipynb
                                                        (121 saved points, sol.V[1:13], sol.c_e[1:5]) is reproduced to 1e-8 (tests/golden/notebook_kats.json) */
  /* opts.stop_function (src/structures.jl:283, called after the eleven built-in checks at every accepted step, src/checks.jl:26; simulate(...; stop_function), src/model_evaluation.jl:32).
     A Julia closure cannot cross the C ABI; its expression can: ONE postfix program g(t, Y, YP, theta) in the PLH_OP_* vocabulary of PLH_VAL_EXPR (n_stop instructions, opcodes in
     stop_ops stored as doubles, operands in stop_args; HOST arrays staged like tdiscon; t = run-local time).  The run ends when g > 0 (g - 0 > eps with the eps of the built-in checks:
     reltol while t < 1 s, else 0), exit flag PLH_FLAG_STOP_FUNCTION, with the interpolation fraction g_prev / (g_prev - g) entering the same linear back-interpolation as the
     built-in bounds (interp_final_points!, src/model_evaluation.jl:369-382; the smallest fraction of all checks that fired wins, as in the reference).  Skipped in :rest runs and
     with check_bounds = 0, like every other check.  n_stop = 0: none. */
  int n_stop; const double* stop_ops; const double* stop_args;
} plh_opts;

/* per-cell, per-run summary == run_info + the printed summary (src/structures.jl:40-44, 678-746) */
typedef struct {
  int flag;          /* exit flag 0..11 or PLH_ERR_* */
  int iterations;    /* run.info.iterations */
  double t_end, V, I, SOC, T_avg;
} plh_run_info;

/* per-cell device counters: the roofline contract of SURVEY.md 8(d) */
typedef struct {
  long long n_steps, n_res, n_jac, n_fact, n_solve, n_newton, n_errfail, n_convfail, sum_kp2, n_init_iters;
  long long cyc[8];   /* shader cycles per phase (residual, jacobian+factor, solve, newton vector ops, step control, init, output, total);
                         filled only by the profiling build (-DPL_PHASE_TIMERS), zero otherwise */
} plh_counters;

/* outputs of plh_integrate; any pointer may be NULL.  Saved points are the reference's per-step pushes of
 * set_vars! (src/save_outputs.jl:11-40): t = 0 of every run + every accepted step; the last point of a run that ended on
 * a bound is the back-interpolated one (src/model_evaluation.jl:369-382). */
typedef struct {
  int max_pts;                       /* row stride of the per-point arrays */
  double *t, *V, *I, *SOC, *T_avg;   /* [n_cells][max_pts] */
  int* n_pts;                        /* [n_cells] */
  double *Y_final, *YP_final;        /* [n_cells][n_states] */
  plh_run_info* run_info;            /* [n_cells][n_runs] */
  plh_counters* counters;            /* [n_cells] */
  double* Y_all;                     /* [n_cells][max_pts][N] or NULL: every saved state vector (reference outputs = :all / sol.Y; 2.4 kB per point) */
} plh_outputs;

/* ---- model handle: replaces petlion()'s generated-function bundle p.funcs (src/structures.jl:315-334) ---- */
int plh_model_create(const plh_model_desc* desc, plh_model_t* out);
void plh_model_destroy(plh_model_t m);
/* Other discretisations (reference src/params.jl:119-136: petlion(...; N_p, N_s, N_n, N_r_p, N_r_n, N_a, N_z)).  The kernels are compiled per grid, like the reference
   generates and caches its functions per model (generate_functions.jl:44-94): the built-in ones for the default 10 / 10 / 10 / 10, any other grid with 2 <= N_p, N_s, N_n,
   N_p + N_s + N_n <= 48, 10 <= N_r_p, N_r_n <= 16 (the two particle grids may differ) as a library built from csrc/variant_tu.hip (petlion.jl_amd/grids.py) and registered here BEFORE
   plh_model_create is called with those dimensions.  Registering the same path twice is a no-op; the library stays loaded for the life of the process. */
int plh_register_grid_library(const char* path);
int plh_n_states(plh_model_t m);     /* p.N.tot  */
int plh_n_diff(plh_model_t m);       /* p.N.diff */
int plh_n_theta(plh_model_t m);      /* length(θ_keys), src/generate_functions.jl:327-363 */
const char* plh_theta_key(plh_model_t m, int i);   /* UTF-8 names identical to the reference Symbols, sorted like θ_keys */
double plh_theta_default(plh_model_t m, int i);    /* chemistry defaults, src/params.jl */
int plh_lds_bytes(plh_model_t m);    /* LDS held by one cell (= one workgroup) of this variant: 160 kB / this = resident cells per CU */
/* p.ind (reference state_indices, src/external.jl:275-365): the named sections of the state vector in storage order, differential states
 * first.  Names are the reference's Symbols (c_e, c_s_avg, T, film, SOH, j, Φ_e, Φ_s, j_s, I); start is 0-based. */
int plh_n_sections(plh_model_t m);
int plh_section(plh_model_t m, int i, const char** name, int* start, int* len);
/* CSC pattern (0-based) of the full N x N Jacobian for a mode == [J_y_sp ; scalar row] of
 * _get_jacobian_combined (src/physics_equations/scalar_residual.jl:500-522).  colptr/rowval may be NULL to query nnz. */
int plh_jac_pattern(plh_model_t m, int mode, int* nnz, int* colptr, int* rowval);
/* CSC pattern (0-based, rows/columns relative to the block) of J_y_alg = J[N_diff : N-1, N_diff : N] at gamma = 0, the generated-function
 * J_y_alg! of seam 1 (src/generate_functions.jl:318-325): N_alg columns, N_alg - 1 rows (the control row is not generated). */
int plh_jac_alg_pattern(plh_model_t m, int mode, int* nnz, int* colptr, int* rowval);
const char* plh_last_error(void);
/* "hipcc=...;clang=...;flags=<hash>;src=<hash>" of this binary (the miscompile guard keys on it), and the number of HIP devices the library can see
 * (0: no GPU or no driver -- hosts use it to decide whether the kernel self-test can run, without a second GPU runtime in the process) */
const char* plh_build_info(void);
int plh_device_count(void);
/* sizeof / offsetof of every struct of this header as the library was compiled, for bindings that mirror them by hand (bindings/julia/PetlionHIP.jl,
 * the ctypes mirror of the tests): out[] receives, per struct in declaration order (plh_model_desc, plh_bounds, plh_run, plh_opts, plh_run_info,
 * plh_counters, plh_outputs): sizeof, number of fields, then the offset of each field.  Returns the number of ints written (or needed, if cap is short). */
int plh_abi_layout(int* out, int cap);

/* ---- batched evaluators (single-cell seam = n_cells 1, PLH_HOST) ---- */
/* initial_guess!(out, SOC, θ, X_applied)  (src/states_definition.jl:80-121): Y[cell][N], Y[I] = 0 */
int plh_initial_guess(plh_model_t m, int n_cells, const double* theta, const double* SOC, double* Y, int ptr_kind, void* stream);
/* R_full(res,t,Y,YP,p,run) = f_diff! ++ f_alg! ++ scalar_residual!  (scalar_residual.jl:558-583) */
int plh_residual(plh_model_t m, int n_cells, const double* theta, const double* Y, const double* YP, int mode, double value,
                 double* F, int ptr_kind, void* stream);
/* J_full(J,t,Y,YP,γ,p,run): nzval[cell][nnz] in the CSC order of plh_jac_pattern  (scalar_residual.jl:588-602) */
int plh_jacobian(plh_model_t m, int n_cells, const double* theta, const double* Y, const double* YP, double cj, int mode,
                 double* nzval, int ptr_kind, void* stream);
/* x = J_full \ b  with the structured (particle-resolvent + block-Thomas + border) solver that replaces KLU
 * (src/model_evaluation.jl:271, 417-428): b[cell][N] in, x out in place */
int plh_linear_solve(plh_model_t m, int n_cells, const double* theta, const double* Y, const double* YP, double cj, int mode,
                     double* b, int ptr_kind, void* stream);
/* the same with n_refine steps of iterative refinement against the factored matrix (plh_opts.refine of the integrator as a stand-alone evaluator) */
int plh_linear_solve_refined(plh_model_t m, int n_cells, const double* theta, const double* Y, const double* YP, double cj, int mode,
                             double* b, int n_refine, int ptr_kind, void* stream);
/* seam 1's split evaluators, with the argument lists of the five generated functions (src/generate_functions.jl:44-94, callers
 * scalar_residual.jl:558-602): f_diff!(out[N_diff], t, Y, YP, θ), f_alg!(out[N_alg - 1], t, Y, YP, θ) (no control row, generate_functions.jl:254),
 * J_y_alg!(nzval[nnz_alg], t, Y, YP, γ, θ) in the CSC order of plh_jac_alg_pattern.  J_y! is plh_jacobian minus the control row's entries
 * (a stub gathers them through the pattern, bindings/julia/SavedModelWriter.jl). */
int plh_residual_diff(plh_model_t m, int n_cells, const double* theta, const double* Y, const double* YP, double* out, int ptr_kind, void* stream);
int plh_residual_alg(plh_model_t m, int n_cells, const double* theta, const double* Y, const double* YP, double* out, int ptr_kind, void* stream);
int plh_jacobian_alg(plh_model_t m, int n_cells, const double* theta, const double* Y, const double* YP, int mode, double* nzval,
                     int ptr_kind, void* stream);
/* newtons_method! (src/model_evaluation.jl:430-480): Y in/out, YP out, status[cell] 0 or PLH_ERR_INIT, iters[cell] */
int plh_init_consistent(plh_model_t m, int n_cells, const double* theta, int mode, double value, double reltol_init,
                        double* Y, double* YP, int* status, int* iters, int ptr_kind, void* stream);

/* ---- the hot path: ensemble integrate == simulate()/simulate!() chains over independent cells ----
 * theta[cell][n_theta]; SOC0[cell]; one protocol (runs[n_runs], host memory) shared by all cells.
 * Y_init / t_init (both NULL for a new solution): continue a previous solution like simulate!(sol, p, ...)
 * (src/model_evaluation.jl:87-97, 206-209): Y_init[cell][n_states] = sol.Y[end], t_init[cell] = sol.t[end],
 * SOC0[cell] = sol.SOC[end]; the first run is then a continuation run (t0 = nextfloat(t_init), tstop at 1 s, :hold works).
 * Y_init WITHOUT t_init (t_init = NULL): simulate(p, ...; initial_states = Y) (src/model_evaluation.jl:15, 102-110, 193-199) -- a NEW solution (t0 = 0) that starts from
 * the caller's state vectors instead of initial_guess!; the algebraic states are re-solved by the consistent initialisation as always; SOC0[cell] = calc_SOC(Y)
 * (src/physics_equations/scalar_residual.jl:95-102: the anode's mean c_s_avg as a stoichiometry fraction), which the caller computes.
 * One wavefront integrates one cell for the whole protocol inside a single kernel launch. */
int plh_integrate(plh_model_t m, int n_cells, const double* theta, const double* SOC0, const double* Y_init, const double* t_init,
                  int n_runs, const plh_run* runs, const plh_opts* opts, const plh_outputs* out, int ptr_kind, void* stream);

/* ---- forward parameter sensitivities next to the states ("parameter-sensitivity (forward) outputs for estimation workflows").  The reference has no such
 * output: its users difference whole simulate() calls (one more simulate() per parameter and direction, with the adaptive step control's noise in the quotient).  Here
 * s_k(t) = dY(t)/d theta[sens_cols[k]] is integrated with the same steps and orders as Y by the staggered-direct method (one linear solve per parameter and accepted step with
 * the factorisation the integrator already holds; csrc/dfn_sens.h).  The guarantee on the states: the integrator takes THE SAME STEPS as plh_integrate -- every counter and
 * exit flag equal in every cell -- and Y, the saved points and run_info agree with it to ~1e-9 relative (measured: bit-identical in all but a few cells per thousand, those at
 * 1e-13 ... 1e-11: the sensitivity instantiation is another compilation of the step loop and may contract a sum differently).  The power-on self-test holds it to 1e-8.
 *   dY_dtheta[cell][k][n_states]  at the end of the last completed run (NaN for a cell whose protocol failed); may be NULL
 *   dV_dtheta[cell][k][max_pts]   at every saved point (the Jacobian of the voltage curve a least-squares fit needs); may be NULL
 *   sens_stat[cell][3]            corrector iterations spent; solves that did not reach the tolerance; steps whose corrector factored the step's own matrix because the
 *                                 integrator's (stale) one did not contract (the integrator's factorisation is saved and copied back: the states do not notice); may be NULL
 * Derivatives are with respect to the absolute value of the parameter.  Saved points at accepted steps / stop times: partial derivatives at fixed time.  The last point of a run
 * that ended on a bound is the reference's linear back-interpolation between the last two accepted points (model_evaluation.jl:369-382), whose fraction depends on theta through
 * the bounded quantity: its derivative -- and what the next run continues from -- includes that shift, i.e. it is the derivative of the end state as simulate() returns it
 * (dV/dtheta = 0 at a voltage bound).  Bounds on V, I, T_avg, eta_plating, c_s_n, c_e, and SOC under a constant current; an SOC bound in another mode or the dfilm bound
 * (a bound on YP) gives NaN from there on.  Protocols: constant or :rest inputs in the modes
 * I, V, P, eta_p, dT, any number of runs, new solutions only; everything else is PLH_E_UNSUPPORTED.  ptr_kind PLH_HOST or PLH_DEVICE (the call is synchronous). */
int plh_integrate_sens(plh_model_t m, int n_cells, const double* theta, const double* SOC0, int n_runs, const plh_run* runs, const plh_opts* opts,
                       const plh_outputs* out, int n_sens, const int* sens_cols, double* dY_dtheta, double* dV_dtheta, int* sens_stat, int ptr_kind, void* stream);

/* ---- compiled input closures.  The reference compiles the user's closure and its symbolic derivatives into its generated control-row functions
 * (differentiate_residual_func, scalar_residual.jl:231-416); the built-in kernels interpret the PLH_VAL_EXPR programs instead.  A closure library is csrc/variant_tu.hip compiled once
 * more for this model's variant and grid with the programs of ONE protocol written out as straight-line device code (petlion.jl_amd/closure_lib.py writes the header and runs
 * hipcc; the library exports its variant table like a grid library, plus plh_closure_digest()).  After it is attached, plh_integrate uses its kernels for every call whose
 * PLH_VAL_EXPR programs (opcodes, operands, derivative columns, in protocol order) hash to the digest the library was built for, and the interpreter for every other call --
 * same arguments, same results (the expression is evaluated with the same operations in the same order).  One library per handle; attaching another replaces it;
 * path = NULL detaches (every closure is interpreted again: the host side verifies a freshly compiled library that way before it relies on it). */
int plh_model_attach_closure_library(plh_model_t m, const char* path);
/* 1 if the handle's last plh_integrate ran the attached closure library's kernels, 0 if it interpreted (or had no closure input) */
int plh_last_integrate_compiled(plh_model_t m);
/* the digest plh_integrate computes for a protocol (0 if it has no PLH_VAL_EXPR run): what a builder embeds in the library */
unsigned long long plh_closure_digest(int n_runs, const plh_run* runs);

/* timing of the last plh_integrate kernel on its stream, measured with HIP events (ms); <0 if unavailable */
double plh_last_kernel_ms(plh_model_t m);

/* pinned host memory for PLH_HOST_ASYNC calls, and the completion point of everything the handle enqueued on `stream` */
int plh_host_alloc(void** p, unsigned long long bytes);
void plh_host_free(void* p);
int plh_synchronize(plh_model_t m, void* stream);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI for the ensemble scatter / gather only ----
 * A communicator joins n_ranks processes, each bound to one GPU.  Rank 0 obtains a 128-byte id with plh_comm_unique_id and hands it to the other
 * ranks by whatever means the host has (MPI, Distributed.jl, a file); every rank then calls plh_comm_create(n_ranks, rank, id, device).
 * plh_ensemble_run (collective: every rank calls it with the same protocol / options; theta, SOC0 and the output arrays are significant on rank 0
 * only and are HOST memory):
 *   1. ncclBroadcast of the ensemble shape, 2. scatter of the parameter rows from rank 0 (grouped ncclSend / ncclRecv; partition = contiguous blocks
 *   cells[r n/G, (r+1) n/G) or cyclic cell mod G, which evens out step-count variance in randomised sweeps), 3. plh_integrate of the local shard on the
 *   rank's GPU (no collective in the data path: cells are independent for the whole trajectory), 4. gather of the per-cell summaries (run_info,
 *   counters, optionally the final states) to rank 0 in the caller's cell order.  rank_ms[n_ranks] (rank 0, may be NULL) receives every rank's
 *   integrate-kernel time: the load-imbalance figure of a randomised sweep.
 * Per-cell protocol arrays (plh_run.value_cell / tf_cell) belong to the protocol: n_cells_total entries indexed by the GLOBAL cell, the same on every rank; each rank
 * integrates with its own shard of them.  An error on any rank (arguments, staging, its plh_integrate) is agreed on between the phases: every rank returns non-zero. */
typedef struct plh_comm_s* plh_comm_t;
#define PLH_PART_BLOCK 0
#define PLH_PART_CYCLIC 1
int plh_comm_unique_id(char id[128]);
int plh_comm_create(int n_ranks, int rank, const char id[128], int device, plh_comm_t* out);
void plh_comm_destroy(plh_comm_t c);
int plh_comm_rank(plh_comm_t c);
int plh_comm_size(plh_comm_t c);
int plh_ensemble_run(plh_comm_t c, plh_model_t m, int n_cells_total, const double* theta, const double* SOC0, int n_runs, const plh_run* runs,
                     const plh_opts* opts, int partition, plh_run_info* run_info, plh_counters* counters, double* Y_final, double* rank_ms);

#ifdef __cplusplus
}
#endif
#endif
