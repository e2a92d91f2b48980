# PetlionHIP.jl -- thin `ccall` layer over libpetlion_hip.so (include/petlion_hip.h) for a PETLION.jl host.
#
# NOT EXECUTED in the build container (no Julia there); kept deliberately mechanical: every function is one ccall.
# Usage sketch (see INTEGRATION.md):
#
#     using PETLION, .PetlionHIP
#     p    = petlion(LCO)                                   # the stock model object: θ, N, opts, bounds
#     h    = PetlionHIP.Model(p)                            # device handle for the same structural options
#     Θ    = PetlionHIP.theta_matrix(h, p, n_cells)         # n_cells × n_theta, θ_keys order of the device library
#     Θ[:, PetlionHIP.key_index(h, :D_sp)] .*= 2 .^ (2rand(n_cells) .- 1)
#     ens  = PetlionHIP.simulate_ensemble(h, p, Θ, [(I = -1,)]; SOC = 1.0)
#     ens.t_end, ens.flag, ens.V[:, i] ...
module PetlionHIP

const lib = get(ENV, "PETLION_HIP_LIB", joinpath(@__DIR__, "..", "..", "petlion.jl_amd", "libpetlion_hip.so"))

const PLH_HOST = Cint(0)
const DSTATE = Dict(:dc_s_p_max => Cint(1), :dc_s_p_min => Cint(2), :dc_s_n_max => Cint(3), :dc_s_n_min => Cint(4), :dc_e_max => Cint(5), :dc_e_min => Cint(6))   # PLH_DSTATE_* (input_methods.jl:190-247)
const MODE = merge(Dict(:I => Cint(0), :V => Cint(1), :dT => Cint(2), :P => Cint(3), :η_p => Cint(4), :res => Cint(5)), Dict(k => Cint(6) for k in keys(DSTATE)))
const VAL_CONST, VAL_HOLD, VAL_REST, VAL_TABLE, VAL_EXPR = Cint(0), Cint(1), Cint(2), Cint(3), Cint(4)

struct ModelDesc
    chemistry::Cint; N_p::Cint; N_s::Cint; N_n::Cint; N_a::Cint; N_z::Cint; N_r_p::Cint; N_r_n::Cint
    temperature::Cint; aging_SEI::Cint; real_bytes::Cint
    precision::Cint      # 0 = fp64, 1 = mixed (fp32 storage of the Newton-matrix factors), 2 = fp64 with the finite-volume rows in the reference's operation order (PLH_PREC_F64_REFORDER)
    device::Cint         # HIP device ordinal, -1 = current
    solid_diffusion::Cint; thermodynamic_factor::Cint; rxn::Cint     # 0 Fickian FDM / 1 quadratic / 2 polynomial ; 0 linear / 1 nonlinear ; 0 BV / 1 MHC
    waves_per_cell::Cint  # 1 (default) or 2 wavefronts per cell
end
struct Bounds      # boundary_stop_conditions, src/structures.jl:237-250
    V_max::Cdouble; V_min::Cdouble; SOC_max::Cdouble; SOC_min::Cdouble; T_max::Cdouble; c_s_n_max::Cdouble
    I_max::Cdouble; I_min::Cdouble; η_plating_min::Cdouble; c_e_min::Cdouble; dfilm_max::Cdouble
end
struct Run
    mode::Cint; value_kind::Cint; value::Cdouble; tf::Cdouble; bounds::Bounds
    n_tab::Cint; closure_id::Cint                             # closure_id: written by the library (compiled closures), pass 0
    tab_t::Ptr{Cdouble}; tab_v::Ptr{Cdouble}                  # VAL_TABLE: piecewise-linear input in run-local time (host arrays)
    value_cell::Ptr{Cdouble}; tf_cell::Ptr{Cdouble}           # per-cell input value / run length ([n_cells] host arrays) or C_NULL
    n_dcol::Cint                                              # VAL_EXPR of the state: number of derivative programs (0 = none)
    dstate::Cint                                              # MODE_DSTATE: which differential state's rate is held (PLH_DSTATE_*), 0 otherwise
    dcol::Ptr{Cint}; dofs::Ptr{Cint}                          # d f / d Y[dcol[k]] = instructions dofs[k]:dofs[k+1] of tab_t / tab_v
end
struct Opts
    abstol::Cdouble; reltol::Cdouble; abstol_init::Cdouble; reltol_init::Cdouble
    maxiters::Cint; check_bounds::Cint; interp_final::Cint; max_order::Cint; jac_every_step::Cint; init_step::Cdouble
    n_tdiscon::Cint; tdiscon::Ptr{Cdouble}     # host array of any length (the caller keeps it alive across the call)
    refine::Cint                                # iterative-refinement steps per linear solve (parity mode), 0 = off
    n_tstops::Cint; tstops::Ptr{Cdouble}       # opts.tstops (run-local times the integrator must hit; src/model_evaluation.jl:292-294), host array
    yp_alg_zero::Cint                           # 1: start the integrator with YP_alg = 0 (step history of the example notebooks' package version), 0 = today's source
    n_stop::Cint; stop_ops::Ptr{Cdouble}; stop_args::Ptr{Cdouble}   # opts.stop_function as a postfix program g(t, Y, YP, θ) (PLH_OP_* opcodes / operands, host arrays): the run ends when
                                                # g > 0, exit flag 12, back-interpolated like a built-in bound (src/checks.jl:26); n_stop = 0: none.  A Julia closure is traced into
                                                # the program the same way an input closure is (closure_program below)
end
struct RunInfo
    flag::Cint; iterations::Cint; t_end::Cdouble; V::Cdouble; I::Cdouble; SOC::Cdouble; T_avg::Cdouble
end
struct Counters
    n_steps::Clonglong; n_res::Clonglong; n_jac::Clonglong; n_fact::Clonglong; n_solve::Clonglong; n_newton::Clonglong
    n_errfail::Clonglong; n_convfail::Clonglong; sum_kp2::Clonglong; n_init_iters::Clonglong; cyc::NTuple{8,Clonglong}
end
struct Outputs
    max_pts::Cint; t::Ptr{Cdouble}; V::Ptr{Cdouble}; I::Ptr{Cdouble}; SOC::Ptr{Cdouble}; T_avg::Ptr{Cdouble}
    n_pts::Ptr{Cint}; Y_final::Ptr{Cdouble}; YP_final::Ptr{Cdouble}; run_info::Ptr{RunInfo}; counters::Ptr{Counters}
    Y_all::Ptr{Cdouble}
end

lasterror() = unsafe_string(ccall((:plh_last_error, lib), Cstring, ()))
check(rc, what) = rc == 0 || error("$what failed ($rc): $(lasterror())")

mutable struct Model
    h::Ptr{Cvoid}
    N::Int; N_diff::Int; θ_keys::Vector{Symbol}
    function Model(p; precision = 0, device = -1, waves_per_cell = 1)   # p::PETLION.model -- reads only p.N and p.numerics
        N = p.N
        chem = Dict(:LCO => 0, :NMC => 1, :NMC_LGM50 => 2)[Symbol(p.numerics.cathode)]      # function name of the cathode system, as in strings_directory_func
        d = Ref(ModelDesc(chem, N.p, N.s, N.n, N.a, N.z, N.r_p, N.r_n, p.numerics.temperature == true, p.numerics.aging == :SEI, 8, precision, device,
                          Dict(:Fickian => 0, :quadratic => 1, :polynomial => 2)[p.numerics.solid_diffusion],
                          p.numerics.thermodynamic_factor === PETLION_thermodynamic_factor_nonlinear(p) ? 1 : 0, p.numerics.rxn_p === PETLION_rxn_MHC(p) ? 1 : 0, waves_per_cell))
        grid = (N.p, N.s, N.n, N.r_p, p.numerics.temperature == true ? N.a : 10, p.numerics.temperature == true ? N.z : 10)
        N.r_n == N.r_p || (grid = (grid..., N.r_n))       # N_r_p != N_r_n: the anode's N_r as a 7th entry (petlion.jl_amd/grids.py::grid7)
        grid == (10, 10, 10, 10, 10, 10) || register_grid(grid_library(grid))     # another discretisation: its kernels are a library of their own (see grid_library)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:plh_model_create, lib), Cint, (Ref{ModelDesc}, Ref{Ptr{Cvoid}}), d, h), "plh_model_create")
        n = ccall((:plh_n_theta, lib), Cint, (Ptr{Cvoid},), h[])
        keys = [Symbol(unsafe_string(ccall((:plh_theta_key, lib), Cstring, (Ptr{Cvoid}, Cint), h[], i - 1))) for i in 1:n]
        m = new(h[], ccall((:plh_n_states, lib), Cint, (Ptr{Cvoid},), h[]), ccall((:plh_n_diff, lib), Cint, (Ptr{Cvoid},), h[]), keys)
        finalizer(x -> ccall((:plh_model_destroy, lib), Cvoid, (Ptr{Cvoid},), x.h), m)
    end
end

# Kernels of another discretisation (reference src/params.jl:119-136).  The grid dimensions are compile-time constants of the device source, so a grid is one more build
# of csrc/variant_tu.hip -- the device-side counterpart of PETLION generating and caching functions per model (generate_functions.jl:44-94).  `grid_library` shells out to
# the Python driver of that build (petlion.jl_amd/grids.py: hipcc, cached under petlion.jl_amd/_grids/), which prints the library path; PETLION_HIP_ROOT = the repository root.
function grid_library(grid; variants = 0:12)
    root = get(ENV, "PETLION_HIP_ROOT", joinpath(@__DIR__, "..", ".."))
    code = "import pkgload; g = pkgload.load().grids; t = g.variant_table(); print(g.library($(grid), [v for v in $(collect(variants)) if t[v][2] == 'false']))"   # (isothermal variants)
    strip(read(setenv(`python -c $code`; dir = root), String))
end
register_grid(path::AbstractString) = check(ccall((:plh_register_grid_library, lib), Cint, (Cstring,), path), "plh_register_grid_library")

# the reference stores the closures themselves in p.numerics (src/params.jl:286); they are compared by identity with the two non-default ones the device instantiates
PETLION_thermodynamic_factor_nonlinear(p) = getfield(parentmodule(typeof(p)), :thermodynamic_factor)
PETLION_rxn_MHC(p) = getfield(parentmodule(typeof(p)), :rxn_MHC)

key_index(m::Model, k::Symbol) = findfirst(==(k), m.θ_keys)
"p.ind of the device layout: state name => 1-based index range into Y (reference state_indices, src/external.jl:275-365)"
function state_indices(m::Model)
    d = Dict{Symbol,UnitRange{Int}}()
    for i in 0:ccall((:plh_n_sections, lib), Cint, (Ptr{Cvoid},), m.h)-1
        nm = Ref{Cstring}(); a = Ref{Cint}(); l = Ref{Cint}()
        check(ccall((:plh_section, lib), Cint, (Ptr{Cvoid}, Cint, Ref{Cstring}, Ref{Cint}, Ref{Cint}), m.h, i, nm, a, l), "plh_section")
        d[Symbol(unsafe_string(nm[]))] = (a[]+1):(a[]+l[])
    end
    d
end
"n_cells × n_theta matrix (row = one cell's θ_tot, update_θ! order of src/generate_functions.jl:364-372)"
theta_matrix(m::Model, p, n_cells) = repeat(permutedims([Float64(p.θ[k]) for k in m.θ_keys]), n_cells, 1)

bounds_of(b; kw...) = Bounds((get(kw, f, getfield(b, f)) for f in (:V_max, :V_min, :SOC_max, :SOC_min, :T_max, :c_s_n_max, :I_max, :I_min, :η_plating_min, :c_e_min, :dfilm_max))...)

const KEEPALIVE = Any[]
function make_run(p, step::NamedTuple)
    name = first(k for k in keys(step) if haskey(MODE, k))
    x = step[name]
    kw = Dict(k => Float64(v) for (k, v) in pairs(step) if k ∉ (name, :tf))
    b = bounds_of(p.bounds; kw...)
    tf = Float64(get(step, :tf, 1e6))
    if x isa Tuple{Vector{Float64},Vector{Float64}}      # (t, values): a tabulated I(t) / V(t) / P(t); the caller keeps the two vectors alive (GC.@preserve)
        return Run(MODE[name], VAL_TABLE, x[2][1], tf, b, length(x[1]), 0, pointer(x[1]), pointer(x[2]), C_NULL, C_NULL, 0, 0, C_NULL, C_NULL)
    end
    if haskey(DSTATE, name)                              # the rate of one differential state held (number or :hold); the device picks the state per cell
        kind, val = x === :hold ? (VAL_HOLD, 0.0) : (VAL_CONST, Float64(x))
        return Run(MODE[name], kind, val, tf, b, 0, 0, C_NULL, C_NULL, C_NULL, C_NULL, 0, DSTATE[name], C_NULL, C_NULL)
    end
    res_x = 0.0
    if name === :res && x isa Tuple                      # res = (x, f): x - f(t, Y, YP, p) = 0 (custom_res!, model_evaluation.jl:155-172)
        res_x, x = Float64(x[1]), x[2]
    end
    if x isa Function                                    # a closure input: its expression as a postfix program (closure_program below)
        ops, args, n_main, dcol, dofs = closure_program(x, p)
        push!(KEEPALIVE, (ops, args, dcol, dofs))        # (the program arrays must outlive the call; emptied by simulate_ensemble when it returns)
        return Run(MODE[name], VAL_EXPR, res_x, tf, b, n_main, 0, pointer(ops), pointer(args), C_NULL, C_NULL, length(dcol), 0, isempty(dcol) ? C_NULL : pointer(dcol), isempty(dcol) ? C_NULL : pointer(dofs))
    end
    x isa Vector{Float64} && return Run(MODE[name], VAL_CONST, x[1], tf, b, 0, 0, C_NULL, C_NULL, pointer(x), C_NULL, 0, 0, C_NULL, C_NULL)   # one value per cell (caller keeps x alive)
    kind, val = x === :hold ? (VAL_HOLD, 0.0) : x === :rest ? (VAL_REST, 0.0) : (VAL_CONST, Float64(x))
    Run(MODE[name], kind, val, tf, b, 0, 0, C_NULL, C_NULL, C_NULL, C_NULL, 0, 0, C_NULL, C_NULL)
end

# Input closures `I = (t, Y, YP, p) -> ...` (input_methods.jl:159-176): PETLION itself traces them with Symbolics to differentiate the control row
# (scalar_residual.jl:248-274); the same traced expression, walked in post-order, is the C ABI's postfix program (PLH_VAL_EXPR, PLH_OP_* of include/petlion_hip.h).
# The Python host mirror does exactly this with operator overloading (petlion.jl_amd/closures.py, tested); this Julia version is its transliteration onto
# SymbolicUtils' expression interface (derivative programs through Symbolics.derivative, as PETLION's differentiate_residual_func does) and has not been executed.
const OPCODE = Dict(:+ => 5, :- => 6, :* => 7, :/ => 8, :sin => 10, :cos => 11, :exp => 12, :log => 13, :sqrt => 14, :^ => 15, :abs => 16, :min => 17, :max => 18,
                    :< => 19, :<= => 20, :> => 21, :>= => 22, :ifelse => 23, :tanh => 24)
function closure_program(f, p)
    S = parentmodule(typeof(p)).Symbolics; SU = S.SymbolicUtils
    θ_sym, Y, YP, t, SOC, I, γ, p_sym, θ_keys = parentmodule(typeof(p)).get_symbolic_vars(p; original_keys = p.cache.θ_keys)   # the tracers PETLION uses (scalar_residual.jl:251)
    ex = S.value(parentmodule(typeof(p)).redefine_func(f)(t, Y, YP, p_sym))
    index_of(v, vec) = findfirst(x -> isequal(S.value(x), v), vec)
    ops = Float64[]; args = Float64[]
    emit(op, a = 0.0) = (push!(ops, op); push!(args, a))
    function walk(e)
        if e isa Number
            emit(0, Float64(e))
        elseif !SU.istree(e)
            isequal(e, S.value(t)) ? emit(1) :
            (k = index_of(e, Y)) !== nothing ? emit(2, k - 1) :
            (k = index_of(e, YP)) !== nothing ? emit(3, k - 1) :
            (k = index_of(e, θ_sym)) !== nothing ? emit(4, k - 1) : error("input closure reads a symbol the device cannot resolve: $e")
        else
            op = nameof(SU.operation(e)); a = SU.arguments(e)
            if op in (:+, :*) && length(a) > 2                    # n-ary sums / products: left fold
                walk(a[1]); for x in a[2:end]; walk(x); emit(OPCODE[op]); end
            elseif op === :- && length(a) == 1
                walk(a[1]); emit(9)
            else
                foreach(walk, a); emit(OPCODE[op])
            end
        end
    end
    walk(ex)
    n_main = length(ops)
    # derivative programs of the control row (PETLION: sparsejacobian of the traced row with respect to Y and YP, scalar_residual.jl:289-291): one program per column the
    # closure reads, behind the main program in the same arrays; column k-1 for Y[k], N + k - 1 for YP[k] of a differential state (include/petlion_hip.h, PLH_VAL_EXPR).
    # A closure that reads YP of an algebraic state keeps PETLION's no-differentiation fallback (n_dcol = 0).
    N = length(Y); Nd = p.N.diff
    vars = S.get_variables(ex)
    dcol = Cint[]; dofs = Cint[n_main]
    reads_alg_yp = any(k -> any(v -> isequal(v, S.value(YP[k])), vars), Nd+1:N)
    if !reads_alg_yp
        for (vec, off) in ((Y, 0), (YP, N)), k in 1:(off == 0 ? N : Nd)
            any(v -> isequal(v, S.value(vec[k])), vars) || continue
            d = S.value(S.simplify(S.derivative(ex, vec[k])))
            (d isa Number && iszero(d)) && continue
            walk(d); push!(dcol, off + k - 1); push!(dofs, length(ops))
        end
    end
    ops, args, n_main, dcol, dofs
end

"""
    simulate_ensemble(m, p, Θ, protocol; SOC=p.opts.SOC, max_pts=2048)

`protocol` = vector of NamedTuples, each the keyword set of one `simulate`/`simulate!` call, e.g.
`[(I = 2, tf = 1800, V_max = 4.1), (V = :hold, I_min = 1/20)]`.  Θ is n_cells × n_theta (row-major is what the C side wants, so
the transposed copy is passed).
"""
function simulate_ensemble(m::Model, p, Θ::Matrix{Float64}, protocol; SOC = p.opts.SOC, max_pts = 2048, Y_init = nothing, t_init = nothing,
                           outputs = p.opts.outputs, refine = 0)
    n = size(Θ, 1)
    runs = [make_run(p, s) for s in protocol]
    o = p.opts
    td = Float64.(o.tdiscon); ts = Float64.(o.tstops)
    opts = Ref(Opts(o.abstol, o.reltol, o.abstol, o.reltol, o.maxiters, o.check_bounds, o.interp_final, 5, 0, 0.0,
                    length(td), isempty(td) ? C_NULL : pointer(td), refine, length(ts), isempty(ts) ? C_NULL : pointer(ts), 0, 0, C_NULL, C_NULL))
    Θt = permutedims(Θ)                                   # column-major n_theta × n_cells == row-major cells
    soc = SOC isa Number ? fill(Float64(SOC), n) : Vector{Float64}(SOC)
    t = zeros(max_pts, n); V = similar(t); I = similar(t); S = similar(t)
    npts = zeros(Cint, n); Y = zeros(m.N, n); YP = zeros(m.N, n)
    info = Matrix{RunInfo}(undef, length(runs), n); cnt = Vector{Counters}(undef, n)
    outs = outputs isa Symbol ? (outputs,) : outputs
    keep_Y = any(x -> x ∈ (:all, :Y, :c_e, :c_s_avg, :T, :film, :SOH, :j, :j_s, :Φ_e, :Φ_s), outs)     # solution_states_logic, src/outputs.jl:107-131
    Tavg = p.numerics.temperature ? zeros(max_pts, n) : Float64[]
    Yall = keep_Y ? zeros(m.N, max_pts, n) : Float64[]                  # sol.Y of every cell: Yall[:, k, i] = state after step k of cell i
    GC.@preserve t V I S npts Y YP info cnt Tavg Yall td ts begin
        out = Ref(Outputs(max_pts, pointer(t), pointer(V), pointer(I), pointer(S), isempty(Tavg) ? C_NULL : pointer(Tavg), pointer(npts), pointer(Y), pointer(YP),
                          pointer(info), pointer(cnt), keep_Y ? pointer(Yall) : C_NULL))
        rc = ccall((:plh_integrate, lib), Cint,
                   (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Run}, Ref{Opts}, Ref{Outputs}, Cint, Ptr{Cvoid}),
                   m.h, n, Θt, soc, Y_init === nothing ? C_NULL : pointer(Y_init), t_init === nothing ? C_NULL : pointer(t_init),
                   length(runs), runs, opts, out, PLH_HOST, C_NULL)
        check(rc, "plh_integrate")
    end
    (t = t, V = V, I = I, SOC = S, T_avg = Tavg, Y_all = Yall, n_pts = npts, Y = Y, YP = YP, run_info = info, counters = cnt,
     flag = [info[end, i].flag for i in 1:n], t_end = [info[end, i].t_end for i in 1:n])
end

"""
    simulate_ensemble_sens(m, p, Θ, protocol, keys; SOC, max_pts)

`simulate_ensemble` with forward sensitivities (`plh_integrate_sens`): `keys` = the θ Symbols to differentiate (entries of `m.θ_keys`).  Adds
`dY_dθ[:, k, i]` = ∂Y/∂θ[keys[k]] of cell i at the end of the protocol and `dV_dθ[:, k, i]` = ∂V/∂θ[keys[k]] at every saved point -- the Jacobian of the voltage curve a
least-squares fit of `keys` needs, from ONE ensemble call instead of 2 length(keys) + 1.  Constant / `:rest` inputs only; the states are bit for bit those of
`simulate_ensemble`.  Derivatives are with respect to the absolute parameter value, at fixed time.
"""
function simulate_ensemble_sens(m::Model, p, Θ::Matrix{Float64}, protocol, keys::Vector{Symbol}; SOC = p.opts.SOC, max_pts = 2048)
    n = size(Θ, 1)
    runs = [make_run(p, s) for s in protocol]
    o = p.opts
    ts = Float64.(o.tstops)
    opts = Ref(Opts(o.abstol, o.reltol, o.abstol, o.reltol, o.maxiters, o.check_bounds, o.interp_final, 5, 0, 0.0, 0, C_NULL, 0, length(ts), isempty(ts) ? C_NULL : pointer(ts), 0, 0, C_NULL, C_NULL))
    cols = Cint[key_index(m, k) - 1 for k in keys]
    ns = length(cols)
    Θt = permutedims(Θ)
    soc = SOC isa Number ? fill(Float64(SOC), n) : Vector{Float64}(SOC)
    t = zeros(max_pts, n); V = similar(t); I = similar(t); S = similar(t)
    npts = zeros(Cint, n); Y = zeros(m.N, n)
    info = Matrix{RunInfo}(undef, length(runs), n); cnt = Vector{Counters}(undef, n)
    Tavg = p.numerics.temperature ? zeros(max_pts, n) : Float64[]
    dY = zeros(m.N, ns, n); dV = zeros(max_pts, ns, n); stat = zeros(Cint, 3, n)
    GC.@preserve t V I S npts Y info cnt Tavg ts cols dY dV stat begin
        out = Ref(Outputs(max_pts, pointer(t), pointer(V), pointer(I), pointer(S), isempty(Tavg) ? C_NULL : pointer(Tavg), pointer(npts), pointer(Y), C_NULL,
                          pointer(info), pointer(cnt), C_NULL))
        rc = ccall((:plh_integrate_sens, lib), Cint,
                   (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Run}, Ref{Opts}, Ref{Outputs}, Cint, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}, Cint, Ptr{Cvoid}),
                   m.h, n, Θt, soc, length(runs), runs, opts, out, ns, cols, dY, dV, stat, PLH_HOST, C_NULL)
        check(rc, "plh_integrate_sens")
    end
    (t = t, V = V, I = I, SOC = S, T_avg = Tavg, n_pts = npts, Y = Y, run_info = info, counters = cnt, dY_dθ = dY, dV_dθ = dV, sens_stat = stat)
end

# ---- seam 1: the five generated functions of p.funcs (src/structures.jl:315-334) as single-cell evaluators ----
function residual!(res::Vector{Float64}, m::Model, Y, YP, θ; mode = :I, value = 0.0)
    check(ccall((:plh_residual, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{Cdouble}, Cint, Ptr{Cvoid}),
                m.h, 1, θ, Y, YP, MODE[mode], value, res, PLH_HOST, C_NULL), "plh_residual")
    res
end
# f_diff!(out[N_diff], t, Y, YP, θ) / f_alg!(out[N_alg-1], t, Y, YP, θ): `out` may be a contiguous range view of `res` (scalar_residual.jl:559-560)
function f_diff!(out, m::Model, Y, YP, θ)
    check(ccall((:plh_residual_diff, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cvoid}),
                m.h, 1, θ, Y, YP, out, PLH_HOST, C_NULL), "plh_residual_diff")
    nothing
end
function f_alg!(out, m::Model, Y, YP, θ)
    check(ccall((:plh_residual_alg, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cvoid}),
                m.h, 1, θ, Y, YP, out, PLH_HOST, C_NULL), "plh_residual_alg")
    nothing
end
function J_full!(nzval::Vector{Float64}, m::Model, Y, YP, γ, θ; mode = :I)
    check(ccall((:plh_jacobian, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cint, Ptr{Cdouble}, Cint, Ptr{Cvoid}),
                m.h, 1, θ, Y, YP, γ, MODE[mode], nzval, PLH_HOST, C_NULL), "plh_jacobian")
    nzval
end
"J_y_alg! (generate_functions.jl:318-325): the block J[N_diff+1:N-1, N_diff+1:N] of the full Jacobian at γ = 0, in the CSC order of that block"
function J_alg!(nzval::Vector{Float64}, m::Model, Y, YP, θ; mode = :I)
    check(ccall((:plh_jacobian_alg, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cint, Ptr{Cvoid}),
                m.h, 1, θ, Y, YP, MODE[mode], nzval, PLH_HOST, C_NULL), "plh_jacobian_alg")
    nzval
end
"0-based CSC pattern (colptr, rowval) of the J_y_alg block: N_alg columns, rows relative to N_diff"
function jac_alg_pattern(m::Model; mode = :I)
    nnz = Ref{Cint}(0)
    ccall((:plh_jac_alg_pattern, lib), Cint, (Ptr{Cvoid}, Cint, Ref{Cint}, Ptr{Cint}, Ptr{Cint}), m.h, MODE[mode], nnz, C_NULL, C_NULL)
    cp = zeros(Cint, m.N - m.N_diff + 1); ri = zeros(Cint, nnz[])
    check(ccall((:plh_jac_alg_pattern, lib), Cint, (Ptr{Cvoid}, Cint, Ref{Cint}, Ptr{Cint}, Ptr{Cint}), m.h, MODE[mode], nnz, cp, ri), "plh_jac_alg_pattern")
    cp, ri
end
function initial_guess!(out::Vector{Float64}, m::Model, SOC, θ)
    check(ccall((:plh_initial_guess, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ref{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cvoid}), m.h, 1, θ, SOC, out, PLH_HOST, C_NULL), "plh_initial_guess")
    out
end
"0-based CSC pattern (colptr, rowval) of [J_y_sp ; control row] for a mode"
function jac_pattern(m::Model; mode = :I)
    nnz = Ref{Cint}(0)
    ccall((:plh_jac_pattern, lib), Cint, (Ptr{Cvoid}, Cint, Ref{Cint}, Ptr{Cint}, Ptr{Cint}), m.h, MODE[mode], nnz, C_NULL, C_NULL)
    cp = zeros(Cint, m.N + 1); ri = zeros(Cint, nnz[])
    check(ccall((:plh_jac_pattern, lib), Cint, (Ptr{Cvoid}, Cint, Ref{Cint}, Ptr{Cint}, Ptr{Cint}), m.h, MODE[mode], nnz, cp, ri), "plh_jac_pattern")
    cp, ri
end

# ---- multi-GPU: one Julia process per GPU (Distributed.jl / MPI.jl workers), RCCL inside the library for the ensemble scatter / gather ----
#     id = myid() == root ? PetlionHIP.unique_id() : nothing          # 128 bytes, handed to the other workers by the host's own means
#     comm = PetlionHIP.Comm(n_ranks, rank, id; device = rank)        # every worker
#     info, cnt, ms = PetlionHIP.ensemble_run(comm, m, p, Θ, protocol; SOC = 1.0, partition = :cyclic)    # Θ significant on rank 0
unique_id() = (id = zeros(UInt8, 128); check(ccall((:plh_comm_unique_id, lib), Cint, (Ptr{UInt8},), id), "plh_comm_unique_id"); id)
mutable struct Comm
    h::Ptr{Cvoid}
    function Comm(n_ranks, rank, id::Vector{UInt8}; device = -1)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:plh_comm_create, lib), Cint, (Cint, Cint, Ptr{UInt8}, Cint, Ref{Ptr{Cvoid}}), n_ranks, rank, id, device, h), "plh_comm_create")
        c = new(h[]); finalizer(x -> ccall((:plh_comm_destroy, lib), Cvoid, (Ptr{Cvoid},), x.h), c)
    end
end
comm_rank(c::Comm) = ccall((:plh_comm_rank, lib), Cint, (Ptr{Cvoid},), c.h)
comm_size(c::Comm) = ccall((:plh_comm_size, lib), Cint, (Ptr{Cvoid},), c.h)
function ensemble_run(c::Comm, m::Model, p, Θ::Matrix{Float64}, protocol; SOC = p.opts.SOC, partition = :block, n_cells = size(Θ, 1), refine = 0)
    runs = [make_run(p, s) for s in protocol]
    o = p.opts; td = Float64.(o.tdiscon); ts = Float64.(o.tstops)
    root = comm_rank(c) == 0
    Θt = root ? permutedims(Θ) : zeros(0, 0)
    soc = root ? (SOC isa Number ? fill(Float64(SOC), n_cells) : Vector{Float64}(SOC)) : Float64[]
    info = Matrix{RunInfo}(undef, length(runs), root ? n_cells : 0); cnt = Vector{Counters}(undef, root ? n_cells : 0); ms = zeros(comm_size(c))
    GC.@preserve td ts Θt soc info cnt ms begin
        opts = Ref(Opts(o.abstol, o.reltol, o.abstol, o.reltol, o.maxiters, o.check_bounds, o.interp_final, 5, 0, 0.0,
                        length(td), isempty(td) ? C_NULL : pointer(td), refine, length(ts), isempty(ts) ? C_NULL : pointer(ts), 0, 0, C_NULL, C_NULL))
        check(ccall((:plh_ensemble_run, lib), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Run}, Ref{Opts}, Cint, Ptr{RunInfo}, Ptr{Counters}, Ptr{Cdouble}, Ptr{Cdouble}),
                    c.h, m.h, n_cells, root ? pointer(Θt) : C_NULL, root ? pointer(soc) : C_NULL, length(runs), runs, opts, partition === :cyclic ? 1 : 0,
                    root ? pointer(info) : C_NULL, root ? pointer(cnt) : C_NULL, C_NULL, root ? pointer(ms) : C_NULL), "plh_ensemble_run")
    end
    info, cnt, ms
end

end # module
