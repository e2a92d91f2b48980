# SavedModelWriter.jl -- seam 1 made turnkey (SURVEY.md 8(f).2): write the reference's generated-function cache
#
#     saved_models/<cathode>_<anode>/<sha1 of the options string>/{info.txt, initial_guess.jl, f_alg.jl, f_diff.jl, J_y.jl, J_y_alg.jl, J_sp.jl}
#
# so that a stock `petlion(LCO; ...)` finds it (load_functions_symbolic, src/generate_functions.jl:44-94) and every later
# `simulate` runs the reference's own IDA/KLU on the evaluators of libpetlion_hip.so.  This is correctness plumbing (config C1): a GPU launch
# per 2.4 kB call is slow by construction; ensembles go through PetlionHIP.simulate_ensemble instead.
#
# NOT EXECUTED in the build container (no Julia there).  What it relies on, all checked against the reference source:
#   * petlion(...; load_funcs = false) builds the model object without generating functions           (src/external.jl:3-33,54)
#   * PETLION.strings_directory_func(p; create_dir = true) names and creates the cache directory      (src/external.jl:417-460)
#   * PETLION.model_info(p) is the text of info.txt; its first line carries the version check          (src/external.jl:395-415, generate_functions.jl:21-32)
#   * the five files must `include` to callables with the positional signatures of build_function      (generate_functions.jl:124,279-280,293-303,313-322)
#   * "J_sp.jl" is a BSON file holding `J_y_sp = (I, J, V, N-1, N)` and `θ_keys`                         (generate_functions.jl:72,135,157)
# The stubs are self-contained (Base.Libc.Libdl + ccall through function pointers), so PETLION needs no new dependency; each file owns one
# device handle for the lifetime of the Julia session.
#
#     include("bindings/julia/PetlionHIP.jl"); include("bindings/julia/SavedModelWriter.jl")
#     using PETLION
#     SavedModelWriter.install(LCO; temperature = false)      # once per structural option set
#     p = petlion(LCO)                                        # now loads the stubs instead of running Symbolics
module SavedModelWriter

using ..PetlionHIP
import PETLION, BSON

const RESIDUAL_SIG = "(Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cdouble, Ptr{Cdouble}, Cint, Ptr{Cvoid})"
const JACOBIAN_SIG = "(Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cint, Ptr{Cdouble}, Cint, Ptr{Cvoid})"

"text shared by the five files: open the library, create the handle for the same structural options"
function prelude(libpath, desc::NTuple{17,Int})
    """
    let
        dl = Base.Libc.Libdl.dlopen($(repr(abspath(libpath))))
        sym(s) = Base.Libc.Libdl.dlsym(dl, s)
        lasterr() = unsafe_string(ccall(sym(:plh_last_error), Cstring, ()))
        desc = Ref{NTuple{17,Cint}}(Cint.($(desc)))          # plh_model_desc: seventeen ints (include/petlion_hip.h)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        ccall(sym(:plh_model_create), Cint, (Ref{NTuple{17,Cint}}, Ref{Ptr{Cvoid}}), desc, h) == 0 || error("plh_model_create: " * lasterr())
        N = Int(ccall(sym(:plh_n_states), Cint, (Ptr{Cvoid},), h[])); Nd = Int(ccall(sym(:plh_n_diff), Cint, (Ptr{Cvoid},), h[]))
        f_res = sym(:plh_residual); f_jac = sym(:plh_jacobian); f_guess = sym(:plh_initial_guess)
        buf = zeros(N)
    """
end

# rows of the residual / entries of the Jacobian are evaluated in mode I (= 0) with value 0: only the control row depends on the mode, and the
# five generated functions never contain it (generate_functions.jl:254: the last row is removed before build_function)
residual_stub(pre, rows) = pre * """
        (out, t, Y, YP, θ) -> begin
            ccall(f_res, Cint, $RESIDUAL_SIG, h[], 1, θ, Y, YP, 0, 0.0, buf, 0, C_NULL) == 0 || error("plh_residual: " * lasterr())
            @inbounds for (k, r) in enumerate($rows); out[k] = buf[r]; end
            nothing
        end
    end
    """

# `sel` = positions, in the N x N CSC nzval of mode I, of the entries this function owns (the control row is skipped).  nzval is a gather
# view in the reference (scalar_residual.jl:510-517), so the device result goes through a dense temporary.
jacobian_stub(pre, nnz_full, sel, takes_gamma) = pre * """
        full = zeros($nnz_full)
        sel = $(repr(sel))
        (nzval, t, Y, YP, γ, θ) -> begin
            ccall(f_jac, Cint, $JACOBIAN_SIG, h[], 1, θ, Y, YP, $(takes_gamma ? "Float64(γ)" : "0.0"), 0, full, 0, C_NULL) == 0 || error("plh_jacobian: " * lasterr())
            @inbounds for (k, q) in enumerate(sel); nzval[k] = full[q]; end
            nothing
        end
    end
    """

guess_stub(pre) = pre * """
        (out, SOC, θ, X_applied) -> begin                     # out has N-1 entries: I is not touched (states_definition.jl:80-121)
            ccall(f_guess, Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ref{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cvoid}), h[], 1, θ, Float64(SOC), buf, 0, C_NULL) == 0 || error("plh_initial_guess: " * lasterr())
            @inbounds for k in 1:N-1; out[k] = buf[k]; end
            nothing
        end
    end
    """

"""
    install(cathode = PETLION.LCO; libpath = PetlionHIP.lib, kwargs...)

Write the cache for `petlion(cathode; kwargs...)`.  Returns the directory.
"""
function install(cathode = PETLION.LCO; libpath = PetlionHIP.lib, kwargs...)
    p0 = PETLION.petlion(cathode; load_funcs = false, kwargs...)
    m = PetlionHIP.Model(p0)
    N, Nd = m.N, m.N_diff
    cp, ri = PetlionHIP.jac_pattern(m; mode = :I)             # 0-based CSC of [J_y_sp ; control row]
    I = Int64[]; J = Int64[]; sel = Int[]; sel_alg = Int[]
    for c in 1:N, q in cp[c]+1:cp[c+1]
        r = ri[q] + 1
        r == N && continue                                    # control row: scalar_jacobian! owns it (scalar_residual.jl:174-229)
        push!(I, r); push!(J, c); push!(sel, q)
        (r > Nd && c > Nd) && push!(sel_alg, q)               # J_y_alg! = nzval of Jac[N_diff+1:end, N_diff+1:end] (generate_functions.jl:318-325)
    end
    J_y_sp = (I, J, ones(Float64, length(I)), N - 1, N)       # findnz(J_sp)..., N-1, N (generate_functions.jl:135,265)
    θ_keys = copy(m.θ_keys)                                   # update_θ! fills θ_tot in this order (generate_functions.jl:364-372)

    n = p0.N
    desc = (Dict(:LCO => 0, :NMC => 1, :NMC_LGM50 => 2)[Symbol(p0.numerics.cathode)], n.p, n.s, n.n, n.a, n.z, n.r_p, n.r_n,
            Int(p0.numerics.temperature == true), Int(p0.numerics.aging == :SEI), 8, 0, -1,     # fp64, current device
            Dict(:Fickian => 0, :quadratic => 1, :polynomial => 2)[p0.numerics.solid_diffusion],
            Int(p0.numerics.thermodynamic_factor === PETLION.thermodynamic_factor), Int(p0.numerics.rxn_p === PETLION.rxn_MHC), 1)
    pre = prelude(libpath, desc)
    dir = PETLION.strings_directory_func(p0; create_dir = true) * "/"
    write(dir * "info.txt", PETLION.model_info(p0))
    write(dir * "initial_guess.jl", guess_stub(pre))
    write(dir * "f_diff.jl", residual_stub(pre, "1:Nd"))
    write(dir * "f_alg.jl", residual_stub(pre, "Nd+1:N-1"))
    write(dir * "J_y.jl", jacobian_stub(pre, length(ri), sel, true))
    write(dir * "J_y_alg.jl", jacobian_stub(pre, length(ri), sel_alg, true))
    BSON.@save dir * "J_sp.jl" J_y_sp θ_keys
    dir
end

end # module
