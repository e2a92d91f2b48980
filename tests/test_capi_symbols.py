"""The HIP C-ABI library loads on a machine without a GPU and exports every symbol include/petlion_hip.h declares;
with no GPU visible the product path fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "petlion_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(plh_[a-z_0-9]+)\s*\(", txt)))


def test_exports_match_header(pkg):
    import __graft_entry__ as g
    g.build_hip()
    lib = pkg._capi.load()
    syms = declared_symbols()
    assert len(syms) >= 16
    for s in syms:
        getattr(lib, s)
    assert sorted(pkg._capi.EXPORTS) == syms


def test_fails_loudly_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg._capi.PetlionHipError) as e:
        pkg.petlion(pkg.LCO)
    assert "no HIP device" in str(e.value) or "HIP" in str(e.value)


def test_unsupported_options_are_refused(pkg):
    with pytest.raises(NotImplementedError):
        pkg.petlion(pkg.LCO, Fickian_method="spectral")
    with pytest.raises(NotImplementedError):
        pkg.petlion(pkg.LCO, rxn_p="MHC")                     # rxn_n must follow (one kinetics for both electrodes is what is instantiated)
    with pytest.raises(NotImplementedError):
        pkg.petlion("LFP")
    with pytest.raises(NotImplementedError):
        pkg.petlion(pkg.LCO, solid_diffusion="Fickian", Fickian_method="spectral")
    with pytest.raises(NotImplementedError):
        pkg.petlion(pkg.LCO, aging="R_film")


def test_julia_binding_matches_header(pkg):
    """bindings/julia/*.jl cannot be executed here (no Julia): check statically that every `:plh_*` symbol they ccall is declared in the header and
    that the Julia mirror structs have as many fields as the C structs they alias (a missing field would shift every later pointer)."""
    jl = "".join(open(os.path.join(ROOT, "bindings", "julia", f)).read() for f in ("PetlionHIP.jl", "SavedModelWriter.jl"))
    used = set(re.findall(r":(plh_[a-z_0-9]+)", jl))
    assert used and used <= set(declared_symbols()), used - set(declared_symbols())
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "petlion_hip.h")).read(), flags=re.S)

    def c_fields(name):
        body = re.search(r"typedef struct \{([^}]*)\} %s;" % name, hdr).group(1)
        n = 0
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                n += len(decl.split(","))
        return n

    def jl_fields(name):
        body = re.search(r"struct %s\b[^\n]*\n(.*?)\nend" % name, jl, flags=re.S).group(1)
        body = re.sub(r"#.*", "", body)
        return len(re.findall(r"::", body))

    for cname, jname in (("plh_model_desc", "ModelDesc"), ("plh_bounds", "Bounds"), ("plh_run", "Run"), ("plh_opts", "Opts"),
                         ("plh_run_info", "RunInfo"), ("plh_counters", "Counters"), ("plh_outputs", "Outputs")):
        assert c_fields(cname) == jl_fields(jname), (cname, c_fields(cname), jl_fields(jname))
    # the ctypes mirrors used by the tests follow the same rule
    cap = pkg._capi
    for cname, st in (("plh_model_desc", cap.ModelDesc), ("plh_run", cap.Run), ("plh_opts", cap.Opts), ("plh_outputs", cap.Outputs)):
        assert c_fields(cname) == len(st._fields_), cname


JL_SIZES = {"Cint": (4, 4), "Cdouble": (8, 8), "Clonglong": (8, 8)}


def _jl_layout(jl, name, known):
    """(size, align, [field offsets]) of a Julia `struct` of isbits fields under the C layout rules Julia uses for ccall."""
    body = re.search(r"struct %s\b[^\n]*\n(.*?)\nend" % name, jl, flags=re.S).group(1)
    body = re.sub(r"#.*", "", body)
    off, offs, amax = 0, [], 1
    for ty in re.findall(r"::\s*([A-Za-z_]+(?:\{[^;\n]*?\})?)\s*(?:;|\n|$)", body):
        m = re.match(r"NTuple\{(\d+),\s*(\w+)\}", ty)
        if ty.startswith("Ptr{") or ty == "Cstring":
            sz, al = 8, 8
        elif m:
            esz, al = JL_SIZES[m.group(2)]
            sz = int(m.group(1)) * esz
        elif ty in JL_SIZES:
            sz, al = JL_SIZES[ty]
        else:
            sz, al, _ = known[ty]
        off = (off + al - 1) // al * al
        offs.append(off)
        off += sz
        amax = max(amax, al)
    return (off + amax - 1) // amax * amax, amax, offs


def test_struct_layouts_match_the_compiled_library(pkg):
    """plh_abi_layout() reports sizeof/offsetof of every struct as compiled into the library; the hand-written mirrors -- the Julia structs of
    bindings/julia/PetlionHIP.jl (laid out here by C rules from their field types) and the ctypes structs of the Python host -- must agree field by field."""
    import __graft_entry__ as g
    g.build_hip()
    lib = pkg._capi.load()
    n = lib.plh_abi_layout(None, 0)
    buf = (C.c_int * n)()
    assert lib.plh_abi_layout(buf, n) == n
    vals, k, layouts = list(buf), 0, []
    for _ in range(7):
        size, nf = vals[k], vals[k + 1]
        layouts.append((size, vals[k + 2:k + 2 + nf])); k += 2 + nf
    assert k == n
    jl = open(os.path.join(ROOT, "bindings", "julia", "PetlionHIP.jl")).read()
    known = {}
    cap = pkg._capi
    ct = {"ModelDesc": cap.ModelDesc, "Bounds": cap.Bounds, "Run": cap.Run, "Opts": cap.Opts, "RunInfo": cap.RunInfo, "Counters": cap.CountersS, "Outputs": cap.Outputs}
    for (size, offs), jname in zip(layouts, ("ModelDesc", "Bounds", "Run", "Opts", "RunInfo", "Counters", "Outputs")):
        jsize, jal, joffs = _jl_layout(jl, jname, known)
        known[jname] = (jsize, jal, joffs)
        assert (jsize, joffs) == (size, list(offs)), (jname, jsize, size, joffs, list(offs))
        st = ct[jname]
        assert C.sizeof(st) == size and [getattr(st, f).offset for f, _ in st._fields_] == list(offs), jname
    assert pkg._capi.RUN_INFO_DTYPE.itemsize == layouts[4][0] and pkg._capi.COUNTERS_DTYPE.itemsize == layouts[5][0]
