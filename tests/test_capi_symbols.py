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
        pkg.petlion(pkg.LCO, solid_diffusion="polynomial")
    with pytest.raises(NotImplementedError):
        pkg.petlion("LFP")
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
