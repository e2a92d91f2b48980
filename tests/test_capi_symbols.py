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
