import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "wave_emu"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import pkgload
    return pkgload.load()


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def emu_model(pkg):
    """the product's device + host source compiled against the TEST-ONLY wave emulator (CPU debugging aid)."""
    import build_emu
    return pkg.petlion(pkg.LCO, _lib_path=build_emu.build())


@pytest.fixture(scope="session")
def hip_model(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    if not os.environ.get("PETLION_HIP_LIB"):          # (an experiment library selected through the environment is used as it is: tools/experiments/)
        g.build_hip()
    return pkg.petlion(pkg.LCO)


@pytest.fixture(scope="session")
def emu_model_nmc(pkg):
    import build_emu
    return pkg.petlion(pkg.NMC, _lib_path=build_emu.build())


@pytest.fixture(scope="session")
def hip_model_nmc(pkg, hip_model):
    return pkg.petlion(pkg.NMC)


@pytest.fixture(scope="session")
def emu_model_sei(pkg):
    import build_emu
    return pkg.petlion(pkg.LCO, aging="SEI", _lib_path=build_emu.build())


@pytest.fixture(scope="session")
def emu_model_nmc_sei(pkg):
    import build_emu
    return pkg.petlion(pkg.NMC, aging="SEI", _lib_path=build_emu.build())


@pytest.fixture(scope="session")
def hip_model_sei(pkg, hip_model):
    return pkg.petlion(pkg.LCO, aging="SEI")


@pytest.fixture(scope="session")
def hip_model_nmc_sei(pkg, hip_model):
    return pkg.petlion(pkg.NMC, aging="SEI")


@pytest.fixture(scope="session")
def emu_model_thermal(pkg):
    import build_emu
    return pkg.petlion(pkg.LCO, temperature=True, _lib_path=build_emu.build())


@pytest.fixture(scope="session")
def hip_model_thermal(pkg, hip_model):
    return pkg.petlion(pkg.LCO, temperature=True)


@pytest.fixture(scope="session")
def emu_model_lgm50_thermal(pkg):
    import build_emu
    return pkg.petlion(pkg.NMC_LGM50, _lib_path=build_emu.build())          # (temperature = true is this chemistry's default)


@pytest.fixture(scope="session")
def hip_model_lgm50_thermal(pkg, hip_model):
    return pkg.petlion(pkg.NMC_LGM50)


F4_OPTIONS = {"quad": dict(solid_diffusion="quadratic"), "poly": dict(solid_diffusion="polynomial"), "nu": dict(thermodynamic_factor="nonlinear"),
              "mhc": dict(rxn_p="MHC", rxn_n="MHC"), "lgm50": dict(cathode="LGM50", temperature=False)}


@pytest.fixture(scope="session")
def emu_models_f4(pkg):
    """SURVEY 8(f).4 model variants on the wave emulator: quadratic / polynomial solid diffusion, nonlinear thermodynamic factor, MHC kinetics"""
    import build_emu
    return {k: pkg.petlion(kw.get("cathode", pkg.LCO), _lib_path=build_emu.build(), **{a: b for a, b in kw.items() if a != "cathode"}) for k, kw in F4_OPTIONS.items()}


@pytest.fixture(scope="session")
def hip_models_f4(pkg, hip_model):
    return {k: pkg.petlion(kw.get("cathode", pkg.LCO), **{a: b for a, b in kw.items() if a != "cathode"}) for k, kw in F4_OPTIONS.items()}
