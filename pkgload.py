"""Import helper: the package directory is literally `petlion.jl_amd/` (the repo's name), which is not a valid dotted
module path, so it is registered under the importable alias `petlion_jl_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
ALIAS = "petlion_jl_amd"


def load():
    if ALIAS in sys.modules:
        return sys.modules[ALIAS]
    pkg_dir = os.path.join(ROOT, "petlion.jl_amd")
    spec = importlib.util.spec_from_file_location(ALIAS, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod
