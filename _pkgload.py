"""Registers the `seed-vc_amd/` directory as the importable package `seedvc_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def load_package():
    if "seedvc_amd" in sys.modules:
        return sys.modules["seedvc_amd"]
    pkg_dir = os.path.join(ROOT, "seed-vc_amd")
    spec = importlib.util.spec_from_file_location(
        "seedvc_amd", os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["seedvc_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
