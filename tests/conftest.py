import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _load_pkg():
    """The package directory is named after the reference (openfoam-2.2.x_amd), which is
    not an importable identifier: load it under the alias `openfoam_amd`."""
    if "openfoam_amd" in sys.modules:
        return sys.modules["openfoam_amd"]
    pkg_dir = os.path.join(ROOT, "openfoam-2.2.x_amd")
    spec = importlib.util.spec_from_file_location(
        "openfoam_amd", os.path.join(pkg_dir, "__init__.py"),
        submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["openfoam_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


_load_pkg()


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.build()
    return oracle_py
