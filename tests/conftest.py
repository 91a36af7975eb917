import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


# every plan the suites submit is validated on the host before it is uploaded (engine.hip check_plan)
os.environ.setdefault("IQHIP_CHECK_PLAN", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """Import iq-tree_amd/ (hyphenated directory) as module `iqtree_amd`."""
    if "iqtree_amd" in sys.modules:
        return sys.modules["iqtree_amd"]
    pkg_dir = os.path.join(ROOT, "iq-tree_amd")
    spec = importlib.util.spec_from_file_location(
        "iqtree_amd", os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["iqtree_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def synth(pkg):
    import importlib
    return importlib.import_module("iqtree_amd.synth")


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_driver
    oracle_driver.lib()
    return oracle_driver


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
