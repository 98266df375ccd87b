import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "temp-fhe-transciphering_amd", "python"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experiments: emulator runs of generators whose kernels are NOT in the product library (measured, "
                                       "not adopted: tools/asmgen/brgen2.py, trgen2.py; lutgen.py's k_lut_eval4 lives in the experiment "
                                       "build only); deselected unless -m experiments is given")
    # the in-tree shared libraries are build artefacts (git-ignored): (re)build them when they are
    # missing or older than their sources; hipcc cross-compiles without a GPU
    import __graft_entry__
    __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


def pytest_collection_modifyitems(config, items):
    """`-m "not gpu"` and `-m gpu` are the two suites the driver runs; the experiments ride in neither (pytest -m experiments)"""
    if "experiments" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="experiment that is not in the library: run with -m experiments")
    for it in items:
        if "experiments" in it.keywords:
            it.add_marker(skip)
