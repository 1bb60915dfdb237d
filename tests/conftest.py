import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The driver runs `pytest -m gpu -x`: one failure hides everything collected behind it.  So the per-row oracle tests gate first (neighbour table, flags, row sets,
# the 29 partials, the normal equations, one optimisation, C1), then the bit-exact stages (levels, fusion, mesh), then the bench-depth and ladder tests, the edge
# cases and the sharded paths, and the long end-to-end schedules (C2 / C3 / C5) LAST — alphabetical collection had them first (round-5 review).
_GPU_ORDER = ["test_gpu_parity", "test_gpu_levels", "test_gpu_fusion", "test_gpu_mesh", "test_gpu_bench_parity", "test_gpu_ladder", "test_gpu_edge_cases",
              "test_gpu_cull", "test_gpu_multi_device", "test_gpu_loader", "test_gpu_configs"]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _GPU_ORDER.index(mod) if mod in _GPU_ORDER else -1          # CPU modules keep their place in front
    items.sort(key=rank)                                                   # stable: the order inside a module is the file's


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure).  Built on demand with g++."""
    from oracle import oracle_py
    oracle_py.build()
    oracle_py.lib()
    return oracle_py
