import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def kats():
    return dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_kats.npz')))


@pytest.fixture(scope='session')
def lih_walker(kats):
    """The reference's canonical LiH test walker, recovered from the 'ne' edges golden:
    ne[0] = r - R_Li with R_Li = 0."""
    return np.asarray(kats['lih_edges_ne'][0], np.float64)


@pytest.fixture(autouse=True)
def _gpu_memory_hygiene(request):
    """GPU tests build contexts with workspaces of tens of GB (32 GB budget per context plus the float64 twin).  Python frees
    them by reference count, but a context caught in a reference cycle -- or in the frames pytest keeps of a failed test --
    stays until the collector runs: collect before every GPU test, and log the free device memory (gpurun_out/gpu_mem.log)
    so that a leak shows up as a trend instead of as an out-of-memory error ten tests later."""
    if request.node.get_closest_marker('gpu') is None:
        yield
        return
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            free, total = torch.cuda.mem_get_info()
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            with open(os.path.join(ROOT, 'gpurun_out', 'gpu_mem.log'), 'a') as f:
                f.write(f'{free / 2 ** 30:8.1f} GiB free of {total / 2 ** 30:.0f}  before {request.node.nodeid}\n')
    except Exception:       # noqa: BLE001 -- bookkeeping only
        pass
    yield
    gc.collect()
