import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return ROOT / 'tests' / 'golden'


# Collection order of the -m gpu run (the driver uses -x): every file that compares a kernel with the oracle / the reference's golden vectors
# comes first, then the size-independent properties, then end-to-end convergence, and the beyond-§8 features (HIP-graph capture) last -- a
# failure in an auxiliary feature must never hide a SURVEY §8 row again (round 2: one flaky threshold in test_gpu_graphs.py stopped the run
# in front of 128 parity tests).
_GPU_ORDER = ('test_gpu_ngp_parity', 'test_gpu_tcnn_parity', 'test_gpu_gs_parity', 'test_gpu_render_parity', 'test_gpu_baseline_size_parity',
              'test_gpu_garden_parity', 'test_gpu_ssim_parity', 'test_gpu_knn_parity', 'test_gpu_gs_densify_parity', 'test_gpu_adam_parity',
              'test_gpu_fused_training_ops', 'test_gpu_ngp_trainer', 'test_gpu_fullsize_properties', 'test_gpu_gs_lifecycle', 'test_gpu_convergence', 'test_gpu_graphs')


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        stem = Path(str(item.fspath)).stem
        if stem in _GPU_ORDER:
            return 1 + _GPU_ORDER.index(stem)
        return 0 if not stem.startswith('test_gpu_') else len(_GPU_ORDER) - 0.5   # CPU tests stay in front; unlisted GPU files just before the graphs' slot
    items.sort(key=rank)   # stable: the order inside a file is kept
