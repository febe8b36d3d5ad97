import os
import sys

import numpy as np
import pytest

# the test harness is the host here: 16 hardware queues for the multi-stream tests, as INTEGRATION.md G tells a host to arrange
# (librbq.so itself no longer touches the environment; rbq_process_defaults() is its opt-in equivalent)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libs():
    """CPU-side libraries (oracle + builder) are built on demand; the HIP library is built by
    __graft_entry__.build() and travels to the GPU box prebuilt."""
    import __graft_entry__ as g
    g.build_cpu_libs()


def make_dataset(n, dim, nlist_gen, seed, normalize=False, uniform=False):
    """Gaussian-mixture synthetic fvecs."""
    rng = np.random.default_rng(seed)
    if uniform:
        return rng.random((n, dim), dtype=np.float32)
    means = rng.standard_normal((nlist_gen, dim)).astype(np.float32)
    comp = rng.integers(0, nlist_gen, n)
    x = means[comp] + 0.35 * rng.standard_normal((n, dim)).astype(np.float32)
    if normalize:
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def build_index(n=2000, dim=64, nlist=16, total_bits=7, metric=0, rotator=1, seed=1234, faster=True,
                uniform=False, normalize=False, data=None):
    import rabitq_rs_amd as rq
    if data is None:
        data = make_dataset(n, dim, max(nlist // 4, 1), seed, normalize=normalize, uniform=uniform)
    built = rq.builder.train(data, nlist, total_bits, metric, rotator, seed, faster, kmeans_iters=5)
    return data, built
