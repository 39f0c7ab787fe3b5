import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import contextlib


@contextlib.contextmanager
def search_path(path: str):
    """Which kernels answer a batch of reads: "seed" (the default: K8s seed-and-compare, K8 for what it leaves) or "walk" (the
    prefilter and the index walk for everything, SLAMEM_SEED_SEARCH=0; the library reads the switch at every call)."""
    old = os.environ.get("SLAMEM_SEED_SEARCH")
    if path == "walk":
        os.environ["SLAMEM_SEED_SEARCH"] = "0"
    else:
        os.environ.pop("SLAMEM_SEED_SEARCH", None)
    try:
        yield
    finally:
        if old is None:
            os.environ.pop("SLAMEM_SEED_SEARCH", None)
        else:
            os.environ["SLAMEM_SEED_SEARCH"] = old
