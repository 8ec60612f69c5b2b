"""pytest configuration: registers the `gpu` marker and makes the repo root importable."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# torch first: it ships its own HIP runtime, and a process must not end up with two of them (libearhip.so
# then binds to the one already loaded — the order bench.py uses; the other order works only as long as
# libearhip touches the GPU before torch does)
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
