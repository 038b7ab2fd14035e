"""pytest configuration: registers the `gpu` marker and puts the repo root and the
package directory (`bounty-matrix-inversion_amd/`, not an importable name) on sys.path."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "bounty-matrix-inversion_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
