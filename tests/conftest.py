import sys
from pathlib import Path

import pytest

REPO_ROOT = Path(__file__).resolve().parent.parent
# tests may use the oracle (test infrastructure); the product package never does
for p in (str(REPO_ROOT), str(REPO_ROOT / "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build once before collection imports
    # the package (hipcc cross-compiles gfx950 without a GPU; re-builds only what is stale)
    import __graft_entry__

    __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir() -> Path:
    return REPO_ROOT / "tests" / "golden"
