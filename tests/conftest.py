import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CLI's prepared-database cache (qpgesture_amd/db_cache.py) must not land in the user's home during a test run
    import tempfile
    os.environ.setdefault("QPG_DB_CACHE_DIR", tempfile.mkdtemp(prefix="qpg_test_cache_"))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
