import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---- measured margins that a pass / fail line hides (encoder error against float32, ...): tests hand them to `note`, the
# run prints them in its summary whatever the capture mode
_NOTES = []


@pytest.fixture
def note():
    return _NOTES.append


def pytest_terminal_summary(terminalreporter):
    if _NOTES:
        terminalreporter.write_sep("-", "measured margins")
        for line in _NOTES:
            terminalreporter.write_line(line)
