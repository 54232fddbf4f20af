import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """How often an input was re-drawn for an exactly-zero logit (tests/util.py REDRAWS): one line in every summary, so
    the figure lands in the driver's GPUTEST record whether or not stdout of passing tests is shown."""
    try:
        from util import REDRAWS
    except Exception:       # the helper module was never imported: nothing re-drew
        return
    total = sum(n for _, n in REDRAWS)
    terminalreporter.write_sep('-', 'exact-zero-logit re-draws: %d in %d test(s)' % (total, len(REDRAWS)))
    for test, n in REDRAWS:
        terminalreporter.write_line('  re-draw x%d: %s' % (n, test))
