import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    import spawn
    spawn.start()          # the helper that starts child programs, before this process touches the GPU (tests/spawn.py)
    _sessionstart_crashtrace(session)


def _sessionstart_crashtrace(session):
    """Native call stacks for crashes inside shared libraries (see tests/capi_client/crashtrace.c);
    installed after pytest's faulthandler, which still prints the Python stack afterwards."""
    import ctypes
    lib = os.path.join(ROOT, "tests", "capi_client", "libcrashtrace.so")
    if os.path.exists(lib) and os.environ.get("MMC_CRASH_TRACE", "1") != "0":
        try:
            ctypes.CDLL(lib).crashtrace_install()
        except OSError:
            pass
