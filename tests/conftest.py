import os
import sys
import tempfile

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def _markers(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "isolated: stresses the runtime (churn, hazards, threads, placement): runs in a child interpreter, after the parity tests")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
        return cache[name]
    return load


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests must FAIL, not skip, on a GPU box whose extension is missing; on a box without a
    # GPU they are simply not selected by the driver (-m "not gpu").  If someone runs the whole
    # suite on a CPU-only machine, skip them with a clear reason.
    items.sort(key=lambda it: 1 if it.get_closest_marker("isolated") is not None else 0)      # stable: parity tests first
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible (gpu-marked test)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---------------------------------------------------------------------------------------------------------------------------
# Native faults.  A SIGABRT inside the HIP runtime or the library used to blank the whole record: the
# interpreter died, faulthandler's Python stacks filled the tail of the log and no pass count was printed.  Three measures:
#   1. every test id is appended to a progress file BEFORE the test runs and handed to a C signal handler
#      (tests/native_fault.c) that prints "which test + native backtrace" as the LAST lines of the log;
#   2. tests marked `isolated` (context churn, stream hazards, threads, stream placement, page-locked caller memory: the ones
#      that stress the runtime rather than check arithmetic) run in a child interpreter each -- a native fault there is ONE
#      failed test with the child's tail in its message, not the end of the session;
#   3. they are also ordered behind every oracle / golden parity test.
PROGRESS = os.environ.get("CF_TEST_PROGRESS") or os.path.join(tempfile.gettempdir(), "cf_pytest_progress_%d.log" % os.getpid())
_fault_lib = None


def _install_native_fault_handler(config):
    global _fault_lib
    import ctypes
    import faulthandler
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    so, src = os.path.join(here, "_native_fault.so"), os.path.join(here, "native_fault.c")
    try:
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so, src], check=True, capture_output=True, timeout=120)
        lib = ctypes.CDLL(so)
        lib.cf_fault_note.argtypes = [ctypes.c_char_p]
        lib.cf_fault_install.argtypes = [ctypes.c_int]
        # pytest redirects fd 2 into a capture file while a test runs: both handlers must write to the REAL stderr, i.e. to the
        # duplicate pytest's own faulthandler plugin made before capturing started (or to our own, made now, with capture suspended)
        fd = None
        try:
            from _pytest.faulthandler import fault_handler_stderr_fd_key
            fd = config.stash.get(fault_handler_stderr_fd_key, None)
        except Exception:                            # noqa: BLE001
            pass
        if fd is None:
            fd = os.dup(2)
        was_on = faulthandler.is_enabled()
        if was_on:
            faulthandler.disable()
        lib.cf_fault_install(fd)                     # first in, last to run: faulthandler chains to the handler it replaced
        if was_on:
            faulthandler.enable(file=fd, all_threads=True)
        _fault_lib = lib
    except Exception as e:                           # noqa: BLE001  (diagnostics only: never a reason to fail a run)
        sys.__stderr__.write("conftest: native fault handler not installed (%s)\n" % e)


@pytest.hookimpl(trylast=True)
def pytest_configure(config):
    _markers(config)
    if not os.environ.get("CF_NO_FAULT_HANDLER"):
        _install_native_fault_handler(config)


def pytest_runtest_logstart(nodeid, location):
    try:
        with open(PROGRESS, "a") as f:
            f.write(nodeid + "\n")
    except OSError:
        pass
    if _fault_lib is not None:
        _fault_lib.cf_fault_note(nodeid.encode())


@pytest.hookimpl(tryfirst=True)
def pytest_pyfunc_call(pyfuncitem):
    """`@pytest.mark.isolated`: run the test in a child interpreter (same pytest, same node id), report its verdict."""
    if pyfuncitem.get_closest_marker("isolated") is None or os.environ.get("CF_TEST_CHILD"):
        return None
    import subprocess
    env = dict(os.environ, CF_TEST_CHILD="1", CF_TEST_PROGRESS=PROGRESS)
    cmd = [sys.executable, "-m", "pytest", pyfuncitem.nodeid, "-x", "-q", "-p", "no:cacheprovider", "--no-header", "-rN"]
    res = subprocess.run(cmd, cwd=str(pyfuncitem.config.rootpath), env=env, capture_output=True, text=True, timeout=1500)
    tail = (res.stdout[-3500:] + "\n" + res.stderr[-2500:]).strip()
    if res.returncode == 0 and " passed" in res.stdout:
        return True
    if res.returncode == 0 and " skipped" in res.stdout:
        pytest.skip("child: " + tail[-300:])
    kind = "died with signal %d" % -res.returncode if res.returncode < 0 else "failed (rc %d)" % res.returncode
    pytest.fail("isolated test %s in its child interpreter:\n%s" % (kind, tail), pytrace=False)
