import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The test process holds two OpenMP runtimes (the oracle's libgomp and the one torch brings), each with a pool as wide as the box
# (256 hardware threads on the GPU box).  Idle threads that SPIN at barriers starve the other pool's; sleeping ones do not.  Set before
# either runtime starts.  (Test infrastructure only: bench.py's cpu_baseline runs the oracle in its own process with its own settings.)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("mp-gadget_amd")


@pytest.fixture(scope="session")
def orc():
    """The strict-IEEE CPU oracle (test infrastructure)."""
    from oracle import oracle as O
    o = O.Oracle(fast=False)
    o.fill_ntab(0, 1.5)
    return o


@pytest.fixture(scope="session")
def engine(pkg):
    """One HIP engine on cuda:0 for the whole session.  Fails (does not skip) if the library or GPU is missing."""
    e = pkg.Engine(0)
    e.gravshort_fill_ntab(0, 1.5)
    yield e
    e.close()


def keep_artifacts_on_failure(body):
    """Debugging aid for the multi-rank tests (several processes on one GPU): when the body fails, the arrays the ranks saved
    under tmp_path are copied to gpurun_out/flake/<test>-<pid>/ before the failure is re-raised, so that a failure seen on the
    GPU box can be analysed afterwards (which rows differ, by how much, where in the box).  Nothing is retried."""
    import functools
    import shutil

    @functools.wraps(body)
    def wrapped(*a, **k):
        try:
            return body(*a, **k)
        except AssertionError:
            tp = k.get("tmp_path") or next((x for x in a if hasattr(x, "iterdir")), None)
            if tp is not None:
                dst = os.path.join(ROOT, "gpurun_out", "flake", "%s-%d" % (body.__name__, os.getpid()))
                try:
                    shutil.copytree(str(tp), dst, dirs_exist_ok=True)
                except OSError:
                    pass
            raise
    return wrapped


def _foreign_device_fault(text):
    """True when a rank died of an HSA queue abort whose faulting kernel (the runtime's "Kernel Name:" note) is none of this library's."""
    import re
    if "HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION" not in text:
        return False
    names = re.findall(r"Kernel Name: (\S+)", text)
    return bool(names) and all("N3mpg" not in n for n in names)


def run_ranks(cmd, env, log_path, timeout=900):
    """subprocess.run of a (multi-)rank helper script; on a non-zero exit the COMPLETE stdout / stderr are written to `log_path`
    (under the test's tmp_path, which keep_artifacts_on_failure copies to gpurun_out/flake/) before the assertion fires: a rank that
    aborts leaves its message far above the launcher's own traceback."""
    import subprocess
    env = dict(env)
    env.setdefault("AMD_LOG_LEVEL", "1")          # HIP runtime errors to stderr
    env.setdefault("TORCH_SHOW_CPP_STACKTRACES", "1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    if r.returncode != 0:
        with open(str(log_path) + ".failed.log", "w") as f:
            f.write("cmd: %s\n---- stdout ----\n%s\n---- stderr ----\n%s\n" % (" ".join(cmd), r.stdout, r.stderr))
        # Several ranks sharing ONE GPU fail once in ~50 runs in the set-up before any kernel of the test proper (DESIGN section 4:
        # a worker SIGABRT in rounds 2-3, a count mismatch in the decomposition on a box's first multi-process use in round 4).
        # The suite runs under -x: ONE repeat, and the first failure is kept in gpurun_out/flake/RETRIED.log and in a warning, so
        # that a repeat is never silent.  A failure that repeats fails the test.
        # Round 6: the repeat is OPT-IN (MPG_TEST_RETRY=1) and off by default - a product-side race that shows once in 50 runs must fail
        # the suite, not hide behind a repeat (profiles/r06b_flake_stress/: 200 clean iterations of the four tests concerned).
        # End of round 6: the worker SIGABRT of rounds 2-3 was caught with its reason (profiles/r06b_flake_stress/README.md, last section):
        # "HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION" in the queue of ONE of two processes sharing the GPU, and the runtime's core-dump note
        # names the kernel - at::native::vectorized_elementwise_kernel<FillFunctor<double>>, PyTorch's own fill, before any kernel of
        # this library had run in that process.  A device fault inside a kernel that is NOT this library's (no "N3mpg" in the mangled
        # name) gets ONE repeat, logged and warned about; a fault in one of ours, an invariant, a wrong number or a time-out never does.
        foreign_fault = _foreign_device_fault(r.stdout + r.stderr)
        if os.environ.get("MPG_TEST_RETRY") != "1" and not foreign_fault:
            assert r.returncode == 0, r.stderr[-3000:]
        import warnings
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        try:
            os.makedirs(os.path.join(root, "gpurun_out", "flake"), exist_ok=True)
            with open(os.path.join(root, "gpurun_out", "flake", "RETRIED.log"), "a") as f:
                f.write("cmd: %s\nrc %d\n---- stderr (tail) ----\n%s\n\n" % (" ".join(cmd), r.returncode, r.stderr[-6000:]))
        except OSError:
            pass
        warnings.warn("multi-rank helper failed once (%s) and was repeated: %s"
                      % ("device fault in a foreign kernel" if foreign_fault else "MPG_TEST_RETRY=1", " ".join(cmd[-3:])))
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return r


class phase_clock:
    """Wall-clock of the phases of a long test, appended to gpurun_out/flake/phases.log (one line per test run): two of ~40 full suites
    of round 2 spent 10 extra minutes inside test_full_size_256_properties[s_clust] and passed; this shows in which phase the next time."""

    def __init__(self, name):
        import time
        self.name, self.t0, self.marks, self.time = name, time.perf_counter(), [], time

    def mark(self, what):
        t = self.time.perf_counter()
        self.marks.append("%s %.1fs" % (what, t - self.t0))
        self.t0 = t

    def write(self):
        try:
            d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "flake")
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "phases.log"), "a") as f:
                f.write("%s pid %d: %s\n" % (self.name, os.getpid(), ", ".join(self.marks)))
        except OSError:
            pass
