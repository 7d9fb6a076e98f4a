"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer.

`make -C oracle asan` builds oracle/rogue_oracle.c with -fsanitize=address,undefined (-fno-sanitize-recover: the first finding aborts the child); child
processes load that library instead of the plain one (ROGUE_ORACLE_SO, libasan preloaded -- the interpreter itself is not instrumented) and run
  * every check of tests/test_oracle_golden.py,
  * a 10^5-env-step random soak over the mini / default / nohide configs through every entry point (tests/oracle_soak.py),
  * the soak again on a MUTANT build (-DORC_MUTANT paths are code too; the mutant machinery multiplied the oracle's paths by 65).
CPU only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")


def _asan_runtime():
    cc = os.environ.get("CC", "gcc")
    try:
        p = subprocess.check_output([cc, "-print-file-name=libasan.so"], text=True).strip()
    except (OSError, subprocess.CalledProcessError):
        return None
    return os.path.realpath(p) if os.path.isabs(p) and os.path.exists(p) else None


def _env(so):
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("no compiler / libasan on this host")
    return dict(os.environ, ROGUE_ORACLE_SO=so, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
                UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", PYTHONPATH=ROOT)


@pytest.fixture(scope="module")
def asan_so():
    if _asan_runtime() is None:
        pytest.skip("no compiler / libasan on this host")
    subprocess.check_call(["make", "-C", ORACLE, "-s", "asan"])
    return os.path.join(ORACLE, "_build", "librogue_oracle_asan.so")


def _clean(r):
    text = r.stdout + r.stderr
    assert "AddressSanitizer" not in text and "runtime error" not in text, text[-4000:]
    assert r.returncode == 0, text[-4000:]


def test_golden_checks_under_asan_ubsan(asan_so):
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q", "-p", "no:cacheprovider"], cwd=ROOT, env=_env(asan_so),
                       capture_output=True, text=True, timeout=600)
    _clean(r)
    assert " passed" in r.stdout


def test_random_soak_under_asan_ubsan(asan_so):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "oracle_soak.py"), "100000"], cwd=ROOT, env=_env(asan_so), capture_output=True, text=True, timeout=600)
    _clean(r)
    assert "SOAK ok" in r.stdout


@pytest.mark.parametrize("mutant", [36, 47])
def test_mutant_paths_under_asan_ubsan(asan_so, mutant):
    """One draw-order mutant of the monster turn and the DistCache mutant: the soak has no expectations, so a mutant runs it to the end like the restatement."""
    subprocess.check_call(["make", "-C", ORACLE, "-s", "asan-mutant", "MUTANT=%d" % mutant])
    so = os.path.join(ORACLE, "_build", "librogue_oracle_asan_m%d.so" % mutant)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "oracle_soak.py"), "20000"], cwd=ROOT, env=_env(so), capture_output=True, text=True, timeout=600)
    _clean(r)
    assert "SOAK ok" in r.stdout
