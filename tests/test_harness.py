"""The test harness itself (tests/conftest.py, "one process per test file"): a test file whose process dies costs
the test that killed it, not the run -- the way the -m gpu suite is driven on a GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

DIES = """
import ctypes
def test_before(): pass
def test_dies(): ctypes.CDLL(None).abort()
def test_after(): pass
"""
EXIT = """
import atexit, ctypes
def test_leaves_a_bomb():
    atexit.register(lambda: ctypes.CDLL(None).abort())
"""
FINE = """
import pytest
def test_fine(): pass
def test_skipped(): pytest.skip("not today")
def test_wrong(): assert 1 == 2
"""


def _run(tmp_path, *extra):
    (tmp_path / "test_a_dies.py").write_text(DIES)
    (tmp_path / "test_b_fine.py").write_text(FINE)
    env = dict(os.environ, GSAGE_TEST_ISOLATE="1", PYTEST_PLUGINS="conftest", GSAGE_DEBUG_ABORT_TRACE="0",
               PYTHONPATH=HERE + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("GSAGE_PYTEST_CHILD", None)
    env.pop("GSAGE_PYTEST_REPORT", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "--rootdir", str(tmp_path),
                        str(tmp_path / "test_a_dies.py"), str(tmp_path / "test_b_fine.py")] + list(extra),
                       cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    return r.returncode, r.stdout.decode(errors="replace")


def test_a_dead_test_process_fails_one_test_and_the_run_goes_on(tmp_path):
    rc, out = _run(tmp_path)
    assert rc == 1, out
    assert "2 failed, 3 passed, 1 skipped" in out, out
    assert "test_a_dies.py::test_dies" in out and "Fatal Python error: Aborted" in out, out
    assert "test_b_fine.py::test_wrong" in out and "assert 1 == 2" in out, out
    assert "test_after" not in out.split("short test summary info")[-1], out      # (ran in a fresh process: passed)


def test_stop_at_first_failure_still_stops(tmp_path):
    rc, out = _run(tmp_path, "-x")
    assert rc == 1 and "1 failed, 1 passed" in out and "stopping after 1 failures" in out, out


def test_a_crash_at_interpreter_exit_fails_the_last_test_of_the_file(tmp_path):
    """all reports arrived, then the child aborted while exiting (where engines are torn down): not green"""
    (tmp_path / "test_c_exit.py").write_text(EXIT)
    (tmp_path / "test_d_fine.py").write_text("def test_ok(): pass\n")
    env = dict(os.environ, GSAGE_TEST_ISOLATE="1", PYTEST_PLUGINS="conftest", GSAGE_DEBUG_ABORT_TRACE="0",
               PYTHONPATH=HERE + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("GSAGE_PYTEST_CHILD", None)
    env.pop("GSAGE_PYTEST_REPORT", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "--rootdir", str(tmp_path),
                        str(tmp_path / "test_c_exit.py"), str(tmp_path / "test_d_fine.py")], cwd=str(tmp_path), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 1 and "crash at interpreter exit" in out, out
    assert "2 passed, 1 error" in out, out                  # (the test itself passed; its file's exit did not)
