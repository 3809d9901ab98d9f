"""CPU tier: the host stages under AddressSanitizer + UndefinedBehaviorSanitizer (the reference has no sanitizer set-up; GPU sanitizers are
not available on the pool, so the device code is covered by bit-exact parity instead).  `make -C yaha_amd/csrc asan` builds host/*.cpp alone into
libyaha_host_asan.so; tests/asan_driver.py runs every golden through it in a child process with the sanitizer runtimes preloaded."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _rt(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], stdout=subprocess.PIPE).stdout.decode().strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_rt("libasan.so") is None or _rt("libubsan.so") is None, reason="sanitizer runtimes not installed")
def test_host_stages_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "yaha_amd", "csrc"), "asan"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=_rt("libasan.so") + ":" + _rt("libubsan.so"), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               YAHA_HIP_LIB=os.path.join(ROOT, "yaha_amd", "csrc", "libyaha_host_asan.so"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_driver.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    err = p.stderr.decode()
    assert p.returncode == 0 and "sanitizer run ok" in p.stdout.decode(), err[-3000:]
    assert "ERROR: AddressSanitizer" not in err and "runtime error:" not in err, err[-3000:]
