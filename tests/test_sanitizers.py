"""AddressSanitizer + UndefinedBehaviorSanitizer over the kernel source (CPU build only: GPU sanitizers are not available
on this pool).  The host simulation runs every fork-join region serially, so this covers index arithmetic, table
lookups, register-row indices and the host side of the C-ABI -- not lane-level hazards, which the GPU parity tests and
the bit-identical spill build cover.  Rodent, humanoid and ant models."""
import os
import subprocess
import sys

import pytest

import helpers as H


def _san(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_san("libasan.so") is None or _san("libubsan.so") is None, reason="sanitizer runtimes not installed")
def test_host_build_is_clean_under_asan_and_ubsan():
    src = os.path.join(H.ROOT, "vnl-brax-imitation_amd", "csrc", "vnl_lib.hip")
    out = os.path.join(H.ROOT, "tests", "hostsim", "_build", "libvnl_hostsim_float_asan.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-g", "-fPIC", "-shared", "-std=c++17", "-fsanitize=address,undefined",
                           "-fno-omit-frame-pointer", "-DVNL_REAL=float", "-I" + os.path.join(H.ROOT, "tests", "hostsim", "stub"),
                           "-x", "c++", src, "-o", out])
    env = dict(os.environ, LD_PRELOAD=f"{_san('libasan.so')} {_san('libubsan.so')}",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([sys.executable, os.path.join(H.ROOT, "tests", "hostsim", "sanitizer_driver.py"), out], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert "rodent ok" in p.stdout and "humanoid ok" in p.stdout and "ant ok" in p.stdout
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-3000:]
