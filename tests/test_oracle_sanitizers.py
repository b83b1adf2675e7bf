"""The CPU oracle under AddressSanitizer + UndefinedBehaviourSanitizer (host code only; `make -C oracle sanitize`).

The checker must itself be free of out-of-bounds accesses and undefined behaviour on awkward frame shapes
(ragged sizes, negative min_disparity, 1..3 disparities, 1-pixel frames). The sanitized build is
single-threaded and unoptimised; its final maps must hash to the values the optimised OpenMP library
produces for the same frames (the checksums below were taken from it)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EXPECTED = [
    "61x37 d[0,24) L1=34 rc=0 checksum=12429655",
    "33x29 d[3,20) L1=9 rc=0 checksum=6e3a3ce6",
    "40x21 d[-4,12) L1=34 rc=0 checksum=c0000000",
    "7x5 d[0,1) L1=34 rc=0 checksum=8f800000",
    "1x1 d[0,3) L1=34 rc=0 checksum=7f800000",
    "23x1 d[0,5) L1=40 rc=0 checksum=13e72e2b",
    "1x19 d[0,2) L1=34 rc=0 checksum=0f800000",
]


def test_oracle_runs_clean_under_asan_ubsan():
    d = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", d, "-s", "sanitize"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([os.path.join(d, "oracle_san")], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr
    assert r.stdout.strip().splitlines() == EXPECTED
