"""The host-only parsers of libzkcensus (.zkey / .wtns / JSON / sha256, csrc/zkc_hostparse.h) under AddressSanitizer + UBSan on the CPU:
truncations, hostile 64-bit section sizes, n = 1 domains, out-of-range coefficients and random mutations (tests/host/parse_asan.cc)."""
import os, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_parsers_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / 'parse_asan')
    cmd = ['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', os.path.join(ROOT, 'tests', 'host', 'parse_asan.cc'), '-o', exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and 'asan' in (b.stderr or '').lower() and 'cannot find' in b.stderr:
        pytest.skip('no sanitizer runtime for g++ on this box')
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS='detect_leaks=1'))
    assert r.returncode == 0 and 'host parsers: ok' in r.stdout, (r.stdout + r.stderr)[-3000:]
