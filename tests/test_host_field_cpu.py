"""Host arithmetic that finishes a proof: the binary-Euclid inversion (csrc/zkc_field.h fp_inv_gcd, csrc/zkc_curve.h xyzz_to_affine_gcd) with which prove_batch_finish makes the
three points of a small pass affine, against the square-and-multiply inversion everything else uses.  Built with hipcc (the headers are HIP host + device code), run on the CPU."""
import os, shutil, subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gcd_inversion_equals_fermat_inversion(tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc on this box')
    exe = str(tmp_path / 'field_inv')
    b = subprocess.run([hipcc, '--offload-arch=gfx950', '-std=c++17', '-O2', '-Wno-unused-result', '-I', os.path.join(ROOT, 'zk-franchise-proof-circuit_amd', 'csrc'), '-I', os.path.join(ROOT, 'include'),
                        os.path.join(ROOT, 'tests', 'host', 'field_inv.hip'), '-o', exe], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'field inversions: ok' in r.stdout, (r.stdout + r.stderr)[-2000:]


def test_pairing_of_the_cpu_verifier(tmp_path):
    """[r5] csrc/zkc_pairing.h (optimal ate, projective sparse lines, shared accumulator, signed-digit loop, cyclotomic final exponentiation, endomorphism subgroup test):
    bilinearity, several pairs on one accumulator = separate loops, cyclotomic = plain squaring, Frobenius maps, and membership in G2 by psi(Q) = [6x^2]Q against [r]Q = infinity
    on subgroup points and on 40 twist points outside the subgroup (tests/host/pairing_host.hip).  Its VALUE is pinned by test_oracle_pinning.py (vk_alphabeta_12)."""
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc on this box')
    exe = str(tmp_path / 'pairing_host')
    b = subprocess.run([hipcc, '--offload-arch=gfx950', '-std=c++17', '-O2', '-Wno-unused-result', '-I', os.path.join(ROOT, 'zk-franchise-proof-circuit_amd', 'csrc'), '-I', os.path.join(ROOT, 'include'),
                        os.path.join(ROOT, 'tests', 'host', 'pairing_host.hip'), '-o', exe], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'pairing host checks: ok' in r.stdout and '40 outside' in r.stdout, (r.stdout + r.stderr)[-2000:]
