"""CPU: the radix-2^29 field and group arithmetic of the hot kernels (csrc/zkc_f29*.h) is host+device code; these programs run it on
the host against the plain 8 x u32 Montgomery reference (zkc_field.h CIOS, zkc_curve.h XYZZ formulas): products, squarings, lazy
add/sub with dominators, the zero test, exits from the limb form, and chains of mixed / full additions and doublings in G1 and G2
including the equal-point and opposite-point cases.  Compiled with hipcc (host pass only is executed; no GPU needed)."""
import os, shutil, subprocess
import pytest
import oracle_lib as ol

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
CSRC = os.path.join(ol.ROOT, 'zk-franchise-proof-circuit_amd', 'csrc')


@pytest.mark.parametrize('name', ['f29_host_test', 'f29_g1_host_test', 'f29_g2_host_test'])
def test_radix29_arithmetic_on_host(name, tmp_path):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip('hipcc not available')
    src = os.path.join(ol.ROOT, 'tools', 'probe', name + '.hip')
    exe = str(tmp_path / name)
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O2', '-std=c++17', '-I' + CSRC, '-I' + os.path.join(ol.ROOT, 'include'), src, '-o', exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert '0 mismatches' in out.stdout, out.stdout
