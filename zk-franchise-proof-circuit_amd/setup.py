"""Test-only key generation: r1cs.build() -> .r1cs -> zkc_setup_from_r1cs -> .zkey + verification_key.json.
Stand-in for `make compile` (circuit/circuit-compiler.sh:80-136) whose outputs are missing blobs."""
import ctypes
import os
from . import _native, r1cs

DEFAULT_SEED = 0x5A4B43454E535553        # "ZKCENSUS" 
_HERE = os.path.dirname(os.path.abspath(__file__))
ARTIFACT_DIR = os.path.join(_HERE, 'build', 'artifacts')


def artifact_paths(nLevels=160, seed=DEFAULT_SEED, directory=None):
    d = directory or ARTIFACT_DIR
    stem = os.path.join(d, 'zkcensus_%d_%x' % (nLevels, seed))
    return stem + '.r1cs', stem + '.zkey', stem + '_vkey.json'


def ensure_test_artifacts(nLevels=160, seed=DEFAULT_SEED, directory=None, force=False):
    """Returns (r1cs_path, zkey_path, vkey_json_path), generating them on first use (about 20-40 s of host time)."""
    r, z, v = artifact_paths(nLevels, seed, directory)
    if not force and all(os.path.exists(p) for p in (r, z, v)):
        return r, z, v
    os.makedirs(os.path.dirname(r), exist_ok=True)
    _, cs = r1cs.build(nLevels)
    cs.write(r)
    err = ctypes.create_string_buffer(512)
    tmpz, tmpv = z + '.tmp%d' % os.getpid(), v + '.tmp%d' % os.getpid()
    rc = _native.load().zkc_setup_from_r1cs(r.encode(), seed, tmpz.encode(), tmpv.encode(), err, 512)
    if rc != 0:
        raise _native.ZkcError(rc, err.value.decode())
    os.replace(tmpz, z); os.replace(tmpv, v)
    return r, z, v
