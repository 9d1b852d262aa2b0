"""Test-only key generation: r1cs.build() -> .r1cs -> zkc_setup_from_r1cs -> .zkey + verification_key.json.
Stand-in for `make compile` (circuit/circuit-compiler.sh:80-136) whose outputs are missing blobs."""
import ctypes
import hashlib
import os
from . import _native, r1cs

DEFAULT_SEED = 0x5A4B43454E535553        # "ZKCENSUS" 
_HERE = os.path.dirname(os.path.abspath(__file__))
ARTIFACT_DIR = os.path.join(_HERE, 'build', 'artifacts')


def artifact_paths(nLevels=160, seed=DEFAULT_SEED, directory=None):
    d = directory or ARTIFACT_DIR
    stem = os.path.join(d, 'zkcensus_%d_%x' % (nLevels, seed))
    return stem + '.r1cs', stem + '.zkey', stem + '_vkey.json'


def _generator_stamp():
    """sha256 over the sources that decide the artifacts' bytes: a key written by an older generator is regenerated, never reused."""
    h = hashlib.sha256()
    for f in (os.path.join(_HERE, 'r1cs.py'), os.path.join(_HERE, 'csrc', 'zkc_setup.hip'), os.path.join(_HERE, 'csrc', 'zkc_fixedbase.h')):
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def ensure_test_artifacts(nLevels=160, seed=DEFAULT_SEED, directory=None, force=False):
    """Returns (r1cs_path, zkey_path, vkey_json_path), generating them on first use (about 20-40 s of host time) and again whenever
    r1cs.py or zkc_setup.hip changed since they were written (stamp file next to them)."""
    r, z, v = artifact_paths(nLevels, seed, directory)
    stamp_path, stamp = z + '.stamp', _generator_stamp()
    fresh = os.path.exists(stamp_path) and open(stamp_path).read().strip() == stamp
    if not force and fresh and all(os.path.exists(p) for p in (r, z, v)):
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
    with open(stamp_path + '.tmp%d' % os.getpid(), 'w') as fh:
        fh.write(stamp)
    os.replace(stamp_path + '.tmp%d' % os.getpid(), stamp_path)
    return r, z, v
