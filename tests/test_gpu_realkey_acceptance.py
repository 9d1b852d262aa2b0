"""GPU: tools/realkey_acceptance.py end to end (steps 1-6) on the build's own nLevels-160 key standing in for Vocdoni's: the reference's inputs_example.json is proved through
it, the public signals must equal the reference's signals.json (they depend on the inputs alone), and the pinned verifier accepts under that key's verification key and rejects
a mangled proof.  With the real proving_key.zkey the same command -- without the overrides -- compares against the published hash and the committed verification_key.json."""
import json, os, sys
import pytest
import oracle_lib as ol

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))


def test_acceptance_script_runs_all_six_steps(tmp_path, capsys):
    import torch  # noqa: F401
    import realkey_acceptance as ra
    from zkcensus_amd import setup
    _, z160, v160 = setup.ensure_test_artifacts(160)
    rep_path = str(tmp_path / 'rep.json')
    rc = ra.main([z160, '--vkey', v160, '--sha256', 'none', '--json', rep_path])
    out = capsys.readouterr().out
    assert rc == 0 and 'ACCEPTED' in out and 'FAIL' not in out, out
    rep = json.load(open(rep_path))
    assert {r['step'] for r in rep['steps']} == {1, 2, 3, 4, 5, 6} and all(r['ok'] for r in rep['steps'])
    # under the REFERENCE's verification key the test key's proof must NOT be accepted: step 6 is a real check, not a formality
    rc = ra.main([z160, '--sha256', 'none'])
    out = capsys.readouterr().out
    assert rc == 1 and 'PASS step 5' in out and "FAIL step 6  the pinned verifier accepts" in out, out
