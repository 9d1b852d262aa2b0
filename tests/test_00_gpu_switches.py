"""GPU: every scheduling / kernel-form switch of the library gives the SAME proof bytes as the default configuration.  Most of them are read once per process
(`static const ... = getenv(...)`), so each configuration runs tests/host/prove_digest.py as a child process (started before this process needs the GPU for anything else) and the
digests are compared.  What is covered: the two LDS-DMA forms of the G2 accumulation, the G2 accumulation held behind the transforms (round 2-3's order), the bucket reduction
on its own stream, two pipeline lanes, round 1's reduction windows and odd ones, buildABC with products for the unit coefficients, unfolded passes with their infinity bases left
in, the buildABC prefetch placements, the two-transform NTT, the one-stage head kernel, the second section tables of a deep pass, folding off.  This file sorts first on purpose (like the Node test): the children are started BEFORE this pytest
process has initialised the GPU -- the GPU boxes refuse to start a program from a process that has."""
import json, os, subprocess, sys
import pytest
import oracle_lib as ol

pytestmark = pytest.mark.gpu
CONFIGS = [
    {}, {'ZKC_G2_ACC': '1', 'ZKC_C_SECTIONS': '13'}, {'ZKC_G2_ACC': '2'}, {'ZKC_G2_ACC_HOLD': '1'}, {'ZKC_REDUCE_STREAM': '1', 'ZKC_INFLIGHT': '40'}, {'ZKC_LANES': '2', 'ZKC_INFLIGHT': '40'},
    {'ZKC_VW_BIG': '1024', 'ZKC_VW_SMALL': '256', 'ZKC_INFLIGHT': '40'}, {'ZKC_VW_BIG': '8192', 'ZKC_VW_SMALL': '512', 'ZKC_VW_G2': '256', 'ZKC_INFLIGHT': '40'},
    {'ZKC_MATVEC_UNITS': '0', 'ZKC_NTT_RADIX': '1'}, {'ZKC_NOFOLD_LISTS': '0', 'ZKC_NO_FOLD': '1'}, {'ZKC_NO_FOLD': '1'}, {'ZKC_MV_PREFETCH_AT_NTT': '1', 'ZKC_INFLIGHT': '8'}, {'ZKC_MATVEC_INLINE': '1', 'ZKC_INFLIGHT': '8'},
    {'ZKC_NTT_SEPARATE': '1'}, {'ZKC_INFLIGHT': '40'}, {'ZKC_DEEP_TABLES': '2', 'ZKC_DEEP_WIRES': '1', 'ZKC_INFLIGHT': '40'},
]


def test_switches_do_not_change_a_byte():
    script = os.path.join(ol.ROOT, 'tests', 'host', 'prove_digest.py')
    digests = []
    for cfg in CONFIGS:
        env = {k: v for k, v in os.environ.items() if not k.startswith('ZKC_')}
        env.update(cfg)
        try:
            r = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=300)
        except OSError as e:                             # the box refused to start a child program from this process
            pytest.skip('cannot start a child process here: %s' % e)
        assert r.returncode == 0, (cfg, r.stderr[-1500:])
        digests.append(json.loads(r.stdout.strip().splitlines()[-1])['sha256'])
    assert len(set(digests)) == 1, [(c, d[:12]) for c, d in zip(CONFIGS, digests)]
