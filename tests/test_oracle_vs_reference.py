"""CPU, build container only: pins the oracle to the REFERENCE's own Python (imported from /root/reference through
oracle/ref_harness.py; skipped where that tree does not exist, e.g. on the GPU box).  Every generator variant the reference's
factory offers (models/networks/__init__.py:11-25) on the HOIG_HOv3 copy, and the default + grid-sample variants on the
HOIG_DexYCB copy (HOIG_DexYCB/models/trainer.py: 13 / 9 / 24 channels, no arm mask): forward outputs, loss terms over two
optimiser steps, every post-step parameter of G and D.  Both sides are fp32 torch-CPU in the same process with the same
thread count, so the comparison is (near-)bit-level."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('HOIG_REFERENCE_ROOT', '/root/reference')

CASES = [('generator_base', 'hov3'), ('generator_spade', 'hov3'), ('generator_spade_attn', 'hov3'),
         ('generator_spade_attn_tiny', 'hov3'), ('generator_spade_attn', 'dexycb'), ('generator_spade', 'dexycb')]


_procs = {}


def _start_all():
    """All six comparisons run concurrently (one process per case: one reference copy per process), two torch threads each."""
    if _procs:
        return
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', HOIG_REF_CHECK_THREADS='2')
    for gen_name, copy in CASES:
        _procs[(gen_name, copy)] = subprocess.Popen([sys.executable, '-m', 'oracle.ref_check', gen_name, copy, '64', '1', '2'],
                                                    cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'HOIG_HOv3', 'models')), reason='reference tree not present')
@pytest.mark.parametrize('gen_name,copy', CASES)
def test_oracle_equals_reference_python(gen_name, copy):
    _start_all()
    out, err = _procs[(gen_name, copy)].communicate(timeout=900)
    assert _procs[(gen_name, copy)].returncode == 0, err[-2000:]
    r = json.loads(out.strip().splitlines()[-1])
    print(r)
    assert r['fwd_max_abs'] <= 1e-6, r
    assert r['loss_max_rel'] <= 1e-5, r
    assert r['post_weight_max_abs'] <= 1e-6, r          # two Adam steps of |dw| ~ 2e-4: identical to fp32 rounding
    assert r['grad_max_rel'] <= 1e-4, r
