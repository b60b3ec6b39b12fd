"""SURVEY 8c(iii) / VERDICT r5 row (g): the quality evidence north_star asks for ("FID/LPIPS within noise"), in the only form that can be
measured offline (TEST INFRASTRUCTURE: imports tests/ and the oracle; tests/quality_metrics.py holds the two metrics and says what
they stand in for).

N seeded samples go through the HIP path's `Trainer.forward` and through the CPU oracle, on the same weights,
  (a) at initialisation, and
  (b) after K optimiser steps made on EACH side from that same state over the same sequence of seeded batches,
and the synthesised target images (`fake_tsf_imgs`) of the two sides are compared by a random-feature Frechet distance
(fid_score.py:146-200's formula) and an LPIPS-shaped multi-layer distance (lpips.py:41-56's form).  The noise floor beside every
number is the ORACLE AGAINST ITSELF with a different intra-op thread count: the same fp32 algorithm with another summation order,
which is all that separates two CPU runs of the reference.  GAN training amplifies that (Adam's first updates have the size of the
learning rate whatever the gradient's magnitude), so after K steps the floor is far from zero -- that is the "noise" north_star's
"within noise" refers to.
    python tools/quality_surrogate.py [--n 256] [--steps 50] [--side 64] [--batch 4] [--save-hip FILE | --load-hip FILE] [--json OUT]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from common import seeded_state, SEEDS                       # noqa: E402
from oracle import hogan_oracle as O                          # noqa: E402
from hoig_amd import synthetic                                # noqa: E402
import quality_metrics as Q                                   # noqa: E402

GEN = 'generator_spade_attn'
EVAL_SEED, TRAIN_SEED = 1000, 2000


def eval_batches(n, side, eb):
    for j in range(0, n, eb):
        yield synthetic.make_inputs(min(eb, n - j), side, seed=EVAL_SEED + j)


def train_batch(k, side, batch):
    return synthetic.make_inputs(batch, side, seed=TRAIN_SEED + k)


def oracle_run(n, side, batch, steps, threads, eb=16, log=None):
    """-> {'init': images, 'k<K>': images for every K in `steps`, 'real': images}: fake_tsf of N eval samples before and after K
    optimiser steps."""
    torch.set_num_threads(threads)
    cfg, sdG, sdD, sdV = seeded_state(GEN)
    ot = O.OracleTrainer(cfg, sdG, sdD, sdV)
    out = {}

    def evaluate():
        fake, real = [], []
        with torch.no_grad():
            for inp in eval_batches(n, side, eb):
                ot.set_prepared_input(inp)
                fake.append(ot.forward()[3].clone())
                real.append(inp['real_tsf'].clone())
        return torch.cat(fake), torch.cat(real)

    t0 = time.time()
    out['init'], out['real'] = evaluate()
    marks = sorted(set(steps))
    for k in range(marks[-1]):
        ot.set_prepared_input(train_batch(k, side, batch))
        ot.optimize_parameters()
        if k + 1 in marks:
            out['k%d' % (k + 1)], _ = evaluate()
    if log:
        log('oracle, %d threads: %d eval samples x %d + %d steps in %.0f s' % (threads, n, len(marks) + 1, marks[-1], time.time() - t0))
    return out


def hip_run(n, side, batch, steps, eb=16, log=None):
    from common import product_trainer
    from hoig_amd import ops
    ops.set_precision(os.environ.get('HOIG_PRECISION', 'bf16x3:f16x2'))
    m = product_trainer(GEN, batch, side)
    out = {}

    def evaluate():
        fake = []
        m.set_eval()
        with torch.no_grad():
            for inp in eval_batches(n, side, eb):
                m.set_input(inp)
                fake.append(m.forward()[3].float().cpu().contiguous().clone())
        m.set_train()
        return torch.cat(fake)

    t0 = time.time()
    out['init'] = evaluate()
    marks = sorted(set(steps))
    for k in range(marks[-1]):
        m.set_input(train_batch(k, side, batch))
        m.optimize_parameters()
        if k + 1 in marks:
            out['k%d' % (k + 1)] = evaluate()
    torch.cuda.synchronize()
    if log:
        log('HIP path (%s): the same in %.0f s' % (os.environ.get('HOIG_PRECISION', 'bf16x3:f16x2'), time.time() - t0))
    return out


def compare(a, b, sd_vgg, dims=Q.FEATURE_DIMS):
    """-> dict: Frechet distance of the two image sets, LPIPS-like distance of the paired images (mean, max)."""
    lp = Q.lpips_like(a, b, sd_vgg)
    return dict(frechet=Q.frechet_between(a, b, dims=dims), lpips_mean=float(lp.mean()), lpips_max=float(lp.max()),
                max_rel=float((a - b).abs().max() / b.abs().max()))


def report(hip, ora, orb, sd_vgg, log, dims=Q.FEATURE_DIMS):
    rows = {}
    fb = lambda x, y: Q.frechet_between(x, y, dims=dims)
    stages = ['init'] + sorted((k for k in ora if k.startswith('k')), key=lambda k: int(k[1:]))
    for stage in stages:
        rows[stage] = dict(hip_vs_oracle=compare(hip[stage], ora[stage], sd_vgg, dims), oracle_vs_oracle=compare(orb[stage], ora[stage], sd_vgg, dims),
                           to_real=dict(hip=fb(hip[stage], ora['real']), oracle=fb(ora[stage], ora['real']),
                                        oracle_other_threads=fb(orb[stage], ora['real'])))
    # scale of the feature space, for reading the absolute numbers: the trace of the oracle's feature covariance
    _, sig = Q.activation_statistics(Q.random_features(ora['init'], dims=dims))
    rows['feature_trace'] = float(np.trace(sig))
    for stage in stages:
        r = rows[stage]
        log('%-8s Frechet  HIP vs oracle %.3e | oracle vs oracle (other thread count) %.3e   [trace of the feature covariance %.3e]'
            % (stage, r['hip_vs_oracle']['frechet'], r['oracle_vs_oracle']['frechet'], rows['feature_trace']))
        log('%-8s LPIPS-like mean / max over pairs:  HIP vs oracle %.3e / %.3e | oracle vs oracle %.3e / %.3e'
            % (stage, r['hip_vs_oracle']['lpips_mean'], r['hip_vs_oracle']['lpips_max'], r['oracle_vs_oracle']['lpips_mean'],
               r['oracle_vs_oracle']['lpips_max']))
        log('%-8s Frechet to the real targets (the FID-shaped number):  HIP %.4f | oracle %.4f | oracle, other thread count %.4f'
            % (stage, r['to_real']['hip'], r['to_real']['oracle'], r['to_real']['oracle_other_threads']))
        log('%-8s max |a - b| / max |b| over all images:  HIP vs oracle %.3e | oracle vs oracle %.3e'
            % (stage, r['hip_vs_oracle']['max_rel'], r['oracle_vs_oracle']['max_rel']))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=256)
    ap.add_argument('--steps', type=int, nargs='+', default=[5, 20, 50], help='optimiser steps after which the N samples are evaluated')
    ap.add_argument('--side', type=int, default=64)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--threads', type=int, nargs=2, default=None, help='intra-op threads of the two oracle runs')
    ap.add_argument('--json', default=None)
    ap.add_argument('--save-hip', default=None, help='run the HIP side only and save its images (a GPU box needs no oracle run for that)')
    ap.add_argument('--oracle-only', action='store_true', help='make the two oracle runs and write them to --oracle-cache (no GPU needed)')
    ap.add_argument('--oracle-cache', default=None, help='file that keeps the two oracle runs (made if absent)')
    ap.add_argument('--load-hip', default=None, help='take the HIP side from a file written by --save-hip (the oracle runs need no GPU)')
    a = ap.parse_args()
    log = lambda s: print(s, flush=True)
    cores = len(os.sched_getaffinity(0))
    ta, tb = a.threads or (cores, max(1, cores // 2 - 1))
    log('# tools/quality_surrogate.py: N = %d eval samples, evaluated at K = %s optimiser steps, %dx%d, batch %d, %s; oracle on %d / %d threads'
        % (a.n, a.steps, a.side, a.side, a.batch, GEN, ta, tb))
    if a.oracle_only:
        ora = oracle_run(a.n, a.side, a.batch, a.steps, ta, log=log)
        orb = oracle_run(a.n, a.side, a.batch, a.steps, tb, log=log)
        torch.save(dict(a=ora, b=orb, args=[a.n, a.steps, a.side, a.batch, ta, tb]), a.oracle_cache)
        return
    if a.load_hip:
        hip = torch.load(a.load_hip)
        assert hip['args'] == [a.n, a.steps, a.side, a.batch], 'the saved HIP run used other arguments: %r' % (hip['args'],)
        log('HIP path (%s): images loaded from %s' % (hip['precision'], a.load_hip))
    else:
        hip = hip_run(a.n, a.side, a.batch, a.steps, log=log)
    if a.save_hip:
        hip['args'] = [a.n, a.steps, a.side, a.batch]
        hip['precision'] = os.environ.get('HOIG_PRECISION', 'bf16x3:f16x2')
        torch.save(hip, a.save_hip)
        return
    if a.oracle_cache and os.path.exists(a.oracle_cache):
        c = torch.load(a.oracle_cache)
        assert c['args'] == [a.n, a.steps, a.side, a.batch, ta, tb], 'the cached oracle runs used other arguments: %r' % (c['args'],)
        ora, orb = c['a'], c['b']
        log('oracle runs (%d / %d threads) loaded from %s' % (ta, tb, a.oracle_cache))
    else:
        ora = oracle_run(a.n, a.side, a.batch, a.steps, ta, log=log)
        orb = oracle_run(a.n, a.side, a.batch, a.steps, tb, log=log)
        if a.oracle_cache:
            torch.save(dict(a=ora, b=orb, args=[a.n, a.steps, a.side, a.batch, ta, tb]), a.oracle_cache)
    sd_vgg = seeded_state(GEN)[3]
    rows = report(hip, ora, orb, sd_vgg, log)
    if a.json:
        json.dump(rows, open(a.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
