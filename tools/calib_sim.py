"""Offline replay of the refinement calibration on the per-walker data of tools/calib_data.py (gpurun_out/calib_*.npz: score,
plain float32 E_loc, float64 E_loc per walker; fixtures and bench-like trajectories): the error model the library's threshold
rule rests on, and what each rule would have refined / left beyond the tolerance.

    python tools/calib_sim.py > profiles/r05_calibration_model.txt                      (first state of the round: single accumulator chain)
    python tools/calib_sim.py gpurun_out/calib_*.npz > profiles/r05_calibration_model_fresh_acc.txt   (final kernels)
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-5
DEFAULT_RHO = 1e-8      # engine.hip: refine_miss (1e-7 until the fresh per-chunk accumulators of the linear kernels)


def load(f):
    d = np.load(f)
    score, e32, e64 = (np.atleast_2d(d[k]) for k in ('score', 'e32', 'e64'))
    rel = np.abs(e32 - e64) / np.maximum(1.0, np.abs(e64))
    return score, rel


def sample_of(score, rel, n=256):
    B = score.shape[1]
    idx = np.arange(min(n, B)) * B // min(n, B)
    q = rel[0, idx] / score[0, idx]
    return np.sort(q[np.isfinite(q) & (score[0, idx] > 0)])


def rule_percentile(score, rel, pct=0.9, target=7e-6):
    """rounds 3-4: threshold = target / (pct-quantile of err / score over the calibration sample of the first call)"""
    q = sample_of(score, rel)
    return target / max(q[int(pct * (len(q) - 1) + 0.5)], 1e-12)


def rule_miss_rate(score, rel, rho=1e-7):
    """round 5 (engine_refine.inl: probe_rethreshold): exponential model err = m score xi; the largest threshold whose kept
    walkers (scores of the first call) miss TOL at an expected rate <= rho"""
    q = sample_of(score, rel)
    m = max(q[int(0.5 * (len(q) - 1) + 0.5)] / np.log(2), q[int(0.9 * (len(q) - 1) + 0.5)] / np.log(10), 1e-14)
    s0 = np.sort(score[0][np.isfinite(score[0]) & (score[0] > 0)])
    acc = np.cumsum(np.exp(-TOL / (m * s0))) / score.shape[1]
    k = int(np.searchsorted(acc, rho, side='right'))
    return (s0[k - 1] if k > 0 else 1.0) if k < len(s0) else 1e9, m


def outcome(score, rel, thr):
    kept = score <= thr
    return (100 * (1 - kept.mean()), rel[kept].max() if kept.any() else 0.0, int((rel[kept] >= TOL).sum()), int((rel[kept] >= 5e-6).sum()))


if __name__ == '__main__':
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', 'calib_*.npz')))
    print('Error model of the float32 forward-Laplacian pass on the MI355X (tools/calib_data.py -> tools/calib_sim.py).')
    print('err = |E_f32 - E_f64| / max(1, |E_f64|) per walker (plain float32, refinement off); score = the predictor of k_final.')
    print()
    print('1. err / score has ONE distribution in every score bin, and it is exponential: quantile / ln(1 / (1 - p)) is the same number m for')
    print('   p = 0.5, 0.9, 0.99, 0.999 (an exponential variable has quantile(p) = m ln(1 / (1 - p))).')
    for f in files:
        score, rel = load(f)
        q = (rel / score).ravel()
        q = q[np.isfinite(q)]
        est = [np.quantile(q, p) / np.log(1 / (1 - p)) for p in (0.5, 0.9, 0.99, 0.999)]
        print(f'   {os.path.basename(f)[6:-4]:30s} n = {rel.size:6d}  plain float32 within 1e-5: {100 * (rel < TOL).mean():6.2f} %  max {rel.max():.1e}   '
              f'm from p50 / p90 / p99 / p99.9 = {est[0]:.2e} {est[1]:.2e} {est[2]:.2e} {est[3]:.2e}   mean {q.mean():.2e}')
    print()
    print('2. Quantiles of err by score bin along the bench trajectories (every quantile doubles when the score doubles).')
    for f in files:
        if 'traj_' not in f:
            continue
        score, rel = load(f)
        s, x = score.ravel(), rel.ravel()
        print(f'   {os.path.basename(f)[6:-4]}')
        edges = [0, 10, 20, 40, 80, 160, 320, 640, 1280, 1e9]
        for a, b in zip(edges, edges[1:]):
            msk = (s >= a) & (s < b)
            if msk.sum() < 5:
                continue
            y = x[msk]
            print(f'     score [{a:5.0f}, {b:10.0f})  {100 * msk.mean():5.1f} % of walkers   err p50 {np.median(y):.1e}  p90 {np.quantile(y, .9):.1e}  p99 {np.quantile(y, .99):.1e}'
                  f'  p99.9 {np.quantile(y, .999):.1e}  max {y.max():.1e}   >= 1e-5: {(y >= TOL).sum():4d}')
    print()
    print('3. Threshold rules replayed: calibrated on a strided 256-walker sample of the FIRST step, applied to every step.')
    print('   columns: threshold, share of walkers refined in float64, largest error among the walkers KEPT in float32, kept walkers >= 1e-5, >= 5e-6')
    for f in files:
        score, rel = load(f)
        print(f'   {os.path.basename(f)[6:-4]}  (n = {rel.size})')
        thr = rule_percentile(score, rel)
        o = outcome(score, rel, thr)
        print(f'     rounds 3-4: 7e-6 / p90(err / score)        thr {thr:8.1f}  refined {o[0]:5.1f} %  kept max {o[1]:.2e}  kept >= 1e-5: {o[2]:3d}  >= 5e-6: {o[3]:4d}')
        for rho in (1e-5, 1e-6, 1e-7, 1e-8, 1e-9):
            thr, m = rule_miss_rate(score, rel, rho)
            o = outcome(score, rel, thr)
            tag = '  <- library default' if rho == DEFAULT_RHO else ''
            print(f'     round 5: miss rate {rho:.0e} (m = {m:.2e})      thr {thr:8.1f}  refined {o[0]:5.1f} %  kept max {o[1]:.2e}  kept >= 1e-5: {o[2]:3d}  >= 5e-6: {o[3]:4d}{tag}')
