"""Offline replay of the refinement calibration on the per-walker data of tools/calib_data.py (gpurun_out/calib_*.npz):
for a rule (percentile of error / score over the calibration sample, target), the threshold it derives, the share of walkers
it sends to float64 and the errors of the walkers it leaves in float32.

    python tools/calib_sim.py [gpurun_out/calib_*.npz]
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def replay(score, e32, e64, pct, target, sample=256):
    """score/e32/e64: [steps, B] (a fixture is one step).  Calibrate on a strided sample of step 0, apply to all steps."""
    rel = np.abs(e32 - e64) / np.maximum(1.0, np.abs(e64))
    B = score.shape[1]
    idx = (np.arange(min(sample, B)) * B // min(sample, B))
    cs = rel[0, idx] / score[0, idx]
    cs = np.sort(cs[np.isfinite(cs) & (score[0, idx] > 0)])
    c = max(cs[int(pct * (len(cs) - 1) + 0.5)], 1e-12)
    thr = min(max(target / c, 1.0), 1e9)
    kept = score <= thr
    out = {'thr': thr, 'refined': 1.0 - kept.mean(), 'kept_max': rel[kept].max() if kept.any() else 0.0,
           'kept_above_1e-5': int((rel[kept] >= 1e-5).sum()), 'n': rel.size}
    return out


if __name__ == '__main__':
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', 'calib_*.npz')))
    for f in files:
        d = np.load(f)
        score, e32, e64 = (np.atleast_2d(d[k]) for k in ('score', 'e32', 'e64'))
        rel = np.abs(e32 - e64) / np.maximum(1.0, np.abs(e64))
        print(f'{os.path.basename(f)[6:-4]:32s} n={rel.size:6d} plain: within 1e-5 {100 * (rel < 1e-5).mean():6.2f} %  max {rel.max():.1e}')
        for pct, target in ((0.9, 7e-6), (0.9, 5e-6), (0.9, 3.5e-6), (0.9, 2.5e-6), (0.99, 7e-6), (0.99, 5e-6), (1.0, 1e-5), (1.0, 7e-6)):
            o = replay(score, e32, e64, pct, target)
            print(f'    pct {pct:4.2f} target {target:.1e}: thr {o["thr"]:9.1f} refined {100 * o["refined"]:5.1f} %  kept max {o["kept_max"]:.2e}  kept >= 1e-5: {o["kept_above_1e-5"]}')
