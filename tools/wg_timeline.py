"""Start / end times (constant 100 MHz counter) of every workgroup of one k_fused2_value sub-step launch: how many run
concurrently, how long each takes, when the last one starts.  python tools/wg_timeline.py [--walkers 4096]"""
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.sampling import synthetic_walkers
from deepqmc_amd.wf import NeuralNetworkWaveFunction

ap = argparse.ArgumentParser(); ap.add_argument('--walkers', type=int, default=4096); ap.add_argument('--opt', action='append', default=[])
args = ap.parse_args()
h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float32, device='cuda:0')
params = wf.init(0, perturb_envelopes=0.05)
eng = wf.engine(params)
for o in args.opt:
    k, v = o.split('='); eng.set_option(k, int(v))
r = torch.as_tensor(synthetic_walkers(h, args.walkers).astype(np.float32), device='cuda:0')
sg, lg = eng.wf_eval(r)
st = {'r': r.clone(), 'log': lg, 'sign': sg, 'age': torch.zeros(args.walkers, dtype=torch.int32, device='cuda:0'),
      'tau': torch.full((1,), 0.3, dtype=torch.float32, device='cuda:0')}
eng.mcmc_steps(st, 30, seed=1)          # warm clocks
eng.set_option('fused_dbg', 3)
eng.mcmc_steps(st, 30, seed=2)          # the stamps of the LAST sub-step survive
torch.cuda.synchronize()
n_ops = len(eng.program.ops)
n = 9 * n_ops + 80 + 1024 + 2 * 8192
out = np.zeros(n)
eng._check(eng.lib.dqmc_debug_read(eng._ctx, -3, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), out.size))
wg = out[9 * n_ops + 80 + 1024:].reshape(8192, 2)
nb = (args.walkers + 3) // 4
wg = wg[:nb]
t0 = wg[:, 0].min()
s, e = (wg[:, 0] - t0) / 100.0, (wg[:, 1] - t0) / 100.0          # microseconds
d = e - s
print('walkers', args.walkers, 'workgroups', nb, 'kernel span us %.1f' % e.max())
print('start  us: p0 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f' % tuple(np.quantile(s, [0, .5, .9, .99, 1])))
print('dur    us: p0 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f' % tuple(np.quantile(d, [0, .5, .9, .99, 1])))
print('end    us: p0 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f' % tuple(np.quantile(e, [0, .5, .9, .99, 1])))
for t in (5, 20, 40, 60, 80, 100, 120, 140):
    print('  running at %3d us: %d' % (t, int(((s <= t) & (e > t)).sum())))
print('by blockIdx quartile: mean start', [round(float(s[k * nb // 4:(k + 1) * nb // 4].mean()), 1) for k in range(4)],
      'mean dur', [round(float(d[k * nb // 4:(k + 1) * nb // 4].mean()), 1) for k in range(4)])
