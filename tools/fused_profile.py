"""Shader-clock profile of the descriptor-driven fused psi kernel (workgroup 0): per wave, cycles spent in each
descriptor of its work list.  Usage: python tools/fused_profile.py [--wt 4]  (prints the plan on stderr)."""
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.sampling import synthetic_walkers
from deepqmc_amd.wf import NeuralNetworkWaveFunction

ap = argparse.ArgumentParser(); ap.add_argument('--wt', type=int, default=0); ap.add_argument('--walkers', type=int, default=4096)
ap.add_argument('--sched', type=int, default=-1); ap.add_argument('--opt', action='append', default=[]); ap.add_argument('--quiet', type=int, default=0); ap.add_argument('--substep', type=int, default=0)
args = ap.parse_args()
h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float32, device='cuda:0')
params = wf.init(0, perturb_envelopes=0.05)
eng = wf.engine(params)
if args.sched >= 0:
    eng.set_option('fused_sched', args.sched)
if args.wt:
    eng.set_option('fused_wt', args.wt)
for o in args.opt:
    k, v = o.split('='); eng.set_option(k, int(v))
eng.set_option('fused_dbg', 1)
if not args.quiet:
    eng.set_option('fused_print', 2)
r = torch.as_tensor(synthetic_walkers(h, args.walkers).astype(np.float32), device='cuda:0')
if args.substep:       # whole Metropolis sub-steps: the last three stamps of each wave are the tail (matrices, determinants, accept)
    sg, lg = eng.wf_eval(r)
    st = {'r': r.clone(), 'log': lg, 'sign': sg, 'age': torch.zeros(args.walkers, dtype=torch.int32, device='cuda:0'),
          'tau': torch.full((1,), 0.3, dtype=torch.float32, device='cuda:0')}
    eng.mcmc_steps(st, 3, seed=1)
for _ in range(0 if args.substep else 3):
    eng.wf_eval(r)
torch.cuda.synchronize()
out = np.zeros(1024)
eng._check(eng.lib.dqmc_debug_read(eng._ctx, -3, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), out.size))
st = out.reshape(4, 256)
t0 = st[:, 0].min()
for w in range(4):
    row = st[w]
    n = int((row > 0).sum())
    print('wave', w, 'start', int(row[0] - t0), 'end', int(row[n - 1] - t0), 'deltas', [int(x) for x in np.diff(row[:n])])
# time between consecutive barriers as seen by wave 0 (level durations) is in the deltas: descriptor k of the plan
# printed on stderr ("[dqmc] wave w desc k ...") took deltas[k] cycles
