"""Per-op shader-clock breakdown of the fused psi kernel (workgroup 0) on the GPU.
Usage: python tools/fused_profile.py [--wt 4]"""
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepqmc_amd import MolecularHamiltonian, Molecule
from deepqmc_amd.sampling import synthetic_walkers
from deepqmc_amd.wf import NeuralNetworkWaveFunction

ap = argparse.ArgumentParser(); ap.add_argument('--wt', type=int, default=4); ap.add_argument('--walkers', type=int, default=4096)
args = ap.parse_args()
h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
wf = NeuralNetworkWaveFunction(h, 'paulinet', dtype=torch.float32, device='cuda:0')
params = wf.init(0, perturb_envelopes=0.05)
eng = wf.engine(params)
eng.set_option('fused_wt', args.wt)
eng.set_option('fused_dbg', 1)
r = torch.as_tensor(synthetic_walkers(h, args.walkers).astype(np.float32), device='cuda:0')
for _ in range(3):
    eng.wf_eval(r)
torch.cuda.synchronize()
n_f = next(k for k, op in enumerate(eng.program.ops) if op.kind == 8) + 1
out = np.zeros(64 + 8 * n_f + 8)
eng._check(eng.lib.dqmc_debug_read(eng._ctx, -3, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), out.size))
d = np.diff(out[:n_f + 1])
print('total cycles (wg 0):', out[n_f] - out[0], ' = %.1f us at 2.4 GHz' % ((out[n_f] - out[0]) / 2400))
for k in range(n_f):
    u = out[64 + 8 * k: 64 + 8 * k + 6]
    if u[0] > 0:
        print('unit', k, [int(x) for x in np.diff(u)])
kinds = {1: 'FEAT_EN', 2: 'FEAT_EE', 3: 'LINEAR', 4: 'SPIN_MEAN', 5: 'CONV', 6: 'EDGE_SUM', 7: 'ROW_SUM', 8: 'ORBITALS'}
# the kernel executes ops in level order; we do not know the order here, so print raw slots
for k, c in enumerate(d):
    print(k, int(c))
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
eng.set_option('fused_dbg', 0)
t0.record()
for _ in range(20):
    eng.wf_eval(r)
t1.record(); torch.cuda.synchronize()
print('wf_eval ms:', t0.elapsed_time(t1) / 20)
