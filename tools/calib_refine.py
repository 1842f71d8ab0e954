"""Per-walker calibration data for the float64 refinement rule (GPU): for every committed parity fixture the plain
float32 E_loc (refine 0), the oracle's value, the node-cancellation ratio (|lap| + |grad|^2) / max(1, |E|) and the
conditioning record kappa (kernels_head.hip) -> gpurun_out/calib/<fixture>.npz.  Analysis happens offline."""
import glob
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from deepqmc_amd.engine import Engine  # noqa: E402
from deepqmc_amd.hamil import MolecularHamiltonian  # noqa: E402
from deepqmc_amd.molecule import Molecule  # noqa: E402
from deepqmc_amd.params import init_params  # noqa: E402
from deepqmc_amd.spec import ANSATZES  # noqa: E402

OUT = os.path.join(ROOT, 'gpurun_out', 'calib')
os.makedirs(OUT, exist_ok=True)
for path in sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'parity_*.npz'))):
    name = os.path.basename(path)[7:-4]
    d = np.load(path)
    meta = json.loads(str(d['meta']))
    mol = Molecule.from_name(meta['molecule'])
    spec = ANSATZES[meta['ansatz']](mol.charges) if meta['ansatz'] == 'transpsiformer' else ANSATZES[meta['ansatz']]()
    h = MolecularHamiltonian(mol=mol)
    tree = init_params(spec, h.n_up, h.n_down, h.n_nuc, seed=meta['param_seed'], perturb_envelopes=meta['perturb_envelopes'])
    eng = Engine(spec, h, tree, dtype=torch.float32, device='cuda:0', norm_eps=meta['norm_eps'])
    eng.set_option('refine', 0)
    r = torch.as_tensor(d['r'], device='cuda:0')
    B = r.shape[0]
    chunk = min(B, 256)
    e, st, kap = [], [], []
    for a in range(0, B, chunk):
        ee, ss = eng.local_energy(r[a:a + chunk], rng=0)
        e.append(ee.double().cpu().numpy())
        st.append(np.stack([ss[k].double().cpu().numpy() for k in ss]))
        kap.append(eng.debug_read('kappa', min(chunk, B - a)))
    e, st, kap = np.concatenate(e), np.concatenate(st, 1), np.concatenate(kap)
    rel = np.abs(e - d['e_loc']) / np.maximum(1, np.abs(d['e_loc']))
    ratio = (np.abs(st[4]) + st[5]) / np.maximum(1, np.abs(e))
    np.savez(os.path.join(OUT, name + '.npz'), e=e, e_ref=d['e_loc'], stats=st, kappa=kap, rel=rel, ratio=ratio, cond=d['cond'])
    q = lambda x: np.quantile(x, [.5, .9, .99, 1.0])
    print(name, 'rel', q(rel), 'ratio', q(ratio), 'kappa', q(kap), 'corr(log rel, log kappa)',
          np.corrcoef(np.log(rel + 1e-12), np.log(kap))[0, 1], 'corr(log rel, log ratio)', np.corrcoef(np.log(rel + 1e-12), np.log(ratio))[0, 1], flush=True)
    eng.close()
