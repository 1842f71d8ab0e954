"""Accuracy / cost of the mixed-precision ECP quadrature on the benzene + ECP parity fixture: V_nl and E_loc errors against the
oracle and the share of float64 pairs for a sweep of the weight threshold ("ecp_heavy_e6").  -> gpurun_out/ecp_sweep.json"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_parity_full import load, DEV
d, meta, h, eng = load('benzene_ecp_psiformer_32')
r = torch.as_tensor(d['r'], device=DEV)
phi = torch.as_tensor(d['ecp_phi'], dtype=torch.float32, device=DEV)
out = []
for heavy in (0, 1000, 10_000, 100_000, 1_000_000, 2_000_000_000):
    eng.set_option('ecp_heavy_e6', heavy)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e, stats = eng.local_energy(r, rng=0, ecp_phi=phi)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c = eng.ecp_counts()
    rel = np.abs(e.double().cpu().numpy() - d['e_loc']) / np.maximum(1.0, np.abs(d['e_loc']))
    vnl = np.abs(stats['hamil/V_nl'].double().cpu().numpy() - d['stats'][3])
    out.append({'ecp_heavy_e6': heavy, 'counts': c, 'f64_share': c['f64'] / max(1, c['f32'] + c['f64']), 'seconds': dt, 'n_refined': eng.last_refined(),
                'e_rel_max': float(rel.max()), 'e_rel_p50': float(np.median(rel)), 'vnl_abs_max': float(vnl.max()), 'vnl_abs_p50': float(np.median(vnl)),
                'e_abs_scale': float(np.abs(d['e_loc']).mean())})
    print(out[-1], flush=True)
eng.set_option('ecp_mixed', 0)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e, stats = eng.local_energy(r, rng=0, ecp_phi=phi)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
rel = np.abs(e.double().cpu().numpy() - d['e_loc']) / np.maximum(1.0, np.abs(d['e_loc']))
out.append({'ecp_mixed': 0, 'seconds': dt, 'n_refined': eng.last_refined(), 'e_rel_max': float(rel.max())})
print(out[-1])
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'ecp_sweep.json'), 'w'), indent=1)
