"""ECP set B (tests/golden/parity_benzene_ecpB_psiformer_256.npz) and set A (…ecp_psiformer_32): E_loc error, pair classes and time
per call for a grid of the two quadrature thresholds ("ecp_heavy_e6", "ecp_dlog_floor_e6")."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_parity_full import load, DEV
for name in ('benzene_ecpB_psiformer_256', 'benzene_ecp_psiformer_32'):
    d, meta, h, eng = load(name)
    r = torch.as_tensor(d['r'], device=DEV)
    phi = torch.as_tensor(d['ecp_phi'], dtype=torch.float32, device=DEV)
    eng.local_energy(r, rng=0, ecp_phi=phi)          # calibrating call
    for heavy in (10000, 3000, 1000):
        for floor in (100, 30, 0):
            eng.set_option('ecp_heavy_e6', heavy); eng.set_option('ecp_dlog_floor_e6', floor)
            eng.local_energy(r, rng=0, ecp_phi=phi)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            e, st = eng.local_energy(r, rng=0, ecp_phi=phi)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            rel = np.abs(e.double().cpu().numpy() - d['e_loc']) / np.maximum(1, np.abs(d['e_loc']))
            vn = np.abs(st['hamil/V_nl'].double().cpu().numpy() - d['stats'][3])
            c = eng.ecp_counts()
            print(f"{name[:14]} heavy {heavy * 1e-6:.0e} floor {floor * 1e-6:.0e}: max {rel.max():.2e} p99 {np.quantile(rel, .99):.2e} >1e-5: {(rel >= 1e-5).sum()}  V_nl abs max {vn.max():.2e}  "
                  f"pairs f32/f64/dropped {c['f32']}/{c['f64']}/{c['dropped']} ({c['f64'] / max(1, c['f32'] + c['f64']):.2f} f64)  {dt * 1e3:.0f} ms", flush=True)
    del eng
