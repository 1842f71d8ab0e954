"""Collect the reference's parameter-free golden vectors into one small fixture.

Run in the build container (needs /root/reference):  python tests/golden/make_golden.py
Writes tests/golden/reference_kats.npz.  Only *data* produced by the reference's own test
suite is gathered (pytest-regressions .npz files under /root/reference/tests); no
reference source is copied.  These are the goldens usable without the reference's PRNG /
haiku parameter initialisation.
"""
import os

import numpy as np

REF = '/root/reference/tests'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_kats.npz')


def main():
    out = {}
    for ms in ('True', 'False'):
        d = np.load(f'{REF}/test_gnn/test_graph_edge_builder_mask_self_{ms}_.npz')
        out[f'graph_edges_mask_self_{ms}'] = d['graph_edges']
    d = np.load(f'{REF}/test_gnn/test_molecular_graph_edge_builder.npz')
    for k in ('ne', 'same', 'anti'):
        out[f'lih_edges_{k}'] = d[k]
    d = np.load(f'{REF}/test_potential/test_pseudo_potentials_LiH_None_.npz')
    for k in d.files:
        out[f'lih_potential_{k}'] = d[k]
    for tag in ('Molecular_', 'Molecular_PP_'):
        d = np.load(f'{REF}/test_hamil/test_init_{tag}.npz')
        for k in d.files:
            out[f'hamil_init_{tag}{k}'] = d[k]
    for name in ('LiH', 'H2O'):
        d = np.load(f'{REF}/test_molecule/test_from_name_{name}_.npz')
        for k in d.files:
            out[f'molecule_{name}_{k}'] = d[k]
    # parameter-dependent goldens, recorded for the day the parameter stream is available
    out['wf_psi_log'] = np.load(f'{REF}/test_wf/test_psi.npz')['log']
    out['wf_psi_sign'] = np.load(f'{REF}/test_wf/test_psi.npz')['sign']
    d = np.load(f'{REF}/test_wf/test_laplace_psi.npz')
    out['wf_lap_log_psis'] = d['lap_log_psis']
    out['wf_quantum_force'] = d['quantum_force']
    out['hamil_E_loc'] = np.load(f'{REF}/test_hamil/test_local_energy_Molecular_.npz')['E_loc']
    d = np.load(f'{REF}/test_wf/test_grad_psi.npz')
    out['wf_grad_conf_coeff_w'] = d['neural_network_wave_function/~/conf_coeff:w']
    # sampler goldens (tests/test_sampling.py): initial state and state/stats after 4 sample() calls
    for tag in ('Metropolis', 'DecorrMetropolis', 'Langevin'):
        d = np.load(f'{REF}/test_sampling/test_sampler_init_{tag}_.npz')
        for k in d.files:
            out[f'sampler_init_{tag}_{k}'] = d[k]
        d = np.load(f'{REF}/test_sampling/test_sampler_sample_{tag}_.npz')
        for k in d.files:
            out[f'sampler_sample_{tag}_{k}'] = d[k]
    np.savez(OUT, **out)
    for k, v in out.items():
        print(k, np.asarray(v).shape)


if __name__ == '__main__':
    main()
