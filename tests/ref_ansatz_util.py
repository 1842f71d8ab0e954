"""The reference's test ansatz (tests/conf/ansatz.yaml) for the HIP path: the parameters the emulated
`hk.transform(...).init(PRNGKey(0))` produces (oracle/ref_test_ansatz.init_params) arranged as the haiku tree of the
reference (module names = the keys of the reference's tests/test_wf/test_grad_psi.npz), plus engine construction."""
import numpy as np
import torch

from deepqmc_amd.engine import Engine
from deepqmc_amd.hamil import MolecularHamiltonian
from deepqmc_amd.molecule import Molecule
from deepqmc_amd.program_featurewise import FeaturewiseSpec, compile_featurewise
from oracle import ref_test_ansatz as rta

WF = 'neural_network_wave_function'
OMNI = f'{WF}/~/omni_net'
GNN = f'{OMNI}/~/electron_gnn'
LAYER = f'{GNN}/~/electron_gnn_layer'
CONV = f'{LAYER}/~/convolution_electron_update_feature/~single_edge_type_update'


def haiku_tree(P):
    t = {f'{WF}/~/conf_coeff': {'w': P['conf_coeff']},
         f'{WF}/~/exponential_envelopes': {'pi': P['pi'], 'zetas': P['zetas']},
         f'{GNN}/~/electron_embedding/ElectronicEmbedding': {'embeddings': P['el_embed']},
         f'{GNN}/~/nuclei_embedding/~/embed': {'embeddings': P['nuc_embed']}}
    for e in ('same', 'anti', 'ne'):
        t[f'{CONV}/w_{e}/linear_0'] = {'w': P[f'w_{e}']}
        t[f'{CONV}/h_{e}/linear_0'] = {'w': P[f'h_{e}_w'], 'b': P[f'h_{e}_b']}
        t[f'{LAYER}/~/g_conv_{e}/linear_0'] = {'w': P[f'g_{e}_w'], 'b': P[f'g_{e}_b']}
    for k in range(3):
        t[f'{OMNI}/~/Jastrow/~/mlp/linear_{k}'] = {'w': P[f'jas_{k}_w'], **({'b': P[f'jas_{k}_b']} if k < 2 else {})}
        t[f'{OMNI}/~/Backflow/~/mlp/linear_{k}'] = {'w': P[f'bf_up_{k}_w'], 'b': P[f'bf_up_{k}_b']}
        t[f'{OMNI}/~/Backflow_1/~/mlp/linear_{k}'] = {'w': P[f'bf_down_{k}_w'], 'b': P[f'bf_down_{k}_b']}
    return {m: {k: np.asarray(v, np.float64) for k, v in leaves.items()} for m, leaves in t.items()}


def reference_test_engine(device, dtype=torch.float64, lib=None, norm_eps=None, partitionable=True):
    """Engine of the reference's test ansatz on LiH with the reference's own initial parameters."""
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    P = rta.init_params(partitionable)
    tree = haiku_tree(P)
    fs = FeaturewiseSpec()
    comp = lambda p_: compile_featurewise(fs, p_, h.n_up, h.n_down, h.n_nuc, h.mol_shells)
    eng = Engine(fs.as_ansatz_spec(), h, tree, dtype=dtype, device=device, lib=lib, norm_eps=norm_eps, compiler=comp)
    return h, P, tree, eng
