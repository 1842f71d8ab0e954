"""python -m deepqmc_amd.codegen [--check]: (re)generate deepqmc_amd/csrc/gen/*.hip; --check only compares."""
import os
import sys

import numpy as np

from . import TARGETS, generate
from .. import spec as spec_mod
from ..molecule import Molecule
from ..hamil import MolecularHamiltonian
from ..params import init_params
from ..program import compile_program

F32_EPS = float(np.finfo(np.float32).eps)


def target_program(mol_name, ansatz):
    mol = Molecule.from_name(mol_name)
    h = MolecularHamiltonian(mol=mol)
    sp = getattr(spec_mod, ansatz)()
    tree = init_params(sp, h.n_up, h.n_down, h.n_nuc, seed=0)
    return compile_program(sp, tree, h.n_up, h.n_down, h.n_nuc, R=np.asarray(mol.coords, np.float64), eps=F32_EPS)


def main(argv):
    check = '--check' in argv
    gen_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'csrc', 'gen')
    os.makedirs(gen_dir, exist_ok=True)
    rc = 0
    for (name, mol_name, ansatz) in TARGETS:
        src = generate(name, target_program(mol_name, ansatz))
        path = os.path.join(gen_dir, f'substep_{name}.hip')
        old = open(path).read() if os.path.exists(path) else None
        if check:
            if old != src:
                print(f'{path}: out of date (run python -m deepqmc_amd.codegen)')
                rc = 1
        elif old != src:
            with open(path, 'w') as f:
                f.write(src)
            print(f'wrote {path} ({len(src.splitlines())} lines)')
        else:
            print(f'{path}: up to date')
    inc = ''.join(f'DQMC_SPEC_KERNEL({name})\n' for (name, _, _) in TARGETS)
    ipath = os.path.join(gen_dir, 'spec_list.inc')
    old = open(ipath).read() if os.path.exists(ipath) else None
    if old != inc:
        if check:
            print(f'{ipath}: out of date'); rc = 1
        else:
            open(ipath, 'w').write(inc)
    return rc


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
