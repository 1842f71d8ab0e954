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
    targets = [(n, m, a, {}) for (n, m, a) in TARGETS]
    # development: DQMC_CODEGEN_VARIANTS="suffix:key=value,key=value;..." adds variants of the FIRST target (same program hash;
    # the library picks one by DQMC_SPEC_VARIANT=<name>); never committed
    for item in filter(None, os.environ.get('DQMC_CODEGEN_VARIANTS', '').split(';')):
        suffix, _, kv = item.partition(':')
        opts = {}
        for pair in filter(None, kv.split(',')):
            k_, _, v_ = pair.partition('=')
            opts[k_] = tuple(int(x) for x in v_.split('/')) if '/' in v_ else (int(v_) if v_.lstrip('-').isdigit() else v_)
        targets.append((f'{TARGETS[0][0]}__{suffix}', TARGETS[0][1], TARGETS[0][2], opts))
    for (name, mol_name, ansatz, opts) in targets:
        src = generate(name, target_program(mol_name, ansatz), **opts)
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
    inc = ''.join(f'DQMC_SPEC_KERNEL({name})\n' for (name, _, _, _) in targets)
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
