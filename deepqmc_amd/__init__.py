"""deepqmc_amd: an MI355X-native local-energy / MCMC evaluator behind DeepQMC's
`hamil.local_energy`, `sampling.*.sample` and `wf.NeuralNetworkWaveFunction` surface.
See DESIGN.md.  The arithmetic lives in the HIP library `deepqmc_amd/csrc/libdqmc_hip.so`
(C-ABI in include/dqmc.h); importing this package does not load it, using an engine does.
"""
from .hamil import MolecularHamiltonian
from .molecule import Molecule
from .types import PhysicalConfiguration, Psi

__all__ = ['MolecularHamiltonian', 'Molecule', 'PhysicalConfiguration', 'Psi']
__version__ = '0.1.0'
