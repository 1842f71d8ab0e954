"""Build-time code generation: plan-specialised HIP kernels written from a layer program (substep.py).
`python -m deepqmc_amd.codegen` rewrites deepqmc_amd/csrc/gen/*.hip for the programs listed in TARGETS."""
from .substep import Unsupported, generate, program_hash  # noqa: F401

# (kernel name, molecule, ansatz): the programs that get a specialised sub-step kernel linked into libdqmc_hip.so
TARGETS = [('lih_paulinet', 'LiH', 'paulinet')]
