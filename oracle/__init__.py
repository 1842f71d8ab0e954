"""CPU oracle: a float64 PyTorch/NumPy restatement of the reference's local-energy / MCMC path.

TEST INFRASTRUCTURE ONLY.  Nothing under `deepqmc_amd/` imports this package; only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may.  Each function
cites the reference file:line it restates.  The reference itself (JAX/haiku, Python>=3.12)
cannot be imported in this environment.

Parity status: PINNED.  The oracle reproduces
  * the reference's parameter-free goldens (edge construction/ordering, Coulomb terms,
    Hamiltonian integers, geometries) -- tests/test_oracle_golden.py, and
  * its parameter-DEPENDENT goldens test_wf/test_psi.npz (log|psi| to 6e-7), test_laplace_psi.npz
    (Laplacian to 6e-8 relative, quantum force to 5e-7), test_hamil/test_local_energy_Molecular_.npz
    (E_loc to 1.4e-7 relative; the reference's own tolerance is 2e-4) and the CI weights of
    test_grad_psi.npz -- tests/test_reference_goldens.py, by emulating JAX's threefry PRNG and
    haiku's initialisation order for the reference's test ansatz (oracle/jaxrng.py,
    oracle/ref_test_ansatz.py), and
  * its sampler goldens test_sampling/test_sampler_{init,sample}_{Metropolis,DecorrMetropolis}_.npz:
    psi of the 10 initial walkers, then 4 x sample(PRNGKey(step)) through oracle/sampling.py driven by
    the emulated jax.random.split/normal/uniform streams -- ages and tau exactly, positions to 1e-12.
  * the Langevin golden test_sampler_sample_Langevin_.npz the same way (oracle/sampling.py: langevin_step,
    clean_force).
Not pinned: the electron initialiser (jax.random.categorical/orthogonal), hk.MultiHeadAttention / LayerNorm /
GLU semantics and the TransPsiformer's nuclei embedding / nuclear head (no reference test instantiates the
Psiformer or TransPsiformer configs), ECP values (oracle/ecp.py restates the formulas; pyscf's coefficient
tables are absent, so the reference's ECP goldens cannot be reproduced -- checked by properties instead).
"""
