#!/usr/bin/env python
"""bench.py -- walker*local-energy evals/sec and ms per VMC step on MI355X.

One "step" is the evaluation-mode VMC iteration of the reference (fit.py:60-113 with
NoOptimizer): `n_sub` Metropolis sub-steps (value-only psi) + one local
energy per walker + the cross-GPU energy mean/variance (one RCCL all-gather of a 56-byte
record).  Workload = BASELINE.json configs[1]: LiH (4 e-), PauliNet ansatz, 4096 walkers per
GPU (weak scaling), synthetic walkers and random-init weights, float32.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (the forward-Laplacian
linear layer) against the exact-f32 MFMA peak; `cpu_baseline` is the PyTorch-CPU oracle timed
on this box's host cores on a bounded sample (a reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from deepqmc_amd import MolecularHamiltonian, Molecule  # noqa: E402
from deepqmc_amd import parallel  # noqa: E402
from deepqmc_amd.sampling import DecorrSampler  # noqa: E402
from deepqmc_amd.wf import NeuralNetworkWaveFunction  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense


def _cpu_worker(args):
    """One single-threaded oracle process: times psi and local-energy evaluations."""
    molname, spec_name, seed, n_sub, budget_s, widx = args
    import torch as _t
    _t.set_num_threads(1)
    from deepqmc_amd.params import init_params
    from deepqmc_amd.sampling import synthetic_walkers
    from deepqmc_amd.spec import ANSATZES
    from oracle import geom, physics
    from oracle import wf as owf
    hamil = MolecularHamiltonian(mol=Molecule.from_name(molname))
    spec = ANSATZES[spec_name]()
    p = owf.to_torch(init_params(spec, hamil.n_up, hamil.n_down, hamil.n_nuc, seed=seed, perturb_envelopes=0.05))
    T = lambda a: _t.as_tensor(np.asarray(a), dtype=_t.float64)
    R, Z = T(hamil.mol.coords), T(hamil.mol.charges)
    r = T(synthetic_walkers(hamil, 256, seed=100 + widx))
    physics.batch_local_energy(p, spec, r[:1], R, Z, hamil.n_up, geom.F32_EPS)      # warm-up
    t0, n_e = time.perf_counter(), 0
    while time.perf_counter() - t0 < 0.6 * budget_s:
        physics.batch_local_energy(p, spec, r[n_e % 256:n_e % 256 + 1], R, Z, hamil.n_up, geom.F32_EPS)
        n_e += 1
    t_e = time.perf_counter() - t0
    t0, n_w = time.perf_counter(), 0
    while time.perf_counter() - t0 < 0.4 * budget_s:
        physics.batch_wave_function(p, spec, r[n_w % 256:n_w % 256 + 1], R, hamil.n_up, geom.F32_EPS)
        n_w += 1
    t_w = time.perf_counter() - t0
    return n_e, t_e, n_w, t_w


def cpu_baseline(molname, spec_name, n_sub, budget_s=20.0):
    """The oracle ("port" of the reference's JAX-CPU path: same per-walker algorithm,
    Hessian-trace Laplacian by forward-over-reverse autodiff, float64 PyTorch) timed on this
    box's host cores: one single-threaded process per core (<= 64), a bounded sample each."""
    import multiprocessing as mp
    cores = max(1, min(os.cpu_count() or 1, 64))
    with mp.get_context('spawn').Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(molname, spec_name, 0, n_sub, budget_s, w) for w in range(cores)])
    eloc_rate = sum(n_e / t_e for n_e, t_e, _, _ in res)      # aggregate over processes
    wf_rate = sum(n_w / t_w for _, _, n_w, t_w in res)
    per_walker_step = n_sub / wf_rate + 1.0 / eloc_rate
    return {
        'value': 1.0 / per_walker_step, 'unit': 'walker*E_loc evals/s (VMC step incl. %d sub-steps)' % n_sub,
        'eloc_only_evals_per_s': eloc_rate, 'psi_evals_per_s': wf_rate, 'cores': cores, 'kind': 'port',
        'sample': f'{sum(x[0] for x in res)} local energies + {sum(x[2] for x in res)} psi evaluations in '
                  f'{budget_s:.0f} s over {cores} single-threaded processes, PyTorch-CPU float64 oracle '
                  f'(the reference JAX-CPU path is not runnable in this image)',
    }


def committed_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 counter passes (run_traffic.sh ->
    tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE in separate --pmc passes, gfx950 x2 correction on
    FETCH_SIZE).  Counters cannot be collected from inside the timed process, so the figure is the
    one measured for the same workload by the profiling run; None if no such profile is committed."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*pmc_hbm_traffic*.json')), reverse=True):
        try:
            prof = json.load(open(path))
        except (OSError, ValueError):
            continue
        if prof.get('workload') == workload and kernel in prof.get('kernels', {}):
            return prof['kernels'][kernel]['hbm_bytes'], os.path.relpath(path, ROOT)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--walkers', type=int, default=4096, help='walkers per GPU')
    ap.add_argument('--n-sub', type=int, default=30, help='Metropolis sub-steps per VMC step (reference preset 30)')
    ap.add_argument('--molecule', default='LiH')
    ap.add_argument('--ansatz', default='paulinet')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f64'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--ecp', action='store_true', help='Gaussian-type ECP on every atom heavier than He with SYNTHETIC '
                    'coefficients (pyscf tables are not available offline): exercises the 12 N n_ecp psi-ratio quadrature')
    ap.add_argument('--fused', type=int, default=1, help='0: one launch per op for psi evaluation')
    ap.add_argument('--overlap', type=int, default=0, help='1: E_loc of step k on a second HIP stream, overlapped with the '
                    'Metropolis sub-steps of step k+1 (software pipelining; same work per step)')
    ap.add_argument('--attention-mfma', type=int, default=-1, help='0: scalar attention kernel, 2: MFMA kernel wherever supported '
                    '(library default 1: MFMA where profitable)')
    ap.add_argument('--fused-version', type=int, default=0, help='1: first fused kernel (in-kernel op interpreter); 2 (library default): descriptor driven')
    ap.add_argument('--fused-wt', type=int, default=0, help='walkers per workgroup tile of the fused psi kernel')
    ap.add_argument('--fused-dbg', type=int, default=0, help='ablation bitmask of the fused kernel (profiling only)')
    ap.add_argument('--fused-occ', type=int, default=0, help='register budget of the fused kernel: workgroups per CU (2..4)')
    ap.add_argument('--fused-lds-kb', type=int, default=0, help='LDS budget (KiB) for the automatic tile choice')
    ap.add_argument('--fused-sched', type=int, default=-1, help='0: keep program order; 1 (library default): reorder ops into full dependency levels (more LDS)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback)'
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=device)

    dtype = torch.float32 if args.dtype == 'f32' else torch.float64
    mol = Molecule.from_name(args.molecule)
    if args.ecp:
        from deepqmc_amd.ecp import ELEMENTS
        tab = lambda z: [2 if z > 2 else 0, [[-1, [[], [[5.4, float(z - 2)]], [[4.6, -4.6]], [[2.7, 5.4]]]],
                                              [0, [[], [], [[1.33, 6.75]]]], [1, [[], [], [[1.25, 0.45]]]]]]
        hamil = MolecularHamiltonian(mol=mol, ecp_type='synthetic',
                                     ecp_tables={ELEMENTS[int(z)]: tab(int(z)) for z in set(mol.charges) if z > 2})
    else:
        hamil = MolecularHamiltonian(mol=mol)
    wf = NeuralNetworkWaveFunction(hamil, args.ansatz, dtype=dtype, device=device)
    params = wf.init(0, perturb_envelopes=0.05)
    eng = wf.engine(params)
    if args.attention_mfma >= 0:
        eng.set_option('attention_mfma', args.attention_mfma)
    if args.fused_version:
        eng.set_option('fused_version', args.fused_version)
    if args.fused_sched >= 0:
        eng.set_option('fused_sched', args.fused_sched)
    if args.fused_occ:
        eng.set_option('fused_occ', args.fused_occ)
    if args.fused_lds_kb:
        eng.set_option('fused_lds_kb', args.fused_lds_kb)
    if args.fused_wt:
        eng.set_option('fused_wt', args.fused_wt)
    eng.set_option('fused', args.fused)
    if args.fused_dbg:
        eng.set_option('fused_dbg', args.fused_dbg)
    B = args.walkers
    sampler = DecorrSampler(hamil, wf, length=args.n_sub)
    state = sampler.init(1000 + rank, params, B)
    loc_ene = hamil.local_energy(wf)

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    def vmc_step(step, state):
        if args.n_sub > 0:
            state, pc, _ = sampler.sample(step * world + rank, state, params)
            r = state['r']
        else:
            r = state['r']
        e, _ = loc_ene(step, params, r)
        stats = parallel.energy_stats(eng, e)
        return state, stats

    # ---- software-pipelined variant: E_loc(k) runs on a second stream while the sub-steps of step k+1 run ----
    pipe = {'e': None, 'stats': None}
    if args.overlap:
        from deepqmc_amd.engine import Engine
        s_e = torch.cuda.Stream(device)
        with torch.cuda.stream(s_e):
            eng_e = Engine(wf.spec, hamil, params, dtype=dtype, device=device)      # its work goes to s_e
        s_main = torch.cuda.current_stream(device)

        def vmc_step_pipelined(step, state):
            state, pc, _ = sampler.sample(step * world + rank, state, params)        # enqueued on the main stream
            if pipe['e'] is not None:                                                # E_loc of the previous step
                with torch.cuda.stream(s_e):
                    pipe['stats'] = parallel.energy_stats(eng_e, pipe['e'])          # syncs s_e only
            r_snap = state['r'].clone()
            r_snap.record_stream(s_e)
            ev = s_main.record_event()
            with torch.cuda.stream(s_e):
                s_e.wait_event(ev)
                pipe['e'], _ = eng_e.local_energy(r_snap)
            return state, pipe['stats']

        def drain():
            with torch.cuda.stream(s_e):
                pipe['stats'] = parallel.energy_stats(eng_e, pipe['e'])
            pipe['e'] = None
            return pipe['stats']

    step_fn = vmc_step_pipelined if args.overlap else vmc_step
    for s in range(args.warmup):
        state, stats = step_fn(s, state)
    if args.overlap:
        stats = drain()
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        state, stats = step_fn(args.warmup + s, state)
    if args.overlap:
        stats = drain()          # the last step's E_loc and reduction are inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = B * world / (elapsed / args.steps)

    # ---- pure E_loc throughput (n_sub = 0), not the headline ----
    r = state['r']
    for _ in range(2):
        loc_ene(None, params, r)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    n_rep = max(5, args.steps)
    for _ in range(n_rep):
        loc_ene(None, params, r)
    torch.cuda.synchronize(device)
    eloc_only = B * world / ((time.perf_counter() - t0) / n_rep)

    # ---- roofline of the dominant kernel: HIP events around every launch, same workload ----
    eng.timing(True)
    eng.timing_reset()
    for s in range(3):
        state, stats = vmc_step(10_000 + s, state)
    torch.cuda.synchronize(device)
    rep = eng.timing_report()
    eng.timing(False)
    names = {'linear': 'k_linear (forward-Laplacian linear layer, v_mfma_f32_16x16x4_f32)',
             'fused_psi': 'k_fused2_value (LDS-resident psi evaluation, v_mfma_f32_16x16x4_f32)',
             'fused_substep': 'k_fused2_value (one launch per Metropolis sub-step: propose + LDS-resident psi + '
                              'determinants + accept, v_mfma_f32_16x16x4_f32)'}
    cands = {k: rep[k] for k in names if k in rep and rep[k]['ms'] > 0}
    dom = max(cands, key=lambda k: cands[k]['ms']) if cands else 'linear'
    lin = rep.get(dom, {'ms': 0.0, 'launches': 0, 'flops': 0.0})
    total_ms = sum(v['ms'] for v in rep.values()) or 1.0
    achieved = lin['flops'] / (lin['ms'] * 1e-3) / 1e12 if lin['ms'] > 0 else 0.0
    traffic, traffic_src = committed_traffic({'linear': 'k_linear', 'fused_psi': 'k_fused2_value', 'fused_substep': 'k_fused2_value'}[dom],
                                             f'{args.molecule}/{args.ansatz}/{B}/{args.dtype}')
    roofline = {
        'bound': 'mfma', 'kernel': names[dom],
        'per_kernel_tflops': {k: v['flops'] / (v['ms'] * 1e-3) / 1e12 for k, v in cands.items()},
        'achieved': achieved, 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / F32_MFMA_PEAK_TFLOPS,
        'traffic': traffic, 'traffic_unit': 'HBM bytes per launch', 'traffic_source': traffic_src,
        'avg_launch_us': 1e3 * lin['ms'] / max(lin['launches'], 1), 'launches_per_step': lin['launches'] / 3,
        'share_of_kernel_time': lin['ms'] / total_ms,
        'kernel_ms_per_step': {k: v['ms'] / 3 for k, v in rep.items()},
    }

    if rank == 0:
        out = {
            'metric': 'walker*local-energy evals/sec (VMC step: n_sub Metropolis sub-steps + E_loc + energy reduction)',
            'value': value, 'unit': 'walker*E_loc evals/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic walkers, random-init weights',
            'config': {'workload': f'{args.molecule} ({hamil.n_elec} e-), {args.ansatz} ansatz, {B} walkers/GPU, '
                                   f'{args.n_sub} Metropolis sub-steps + local energy + RCCL energy stats',
                       'walkers_per_gpu': B, 'n_sub': args.n_sub, 'parallelism': f'walker-dp{world}'},
            'eloc_only_evals_per_s': eloc_only,
            'energy': stats,
            'flops_per_eloc': (3 * hamil.n_elec + 2) * eng.program.flops_per_walker,
            'roofline': roofline,
        }
        if args.ecp:
            out['data'] += ', synthetic ECP coefficients'
            out['config']['workload'] += ' + Gaussian-type ECP (12-point quadrature)'
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.molecule, args.ansatz, args.n_sub)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
