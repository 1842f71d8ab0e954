#!/usr/bin/env python
"""bench.py -- walker*local-energy evals/sec and ms per VMC step on MI355X.

One "step" is the evaluation-mode VMC iteration of the reference (fit.py:60-113 with
NoOptimizer): `n_sub` Metropolis sub-steps (value-only psi) + one local
energy per walker + the cross-GPU energy mean/variance (one RCCL all-gather of a 56-byte
record).  Workload = BASELINE.json configs[1]: LiH (4 e-), PauliNet ansatz, 4096 walkers per
GPU (weak scaling), synthetic walkers and random-init weights, float32.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...      (re-executes itself under torch.distributed.run, one rank per GPU, RCCL)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

The walkers of an N-GPU run are the reference's split of ONE global batch of N x `--walkers` walkers
(`electron_batch_size // device_count`, sampling_utils.py:253-262 -> parallel.shard_bounds).  The timed
region is `--steps` VMC steps bracketed by barrier + synchronize, repeated back to back until >= 10 s of
steady state have been measured (at least 10 blocks); `ms_per_step` is the median block (max over ranks
per block), min / max are reported beside it.

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
F64_MFMA_PEAK_TFLOPS = 78.6     # v_mfma_f64_16x16x4_f64: 2048 flops in 64 cycles per SIMD (measured, tools/ubench/mfma_rate.hip) x 1024 SIMDs x 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: v_mfma_f32_16x16x32_bf16 / 32x32x16, dense (no 2:1 sparsity)


def _cpu_eloc_fns(molname, spec_name, dtype_name):
    """Batched oracle closures: vmap over walkers of the per-walker local energy (forward-over-reverse Hessian
    trace, `jacfwd(grad(log|psi|))`) and of psi -- the CPU stand-in SURVEY.md section 8d specifies."""
    import torch as _t
    from torch.func import grad, jacfwd, vmap
    from deepqmc_amd.params import init_params
    from deepqmc_amd.spec import ANSATZES
    from oracle import geom, physics
    from oracle import wf as owf
    dt = _t.float32 if dtype_name == 'f32' else _t.float64
    mol = Molecule.from_name(molname)
    hamil = MolecularHamiltonian(mol=mol)
    spec = ANSATZES[spec_name](mol.charges) if spec_name == 'transpsiformer' else ANSATZES[spec_name]()
    p = owf.to_torch(init_params(spec, hamil.n_up, hamil.n_down, hamil.n_nuc, seed=0, perturb_envelopes=0.05), dtype=dt)
    T = lambda a: _t.as_tensor(np.asarray(a), dtype=dt)
    R, Z = T(hamil.mol.coords), T(hamil.mol.charges)
    eps = geom.F32_EPS

    def logpsi(flat):
        return owf.wave_function(p, spec, flat.reshape(-1, 3), R, hamil.n_up, eps)[1]

    def eloc(rw):
        x = rw.reshape(-1)
        g = grad(logpsi)(x)
        lap = _t.diagonal(jacfwd(grad(logpsi))(x)).sum()
        return (-0.5 * (lap + (g ** 2).sum()) + physics.electronic_potential(rw, eps) + physics.local_potential(rw, R, Z)
                + physics.nuclear_energy(R, Z))

    return hamil, T, vmap(eloc), vmap(lambda rw: logpsi(rw.reshape(-1)))


def _cpu_time(fn, x, budget_s, min_rep=3):
    fn(x)                                       # warm-up (first call traces the functorch transforms)
    ts = []
    t_end = time.perf_counter() + budget_s
    while len(ts) < min_rep or (time.perf_counter() < t_end and len(ts) < 50):
        t0 = time.perf_counter()
        fn(x)
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), len(ts)


def _cpu_worker(args):
    """One oracle process with `threads` intra-op threads: median time of a batched E_loc and a batched psi call."""
    molname, spec_name, dtype_name, batch, threads, budget_s, widx = args
    import torch as _t
    _t.set_num_threads(threads)
    from deepqmc_amd.sampling import synthetic_walkers
    hamil, T, f_eloc, f_psi = _cpu_eloc_fns(molname, spec_name, dtype_name)
    r = T(synthetic_walkers(hamil, batch, seed=100 + widx))
    with _t.no_grad():
        t_e, n_e = _cpu_time(f_eloc, r, 0.6 * budget_s)
        t_w, n_w = _cpu_time(f_psi, r, 0.3 * budget_s)
    return batch / t_e, batch / t_w, n_e, n_w


def cpu_baseline_measure(molname, spec_name, n_sub, dtype_name='f32', batch=256, budget_s=8.0):
    """The oracle ("port": the reference's per-walker algorithm restated in PyTorch -- NOT the reference JAX-CPU
    path, which cannot run in this image) timed on this box's host cores in the reference's production dtype,
    batched with torch.func.vmap.  Two layouts are timed on a bounded sample and the faster one is reported:
    one process with up to 16 intra-op threads, and one process per 8 cores (at most 24) with 8 threads each
    (hundreds of intra-op threads on these small tensors only thrash)."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    layouts = [(1, min(cores, 16))]
    if cores >= 16:
        layouts.append((min(cores // 8, 24), 8))
    best = None
    for n_proc, threads in layouts:
        jobs = [(molname, spec_name, dtype_name, batch, threads, budget_s, w) for w in range(n_proc)]
        if n_proc == 1:
            res = [_cpu_worker(jobs[0])]
        else:
            with mp.get_context('spawn').Pool(n_proc) as pool:
                try:
                    res = pool.map_async(_cpu_worker, jobs).get(timeout=20 * budget_s)
                except mp.TimeoutError:
                    continue
        eloc_rate, wf_rate = sum(x[0] for x in res), sum(x[1] for x in res)
        cand = {'eloc_rate': eloc_rate, 'wf_rate': wf_rate, 'n_proc': n_proc, 'threads': threads,
                'n_eloc_calls': sum(x[2] for x in res), 'n_psi_calls': sum(x[3] for x in res)}
        if best is None or cand['eloc_rate'] > best['eloc_rate']:
            best = cand
    per_walker_step = n_sub / best['wf_rate'] + 1.0 / best['eloc_rate']
    return {
        'value': 1.0 / per_walker_step, 'unit': 'walker*E_loc evals/s (VMC step incl. %d sub-steps)' % n_sub,
        'eloc_only_evals_per_s': best['eloc_rate'], 'psi_evals_per_s': best['wf_rate'],
        'cores': best['n_proc'] * best['threads'], 'host_logical_cores': cores, 'kind': 'port',
        'sample': f"oracle-CPU stand-in (not the reference JAX-CPU path): torch.func.vmap over {batch} walkers of the per-walker "
                  f"local energy (jacfwd(grad) Laplacian) and of psi, {dtype_name}, {best['n_proc']} process(es) x {best['threads']} "
                  f"threads, median of {best['n_eloc_calls']} E_loc + {best['n_psi_calls']} psi batched calls "
                  f"(~{budget_s:.0f} s per layout; layouts tried (processes, threads): {layouts})",
    }


def cpu_baseline(molname, spec_name, n_sub, dtype_name='f32', timeout_s=240.0):
    """Runs cpu_baseline_measure in a child process with a hard wall-clock limit, so that a slow host can never
    keep the GPU numbers from being printed."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--molecule', molname, '--ansatz', spec_name,
           '--n-sub', str(n_sub), '--dtype', dtype_name]
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
        if p.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {'value': None, 'kind': 'port', 'error': (p.stderr or 'no output')[-400:]}
    except subprocess.TimeoutExpired:
        return {'value': None, 'kind': 'port', 'error': f'CPU baseline did not finish within {timeout_s:.0f} s'}


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
    ap.add_argument('--equilibrate', type=int, default=400, help='untimed Metropolis sub-steps before the warm-up steps, so that the timed '
                    'steps see |psi|^2-distributed walkers (the reference equilibrates its sampler before training, sampling_utils.py:133-170); 0: off')
    ap.add_argument('--ecp', action='store_true', help='Gaussian-type ECP on every atom heavier than He with SYNTHETIC '
                    'coefficients (pyscf tables are not available offline): exercises the 12 N n_ecp psi-ratio quadrature')
    ap.add_argument('--fused', type=int, default=1, help='0: one launch per op for psi evaluation')
    ap.add_argument('--overlap', type=int, default=0, help='1: E_loc of step k on a second HIP stream, overlapped with the '
                    'Metropolis sub-steps of step k+1 (software pipelining; same work per step)')
    ap.add_argument('--attention-mfma', type=int, default=-1, help='0: scalar attention kernel, 2: MFMA kernel wherever supported '
                    '(library default 1: MFMA where profitable)')
    ap.add_argument('--fused-wt', type=int, default=0, help='walkers per workgroup tile of the fused psi kernel')
    ap.add_argument('--fused-dbg', type=int, default=0, help='ablation bitmask of the fused kernel (profiling only)')
    ap.add_argument('--fused-occ', type=int, default=0, help='register budget of the fused kernel: workgroups per CU (2..4)')
    ap.add_argument('--fused-lds-kb', type=int, default=0, help='LDS budget (KiB) for the automatic tile choice')
    ap.add_argument('--fused-sched', type=int, default=-1, help='0: keep program order; 1 (library default): reorder ops into full dependency levels (more LDS)')
    ap.add_argument('--states', type=int, default=1, help='electronic states (BASELINE configs[4]: 3): one parameter set and one '
                    'Markov-chain ensemble per state, local energy of every state, the S x S psi-ratio matrix (S^2 value-only '
                    'psi evaluations per walker, loss/overlap.py:40-99) and the overlap penalty with its all-reduce')
    ap.add_argument('--refine', type=int, default=-1, help='float64 refinement of ill-conditioned walkers: 0 off, 1 flagged walkers '
                    '(library default), 2 whole E_loc pass in float64')
    ap.add_argument('--opt', action='append', default=[], help='library option name=value (dqmc_set_option), repeatable')
    ap.add_argument('--repeats', type=int, default=0, help='timed blocks of --steps steps (0: as many as fill --min-seconds, at least 10)')
    ap.add_argument('--min-seconds', type=float, default=10.0, help='steady-state time the timed blocks must cover')
    ap.add_argument('--torch-reduce', action='store_true', help='reduce the energy statistics through torch.distributed on the host '
                    '(default: one ncclAllGather inside the library on the context\'s stream)')
    ap.add_argument('--emulated', action='store_true', help='TEST ONLY: CPU SIMT emulation of the kernels + gloo (exercises the '
                    'launch / shard / reduce path without a GPU; the numbers mean nothing)')
    ap.add_argument('--cpu-baseline-only', action='store_true', help='internal: print the cpu_baseline JSON object and exit')
    args = ap.parse_args()

    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_measure(args.molecule, args.ansatz, args.n_sub, args.dtype)))
        return
    if args.gpus < 1:
        sys.exit('bench.py: --gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # launched as plain `python bench.py --gpus N`: become N ranks (one process per GPU) under torch.distributed.run
        n_dev = args.gpus if args.emulated else (torch.cuda.device_count() if torch.cuda.is_available() else 0)
        if n_dev < args.gpus:
            sys.exit(f'bench.py: --gpus {args.gpus} requested but only {n_dev} GPU device(s) are visible on this node')
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    # stdout carries ONE JSON line.  Libraries underneath write there too (this RCCL build prints a version banner through C
    # stdio when a communicator is created, flushed at exit -- after the JSON line): file descriptor 1 points at stderr for
    # the rest of the run, and the result line goes out through a private duplicate of the original descriptor.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} does not match the launcher\'s WORLD_SIZE={world}')
    lib = None
    if args.emulated:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from simt_util import emu_lib
        lib = emu_lib()
        device = torch.device('cpu')
    else:
        if not torch.cuda.is_available():
            sys.exit('bench.py needs a GPU (there is no CPU fallback; --emulated is a test harness)')
        if local_rank >= torch.cuda.device_count():
            sys.exit(f'bench.py: rank {rank} needs GPU {local_rank} but only {torch.cuda.device_count()} device(s) are visible')
        torch.cuda.set_device(local_rank)
        device = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.emulated:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)

    def sync():
        if device.type == 'cuda':
            torch.cuda.synchronize(device)

    t_start = time.perf_counter()

    def log(msg):
        if rank == 0:
            print(f'[bench {time.perf_counter() - t_start:7.1f} s] {msg}', file=sys.stderr, flush=True)

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
    wf = NeuralNetworkWaveFunction(hamil, args.ansatz, dtype=dtype, device=device, lib=lib)
    params = wf.init(0, perturb_envelopes=0.05)
    eng = wf.engine(params)
    if args.attention_mfma >= 0:
        eng.set_option('attention_mfma', args.attention_mfma)
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
    # this rank's shard of the global batch: electron_batch_size // device_count (sampling_utils.py:253-262)
    lo, hi = parallel.shard_bounds(B * world, rank, world)
    from deepqmc_amd.sampling import synthetic_walkers

    def shard_initializer(hamil_, n, seed):
        assert n == hi - lo
        return synthetic_walkers(hamil_, B * world, seed=seed)[lo:hi]

    sampler = DecorrSampler(hamil, wf, length=args.n_sub, sample_initializer=shard_initializer, in_place=not args.overlap)
    state = sampler.init(1000, params, B)
    loc_ene = hamil.local_energy(wf)
    if args.refine >= 0:
        eng.set_option('refine', args.refine)
    for kv in args.opt:
        eng.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    S = args.states
    if S > 1:
        from deepqmc_amd import loss
        from deepqmc_amd.sampling import MultiElectronicStateSampler
        params_s = [params] + [wf.init(s, perturb_envelopes=0.05) for s in range(1, S)]
        ms_sampler = MultiElectronicStateSampler(sampler, S)
        ms_state = ms_sampler.init(1000, params_s, B)
        if args.refine >= 0:
            for p_ in params_s:
                wf.engine(p_).set_option('refine', args.refine)
        ones = torch.ones(1, S, B, dtype=torch.float64, device=device)
    # Energy reduction: inside the library -- record + ONE ncclAllGather on the context's stream over an RCCL
    # communicator of the library's own + one D2H of the gathered records (dqmc_energy_stats_allgather).  The host path
    # (record D2H -> torch.distributed.all_gather -> merge) remains for the emulated test harness and as the fallback if the
    # communicator cannot be created; the JSON line says which one ran (`config.reduction`).
    comm = None
    if not args.emulated and not args.torch_reduce:
        try:
            comm = parallel.RcclCommunicator(rank, world, device)
        except Exception as exc:       # noqa: BLE001 -- whatever RCCL / ctypes raises: report and fall back
            log(f'in-library RCCL communicator unavailable ({exc!r}); reducing through torch.distributed')
    if world > 1:                      # every rank must take the same path
        ok = torch.tensor([1 if comm is not None or args.emulated or args.torch_reduce else 0], dtype=torch.int32,
                          device=device)
        torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
        if int(ok.item()) == 0 and comm is not None:
            comm.close()
            comm = None

    def reduce_stats(engine, e):
        if comm is not None:
            return parallel.energy_stats_inlib(engine, e, comm)
        return parallel.energy_stats(engine, e)

    if comm is not None:
        n_ranks_seen = comm.count()
        reduce_stats(eng, torch.zeros(8, dtype=dtype, device=device))        # one real collective before anything is timed
    else:
        n_ranks_seen = len(parallel.all_gather_records(np.zeros(7), device if not args.emulated else 'cpu'))
    if world > 1:
        assert n_ranks_seen == torch.distributed.get_world_size() == world

    def barrier():
        sync()
        if world > 1:
            torch.distributed.barrier()
        sync()

    def vmc_step_states(step, state):
        """Multi-state step: sample every state, E_loc[1,S,B], psi ratios [1,S,S,B], overlap penalty (one all-reduce)."""
        state, pc, _ = ms_sampler.sample(step * world + rank, state, params_s)
        r = pc.r[None]
        E, _ = loss.compute_local_energy(step, hamil, wf, params_s, r)
        ratio, _ = loss.compute_psi_ratio(wf, params_s, r)
        pen, info = loss.compute_mean_overlap(ratio, ones)
        stats = reduce_stats(eng, E[0, 0].contiguous())
        stats['overlap/penalty'] = float(pen)
        return state, stats

    refined = []           # walkers re-evaluated in float64 per step (host-side counter of the library, no sync)

    def vmc_step(step, state):
        if S > 1:
            return vmc_step_states(step, state)
        if args.n_sub > 0:
            state, pc, _ = sampler.sample(step * world + rank, state, params)
            r = state['r']
        else:
            r = state['r']
        e, _ = loc_ene(step, params, r)
        refined.append(eng.last_refined())
        stats = reduce_stats(eng, e)
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
                    pipe['stats'] = reduce_stats(eng_e, pipe['e'])          # syncs s_e only
            r_snap = state['r'].clone()
            r_snap.record_stream(s_e)
            ev = s_main.record_event()
            with torch.cuda.stream(s_e):
                s_e.wait_event(ev)
                pipe['e'], _ = eng_e.local_energy(r_snap)
            return state, pipe['stats']

        def drain():
            with torch.cuda.stream(s_e):
                pipe['stats'] = reduce_stats(eng_e, pipe['e'])
            pipe['e'] = None
            return pipe['stats']

    step_fn = vmc_step_pipelined if args.overlap else vmc_step
    if S > 1:
        assert not args.overlap, '--overlap is a single-state arrangement'
        state = ms_state
    stats = None
    if args.equilibrate > 0 and args.n_sub > 0:
        # atom-centred Gaussian walkers sit near the nodes of psi far more often than |psi|^2-distributed ones: burn in
        # with plain sub-steps (no local energy) before anything is timed or reported
        burn = DecorrSampler(hamil, wf, length=min(50, args.equilibrate))
        for k in range((args.equilibrate + burn.length - 1) // burn.length):
            if S > 1:
                state = [burn.sample(900_000 + k * S + s_, state[s_], params_s[s_])[0] for s_ in range(S)]
            else:
                state = burn.sample(900_000 + k, state, params)[0]
        sync()
        log(f'equilibrated with {args.equilibrate} sub-steps')
    for s in range(args.warmup):
        state, stats = step_fn(s, state)
    if args.overlap:
        stats = drain()

    def timed_block(first_step):
        nonlocal state, stats
        barrier()
        t0 = time.perf_counter()
        for s in range(args.steps):
            state, stats = step_fn(first_step + s, state)
        if args.overlap:
            stats = drain()          # the last step's E_loc and reduction are inside the timed region
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    log('warm-up done; timing')
    counters0 = eng.refine_counters()
    blocks = [timed_block(args.warmup)]
    n_blocks = args.repeats if args.repeats > 0 else max(10, int(np.ceil(args.min_seconds / max(blocks[0], 1e-6))))
    n_blocks = min(n_blocks, 2000)
    if world > 1:            # every rank must run the same number of blocks
        t = torch.tensor([n_blocks], dtype=torch.int64, device=device)
        torch.distributed.broadcast(t, 0)
        n_blocks = int(t.item())
    for k in range(1, n_blocks):
        blocks.append(timed_block(args.warmup + k * args.steps))
    log(f'{len(blocks)} timed blocks done')
    elapsed = float(np.median(blocks))
    ms_per_step = 1e3 * elapsed / args.steps
    refined_timed = list(refined[args.warmup:]) if S == 1 and not args.overlap else []
    refine_state = eng.refine_info()
    counters1 = eng.refine_counters()
    # which mode the timed local-energy calls ran in (dqmc_refine_counters): mixed float32 + float64 twin, or whole batch in float64
    refine_modes = {k: counters1[k] - counters0[k] for k in counters1}
    refine_modes['mixed_calls'] = refine_modes['calls'] - refine_modes['direct_f64_calls']
    value = S * B * world / (elapsed / args.steps)        # every state's walkers get a local energy per step

    # ---- the same loop with the float64 refinement switched off (secondary figure; plain float32 arithmetic) ----
    ms_refine_off = None
    if not args.emulated and args.refine < 0 and args.dtype == 'f32' and S == 1 and not args.overlap:
        eng.set_option('refine', 0)
        timed_block(10_000)
        off = [timed_block(10_000 + (k + 1) * args.steps) for k in range(max(3, min(len(blocks), 10)))]
        ms_refine_off = 1e3 * float(np.median(off)) / args.steps
        eng.set_option('refine', 1)
    eloc_only, roofline = None, None
    if not args.emulated:       # (the emulated test harness only exercises launch / shard / reduce)
        # ---- pure E_loc throughput (n_sub = 0), not the headline ----
        r = state['r'] if S == 1 else state[0]['r']
        for _ in range(2):
            loc_ene(0, params, r)
        sync()
        t0 = time.perf_counter()
        n_rep = max(5, args.steps)
        for k in range(n_rep):
            loc_ene(k, params, r)
        sync()
        eloc_only = B * world / ((time.perf_counter() - t0) / n_rep)

        # ---- roofline of the dominant kernel: HIP events around every launch, same workload ----
        eng.timing(True)
        eng.timing_reset()
        for s in range(3):
            state, stats_t = vmc_step(10_000_000 + s, state)
            stats = stats_t or stats
        sync()
        rep = eng.timing_report()
        eng.timing(False)
        names = {'linear': 'k_linear / k_linear_bf (forward-Laplacian linear layer: v_mfma_f32_16x16x4_f32, or float32 products as nine '
                           'v_mfma_f32_16x16x32_bf16 on three-piece bf16 splits where option linear_bf selects them)',
                 'fused_psi': 'k_fused2_value (LDS-resident psi evaluation; float32 layers as v_mfma_f32_16x16x32_bf16 x6 on '
                              'three-piece bf16 splits, shallow layers v_mfma_f32_16x16x4_f32)',
                 'fused_substep': 'k_fused2_value (one launch per Metropolis sub-step: propose + LDS-resident psi + determinants + '
                                  'accept; float32 layers as v_mfma_f32_16x16x32_bf16 x6 on three-piece bf16 splits of the operands, '
                                  'shallow layers v_mfma_f32_16x16x4_f32)'}
        spec_name = eng.substep_kernel() if hasattr(eng, 'substep_kernel') else ''
        if spec_name:
            names['fused_substep'] = (f'{spec_name} (plan-specialised sub-step kernel, deepqmc_amd/csrc/gen: one wave per tile of 4 walkers, activations '
                                      'in registers, weights streamed through an LDS ring; float32 layers as v_mfma_f32_16x16x32_bf16 x6 on '
                                      'three-piece bf16 splits)')
        cands = {k: rep[k] for k in names if k in rep and rep[k]['ms'] > 0}
        dom = max(cands, key=lambda k: cands[k]['ms']) if cands else 'linear'
        lin = rep.get(dom, {'ms': 0.0, 'launches': 0, 'flops': 0.0})
        total_ms = sum(v['ms'] for v in rep.values()) or 1.0       # (float32 context + its float64 twin)
        achieved = lin['flops'] / (lin['ms'] * 1e-3) / 1e12 if lin['ms'] > 0 else 0.0
        traffic, traffic_src = committed_traffic({'linear': 'k_linear', 'fused_psi': 'k_fused2_value', 'fused_substep': (spec_name.split('<')[0] if spec_name else 'k_fused2_value')}[dom],
                                                 f'{args.molecule}/{args.ansatz}/{B}/{args.dtype}')
        roofline = {
            'bound': 'mfma', 'kernel': names[dom],
            'per_kernel_tflops': {k: v['flops'] / (v['ms'] * 1e-3) / 1e12 for k, v in cands.items()},
            'achieved': achieved, 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / F32_MFMA_PEAK_TFLOPS,
            # `achieved` / `frac` price ALGORITHMIC flops (the dense T = 3N + 2 lanes of SURVEY.md section 8d).  What the launches
            # actually multiply (8 compact lanes on edge rows, per-walker pieces once per walker: dqmc_timing_get_executed) is
            # `executed`; executed / peak is matrix-pipe utilisation, not a roofline fraction -- for N2 / FermiNet the two
            # differ by ~2x, and the events of launches that overlap on four streams are summed (an upper bound on both)
            'executed': {k: v.get('flops_executed', v['flops']) / (v['ms'] * 1e-3) / 1e12 for k, v in cands.items()},
            'executed_frac_of_peak': (lin.get('flops_executed', lin['flops']) / (lin['ms'] * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS) if lin['ms'] > 0 else 0.0,
            # (counters cannot be collected inside the timed process: the figure is read from the committed rocprofv3 --pmc passes
            # of the same workload, not measured by this run -- hence the key's name; `traffic` itself stays null)
            'traffic': None, 'traffic_from_profile': traffic, 'traffic_unit': 'HBM bytes per launch', 'traffic_source': traffic_src,
            'avg_launch_us': 1e3 * lin['ms'] / max(lin['launches'], 1), 'launches_per_step': lin['launches'] / 3,
            'share_of_kernel_time': lin['ms'] / total_ms,
            'kernel_ms_per_step': {k: v['ms'] / 3 for k, v in rep.items() if not k.startswith('f64.')},
        }
        # float64 refinement twin (records "f64.<name>"): where most of the kernel time of a step is float64 -- the attention
        # ansatzes of BASELINE configs[3..4] -- the dominant float64 kernel is priced against the float64 MFMA peak and becomes
        # the primary `roofline`; the float32 figure stays beside it
        f64 = {k[4:]: v for k, v in rep.items() if k.startswith('f64.') and v['ms'] > 0}
        f64_ms = sum(v['ms'] for v in f64.values())
        roofline['f64_share_of_kernel_time'] = f64_ms / total_ms
        f64_c = {k: v for k, v in f64.items() if v['flops'] > 0}
        if f64_c:
            d64 = max(f64_c, key=lambda k: f64_c[k]['ms'])
            a64 = f64_c[d64]['flops'] / (f64_c[d64]['ms'] * 1e-3) / 1e12
            roofline_f64 = {
                'bound': 'mfma', 'kernel': {'linear': 'k_linear<double> (forward-Laplacian linear layer, v_mfma_f64_16x16x4_f64; split-group tiles '
                                                      'for 96 / 128 lanes)', 'attention': 'k_attention_mfma<double> (v_mfma_f64_16x16x4_f64)'}.get(d64, d64),
                'per_kernel_tflops': {k: v['flops'] / (v['ms'] * 1e-3) / 1e12 for k, v in f64_c.items()},
                'achieved': a64, 'peak': F64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': a64 / F64_MFMA_PEAK_TFLOPS, 'traffic': None,
                'executed': {k: v.get('flops_executed', v['flops']) / (v['ms'] * 1e-3) / 1e12 for k, v in f64_c.items()},
                'avg_launch_us': 1e3 * f64_c[d64]['ms'] / max(f64_c[d64]['launches'], 1), 'launches_per_step': f64_c[d64]['launches'] / 3,
                'share_of_kernel_time': f64_c[d64]['ms'] / total_ms, 'f64_share_of_kernel_time': f64_ms / total_ms,
                'kernel_ms_per_step': {k: v['ms'] / 3 for k, v in f64.items()}}
            if f64_ms > 0.5 * total_ms:
                roofline_f64['float32_kernels'] = roofline
                roofline = roofline_f64
            else:
                roofline['float64_twin'] = roofline_f64
        if dom.startswith('fused') and args.dtype == 'f32' and 'matrix_pipe' not in roofline and roofline.get('peak') == F32_MFMA_PEAK_TFLOPS:
            # `achieved` counts ALGORITHMIC float32 flops and `peak` is the float32 MFMA peak (the arithmetic the path
            # delivers).  The fused kernel executes most of them on the bf16 pipe at six bf16 MFMA flops per float32 flop:
            roofline['matrix_pipe'] = {
                'note': 'float32 products as six bf16 MFMAs per 16x16x32 block (operands split into three bf16 pieces); '
                        'executed bf16-pipe flops <= 6 x algorithmic',
                'bf16_dense_peak_tflops': BF16_MFMA_PEAK_TFLOPS,
                'executed_tflops_upper': 6 * achieved, 'pipe_frac_upper': 6 * achieved / BF16_MFMA_PEAK_TFLOPS}

    if rank == 0:
        out = {
            'metric': 'walker*local-energy evals/sec (VMC step: n_sub Metropolis sub-steps + E_loc + energy reduction)',
            'value': value, 'unit': 'walker*E_loc evals/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'timed_blocks': len(blocks), 'timed_seconds': float(np.sum(blocks)),
            'ms_per_step_min': 1e3 * float(np.min(blocks)) / args.steps, 'ms_per_step_max': 1e3 * float(np.max(blocks)) / args.steps,
            'n_ranks_seen': n_ranks_seen, 'ms_per_step_refine_off': ms_refine_off,
            'dtype': args.dtype, 'data': 'synthetic walkers, random-init weights',
            'config': {'workload': f'{args.molecule} ({hamil.n_elec} e-), {args.ansatz} ansatz, '
                                   + (f'{S} electronic states x ' if S > 1 else '') + f'{B} walkers/GPU, '
                                   f'{args.n_sub} Metropolis sub-steps + local energy + RCCL energy stats'
                                   + (f' + {S}x{S} psi-ratio matrix + overlap penalty' if S > 1 else ''),
                       'walkers_per_gpu': B, 'n_sub': args.n_sub, 'states': S, 'parallelism': f'walker-dp{world}',
                       'reduction': ('in-library: dqmc_energy_stats_allgather (one ncclAllGather of 7 doubles per rank on the '
                                     'context stream)' if comm is not None else 'host: torch.distributed.all_gather of the 7-double record'),
                       'refine': {-1: 'library default (1: float64 re-evaluation of flagged walkers, self-calibrated threshold)', 0: 'off',
                                  1: 'flagged walkers', 2: 'whole E_loc pass in float64'}[args.refine],
                       # what the library actually did during the timed steps (dqmc_last_refined / dqmc_refine_info)
                       'refine_engaged': {**refine_state, 'timed_calls_by_mode': refine_modes,
                                          'walkers_refined_per_step_mean': float(np.mean(refined_timed)) if refined_timed else None,
                                          'walkers_refined_per_step_max': int(np.max(refined_timed)) if refined_timed else None,
                                          'fraction_refined': float(np.mean(refined_timed)) / B if refined_timed else None}},
            'eloc_only_evals_per_s': eloc_only,
            'energy': stats,
            'flops_per_eloc': (3 * hamil.n_elec + 2) * eng.program.flops_per_walker,
            'roofline': roofline,
        }
        if args.emulated:
            out['data'] = 'EMULATED on the CPU (test harness): not a measurement'
        if args.ecp:
            out['data'] += ', synthetic ECP coefficients'
            out['config']['workload'] += ' + Gaussian-type ECP (12-point quadrature)'
        log('GPU sections done; CPU baseline')
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.molecule, args.ansatz, args.n_sub, args.dtype)
        result_out.write(json.dumps(out) + '\n')
        result_out.flush()
    if comm is not None:
        comm.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
