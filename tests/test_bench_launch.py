"""bench.py's own multi-rank launch path on the CPU: `python bench.py --gpus 2` must re-execute itself under
torch.distributed.run (one process per rank), shard ONE global walker batch with the reference's
`electron_batch_size // device_count` split, reduce the energy statistics with one all-gather, and print one JSON
line from rank 0.  The kernels run in the SIMT emulator and the collective over gloo (`--emulated`, a test harness);
on the GPU box the same code path runs RCCL."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--emulated', '--walkers', '4', '--steps', '1', '--warmup', '0',
           '--n-sub', '1', '--repeats', '1', '--no-cpu-baseline', '--equilibrate', '0', *extra]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)


def test_self_launch_two_ranks():
    from simt_util import emu_lib
    emu_lib()                                    # build the emulated library once, before two ranks race for it
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p2 = run_bench('--gpus', '2', env=env)
    assert p2.returncode == 0, p2.stderr[-2000:]
    lines = [l for l in p2.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                       # rank 0 only
    out2 = json.loads(lines[0])
    assert out2['n_gpus'] == 2 and out2['n_ranks_seen'] == 2 and out2['scaling'] == 'weak'
    # the two shards are the halves of the single-process global batch of 8 walkers: after one sub-step with
    # per-rank noise the energies differ, but the walker COUNT behind the merged statistics must be the global one
    p1 = run_bench('--gpus', '1', '--walkers', '8', env=env)
    assert p1.returncode == 0, p1.stderr[-2000:]
    out1 = json.loads([l for l in p1.stdout.splitlines() if l.startswith('{')][0])
    assert out1['n_ranks_seen'] == 1
    for k in ('local_energy/mean', 'local_energy/std', 'local_energy/min', 'local_energy/max'):
        assert k in out2['energy'] and k in out1['energy']


def test_too_many_gpus_is_a_clear_error():
    """--gpus N on a node with fewer devices: a one-line message, not an assert / traceback."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert 'only' in p.stderr and 'GPU device(s) are visible' in p.stderr and 'Traceback' not in p.stderr
    env['WORLD_SIZE'] = '3'
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and 'WORLD_SIZE=3' in p.stderr and 'Traceback' not in p.stderr
