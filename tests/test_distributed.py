"""Walker data parallelism on CPU: 2 ranks over gloo, each with its own shard; ONE all-gather of
the 56-byte energy record + Chan merge must equal the single-process statistics
(reference observable.py:474-479 / parallel.py:175-225)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepqmc_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, e_all, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from deepqmc_amd.engine import Engine
    from deepqmc_amd.hamil import MolecularHamiltonian
    from deepqmc_amd.molecule import Molecule
    from deepqmc_amd.params import init_params
    from deepqmc_amd.spec import paulinet
    from simt_util import emu_lib
    h = MolecularHamiltonian(mol=Molecule.from_name('LiH'))
    eng = Engine(paulinet(), h, init_params(paulinet(), 2, 2, 2), dtype=torch.float64, device='cpu', lib=emu_lib())
    lo, hi = parallel.shard_bounds(len(e_all), rank, world)
    stats = parallel.energy_stats(eng, torch.as_tensor(e_all[lo:hi]).contiguous())
    out[rank] = stats
    dist.destroy_process_group()


def test_two_rank_energy_stats():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(0)
    e_all = rng.standard_normal(64) * 3 - 8
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), e_all, out), nprocs=2, join=True)
    for rank in (0, 1):
        st = out[rank]
        np.testing.assert_allclose(st['local_energy/mean'], e_all.mean(), rtol=1e-12)
        np.testing.assert_allclose(st['local_energy/std'], e_all.std(), rtol=1e-12)
        assert st['local_energy/min'] == e_all.min() and st['local_energy/max'] == e_all.max()


def test_shard_bounds():
    assert parallel.shard_bounds(4096 * 8, 3, 8) == (3 * 4096, 4 * 4096)
    import pytest
    with pytest.raises(ValueError):
        parallel.shard_bounds(10, 0, 3)


def _overlap_worker(rank, world, port, ratio, weight, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from deepqmc_amd import loss
    B = ratio.shape[-1]
    lo, hi = parallel.shard_bounds(B, rank, world)
    ov, stats = loss.compute_mean_overlap(torch.as_tensor(ratio[..., lo:hi]).contiguous(), torch.as_tensor(weight[..., lo:hi]).contiguous())
    out[rank] = (float(ov), stats['overlap/pairwise/mean'].numpy())
    dist.destroy_process_group()


def test_two_rank_mean_overlap():
    """compute_mean_overlap (loss/overlap.py:124-149: all_device_mean over every rank's walkers) on two gloo ranks
    equals the single-process value."""
    from deepqmc_amd import loss
    rng = np.random.default_rng(1)
    ratio = rng.standard_normal((1, 3, 3, 16))
    ratio[:, np.arange(3), np.arange(3)] = 1.0
    weight = rng.random((1, 3, 16)) + 0.5
    ref_ov, ref_stats = loss.compute_mean_overlap(torch.as_tensor(ratio), torch.as_tensor(weight))
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_overlap_worker, args=(2, _free_port(), ratio, weight, out), nprocs=2, join=True)
    for rank in (0, 1):
        np.testing.assert_allclose(out[rank][0], float(ref_ov), rtol=1e-12)
        np.testing.assert_allclose(out[rank][1], ref_stats['overlap/pairwise/mean'].numpy(), rtol=1e-12, atol=1e-15)
