"""world_size-2 gloo tests of the sharding path (CPU): catalog broadcast from rank 0, disjoint
(time, freq) blocks per rank, gather + assembly == single-process result.  The block compute is
injected (the CPU oracle here; on GPUs it is GPUSimulationEngine.simulate with time_idx/freq_idx)."""

import os
import socket

import numpy as np
import pytest

from fftvis_amd import parallel, synth
from tests.helpers import oracle_simulate


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = synth.make_config("C1", nsrc=60, nfreq=6, ntimes=4)
        cfg["polarized"] = True
        src = (cfg["ra"], cfg["dec"], cfg["fluxes"]) if rank == 0 else (None, None, None)
        ra, dec, fl = parallel.broadcast_catalog(*src, src=0)
        np.testing.assert_array_equal(ra, cfg["ra"])
        np.testing.assert_array_equal(fl, cfg["fluxes"])
        cfg.update(ra=ra, dec=dec, fluxes=fl)

        def compute_block(tsl, fsl):
            sub = dict(cfg, freqs=cfg["freqs"][fsl], times=cfg["times"][tsl], fluxes=fl[:, fsl])
            return oracle_simulate(sub)

        vis = parallel.simulate_sharded(compute_block, 6, 4, gather_to=0)
        if rank == 0:
            q.put(vis)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    vis = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = synth.make_config("C1", nsrc=60, nfreq=6, ntimes=4)
    cfg["polarized"] = True
    ref = oracle_simulate(cfg)
    assert vis.shape == ref.shape == (6, 4, 2, 2, 21)
    np.testing.assert_allclose(vis, ref, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("world,nf,nt", [(2, 64, 10), (4, 64, 10), (8, 128, 20), (8, 256, 60), (8, 4, 3), (3, 7, 5)])
def test_shard_blocks_cover_plane_once(world, nf, nt):
    blocks = parallel.shard_blocks(world, nf, nt)
    assert len(blocks) == world
    cover = np.zeros((nt, nf), int)
    for mine in blocks:
        for b in mine:
            cover[b[0], b[1]] += 1
    assert (cover == 1).all()
