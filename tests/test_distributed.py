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


def _run_ranks(target, world, nresults, attempts=2, wait_s=120):
    """Start `world` ranks of `target(rank, world, port, queue)` and collect `nresults` queue items.  The rendezvous port
    is picked by binding port 0 and closing it again, which another process can win in between (seen once: a rank never
    joined and the collection sat out its timeout): a failed start -- a rank that exits early, or nothing in the queue
    in time -- is retried once on a fresh port before it counts as a failure."""
    import queue as queue_mod

    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    last = None
    for _ in range(attempts):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = []
        try:
            for _i in range(nresults):
                got.append(q.get(timeout=wait_s))
        except queue_mod.Empty:
            last = "no result within %d s (exit codes %s)" % (wait_s, [p.exitcode for p in procs])
            import sys

            print("test_distributed: %s: %s; retrying on a fresh port" % (getattr(target, "__name__", target), last), file=sys.stderr)
            for p in procs:
                if p.is_alive():
                    p.kill()  # (exactly the processes started here)
                p.join(timeout=30)
            continue
        codes = []
        for p in procs:
            p.join(timeout=60)
            codes.append(p.exitcode)
        if all(c == 0 for c in codes):
            return got
        last = "exit codes %s" % codes
    raise AssertionError("ranks failed twice: " + str(last))


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

        parallel._PIECE_BYTES = 1000  # blocks of 4 to 8 KB travel as several point-to-point pieces
        vis = parallel.simulate_sharded(compute_block, 6, 4, gather_to=0)
        if rank == 0:
            q.put(vis)
    finally:
        dist.destroy_process_group()


def _shm_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert parallel._all_ranks_on_one_host()
        blocks = parallel.shard_blocks_weighted(world, np.linspace(1e8, 2e8, 6), 4)

        def one_call(scale):
            res = parallel._SharedResult((6, 4, 2, 2, 5), np.complex128, owner=0)
            assert not os.path.exists(res.path) and res.array.shape == (6, 4, 2, 2, 5)  # mapped by all, name gone
            for tsl, fsl in blocks[rank]:  # every rank fills its own slice of the ONE array
                f, t = np.meshgrid(np.arange(6)[fsl], np.arange(4)[tsl], indexing="ij")
                res.array[fsl, tsl] = scale * (100 * f + t)[:, :, None, None, None] * (1 + 1j)
            return res.path, res.seal()

        p1, out1 = one_call(1.0)
        first = np.array(out1) if rank == 0 else None
        p2, out2 = one_call(2.0)   # the caller still holds out1: a NEW segment
        assert p2 != p1
        if rank == 0:
            assert np.array_equal(out1, first)  # untouched by the second call
        del out1, out2
        p3, out3 = one_call(3.0)   # the previous result was dropped: ITS segment is delivered into again (warm pages)
        assert p3 == p2
        if rank == 0:
            q.put((first, np.array(out3), os.path.exists(p1) or os.path.exists(p2)))
        else:
            assert out3 is None
    finally:
        dist.destroy_process_group()


def _flux_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for pol_sky in (False, True):
            cfg = synth.make_config("C1", nsrc=50, nfreq=10, ntimes=2)
            _, _, fl = synth.catalog(50, cfg["freqs"], 3, polarized_sky=pol_sky)
            ranges = [(0, 10), (4, 9), None][:world]
            args = (cfg["ra"], cfg["dec"], fl) if rank == 0 else (None, None, None)
            cat = parallel.broadcast_catalog_device(*args, True, 2, torch.device("cpu"), freq_ranges=lambda n: ranges)
            full = parallel.broadcast_catalog_device(*args, True, 2, torch.device("cpu"))
            assert torch.equal(cat.eq, full.eq) and cat.flux.shape == full.flux.shape
            fr = ranges[rank]
            if fr is not None:
                assert torch.equal(cat.flux[:, fr[0]:fr[1]], full.flux[:, fr[0]:fr[1]])
            if rank != 0:  # nothing else arrived
                mask = torch.ones(10, dtype=torch.bool)
                if fr is not None:
                    mask[fr[0]:fr[1]] = False
                assert not cat.flux[:, mask].any()
        q.put(rank)
    finally:
        dist.destroy_process_group()


def test_flux_travels_as_the_channels_each_rank_needs():
    """SURVEY 8e / VERDICT r3 missing #5: the catalog's unit vectors are broadcast, the flux goes to every rank as one
    point-to-point piece holding only the channels its blocks cover (real and 2 x 2 complex coherencies): those columns
    equal the broadcast's, the rest of the rank's full-width tensor stays zero."""
    got = sorted(_run_ranks(_flux_worker, 3, 3))
    assert got == [0, 1, 2]


def test_shared_result_is_filled_by_every_rank_and_leaves_no_file():
    """N > 1, one node: the result of a sharded run is ONE array in shared memory; every rank delivers its blocks into
    its own slice, rank 0 ends up with the assembled array, the /dev/shm name is gone as soon as every rank has mapped
    it, and a later call of the same shape reuses the segment (warm pages) once the earlier result has been dropped --
    never while the caller still holds it."""
    ((vis, vis3, still_there),) = _run_ranks(_shm_worker, 2, 1)
    assert not still_there
    f, t = np.meshgrid(np.arange(6), np.arange(4), indexing="ij")
    want = np.broadcast_to((100 * f + t)[:, :, None, None, None] * (1 + 1j), vis.shape)
    np.testing.assert_array_equal(vis, want)
    np.testing.assert_array_equal(vis3, 3.0 * want)


def test_two_rank_sharding_matches_single_process():
    (vis,) = _run_ranks(_worker, 2, 1)
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


@pytest.mark.parametrize("world,nf,nt", [(1, 128, 20), (2, 128, 20), (3, 128, 20), (4, 128, 20), (8, 128, 20),
                                         (8, 256, 60), (8, 4, 3), (3, 7, 5), (7, 64, 2), (8, 3, 1), (5, 1, 1)])
def test_weighted_shards_cover_plane_once_and_balance(world, nf, nt):
    """GPU runs shard by cost (nu^2 grid size), time-major: one block per rank, the plane covered once, and
    no rank more than a few per cent over the mean once the job has a few dozen slices per rank."""
    freqs = np.linspace(100e6, 200e6, nf)
    blocks = parallel.shard_blocks_weighted(world, freqs, nt)
    assert len(blocks) == world and all(len(b) <= 1 for b in blocks)
    cover = np.zeros((nt, nf), int)
    cost = parallel.slice_cost(freqs)
    loads = []
    for mine in blocks:
        for tsl, fsl in mine:
            cover[tsl, fsl] += 1
            loads.append((tsl.stop - tsl.start) * cost[fsl].sum())
    assert (cover == 1).all()
    if nt * nf >= 64 * world:
        assert max(loads) / (nt * cost.sum() / world) < 1.06
    # by count the top half of the band would cost ~1.9x the bottom half
    if (world, nf, nt) == (2, 128, 20):
        assert blocks[0][0][0] == slice(0, 10) and blocks[0][0][1] == slice(0, 128)  # time-major split
    if (world, nf, nt) == (8, 128, 20):
        f_cut = blocks[0][0][1].stop
        assert 75 <= f_cut <= 85  # 4 time parts x 2 frequency parts; the cut sits above the middle channel
