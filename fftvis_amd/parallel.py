"""Multi-GPU sharding of a simulation: one process per GPU, independent (time, frequency) blocks.

The reference shards the same way over Ray workers (src/fftvis/cpu/cpu_simulate.py:711-713,
800-847 with ``get_task_chunks``, core/utils.py:122-187): blocks are disjoint, so there is no
data-path collective.  Communication here is (1) one broadcast of the source catalog from rank 0
(``torch.distributed`` -- RCCL over xGMI with the "nccl" backend, gloo on CPU) and (2) an optional
gather of the finished blocks to rank 0.
"""

from __future__ import annotations

import numpy as np

from .core.utils import get_task_chunks


def shard_blocks(world: int, nfreqs: int, ntimes: int):
    """Per rank, the list of (time_slice, freq_slice) blocks it owns; together they cover the
    (ntimes, nfreqs) plane exactly once.

    Follows the reference's chunker.  When it decides the job is too small to split
    (ntasks < 2 * world) rank 0 takes everything, like the reference falling back to one process
    (core/utils.py:160-162).  For awkward shapes the chunker can emit more chunks than workers
    (e.g. 3 workers, 7 channels, 5 times -> 7 chunks); the reference's ``zip`` over workers
    (cpu_simulate.py:800) would silently drop the surplus, here they are dealt round-robin so
    nothing is left uncomputed."""
    _, fchunks, tchunks, _, _ = get_task_chunks(world, nfreqs, ntimes)
    blocks = [[] for _ in range(world)]
    for i, (fc, tc) in enumerate(zip(fchunks, tchunks)):
        t0, t1, _ = tc.indices(ntimes)
        f0, f1, _ = fc.indices(nfreqs)
        if t1 > t0 and f1 > f0:
            blocks[i % world].append((slice(t0, t1), slice(f0, f1)))
    return blocks


def broadcast_catalog(ra, dec, fluxes, src: int = 0, device=None):
    """Broadcast (ra, dec, fluxes) from ``src`` to every rank; non-source ranks pass None.
    Returns numpy arrays (CPU group) -- for device-resident hand-over use bench.py's pattern of
    broadcasting device tensors and ``SimHandle.set_sources_device``."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        fl = np.ascontiguousarray(fluxes)
        meta = [(len(ra), fl.shape, str(fl.dtype))]
    dist.broadcast_object_list(meta, src=src)
    nsrc, fshape, fdt = meta[0]
    dev = device if device is not None else "cpu"
    out = []
    for arr, shape, dt in ((ra, (nsrc,), "float64"), (dec, (nsrc,), "float64"), (fluxes, fshape, fdt)):
        if rank == src:
            t = torch.from_numpy(np.ascontiguousarray(arr, dtype=dt)).to(dev)
        else:
            t = torch.empty(shape, dtype=getattr(torch, dt), device=dev)
        dist.broadcast(t, src=src)
        out.append(t.cpu().numpy())
    return tuple(out)


def simulate_sharded(compute_block, nfreqs: int, ntimes: int, gather_to: int | None = 0):
    """Run ``compute_block(time_slice, freq_slice) -> ndarray`` (final layout, leading axes
    (nf_here, nt_here)) on each of this rank's blocks; optionally gather and assemble on
    ``gather_to``.

    Returns the assembled (nfreqs, ntimes, ...) array on ``gather_to`` (or this rank's
    [(block, array), ...] list when ``gather_to`` is None), None elsewhere."""
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    blocks = shard_blocks(world, nfreqs, ntimes)
    parts = [(blk, compute_block(*blk)) for blk in blocks[rank]]
    if gather_to is None:
        return parts
    gathered = [None] * world if rank == gather_to else None
    dist.gather_object(parts, gathered, dst=gather_to)
    if rank != gather_to:
        return None
    first = next(p for plist in gathered for _, p in plist)
    vis = np.zeros((nfreqs, ntimes) + first.shape[2:], dtype=first.dtype)
    for plist in gathered:
        for (tsl, fsl), p in plist:
            vis[fsl, tsl] = p  # reference: vis[tc][..., fc] = future (cpu_simulate.py:846-847)
    return vis
