"""Multi-GPU sharding of a simulation: one process per GPU, independent (time, frequency) blocks.

The reference shards the same way over Ray workers (src/fftvis/cpu/cpu_simulate.py:711-713,
800-847 with ``get_task_chunks``, core/utils.py:122-187): blocks are disjoint, so there is no
data-path collective.  Communication here is (1) one broadcast of the source catalog from rank 0
(``torch.distributed`` -- RCCL over xGMI with the "nccl" backend, gloo on CPU), received straight
into device memory and handed to the engine as device pointers, and (2) an optional gather of the
finished blocks to rank 0.
"""

from __future__ import annotations

import numpy as np

from .core.utils import get_task_chunks


def shard_blocks(world: int, nfreqs: int, ntimes: int):
    """Per rank, the list of (time_slice, freq_slice) blocks it owns; together they cover the
    (ntimes, nfreqs) plane exactly once.

    Follows the reference's chunker (equal COUNTS of slices).  When it decides the job is too small to
    split (ntasks < 2 * world) rank 0 takes everything, like the reference falling back to one
    process (core/utils.py:160-162).  For awkward shapes the chunker can emit more chunks than workers
    (e.g. 3 workers, 7 channels, 5 times -> 7 chunks); the reference's ``zip`` over workers
    (cpu_simulate.py:800) would silently drop the surplus, here they are dealt round-robin so
    nothing is left uncomputed.  The GPU engine's cost per slice grows like nu^2, so GPU runs use
    ``shard_blocks_weighted`` instead."""
    _, fchunks, tchunks, _, _ = get_task_chunks(world, nfreqs, ntimes)
    blocks = [[] for _ in range(world)]
    for i, (fc, tc) in enumerate(zip(fchunks, tchunks)):
        t0, t1, _ = tc.indices(ntimes)
        f0, f1, _ = fc.indices(nfreqs)
        if t1 > t0 and f1 > f0:
            blocks[i % world].append((slice(t0, t1), slice(f0, f1)))
    return blocks


def slice_cost(freqs, fixed: float = 0.1, power: float = 2.0) -> np.ndarray:
    """Relative cost of one (time, channel) slice on the GPU engine: the fine grid has
    ~(2 sigma b_max nu / c)^2 cells, and spread output, FFT passes and gather input all scale with
    it (HERA's 100-200 MHz band: the top channel costs 4x the bottom one); ``fixed`` (in units of
    the top channel's grid cost) stands for the per-slice work that does not.  The exponent is
    measured (``bench.py --as-rank R --of-ranks 8``: the two frequency parts of an 8-rank job on the
    same GPU).  Round 2's passes wanted nu^2.4 (FFT lengths step, the longer rows fold more residues);
    since the column plan and the source disc of round 3 the y-pass and the stores no longer follow
    the FFT length, and with 2.4 the parts of C3 (cut at channel 83) ran 120.2 and 110.1 ms, those of
    C4 (cut at 164) 862 and 787 ms: both 1.09 : 1 -- what plain nu^2 with the same ``fixed`` predicts."""
    f = np.abs(np.asarray(freqs, dtype=float))
    return (f / f.max()) ** power + fixed


def _cut_by_weight(w: np.ndarray, parts: int):
    """Cut range(len(w)) into ``parts`` consecutive non-empty runs of (nearly) equal total weight."""
    n = len(w)
    parts = max(1, min(parts, n))
    c = np.concatenate([[0.0], np.cumsum(w)])
    edges = [0]
    for k in range(1, parts):
        target = c[-1] * k / parts
        e = int(np.searchsorted(c, target))
        if e > 0 and abs(c[e - 1] - target) <= abs(c[min(e, n)] - target):
            e -= 1
        e = max(edges[-1] + 1, min(e, n - (parts - k)))  # every run keeps at least one element
        edges.append(e)
    edges.append(n)
    return [(edges[i], edges[i + 1]) for i in range(parts)]


def shard_blocks_weighted(world: int, freqs, ntimes: int, nsrc: int | None = None):
    """One (time_slice, freq_slice) block per rank (empty list for surplus ranks), TIME-MAJOR and
    balanced by ``slice_cost``: the ranks form an (a x b) grid, a time parts x b frequency parts
    with a b <= world; time parts have (nearly) equal length -- rotation, horizon cut and az/za are
    per-time work that then amortises over a rank's whole frequency range -- and
    the frequency cuts equalise the summed nu^2 cost, not the channel count.  Among the factorisations
    the one with the smallest largest block wins; ties go to more time parts.  ``nsrc`` (catalog size)
    raises the per-slice share that does not grow with frequency -- beam, coherency and the spread's walk
    over the sources: measured on HERA-350, 8-rank blocks, 0.12 of the top channel's grid cost at 1e5
    sources (C3: blocks within 2 %) and 0.17 at 1e6 (C4: 6 % apart with 0.1)."""
    freqs = np.atleast_1d(np.asarray(freqs, dtype=float))
    nf = len(freqs)
    w = slice_cost(freqs, fixed=0.1 if nsrc is None else 0.115 + 5.5e-8 * float(nsrc))
    best = None
    for a in range(1, min(world, max(ntimes, 1)) + 1):
        b = min(world // a, nf)
        if b < 1:
            continue
        tparts = _cut_by_weight(np.ones(max(ntimes, 1)), a)
        fparts = _cut_by_weight(w, b)
        worst = max(t1 - t0 for t0, t1 in tparts) * max(w[f0:f1].sum() for f0, f1 in fparts)
        key = (worst, -a)
        if best is None or key < best[0]:
            best = (key, tparts, fparts)
    _, tparts, fparts = best
    blocks = [[] for _ in range(world)]
    r = 0
    for t0, t1 in tparts:
        for f0, f1 in fparts:
            if ntimes > 0 and t1 > t0 and f1 > f0:
                blocks[r].append((slice(t0, t1), slice(f0, f1)))
            r += 1
    return blocks


def broadcast_catalog(ra, dec, fluxes, src: int = 0, device=None):
    """Broadcast (ra, dec, fluxes) from ``src`` to every rank; non-source ranks pass None.
    Returns numpy arrays (host copies; for the device-resident hand-over see
    ``broadcast_catalog_device``)."""
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


class DeviceCatalog:
    """A source catalog resident on this rank's GPU in the engine's layout: ``eq`` (3, nsrc) equatorial
    unit vectors and ``flux`` (nsrc, nfreq) real / (nsrc, nfreq, 2, 2) complex coherency (already x0.5,
    reference cpu/utils.py:26-80), as torch tensors of the run's precision.  ``GPUSimulationEngine.simulate``
    takes it as ``catalog_device=`` and hands the pointers to ``fv_sim_set_sources(on_device=1)``."""

    def __init__(self, eq, flux, polarized_sky: bool):
        self.eq, self.flux, self.polarized_sky = eq, flux, bool(polarized_sky)
        self.nsrc, self.nfreq = int(eq.shape[1]), int(flux.shape[1])


def broadcast_catalog_device(ra, dec, fluxes, polarized: bool, precision: int, device, src: int = 0,
                             via_host: bool = False, freq_ranges=None) -> DeviceCatalog:
    """Rank ``src`` prepares the catalog (``prepare_source_catalog``, unit vectors), every rank receives
    it INTO DEVICE MEMORY with one broadcast per array -- RCCL over xGMI under the "nccl" backend; with a
    CPU group (gloo rehearsal on a one-GPU box, ``via_host=True``) through a host staging copy.  The only
    collective of a sharded run (24 B/source + 8 nf B/source, 2 GB at C4).

    ``freq_ranges``: ``callable(nsrc) -> [(f_lo, f_hi) | None per rank]`` -- the channels each rank's blocks cover.
    The flux then travels as ONE point-to-point piece per rank holding only its channels (batched isend / irecv: rank
    ``src`` feeds its xGMI links in parallel) instead of a broadcast of every channel to everybody: an 8-rank job cut
    into 4 time x 2 frequency parts moves half the bytes per rank.  Every rank still holds a full-width flux tensor
    (288 GB of HBM: the engine indexes channels by their catalog position); the columns it was not sent are zero and
    never read."""
    import torch
    import torch.distributed as dist

    from .core.coords import eq_unit_vectors
    from .core.utils import prepare_source_catalog

    rank = dist.get_rank()
    rdt = torch.float32 if precision == 1 else torch.float64
    cdt = torch.complex64 if precision == 1 else torch.complex128
    meta = [None]
    if rank == src:
        coh, pol_sky = prepare_source_catalog(np.asarray(fluxes), polarized)
        meta = [(int(len(ra)), tuple(coh.shape), bool(pol_sky))]
    dist.broadcast_object_list(meta, src=src)
    nsrc, fshape, pol_sky = meta[0]
    fdt = cdt if pol_sky else rdt
    if rank == src:
        rr = np.asarray(ra).astype(np.float32 if precision == 1 else np.float64).astype(float)
        dd = np.asarray(dec).astype(np.float32 if precision == 1 else np.float64).astype(float)
        eq = torch.from_numpy(eq_unit_vectors(rr, dd)).to(device, rdt)
        flux = torch.from_numpy(np.ascontiguousarray(coh)).to(device, fdt)
    else:
        eq = torch.empty((3, nsrc), dtype=rdt, device=device)
        flux = torch.empty(fshape, dtype=fdt, device=device)
    def bcast(t):
        tv = torch.view_as_real(t) if t.is_complex() else t  # RCCL has no complex type: the same bytes as reals
        if via_host:
            h = tv.cpu()
            dist.broadcast(h, src=src)
            tv.copy_(h)
        else:
            dist.broadcast(tv, src=src)

    bcast(eq)
    if freq_ranges is None:
        bcast(flux)
    else:
        world = dist.get_world_size()
        ranges = freq_ranges(nsrc)
        assert len(ranges) == world
        if rank != src:
            flux.zero_()
        ops, keep, mine = [], [], None
        for r, fr in enumerate(ranges):
            if r == src or fr is None or fr[1] <= fr[0]:
                continue
            if rank == src:  # the rank's channels, packed
                piece = flux[:, fr[0]:fr[1]].contiguous()
                pv = torch.view_as_real(piece) if piece.is_complex() else piece
                pv = pv.cpu() if via_host else pv
                keep.append(pv)
                ops.append(dist.P2POp(dist.isend, pv, r))
            elif rank == r:
                shape = (nsrc, fr[1] - fr[0]) + tuple(fshape[2:])
                piece = torch.empty(shape, dtype=fdt, device="cpu" if via_host else device)
                pv = torch.view_as_real(piece) if piece.is_complex() else piece
                mine = (fr, piece)
                ops.append(dist.P2POp(dist.irecv, pv, src))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if mine is not None:
            fr, piece = mine
            flux[:, fr[0]:fr[1]].copy_(piece)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    return DeviceCatalog(eq, flux, pol_sky)


def simulate_sharded(compute_block, nfreqs: int, ntimes: int, gather_to: int | None = 0, blocks=None):
    """Run ``compute_block(time_slice, freq_slice) -> ndarray`` (final layout, leading axes
    (nf_here, nt_here)) on each of this rank's blocks; optionally gather and assemble on
    ``gather_to``.  ``blocks`` = per-rank block lists (default: the reference's count-based chunker).

    Returns the assembled (nfreqs, ntimes, ...) array on ``gather_to`` (or this rank's
    [(block, array), ...] list when ``gather_to`` is None), None elsewhere."""
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    if blocks is None:
        blocks = shard_blocks(world, nfreqs, ntimes)
    parts = [(blk, compute_block(*blk)) for blk in blocks[rank]]
    if gather_to is None:
        return parts
    # Only the block index travels as a Python object; the visibilities go as flat real tensors, block by
    # block and in pieces of <= 256 MiB (C4: 7.5 GB per rank -- pickling that through gather_object would
    # hold three copies of it), point to point into rank ``gather_to``: RCCL send / recv of device staging
    # buffers under "nccl", host tensors under gloo.
    index = [(blk, p.shape, str(p.dtype)) for blk, p in parts]
    gathered = [None] * world if rank == gather_to else None
    dist.gather_object(index, gathered, dst=gather_to)
    if rank != gather_to:
        for _, p in parts:
            _send_array(p, gather_to)
        return None
    vis = None
    for src, plist in enumerate(gathered):
        for n, ((tsl, fsl), shape, dt) in enumerate(plist):
            p = parts[n][1] if src == rank else _recv_array(shape, np.dtype(dt), src)
            if vis is None:
                vis = np.zeros((nfreqs, ntimes) + tuple(shape[2:]), dtype=dt)
            vis[fsl, tsl] = p  # reference: vis[tc][..., fc] = future (cpu_simulate.py:846-847)
    return vis


_PIECE_BYTES = 256 * 2**20


def _flat_real(a: np.ndarray) -> np.ndarray:
    """The bytes of a contiguous real / complex array as a flat real array (RCCL has no complex type)."""
    a = np.ascontiguousarray(a)
    return a.view(a.real.dtype).reshape(-1)


def _staging_device():
    import torch
    import torch.distributed as dist

    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else None


def _send_array(a: np.ndarray, dst: int):
    import torch
    import torch.distributed as dist

    flat, dev = _flat_real(a), _staging_device()
    step = max(1, _PIECE_BYTES // flat.itemsize)
    for i in range(0, flat.size, step):
        t = torch.from_numpy(flat[i:i + step])
        dist.send(t.to(dev) if dev is not None else t, dst=dst)


def _recv_array(shape, dtype: np.dtype, src: int) -> np.ndarray:
    import torch
    import torch.distributed as dist

    out = np.empty(shape, dtype=dtype)
    flat, dev = _flat_real(out), _staging_device()
    step = max(1, _PIECE_BYTES // flat.itemsize)
    for i in range(0, flat.size, step):
        piece = flat[i:i + step]
        if dev is None:
            dist.recv(torch.from_numpy(piece), src=src)
        else:
            t = torch.empty(piece.size, dtype=torch.from_numpy(piece).dtype, device=dev)
            dist.recv(t, src=src)
            piece[:] = t.cpu().numpy()
    return out


def _all_ranks_on_one_host() -> bool:
    import socket

    import torch.distributed as dist

    names = [None] * dist.get_world_size()
    dist.all_gather_object(names, socket.gethostname())
    return len(set(names)) == 1


_WARM = {}  # (owner, shape, dtype) -> the segment this process mapped for the previous call of that shape


class _SharedResult:
    """The whole (nfreqs, ntimes, ...) result of a sharded run as ONE array in shared memory (a file under /dev/shm
    mapped by every rank of the node).  Rank ``owner`` creates it and tells the others its name; once every rank has
    mapped it the name is removed from the file system (nothing is left behind whatever happens next) and the
    mappings are the only thing that keeps the pages.

    Segments are REUSED: fresh tmpfs pages are allocated one by one under the file's lock -- 3.5 GB/s for sixteen
    threads together, measured; a 10-GB result would cost 3 s before the first byte arrives, however many ranks share
    the work -- so every rank keeps its mapping of the last result of each shape, and the owner hands the same segment
    out again once the array it returned last time has been collected (while the caller still holds it, a new segment
    is made).  A sequence of calls on one observation shape pays the page allocation once, like the engine's plans.
    ``array`` is this rank's view for delivering its blocks; ``seal()`` ends the call."""

    def __init__(self, shape, dtype, owner: int):
        import torch.distributed as dist

        self.owner, self.rank = owner, dist.get_rank()
        self.shape, self.dtype = tuple(int(x) for x in shape), np.dtype(dtype)
        self.handed = None  # owner: weak reference to the array the caller got
        self.path, self.map = None, None
        key = (owner, self.shape, self.dtype.str)
        warm = _WARM.get(key)
        box = [None]
        if self.rank == owner and warm is not None and (warm.handed is None or warm.handed() is None):
            box = [warm.path]
        dist.broadcast_object_list(box, src=owner)
        have = warm is not None and box[0] is not None and warm.path == box[0]
        flags = [None] * dist.get_world_size()
        dist.all_gather_object(flags, bool(have))
        if all(flags):  # every rank still maps that segment: deliver into it again
            self.path, self.map = warm.path, warm.map
        else:
            self._create()
        _WARM[key] = self
        self.array = np.frombuffer(self.map, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)

    def _create(self):
        import mmap
        import os
        import uuid

        import torch.distributed as dist

        nbytes = max(int(np.prod(self.shape)) * self.dtype.itemsize, 1)
        box = [None]
        fd = -1
        if self.rank == self.owner:
            box = [f"/dev/shm/fftvis_amd_{os.getpid()}_{uuid.uuid4().hex}"]
            fd = os.open(box[0], os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
            try:
                os.ftruncate(fd, nbytes)
                # the pages are allocated HERE, in one sweep by one process (measured: ~5 GB/s), not by the page faults
                # of every rank's pinning helper racing for the file's lock (1.8 GB/s for two ranks together)
                os.posix_fallocate(fd, 0, nbytes)
            except BaseException:
                os.close(fd)
                os.unlink(box[0])
                raise
        dist.broadcast_object_list(box, src=self.owner)
        self.path = box[0]
        try:
            if self.rank != self.owner:
                fd = os.open(self.path, os.O_RDWR)
            self.map = mmap.mmap(fd, nbytes, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
        finally:
            if fd >= 0:
                os.close(fd)
            dist.barrier()  # everybody has mapped it (or failed): the name can go
            if self.rank == self.owner:
                os.unlink(self.path)
            dist.barrier()  # ... and is gone when any rank returns (a fresh segment only: warm reuse skips all of this)

    def seal(self):
        """Every rank: all blocks are in.  Returns the result on the owner (an array over the shared pages; the
        segment is not reused while it lives), None elsewhere."""
        import weakref

        import torch.distributed as dist

        dist.barrier()
        self.array = None
        if self.rank != self.owner:
            return None
        out = np.frombuffer(self.map, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)
        self.handed = weakref.ref(out)
        return out


def release_shared_results():
    """Drop this process's mappings of the shared results of earlier sharded calls (the segments kept warm for the
    next call of the same shape).  An array the owner still holds keeps its pages; everything else is returned to the
    system.  Call it on every rank, or on none: the next call re-creates what is missing."""
    _WARM.clear()


def simulate_vis_sharded(device, gather_to: int | None = 0, via_host: bool = False, gather: str = "auto", **kw):
    """``simulate_vis`` across the ranks of the initialised process group, one GPU per rank.

    Rank 0's ``ra / dec / fluxes`` are broadcast into every rank's device memory
    (``broadcast_catalog_device``); every other argument must be the same on all ranks (they are small:
    array, beams, frequencies, times).  Each rank simulates its cost-balanced block
    (``shard_blocks_weighted``) through ``GPUSimulationEngine.simulate(time_idx=, freq_idx=)`` on
    ``device`` (its local GPU index); blocks are disjoint, so nothing is reduced -- rank ``gather_to``
    assembles them like the reference's ``vis[tc][..., fc] = future`` (cpu_simulate.py:843-847).

    ``gather``: how the blocks reach ``gather_to``.  ``"shm"`` (what ``"auto"`` picks when every rank runs on the
    same host -- one node, one process per GPU): the result is one array in shared memory and every rank's engine
    delivers its block STRAIGHT into its slice of it (``simulate(out=vis[fsl, tsl], out_shared=True)``): pinned in
    place run by run, filled from the rank's copy stream while its later time steps still compute -- the single-GPU
    host path, per rank, with no assembly step at all.  ``"p2p"`` (ranks on several hosts): blocks travel to
    ``gather_to`` as point-to-point tensors."""
    import torch

    from .core.coords import julian_dates
    from .wrapper import create_simulation_engine

    beam = kw.pop("beam")
    kw["beam_list"] = list(beam) if isinstance(beam, (list, tuple)) else [beam]
    polarized, precision = bool(kw.get("polarized", False)), int(kw.get("precision", 2))
    max_memory, min_chunks = kw.pop("max_memory", np.inf), kw.pop("min_chunks", 1)
    import torch.distributed as dist

    freqs = np.asarray(kw["freqs"])
    ntimes = len(julian_dates(kw["times"]))

    def channels_of_rank(nsrc):  # what every rank's blocks cover: the flux columns it is sent
        out = []
        for blist in shard_blocks_weighted(dist.get_world_size(), freqs, ntimes, nsrc):
            out.append((min(f.start for _, f in blist), max(f.stop for _, f in blist)) if blist else None)
        return out

    cat = broadcast_catalog_device(kw.pop("ra", None), kw.pop("dec", None), kw.pop("fluxes", None), polarized,
                                   precision, torch.device("cuda", int(device)), via_host=via_host,
                                   freq_ranges=channels_of_rank)
    engine = create_simulation_engine("gpu", device=int(device))
    if "nchunks" not in kw:  # the wrapper's memory knobs (reference wrapper.py:292-302), against this rank's device
        from .wrapper import device_chunks

        nfeed = 2 if polarized else 1
        kw["nchunks"] = device_chunks(int(device), max_memory, min_chunks, kw["beam_list"], nfeed, nfeed,
                                      len(kw["ants"]), cat.nsrc, precision, kw.get("source_buffer", 1.0), len(freqs))

    def compute_block(tsl, fsl, out=None):
        return engine.simulate(ra=None, dec=None, fluxes=None, catalog_device=cat, time_idx=tsl, freq_idx=fsl,
                               out=out, out_shared=out is not None, **kw)

    blocks = shard_blocks_weighted(dist.get_world_size(), freqs, ntimes, cat.nsrc)
    if gather not in ("auto", "shm", "p2p"):
        raise ValueError(f"gather must be 'auto', 'shm' or 'p2p', got {gather!r}")
    use_shm = gather_to is not None and gather != "p2p" and _all_ranks_on_one_host()
    if gather == "shm" and gather_to is not None and not use_shm:
        raise ValueError("gather='shm' needs every rank on the same host")
    if not use_shm:
        return simulate_sharded(compute_block, len(freqs), ntimes, gather_to, blocks=blocks)
    nbls = len(kw["baselines"]) if kw.get("baselines") is not None else None
    if nbls is None:  # the reference's default: one baseline per redundant group (cpu_simulate.py:614-616)
        from .core.utils import get_pos_reds

        nbls = len(get_pos_reds({k: np.asarray(v) for k, v in kw["ants"].items()}, include_autos=True))
    tail = (2, 2, nbls) if polarized else (nbls,)
    res = _SharedResult((len(freqs), ntimes) + tail, np.complex64 if precision == 1 else np.complex128, gather_to)
    for tsl, fsl in blocks[dist.get_rank()]:
        compute_block(tsl, fsl, out=res.array[fsl, tsl])
    return res.seal()
