"""Worker of tests/test_gpu_scale.py::test_sharded_run_two_ranks_on_the_gpu (launched by
torch.distributed.run, gloo rendezvous, every rank on device 0): runs parallel.simulate_vis_sharded
through the GPU engine and saves rank 0's assembled visibilities."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import fftvis_amd  # noqa: E402
from fftvis_amd import parallel, synth  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = synth.make_config("C2", nsrc=2500, nfreq=12, ntimes=6)
    cfg["polarized"] = True
    freqs = cfg["freqs"]
    cfg["beam"] = fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(freqs, nza=46, naz=90), freqs)
    if rank != 0:  # only rank 0 holds the catalog; the others receive it into device memory
        cfg["ra"] = cfg["dec"] = cfg["fluxes"] = None
    # one node: every rank delivers its block straight into ONE result in shared memory ...
    vis = parallel.simulate_vis_sharded(device=0, gather_to=0, via_host=True, **cfg)
    # ... and the route ranks on several hosts take: blocks travel to rank 0 as point-to-point tensors
    vis_p2p = parallel.simulate_vis_sharded(device=0, gather_to=0, via_host=True, gather="p2p", **cfg)
    if rank == 0:
        blocks = parallel.shard_blocks_weighted(world, freqs, 6)
        flat = [(b[0][0].start, b[0][0].stop, b[0][1].start, b[0][1].stop) for b in blocks if b]
        np.savez(sys.argv[1], vis=np.array(vis), vis_p2p=vis_p2p, blocks=np.array(flat))
    else:
        assert vis is None and vis_p2p is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
