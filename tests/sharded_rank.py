"""One rank of a sharded DeviceChains job (helper of tests/test_gpu_sharded.py, not a test module):
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tests/sharded_rank.py OUTDIR BACKEND
BACKEND gloo: all ranks share GPU 0 (the GPU boxes of the build have one GPU).  BACKEND nccl (= RCCL): one GPU per rank
(cuda:LOCAL_RANK), the collectives run over xGMI -- needs as many GPUs as ranks."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch.distributed as dist


def job(nlocal, offset, total, dist_, out, device=0):
    """The job of test_gpu_sharded: `total` chains = total/4 ladders x 4 rungs, ladders spanning the ranks."""
    from conftest import golden
    from test_gpu_chains import SETUPS, make_targets
    from bayhunter_amd.device_chains import DeviceChains
    su = SETUPS["exp"]
    init = dict(su["init"], iter_burnin=240, iter_main=160, maxmodels=40, savepath=out)
    nl = total // 4
    gid = offset + np.arange(nlocal)
    betas = (1.0 / np.geomspace(1.0, 12.0, 4))[gid // nl]
    dc = DeviceChains(make_targets(golden("chain_golden.npz")), nlocal, init, su["priors"], seed=2024, betas=betas,
                      ladder=gid % nl, swap_every=10, dist=dist_, device=device)
    assert dc.chain_offset == offset
    dc.run()
    dc.save()
    return dc


if __name__ == "__main__":
    out, backend = sys.argv[1], sys.argv[2]
    device = 0
    if backend == "nccl":      # deliberately NO torch.cuda.set_device: DeviceChains(device=) must place everything itself
        import torch
        device = int(os.environ.get("LOCAL_RANK", "0"))
        dist.init_process_group(backend, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    total = 24
    nlocal = total // world
    dc = job(nlocal, rank * nlocal, total, dist, out, device)
    print("rank %d: sweeps %d, accepted swaps %d" % (rank, dc.sweep, dc.nswaps), flush=True)
    dist.barrier()
    dist.destroy_process_group()
