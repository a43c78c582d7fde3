#!/usr/bin/env python3
"""Dry run of the sharded parallel-tempering flow on a box with ONE GPU: N ranks (gloo), all on GPU 0,
rank r holds temperature rung r of every ladder (BASELINE configs[4] layout).  Checks that every rank
computes the same exchange and that the multiset of temperatures is conserved (dev tool):
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/gpu_pt_dryrun.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import bayhunter_amd as bh
from bayhunter_amd.device_chains import DeviceChains

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "chain_golden.npz"))
t1 = bh.RayleighDispersionPhase(g["xsw"], g["ysw"])
t2 = bh.PReceiverFunction(g["xrf"], g["yrf"])
t2.moddata.plugin.set_modelparams(gauss=1.0, p=6.4)
priors = dict(vpvs=(1.4, 2.1), layers=(1, 10), vs=(2, 5), z=(0, 60), rfnoise_corr=(0.35, 0.75),
              rfnoise_sigma=(1e-5, 0.05), swdnoise_corr=0., swdnoise_sigma=(1e-5, 0.1))
init = dict(nchains=1, iter_burnin=300, iter_main=100, acceptance=(40, 45), thickmin=0.1, lvz=0.1, hvz=None,
            rcond=None, maxmodels=20)
C = 16                                            # ladders; this rank's chains all start at rung `rank`
betas = np.full(C, (1.0 / np.geomspace(1.0, 8.0, world))[rank])
dc = DeviceChains(bh.JointTarget([t1, t2]), C, init, priors, seed=100 + rank, betas=betas, ladder=np.arange(C),
                  swap_every=25, dist=dist).run()
mine = torch.tensor(dc.state_host()["beta"])
allb = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(allb, mine)
allb = torch.stack(allb).numpy()                  # [rank, ladder]
ok = all(np.allclose(np.sort(allb[:, l]), np.sort(1.0 / np.geomspace(1.0, 8.0, world))) for l in range(C))
print("rank %d: sweeps %d, accepted swaps (global) %d, temperatures conserved per ladder: %s" % (rank, dc.sweep, dc.nswaps, ok), flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok and dc.nswaps > 0 else 1)
