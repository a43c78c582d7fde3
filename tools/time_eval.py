import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bayhunter_amd import engine as E
class A: pass
args=A(); args.batch=4096; args.layers=10; args.steps=40; args.warmup=3; args.no_parity=True; args.full=False; args.no_cpu_baseline=True; args.no_rf_roofline=True
eng=E.Engine(0)
dev=torch.device('cuda',0); torch.cuda.set_device(0)
import torch.distributed as dist
out={}
for w in sys.argv[1:] or ['c2','c3','c2g']:
    for search in ('fast','reference'):
        for pre in (1,0):
            eng.set_swd_search(search); eng.set_swd_prescan(pre)
            r=bench.run_eval(args,eng,0,1,dist,dev,w,False,with_cpu=False,rf_roof=False)
            print(w,search,'prescan',pre,'ms/step %.3f'%r['ms_per_step'],'kernel',{k:round(v,3) for k,v in r['kernel_ms_per_step'].items()}, flush=True)
