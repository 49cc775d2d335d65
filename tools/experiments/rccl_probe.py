import sys, os
sys.path.insert(0, '.')
import numpy as np
from tests import mp_common as mpc

def w(rank, world):
    import torch, torch.distributed as dist
    from fluca_amd import capi, poisson as flp
    n=(24,20,16); ranks=(1,1,2); bc=[1,1,1,1,4,1]
    d=mpc.decomp_of(capi,n,ranks,rank)
    P=flp.Poisson.uniform(n,[(0,1),(0,1),(0,0.5)],bc,1e-3,decomp=d)
    idb=[flp.rccl_unique_id() if rank==0 else None]
    dist.broadcast_object_list(idb,src=0)
    print(rank,"init rccl...",flush=True)
    P.comm_init_rccl(idb[0],rank,world)
    print(rank,"rccl ok",flush=True)
    x=torch.ones(P.ncell,dtype=torch.float64,device="cuda")*(rank+1)
    y=P.apply(x); torch.cuda.synchronize()
    print(rank,"apply ok",float(y.abs().max()),flush=True)
    b=torch.rand(P.ncell,dtype=torch.float64,device="cuda")-0.5
    xs,info=P.solve(b,maxit=50)
    print(rank,"solve",info["iters"],info["reason"],flush=True)
    P.close()
if __name__ == "__main__":
    os.environ["NCCL_DEBUG"]="WARN"
    mpc.run_ranks(2,w,timeout=150)
    print("RCCL same-device OK")
