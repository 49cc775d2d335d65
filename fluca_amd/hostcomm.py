"""Host-staged halo transport over a torch.distributed (gloo) group: the callbacks fl_poisson_comm_init_host expects.

A rehearsal / test transport -- faces travel device -> pinned host -> gloo -> pinned host -> device; the production wire is
RCCL Send/Recv inside the library (fl_poisson_comm_init_rccl)."""
import numpy as np
import torch
import torch.distributed as dist


def gloo_exchange(msgs):
    """msgs: list of (peer, sendtag, recvtag, send ndarray | None, recv ndarray | None) -- the fl_exchange_fn contract."""
    reqs, keep = [], []
    for peer, stag, rtag, s, r in msgs:
        if r is not None:
            t = torch.from_numpy(r)
            reqs.append(dist.irecv(t, src=int(peer), tag=int(rtag)))
        if s is not None:
            t = torch.from_numpy(np.ascontiguousarray(s).copy())
            keep.append(t)
            reqs.append(dist.isend(t, dst=int(peer), tag=int(stag)))
    for q in reqs:
        q.wait()


def gloo_allreduce(vals):
    t = torch.from_numpy(vals)
    dist.all_reduce(t)
