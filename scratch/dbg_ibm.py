import sys; sys.path.insert(0,'.')
import numpy as np, torch
from tests.test_gpu_ibm import Ibm, sphere_markers
from tests.gpu_common import *
n=(24,20,18); bc=[V]*6
box=[(0.,1.)]*3
P,g=make_pair(n,bc,box=box)
rng=np.random.default_rng(4)
L=300
X=sphere_markers(L,(0.5,0.5,0.5),0.3)
X[0][:5] = [0.01, 0.99, 0.5, 0.02, 0.97]
X[1][:5] = [0.5, 0.5, 0.01, 0.98, 0.03]
X[2][:5] = [0.03, 0.5, 0.99, 0.5, 0.5]
for kind in (0,1):
    m=Ibm(P,kind,X)
    F=rng.standard_normal((3,L)); dV=rng.uniform(0.5,1.5,L)*1e-3
    f0=np.zeros((3,g.ncell))
    f=host(m.spread(dev(F),dev(dV),dev(f0),3)).reshape(3,-1)
    ref=g.ibm_spread(kind,X,dV,F,f0.copy())
    d=abs(f-ref)
    print(kind,"maxdiff",d.max(),"refmax",abs(ref).max(), "nbad",(d>1e-10).sum(), "sum f",f.sum(), ref.sum())
    bad=np.argwhere(d>1e-10)[:10]
    for c,cell in bad:
        k=cell//(n[0]*n[1]); j=(cell//n[0])%n[1]; i=cell%n[0]
        print("  comp",c,"cell",(i,j,k),f[c,cell],ref[c,cell])
    # per-marker test
    for l in [0,1,5,100]:
        Fl=np.zeros((3,L)); Fl[0,l]=1.0
        f=host(m.spread(dev(Fl),dev(np.ones(L)),dev(f0),3)).reshape(3,-1)
        ref=g.ibm_spread(kind,X,np.ones(L),Fl,f0.copy())
        print("   marker",l,"diff",abs(f-ref).max(), "nnz gpu",(f[0]!=0).sum(),"nnz ref",(ref[0]!=0).sum())
    m.close()
