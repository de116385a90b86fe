"""Random solver configurations of every family: the root lists must not depend on the number of lanes per task that
es_worker_run uses for its speculative mid-point evaluation (GPU box; not a test)."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import eigensolver_amd as E
from eigensolver_amd import _lib
ctx=_lib.Context(0)
rng=np.random.default_rng(11)
bad=0; n=0
def mk(i):
    c=i%6
    if c==0: return E.CylinderNonUniformFlow(U_i0=float(rng.uniform(-0.8,0.8)), width=float(rng.choice([0.6,0.9,1.5,3.0,1e5])), ctx=ctx)
    if c==1: return E.CylinderNonUniformDensity(width=float(rng.choice([0.9,1.25,1.75,3.0])), photospheric=bool(rng.integers(2)), ctx=ctx)
    if c==2: return E.CylinderRotationalFlow(v_twist=float(rng.choice([0.05,0.1,0.25])), power=float(rng.choice([0.8,1.0,1.25])), variant=str(rng.choice(["kink_fast","kink_slow","sausage","sausage_slow"])), ctx=ctx)
    if c==3: return E.SlabNonUniformFlow(U_i0=float(rng.uniform(0.1,0.9)), width=float(rng.choice([1.0,1.5,3.0,1e5])), ctx=ctx)
    if c==4: return E.SlabNonUniformDensity(width=float(rng.choice([0.9,1.5,3.0])), coronal=bool(rng.integers(2)), ctx=ctx)
    return E.SlabUniformFlow(ctx=ctx)
for i in range(60):
    s=mk(i)
    ks=np.sort(rng.uniform(0.05,4.0,int(rng.integers(3,40))))
    npb=int(rng.integers(8,60))
    res={}
    for ll in ("0", str(int(rng.integers(1,7)))):
        os.environ["ES_WORKER_LOG_LANES"]=ll
        res[ll]=s.solve(ks, npb) if not isinstance(s,E.SlabUniformFlow) else s.solve(ks)
    a,b=list(res.values())
    for mode in a:
        n+=1
        if not (np.array_equal(a[mode][0],b[mode][0]) and np.array_equal(a[mode][1],b[mode][1])):
            bad+=1; print("MISMATCH",type(s).__name__,mode,len(a[mode][0]),len(b[mode][0]))
    s.close()
print("compared",n,"mode results; mismatches",bad)
