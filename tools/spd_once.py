import sys, numpy as np
sys.path.insert(0,'/root/repo')
import __graft_entry__ as e
vsl=e.load_package()
ctx=vsl.Context(0)
n,bw=5988,221
rng=np.random.default_rng(11)
S=np.zeros((n,n))
for d in range(1,bw+1):
    v=rng.standard_normal(n-d); S[np.arange(d,n),np.arange(0,n-d)]=v
S=S+S.T; S[np.arange(n),np.arange(n)]=np.abs(S).sum(1)+1
b=rng.standard_normal(n)
x=ctx.spd_solve(S,b,bw)
print("res", np.linalg.norm(S@x-b)/np.linalg.norm(b))
