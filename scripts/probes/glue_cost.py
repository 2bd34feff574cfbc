import numpy as np, time
S=32; n=36960
rng=np.random.default_rng(0)
kp=rng.random((n,2))*300; is3d=rng.random(n)<0.9; sid=np.repeat(np.arange(S,dtype=np.int32), n//S)
ok=rng.random(n)<0.98
noise=rng.normal(0,0.5,(1<<17,2)); fl=rng.random((S,2))
def t(f,k=50):
    f(); t0=time.perf_counter()
    for _ in range(k): f()
    return (time.perf_counter()-t0)/k*1e6
print("proj", t(lambda: kp + fl[sid] + noise[100:100+n]))
new=kp+1
print("filter", t(lambda: (new[ok], is3d[ok], sid[ok], int(ok.sum()))))
keep=rng.random(n)>=0.15
print("cull (rng+filter)", t(lambda: (rng.random(n)>=0.15, kp[keep], is3d[keep], sid[keep])))
m=int(0.15*n); fresh=(rng.random((m,2))*300).astype(np.int64); fsid=np.sort(rng.integers(0,S,m)).astype(np.int32)
kp2,is2,sid2=kp[keep],is3d[keep],sid[keep]
def merge():
    a=np.searchsorted(sid2,np.arange(S+1)); b=np.searchsorted(fsid,np.arange(S+1)); fr=fresh.astype(np.float64)
    k=np.concatenate([x for s_ in range(S) for x in (kp2[a[s_]:a[s_+1]], fr[b[s_]:b[s_+1]])])
    i=np.concatenate([x for s_ in range(S) for x in (is2[a[s_]:a[s_+1]], np.zeros(b[s_+1]-b[s_],dtype=bool))])
    s=np.concatenate([x for s_ in range(S) for x in (sid2[a[s_]:a[s_+1]], fsid[b[s_]:b[s_+1]])])
    return k,i,s
print("merge", t(merge))
print("wrapper prep (ascontiguous etc.)", t(lambda: (np.ascontiguousarray(kp,dtype=np.float64).reshape(-1,2), np.ascontiguousarray(sid,dtype=np.int32), np.ascontiguousarray(is3d,dtype=np.uint8), np.empty((n,2)), np.zeros(n,dtype=np.uint8))))
st=ok.view(np.uint8)
print("where", t(lambda: np.where(ok[:,None], new, kp)))
