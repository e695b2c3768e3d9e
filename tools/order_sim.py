"""Tuning aid: list scheduling of measured per-window busy times (gpurun_out/window_cost_<case>.npz from tools/window_cost.py) on the
window kernel's slots under several processing orders.   python tools/order_sim.py"""
import numpy as np, heapq, sys
def sim(t, order, slots):
    h=[0.0]*slots; heapq.heapify(h); end=0
    for i in order:
        s=heapq.heappop(h); e=s+t[i]; end=max(end,e); heapq.heappush(h,e)
    return end
def cost8(hdr,cm):
    status=hdr[:,0]&0xFF; why=(hdr[:,0]>>8)&0xFF; heavy=hdr[:,2]; nsurv=hdr[:,4]; done=cm[:,0]; mlive=cm[:,1]; K=hdr[:,1]; N=hdr[:,3]
    c=np.where(done==1, np.where(mlive<=5,3,np.minimum(23,(3*mlive)//2-3)), 13+np.where(nsurv>600,6,0))
    c=np.where(heavy>0,26,c)
    c=np.where(status!=1,31,c)
    return c
for case,slots in (("bench",3824),("bench60",3824),("bench4",3824)):
    d=np.load(f"gpurun_out/window_cost_{case}.npz"); t=d["phase"].sum(axis=1)*1000; n=len(t)
    hdr=d["hdr"]; cm=d["cmp"]; heavy=hdr[:,2]>0; status=hdr[:,0]&0xFF
    first=np.where(status!=1, True, heavy)
    cur=np.concatenate([np.where(first)[0], np.where(~first)[0][::-1]])
    rng=np.random.default_rng(1); nf=np.where(~first)[0]; rng.shuffle(nf); cur2=np.concatenate([np.where(first)[0], nf])
    c=cost8(hdr,cm); new=np.argsort(-c,kind="stable")
    ideal=np.argsort(-t)
    print(case,"lower bound",t.sum()/slots,"measured kernel",d["kernel_ms"][1],"sim current",sim(t,cur,slots),"sim current(random)",sim(t,cur2,slots),"sim classes",sim(t,new,slots),"sim LPT oracle",sim(t,ideal,slots))
print("--- what limits the class order")
d=np.load("gpurun_out/window_cost_bench.npz"); t=d["phase"].sum(axis=1)*1000; hdr=d["hdr"]; cm=d["cmp"]; c=cost8(hdr,cm); b=d["builds"]
c2=np.where(b>1,28,c); print("classes + multi-build known:",sim(t,np.argsort(-c2,kind="stable"),3824))
# tail analysis of class order
def sim_ends(t, order, slots):
    h=[0.0]*slots; heapq.heapify(h); ends=np.zeros(len(t)); starts=np.zeros(len(t))
    for i in order:
        s=heapq.heappop(h); e=s+t[i]; ends[i]=e; starts[i]=s; heapq.heappush(h,e)
    return starts,ends
o=np.argsort(-c,kind="stable"); s,e=sim_ends(t,o,3824)
last=np.argsort(-e)[:15]
for i in last: print(" start",round(s[i],2),"end",round(e[i],2),"t",round(t[i],2),"class",c[i],"builds",b[i],"mlive",cm[i,1],"done",cm[i,0],"nsurv",hdr[i,4],"ncomp",hdr[i,5],"nvar",d["nvar"][i])
print("queue drained at", s.max())
for cl in sorted(set(c.tolist()),reverse=True):
    m=c==cl; print(" class",cl,"n",m.sum(),"mean t",round(t[m].mean(),2),"p99",round(np.percentile(t[m],99),2),"max",round(t[m].max(),2),"start range",round(s[m].min(),2),round(s[m].max(),2))
