"""Per-tile map of the binned gridding kernel's apply-loop cost on the metric trajectory (512^2 grid, 402 golden-angle spokes),
simulated on the CPU: visits per 2x2-block thread and batch; a wave runs max-over-lanes, a workgroup max-over-waves.
Prints, per ring of tiles (Chebyshev distance from the k-space centre), the share of useful visits, of wave iterations and of
workgroup iterations.  usage: python tools/probe/lane_util_map.py [records per batch]"""
import numpy as np, sys
n=512; h=256; rmax=255; W=2.0; T=32; NREC=int(sys.argv[1]) if len(sys.argv)>1 else 256
npe=402
PHI=np.float32(1.9416089796736116)
pe=np.arange(npe,dtype=np.float32)
t=np.fmod((PHI*pe).astype(np.float32).astype(np.float64), 2*np.pi)
c=np.cos(t).astype(np.float32); s=np.sin(t).astype(np.float32)
def tile(tx,ty):
    x0=tx*T-h; y0=ty*T-h; eps=0.01
    bxlo=x0-W-eps; bxhi=x0+T-1+W+eps; bylo=y0-W-eps; byhi=y0+T-1+W+eps
    with np.errstate(divide='ignore'):
        ic=np.where(c!=0,1/c.astype(np.float64),1e30); isn=np.where(s!=0,1/s.astype(np.float64),1e30)
    xa,xb=bxlo*ic,bxhi*ic; ya,yb=bylo*isn,byhi*isn
    lo=np.maximum(np.maximum(np.minimum(xa,xb),np.minimum(ya,yb)),-rmax); hi=np.minimum(np.minimum(np.maximum(xa,xb),np.maximum(ya,yb)),rmax)
    rlo=np.ceil(lo).astype(int); rhi=np.floor(hi).astype(int)
    ok=(lo<=hi)&(rhi>=rlo)
    segs=[(j,rlo[j],rhi[j]-rlo[j]+1) for j in range(npe) if ok[j]]
    batches=[]; cur=[]; cnt=0
    for sg in segs:
        if cnt+sg[2]>NREC and cur: batches.append(cur); cur=[]; cnt=0
        cur.append(sg); cnt+=sg[2]
    if cur: batches.append(cur)
    V=WI=WG=0
    for b in batches:
        H=np.zeros((T+4,T+4),int)
        for j,r0,l in b:
            r=np.arange(r0,r0+l).astype(np.float32)
            fx=np.floor(r*c[j]).astype(int)-x0+2; fy=np.floor(r*s[j]).astype(int)-y0+2
            m=(fx>=0)&(fx<T+4)&(fy>=0)&(fy<T+4)
            np.add.at(H,(fy[m],fx[m]),1)
        # block (by,bx) sees cells rows 2by..2by+4, cols 2bx..2bx+4 (shifted by the +2 above)
        S=np.zeros((T+5,T+5),int); S[1:,1:]=H.cumsum(0).cumsum(1)
        idx=2*np.arange(T//2)
        cntb=S[idx[:,None]+5,idx[None,:]+5]-S[idx[:,None],idx[None,:]+5]-S[idx[:,None]+5,idx[None,:]]+S[idx[:,None],idx[None,:]]
        wm=[cntb[4*w:4*w+4,:].max() for w in range(4)]
        V+=cntb.sum(); WI+=sum(wm)*64; WG+=4*max(wm)*64
    return V,WI,WG,len(batches)
rings={}
for ty in range(16):
    for tx in range(16):
        v,wi,wg,nb=tile(tx,ty)
        ring=max(abs(tx-7.5),abs(ty-7.5))-0.5
        a=rings.setdefault(int(ring),[0,0,0,0,0]); a[0]+=v; a[1]+=wi; a[2]+=wg; a[3]+=nb; a[4]+=1
TV=sum(a[0] for a in rings.values()); TW=sum(a[1] for a in rings.values()); TG=sum(a[2] for a in rings.values())
print(f"records/batch {NREC}: lane util wave-max {TV/TW:.3f}, wg-max {TV/TG:.3f}")
for r in sorted(rings):
    v,wi,wg,nb,nt=rings[r]
    print(f" ring {r}: tiles {nt:3d} batches {nb:5d} visits {100*v/TV:5.1f}% wave-iters {100*wi/TW:5.1f}% wg-iters {100*wg/TG:5.1f}%  util wave {v/max(wi,1):.3f} wg {v/max(wg,1):.3f}")
