"""CPU simulation of the binned gridding kernel's apply-loop lane utilisation on the metric trajectory (512^2 grid, 402 golden-angle spokes):
per tile, per record batch, visits per 2x2-block thread; a wave runs max-over-lanes.  Compares batches of consecutive accepted spokes
(what the kernel does) with batches stratified by the spokes' perpendicular offset.  usage: python tools/probe/lane_util_sim.py [records per batch]
Result: consecutive golden-angle spokes are already well spread; stratifying does not help."""
import numpy as np, sys
n=512; h=256; rmax=255; W=2.0; CW=2; T=32; NREC=int(sys.argv[1]) if len(sys.argv)>1 else 256
npe=402
PHI=np.float32(1.9416089796736116)
pe=np.arange(npe,dtype=np.float32)
t=np.fmod((PHI*pe).astype(np.float32).astype(np.float64), 2*np.pi)
c=np.cos(t); s=np.sin(t)
def tile_stats(tx,ty,mode):
    x0=tx*T-h; y0=ty*T-h
    eps=0.01
    bxlo=x0-W-eps; bxhi=x0+T-1+W+eps; bylo=y0-W-eps; byhi=y0+T-1+W+eps
    segs=[]
    for j in range(npe):
        ic=1/c[j] if c[j]!=0 else 1e30; isn=1/s[j] if s[j]!=0 else 1e30
        xa,xb=bxlo*ic,bxhi*ic; ya,yb=bylo*isn,byhi*isn
        lo=max(min(xa,xb),min(ya,yb),-rmax); hi=min(max(xa,xb),max(ya,yb),rmax)
        if lo<=hi:
            rlo=int(np.ceil(lo)); rhi=int(np.floor(hi))
            if rhi>=rlo: segs.append((j,rlo,rhi-rlo+1))
    if not segs: return 0,0,0
    xc=x0+T/2-0.5; yc=y0+T/2-0.5
    if mode=='strat':
        off=[xc*s[j]-yc*c[j] for j,_,_ in segs]
        order=np.argsort(off,kind='stable')
        total=sum(l for _,_,l in segs)
        nb=max(1,int(np.ceil(total/(NREC*0.9))))
        while True:
            batches=[[] for _ in range(nb)]
            for p,i in enumerate(order): batches[p%nb].append(segs[i])
            if max(sum(l for _,_,l in b) for b in batches)<=NREC: break
            nb+=1
    else:
        batches=[]; cur=[]; cnt=0
        for sg in segs:
            if cnt+sg[2]>NREC and cur: batches.append(cur); cur=[]; cnt=0
            cur.append(sg); cnt+=sg[2]
        if cur: batches.append(cur)
    # per batch visits per thread
    tot_wave_iters=0; tot_visits=0; tot_wg_iters=0
    for b in batches:
        cnt=np.zeros((T//2,T//2),int)
        for j,rlo,l in b:
            r=np.arange(rlo,rlo+l)
            kx=(r.astype(np.float32)*np.float32(c[j])); ky=(r.astype(np.float32)*np.float32(s[j]))
            fx=np.floor(kx).astype(int)-x0; fy=np.floor(ky).astype(int)-y0   # tile-relative base cell
            # block bx (points 2bx,2bx+1) sees cells fx in [2bx-2, 2bx+2]
            for f_x,f_y in zip(fx,fy):
                bxs=[bx for bx in range(max(0,(f_x-2+1)//2), min(T//2-1,(f_x+2)//2)+1) if 2*bx-2<=f_x<=2*bx+2]
                bys=[by for by in range(max(0,(f_y-2+1)//2), min(T//2-1,(f_y+2)//2)+1) if 2*by-2<=f_y<=2*by+2]
                for by in bys:
                    for bx in bxs: cnt[by,bx]+=1
        # waves: wave w covers block rows 4w..4w+3
        wm=[cnt[4*w:4*w+4,:].max() for w in range(4)]
        tot_wave_iters+=sum(wm); tot_visits+=cnt.sum(); tot_wg_iters+=4*max(wm)
    return tot_visits, tot_wave_iters*64, tot_wg_iters*64
for mode in ('seq','strat'):
    V=WI=WG=0
    res={}
    for (tx,ty) in [(8,8),(9,8),(10,8),(12,8),(15,8),(10,10),(12,12),(14,13),(9,11),(11,14)]:
        v,wi,wg=tile_stats(tx,ty,mode); V+=v; WI+=wi; WG+=wg
        res[(tx,ty)]=(v, round(v/max(wi,1),3), round(v/max(wg,1),3))
    print(mode, NREC, 'lane util (wave max)', round(V/WI,3), 'incl. barrier wait (wg max)', round(V/WG,3))
    print(res)
