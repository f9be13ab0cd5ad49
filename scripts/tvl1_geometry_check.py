# replicate k_tvl1_step_fused's geometry (geom bits) and verify: exact partition, dependency cone inside tile-or-outside-image,
# tile count <= launched grid
import itertools, random
TW, TH = 64, 32
def tiles(w,h,K,n_iters,geom):
    Kh = min(K,n_iters) if geom&2 else K
    SWh, SHh = TW-2*Kh, TH-2*Kh
    shift = bool(geom&1)
    ntx = max(1,(w-2*Kh+SWh-1)//SWh) if shift else (w+SWh-1)//SWh
    nty = (h+SHh-1)//SHh
    out=[]
    for t in range(ntx*nty):
        ty, tx = divmod(t, ntx)
        xs = tx*SWh if shift else tx*SWh-Kh
        ys = ty*SHh-Kh
        out.append((xs,ys,shift and tx==0, shift and tx==ntx-1))
    return Kh,out
def grid(w,h,K,geom,split_warp=True):
    tiles_x=(w+(64-2*K)-1)//(64-2*K); tiles_y=(h+(TH-2*K)-1)//(TH-2*K)
    g=tiles_x*tiles_y
    if geom&1 and split_warp: g=max(1,(w-2*K+(64-2*K)-1)//(64-2*K))*tiles_y
    return g
def check(w,h,K,n,geom):
    Kh,ts=tiles(w,h,K,n,geom)
    assert len(ts)<=grid(w,h,K,geom), (w,h,K,n,geom,len(ts),grid(w,h,K,geom))
    own={}
    for (xs,ys,lo,hi) in ts:
        for ly in range(Kh,TH-Kh):
            gy=ys+ly
            if not(0<=gy<h): continue
            for lx in range(TW):
                gx=xs+lx
                if not (0<=gx<w): continue
                if not ((lx>=Kh or lo) and (lx<TW-Kh or hi)): continue
                assert (gx,gy) not in own, ("double",w,h,K,n,geom,gx,gy)
                own[(gx,gy)]=1
                # cone
                x0=max(gx-Kh,0); x1=min(gx+Kh,w-1); y0=max(gy-Kh,0); y1=min(gy+Kh,h-1)
                assert xs<=x0 and x1<xs+TW and ys<=y0 and y1<ys+TH, ("cone",w,h,K,n,geom,gx,gy,xs,ys)
    assert len(own)==w*h, ("missing",w,h,K,n,geom,len(own),w*h)
random.seed(1)
cases=[(786,442),(1920,1080),(983,553),(16,16),(57,40),(64,64),(56,24),(60,28),(65,33),(120,442),(300,200),(97,61),(224,224),(8,8),(1,1),(63,5),(128,31)]
cases+= [(random.randint(1,400),random.randint(1,200)) for _ in range(60)]
cnt=0
for (w,h) in cases:
    for K in (1,2,3,4,5,6,7,12):
        for n in range(1,K+1):
            for geom in (0,1,2,3):
                if w*h>400000 and (K not in (4,) or n not in (2,4)): continue
                check(w,h,K,n,geom); cnt+=1
print("ok",cnt)
for (w,h) in [(1920,1080),(1536,864),(1229,691),(983,553),(786,442)]:
    print(w,h,[ (len(tiles(w,h,4,n,g)[1])) for n in (4,2) for g in (0,1,2,3)])
