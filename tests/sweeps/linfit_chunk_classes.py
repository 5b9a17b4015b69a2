import numpy as np, sys
f32=np.float32
P=64*200; N=128; kappa=f32(3.0)
rng=np.random.default_rng(3)
k=np.arange(N)
bg=1000.0+5.0*np.sin(k); gain=1.0+0.02*np.cos(1.7*k); sig=30.0*(1.0+0.5*(k%7)/6.0)
sky=200.0*rng.uniform(0,1,P)
v=bg[None,:]+gain[None,:]*sky[:,None]+sig[None,:]*rng.standard_normal((P,N))
uo=rng.uniform(size=(P,N)); um=rng.uniform(size=(P,N))
v=np.where(uo<0.004,v+300.0+19700.0*um,np.where(uo<0.005,v-(100.0+800.0*um),v))
ys=np.sort(v.astype(f32),axis=1)
alive=np.ones((P,N),bool)
active=np.ones(P,bool)
CH=int(sys.argv[1]) if len(sys.argv)>1 else 8
for it in range(1,13):
    m=alive.sum(1)
    rank=np.cumsum(alive,1)-1
    # fp64 fit is fine for class statistics
    y=np.where(alive,ys,0).astype(np.float64)
    ym=y.sum(1)/m
    xm=(m-1)/2.0
    dx=np.where(alive,rank-xm[:,None],0.0)
    dy=np.where(alive,ys-ym[:,None],0.0)
    xsd=np.sqrt((dx*dx).sum(1)/m); ysd=np.sqrt((dy*dy).sum(1)/m)
    slope=(dx*dy).sum(1)/(xsd*ysd*(m+1))*ysd/xsd
    icpt=ym-slope*xm
    r=np.where(alive,ys-(rank*slope[:,None]+icpt[:,None]),0.0)
    sg=np.abs(r).sum(1)/m
    rej=alive&((-r>kappa*sg[:,None])|(r>kappa*sg[:,None]))&active[:,None]
    # wave-level classes
    A=alive.reshape(-1,64,N//CH,CH); act=active.reshape(-1,64)
    full=A.all(3); empty=(~A).all(3)
    aa=(full|~act[:,:,None]).all(1); ad=(empty|~act[:,:,None]).all(1)
    mixed=~aa&~ad
    R=rej.reshape(-1,64,N//CH,CH).any(3).any(1)
    wav=act.any(1)
    print("iter %2d active px %5.1f%%  waves active %5.1f%%  per active wave: aa %.1f dead %.1f mixed %.1f  chunks with a reject %.1f (of %d)  lanes active/wave %.1f"%(it,100*active.mean(),100*wav.mean(),aa[wav].sum(1).mean(),ad[wav].sum(1).mean(),mixed[wav].sum(1).mean(),R[wav].sum(1).mean(),N//CH,act[wav].sum(1).mean()))
    nrej=rej.sum(1)
    alive&=~rej
    active&=~((nrej==0)|(m<3))
