import numpy as np
from numpy.polynomial import chebyshev as C
np.set_printoptions(precision=17)
H=np.pi/2*1.0005   # a little beyond pi/2: the reduced argument can exceed it by rounding
def fit(f, w, deg, iters=60):
    # Lawson-style iteratively reweighted least squares towards minimax of w*(P-f) on u in [0,H^2]
    k=np.arange(4000); u=(H*H)*(0.5-0.5*np.cos(np.pi*(k+0.5)/4000))
    F=f(u); W=w(u); lw=np.ones_like(u)
    A=np.vander(u,deg+1,increasing=True)
    for _ in range(iters):
        sw=np.sqrt(lw)*W
        c,*_=np.linalg.lstsq(A*sw[:,None],F*sw,rcond=None)
        e=np.abs(W*(A@c-F)); lw=lw*(e/e.max()+1e-3); lw/=lw.sum()
    return c,e.max()
fs=lambda u: np.where(u>1e-12,(np.sin(np.sqrt(u))/np.sqrt(np.maximum(u,1e-300))-1)/np.maximum(u,1e-300),-1/6)
fc=lambda u: np.where(u>1e-8,(np.cos(np.sqrt(u))-1+u/2)/np.maximum(u*u,1e-300),1/24)
cs,es=fit(fs,lambda u:u**1.5,3)
cc,ec=fit(fc,lambda u:u**2,2)
print('sin coeffs (u^0..u^3 of r^3 term):',cs, 'max abs err (double)',es)
print('cos coeffs (u^0..u^2 of r^4 term):',cc, 'max abs err (double)',ec)
# float32 evaluation with fma emulation
def f32(x): return np.float32(x)
def fma(a,b,c): return np.float32(np.float64(a)*np.float64(b)+np.float64(c))
cs32=[f32(x) for x in cs]; cc32=[f32(x) for x in cc]
r=np.linspace(-H,H,2000001).astype(np.float32)
r2=(r*r).astype(np.float32)
p=fma(cs32[3],r2,cs32[2]); p=fma(p,r2,cs32[1]); p=fma(p,r2,cs32[0]); sp=fma(p,(r2*r).astype(np.float32),r)
q=fma(cc32[2],r2,cc32[1]); q=fma(q,r2,cc32[0]); cp=fma(q,(r2*r2).astype(np.float32),fma(np.float32(-0.5),r2,np.float32(1)))
s_true=np.sin(r.astype(np.float64)); c_true=np.cos(r.astype(np.float64))
print('float32 eval: max |sin err|',np.abs(sp-s_true).max(),' max rel sin err',np.max(np.abs(sp-s_true)/np.maximum(np.abs(s_true),1e-30)), ' max |cos err|',np.abs(cp-c_true).max())
print('sin2 err',np.abs(sp.astype(np.float64)**2-s_true**2).max(),'sc err',np.abs(sp.astype(np.float64)*cp-s_true*c_true).max())
for name,arr in (('sin',cs32),('cos',cc32)):
    print(name,[repr(float(x)) for x in arr])
# compare: current scheme accuracy on [-pi/4,pi/4]
r=np.linspace(-np.pi/4,np.pi/4,1000001).astype(np.float32); r2=(r*r).astype(np.float32)
sp=fma(fma(fma(f32(-1.9515295891e-4),r2,f32(8.3321608736e-3)),r2,f32(-1.6666654611e-1)),(r2*r).astype(np.float32),r)
cp=fma(fma(fma(f32(2.443315711809948e-5),r2,f32(-1.388731625493765e-3)),r2,f32(4.166664568298827e-2)),(r2*r2).astype(np.float32),fma(f32(-0.5),r2,f32(1)))
print('current polys on [-pi/4,pi/4]: sin err',np.abs(sp-np.sin(r.astype(np.float64))).max(),'cos err',np.abs(cp-np.cos(r.astype(np.float64))).max())
