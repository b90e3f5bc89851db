import numpy as np
np.random.seed(0)
LD=np.longdouble
def truth(z, sl):
    z=LD(z); az=np.abs(z); a=LD(sl)/4*(az/3)**LD(-1.5)
    phi=np.arccos(a); return 2*np.sign(z)/3*az*(1+np.cos(2*LD(np.pi)*0+2*np.arccos(LD(-1))/3-2*phi/3))
def ref_double(z, sl):
    az=np.abs(z); a=sl/4*(az/3)**(-1.5); phi=np.arccos(a)
    return 2*np.sign(z)/3*az*(1+np.cos(2*np.pi/3-2*phi/3))
# fit P(e): delta = e*P(e), where 4 d^3 + 6 d^2 = m = e^2
def delta_exact(m):
    m=LD(m); a=1-m; w=np.cos(np.arccos(-a)/3); return w-LD(0.5)
es=np.linspace(1e-6,1,4001); ds=np.array([float(delta_exact(e*e)) for e in es])
for deg in (2,3,4,5):
    # chebyshev-ish least squares fit of ds/es in e
    c=np.polynomial.chebyshev.chebfit(2*es-1, ds/es, deg)
    P=np.polynomial.chebyshev.cheb2poly(c)  # in variable x=2e-1
    approx=np.polynomial.chebyshev.chebval(2*es-1,c)*es
    print(deg, 'max rel err', np.max(np.abs(approx-ds)/ds))
deg=3
c=np.polynomial.chebyshev.chebfit(2*es-1, ds/es, deg)
# convert to monomial in e
px=np.polynomial.chebyshev.cheb2poly(c) # coefficients in x
# x = 2e-1 -> polynomial in e
pe=np.polynomial.polynomial.Polynomial(px)(np.polynomial.polynomial.Polynomial([-1,2]))
coef=pe.coef
print('coef (monomial in e, low->high):', [repr(float(v)) for v in coef])
def mine(z, sl, iters=3, f32rcp=True):
    az=np.abs(z); t=az*(1.0/3.0); r=1/np.sqrt(t); a=(sl/4)*r*r*r
    m=np.maximum(1-a,0.0); e=np.sqrt(m.astype(np.float32)).astype(np.float64)
    d=e*np.polynomial.polynomial.polyval(e,coef)
    for it in range(iters):
        g=(4*d+6)*d*d-m
        gp=12*d*(d+1)
        inv=(1/gp.astype(np.float32)).astype(np.float64) if f32rcp else 1/gp
        inv=np.where(gp>0,inv,0.0)
        d=d-g*inv
    w=0.5+d
    return np.sign(z)*4*t*w*w
sl=1.0
p=54**(1/3)*(2*sl)**(2/3)/4
z=np.concatenate([np.random.uniform(p,6,2000000), p*(1+np.random.uniform(0,1e-4,200000))])
tr=truth(z,sl)
for name,v in (('ref double',ref_double(z,sl)),('mine 2it',mine(z,sl,2)),('mine 3it',mine(z,sl,3)),('mine 4it',mine(z,sl,4))):
    err=np.abs((LD(v)-tr)/np.abs(LD(z))).astype(float)
    print(name,'max err/|z| %.3e'%err.max(),' 99.99pct %.3e'%np.quantile(err,0.9999), 'mean %.2e'%err.mean())
d=np.abs(mine(z,sl,3)-ref_double(z,sl))/np.abs(z)
print('mine3 vs ref double: max %.3e'%d.max(), 'count>1e-12:', (d>1e-12).sum(), 'count>1e-13', (d>1e-13).sum())
a=sl/4*(np.abs(z)/3)**(-1.5)
big=d>1e-13
print('1-a of those:', np.sort(1-a[big])[:10], np.sort(1-a[big])[-5:] if big.any() else None)
