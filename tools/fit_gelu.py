import numpy as np
from scipy.special import erfc, erf
from scipy.optimize import least_squares
z = np.linspace(0, 4.2, 20001)
Q = -np.log2(erfc(z))          # erfc(z) = 2^-Q(z)
x = z*np.sqrt(2)
def gelu_exact(x): return 0.5*x*(1+erf(x/np.sqrt(2)))
for deg in (4,5,6,7):
    # fit Q(z) = sum_{i=1..deg} c_i z^i minimizing weighted error: d(erfc) = erfc*ln2*dQ ; gelu err = 0.5*x*d(erfc)
    w = erfc(z)*np.log(2)*0.5*np.maximum(x, 0.05)/np.maximum(gelu_exact(x),1e-3)  # relative gelu error on the positive side
    w2 = erfc(z)*np.log(2)   # abs erf error
    V = np.stack([z**i for i in range(1,deg+1)],1)
    W = np.maximum(w, w2*0.3)
    c,*_ = np.linalg.lstsq(V*W[:,None], Q*W, rcond=None)
    # iterate towards minimax (Lawson)
    lw = np.ones_like(z)
    for it in range(60):
        c,*_ = np.linalg.lstsq(V*(W*lw)[:,None], Q*W*lw, rcond=None)
        r = np.abs((V@c-Q)*W); lw = lw*(r/r.max()+1e-3)**0.5; lw/=lw.max()
    c32 = c.astype(np.float32)
    def gelu_apx(xx):
        xx = xx.astype(np.float32); zz = np.abs(xx)*np.float32(0.70710678)
        zz = np.minimum(zz, np.float32(4.2))
        p = np.zeros_like(zz)
        for ci in c32[::-1]: p = p*zz + ci
        p = p*zz
        E = np.exp2(-p).astype(np.float32)
        r_ = np.float32(0.5)*xx*E
        return np.where(xx>=0, xx-r_, r_)
    xs = np.linspace(-8,8,400001)
    ga = gelu_apx(xs).astype(np.float64); ge = gelu_exact(xs)
    abs_err = np.abs(ga-ge)
    rel = abs_err/np.maximum(np.abs(ge),1e-30)
    m = np.abs(xs)>1e-3
    print(deg, 'max abs', abs_err.max(), 'max rel (|gelu|>1e-4)', rel[np.abs(ge)>1e-4].max(), 'erf abs err', np.abs((2**-(V@c))-erfc(z)).max())
    print('   coeffs', ', '.join(f'{v:.9g}' for v in c32))
