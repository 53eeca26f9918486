import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as Po
np.set_printoptions(precision=17)

def cheb_fit(f, a, b, deg, n=4000):
    # least-squares on Chebyshev nodes ~ near minimax
    k = np.arange(n); x = np.cos(np.pi*(k+0.5)/n); s = 0.5*(b-a)*x + 0.5*(b+a)
    c = Ch.chebfit(x, f(s), deg)
    # convert to monomial in s
    p = Ch.cheb2poly(c)  # poly in x
    # x = (2s - (a+b))/(b-a)
    P = np.poly1d(p[::-1]); lin = np.poly1d([2/(b-a), -(a+b)/(b-a)])
    Q = P(lin)
    return Q.coeffs[::-1]  # ascending in s

def f32_horner(coefs, s):
    s = s.astype(np.float32); r = np.float32(coefs[-1])*np.ones_like(s)
    for c in coefs[-2::-1]:
        r = (r.astype(np.float64)*s.astype(np.float64) + np.float64(np.float32(c))).astype(np.float32)  # emulate fma (single rounding)
    return r

# atan: atan(t) = t * P(t^2), t in [0,1]
fa = lambda s: np.where(s>0, np.arctan(np.sqrt(s))/np.sqrt(np.maximum(s,1e-300)), 1.0)
for deg in (7,8,9,10):
    c = cheb_fit(fa, 0.0, 1.0, deg)
    t = np.linspace(0,1,2000001); s=(t.astype(np.float32)*t.astype(np.float32))
    r = f32_horner(c, s); at = (r.astype(np.float64)*t.astype(np.float32)).astype(np.float32)
    err = np.abs(at.astype(np.float64)-np.arctan(t.astype(np.float32).astype(np.float64)))
    print("atan deg",deg,"max abs err %.3g"%err.max(), "max ulp-ish rel %.3g"%np.max(err[1:]/np.arctan(t[1:])))
c_atan = cheb_fit(fa,0,1,9); print("atan coefs", [float(np.float32(x)) for x in c_atan])

# asin: for |x|<=0.5: asin(x) = x + x*z*P(z), z=x^2 ; P(z) = (asin(x)/x - 1)/z  on z in [0,0.25]
def fs(z):
    x=np.sqrt(np.maximum(z,1e-300)); 
    return np.where(z>1e-12,(np.arcsin(x)/x-1)/np.maximum(z,1e-300), 1/6.0)
for deg in (3,4,5):
    c = cheb_fit(fs,0.0,0.25,deg)
    x=np.linspace(0,0.5,1000001).astype(np.float32); z=(x*x)
    p=f32_horner(c,z); r=(x.astype(np.float64)+x.astype(np.float64)*z*p).astype(np.float32)
    err=np.abs(r.astype(np.float64)-np.arcsin(x.astype(np.float64)))
    # upper branch: x in (0.5,1]: z=(1-x)/2, s=sqrt(z), asin = pi/2 - 2*(s + s*z*P(z))
    xu=np.linspace(0.5,1,1000001).astype(np.float32); zu=((1-xu)*np.float32(0.5)).astype(np.float32); su=np.sqrt(zu).astype(np.float32)
    pu=f32_horner(c,zu); ru=(np.pi/2-2*(su.astype(np.float64)+su.astype(np.float64)*zu*pu)).astype(np.float32)
    erru=np.abs(ru.astype(np.float64)-np.arcsin(xu.astype(np.float64)))
    print("asin deg",deg,"max abs err lo %.3g hi %.3g"%(err.max(),erru.max()))
c_asin=cheb_fit(fs,0,0.25,4); print("asin coefs",[float(np.float32(x)) for x in c_asin])

# sin/cos on [-pi/4,pi/4]: sin(r)= r + r*z*S(z); cos(r)=1 - z/2 + z*z*C(z), z=r^2
def fsn(z):
    r=np.sqrt(np.maximum(z,1e-300)); return np.where(z>1e-12,(np.sin(r)/r-1)/np.maximum(z,1e-300),-1/6.0)
def fcs(z):
    r=np.sqrt(np.maximum(z,1e-300)); return np.where(z>1e-8,(np.cos(r)-1+z/2)/np.maximum(z*z,1e-300),1/24.0)
zmax=(np.pi/4)**2*1.02
for deg in (2,3):
    cs=cheb_fit(fsn,0,zmax,deg); cc=cheb_fit(fcs,0,zmax,deg)
    r=np.linspace(-np.pi/4,np.pi/4,2000001).astype(np.float32); z=r*r
    sn=(r.astype(np.float64)+r.astype(np.float64)*z*f32_horner(cs,z)).astype(np.float32)
    cn=(1-0.5*z.astype(np.float64)+z.astype(np.float64)*z*f32_horner(cc,z)).astype(np.float32)
    print("sincos deg",deg,"sin err %.3g cos err %.3g"%(np.abs(sn-np.sin(r.astype(np.float64))).max(),np.abs(cn-np.cos(r.astype(np.float64))).max()))
cs=cheb_fit(fsn,0,zmax,3); cc=cheb_fit(fcs,0,zmax,3)
print("sin coefs",[float(np.float32(x)) for x in cs]); print("cos coefs",[float(np.float32(x)) for x in cc])

print("---- refit sin/cos with series-defined targets")
fsn2=lambda z: -1/6 + z/120 - z**2/5040 + z**3/362880 - z**4/39916800 + z**5/6227020800
fcs2=lambda z: 1/24 - z/720 + z**2/40320 - z**3/3628800 + z**4/479001600 - z**5/87178291200
for deg in (2,3):
    cs=cheb_fit(fsn2,0,zmax,deg); cc=cheb_fit(fcs2,0,zmax,deg)
    r=np.linspace(-np.pi/4*1.01,np.pi/4*1.01,2000001).astype(np.float32); z=r*r
    sn=(r.astype(np.float64)+r.astype(np.float64)*z*f32_horner(cs,z)).astype(np.float32)
    cn=((1-0.5*z.astype(np.float64)).astype(np.float32).astype(np.float64)+z.astype(np.float64)*z*f32_horner(cc,z)).astype(np.float32)
    print("sincos deg",deg,"sin err %.3g cos err %.3g"%(np.abs(sn-np.sin(r.astype(np.float64))).max(),np.abs(cn-np.cos(r.astype(np.float64))).max()))
    print(" sin coefs",[float(np.float32(x)) for x in cs]); print(" cos coefs",[float(np.float32(x)) for x in cc])
c8=cheb_fit(fa,0,1,8); print("atan8 coefs",[float(np.float32(x)) for x in c8])
# Cody-Waite constants for pi/2
import struct
def f32(x): return float(np.float32(x))
hi=f32(np.pi/2); hi_trunc=struct.unpack('f',struct.pack('I',struct.unpack('I',struct.pack('f',np.float32(np.pi/2)))[0]&0xFFFFF000))[0]
mid=np.pi/2-hi_trunc; mid_t=struct.unpack('f',struct.pack('I',struct.unpack('I',struct.pack('f',np.float32(mid)))[0]&0xFFFFF000))[0]
lo=np.pi/2-hi_trunc-mid_t
print("pio2 hi(trunc12) %.10e mid %.10e lo %.10e"%(hi_trunc,mid_t,f32(lo)))
