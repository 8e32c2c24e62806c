import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
from modppl_amd import capi
import math
L = capi.load()
dp = C.POINTER(C.c_double)
rng = np.random.default_rng(5)
def fma(a,b,c): return math.fma(a,b,c) if hasattr(math,'fma') else None
# python 3.10 has no math.fma: emulate with fractions
from fractions import Fraction
def fma(a,b,c):
    return float(Fraction(a)*Fraction(b)+Fraction(c))
res = {}
for trial in range(20):
    A = rng.normal(size=(16,4)) * np.exp(rng.normal(size=(16,4))*3)
    B = rng.normal(size=(4,16)) * np.exp(rng.normal(size=(4,16))*3)
    Cm = rng.normal(size=(16,16)) * np.exp(rng.normal(size=(16,16))*3)
    D = np.zeros((16,16))
    rc = L.mp_probe_mfma_f64(A.ctypes.data_as(dp), B.ctypes.data_as(dp), Cm.ctypes.data_as(dp), D.ctypes.data_as(dp), 0)
    assert rc == 0
    cands = {}
    def chain(order, fused=True):
        out = np.zeros((16,16))
        for i in range(16):
            for j in range(16):
                acc = Cm[i,j]
                for k in order:
                    acc = fma(A[i,k], B[k,j], acc) if fused else acc + A[i,k]*B[k,j]
                out[i,j] = acc
        return out
    cands['fma k=0,1,2,3 from C'] = chain([0,1,2,3])
    cands['fma k=3,2,1,0 from C'] = chain([3,2,1,0])
    cands['mul+add k asc'] = chain([0,1,2,3], False)
    ex = np.zeros((16,16))
    for i in range(16):
        for j in range(16):
            ex[i,j] = float(sum(Fraction(A[i,k])*Fraction(B[k,j]) for k in range(4)) + Fraction(Cm[i,j]))
    cands['exact sum, one rounding'] = ex
    # products summed exactly then added to C with one rounding? same as exact. pairwise: (p0+p1)+(p2+p3)+C variants
    pw = np.zeros((16,16))
    for i in range(16):
        for j in range(16):
            s01 = fma(A[i,0],B[0,j], A[i,1]*B[1,j])
            pw[i,j] = 0
    for name, v in cands.items():
        res[name] = res.get(name, 0) + int((v != D).sum())
print({k: v for k, v in res.items()}, "of", 20*256)
