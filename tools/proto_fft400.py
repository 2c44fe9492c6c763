"""Numpy prototype of the radix-20x20 (PFA 4x5 inside) two-real-frames-per-complex-FFT scheme
used by audio_amd/csrc/melspec400.h.  Validates index maps only."""
import numpy as np

def dft4(a, b, c, d):
    s0, s1 = a + c, a - c
    s2, s3 = b + d, b - d
    return s0 + s2, s1 - 1j * s3, s0 - s2, s1 + 1j * s3

C1, C2 = np.cos(2*np.pi/5), np.cos(4*np.pi/5)
S1, S2 = np.sin(2*np.pi/5), np.sin(4*np.pi/5)
def dft5(x0, x1, x2, x3, x4):
    t1, t2, t3, t4 = x1 + x4, x2 + x3, x1 - x4, x2 - x3
    m1 = x0 + C1*t1 + C2*t2
    m2 = x0 + C2*t1 + C1*t2
    u1 = S1*t3 + S2*t4
    u2 = S2*t3 - S1*t4
    return x0 + t1 + t2, m1 - 1j*u1, m2 - 1j*u2, m2 + 1j*u2, m1 + 1j*u1

def dft20(x):
    """x: list of 20 complex -> list of 20 (natural order in and out)."""
    t = [[None]*5 for _ in range(4)]
    for n2 in range(5):
        o = dft4(*[x[(5*n1 + 4*n2) % 20] for n1 in range(4)])
        for k1 in range(4):
            t[k1][n2] = o[k1]
    X = [None]*20
    for k1 in range(4):
        o = dft5(*t[k1])
        for k2 in range(5):
            X[(5*k1 + 16*k2) % 20] = o[k2]
    return X

rng = np.random.default_rng(0)
x = rng.standard_normal(20) + 1j*rng.standard_normal(20)
assert np.allclose(dft20(list(x)), np.fft.fft(x))

# 400 = 20 x 20, n = r + 20 q ; k = s + 20 u
xa, xb = rng.standard_normal(400), rng.standard_normal(400)
z = xa + 1j*xb
Y = np.zeros((20, 20), complex)      # [r][s]
for r in range(20):
    Y[r] = dft20([z[r + 20*q] for q in range(20)])
    Y[r] *= np.exp(-2j*np.pi*r*np.arange(20)/400)
Z = np.zeros(400, complex)
for s in range(20):
    o = dft20([Y[r][s] for r in range(20)])
    for u in range(20):
        Z[s + 20*u] = o[u]
assert np.allclose(Z, np.fft.fft(z))
# unpack with the partner-lane rule
Zl = Z.reshape(20, 20).T             # Zl[s][u] = Z[s+20u]
A = np.fft.rfft(xa); B = np.fft.rfft(xb)
for s in range(20):
    partner = (20 - s) % 20
    G = Zl[partner]
    us = range(11) if s == 0 else range(10)
    for u in us:
        k = s + 20*u
        zk = Zl[s][u]
        if s == 0:
            zc = Zl[0][0] if u == 0 else G[20 - u]
        else:
            zc = G[19 - u]
        zc = np.conj(zc)
        a = (zk + zc)/2; b = (zk - zc)/(2j)
        assert np.allclose(a, A[k]) and np.allclose(b, B[k]), (s, u)
print("proto OK")
