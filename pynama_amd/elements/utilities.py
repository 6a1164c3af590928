"""1-D quadrature rules and tensor Gauss-point lists.

Mirrors the reference module ``src/elements/utilities.py`` (names, signatures, return types):
``gaussPoints`` :43-61, ``lobattoPoints`` :63-92, ``generateGaussPoints2D/3D`` :15-41.
Set-up only (host, numpy); the tables built from these rules are uploaded once to the GPU.
"""
from collections import namedtuple

import numpy as np


class GaussPoint2D(namedtuple('GaussPoint', ['r', 's', 'w'])):
    __slots__ = ()


class GaussPoint3D(namedtuple('GaussPoint', ['r', 's', 't', 'w'])):
    __slots__ = ()


def _legendre(n, x):
    """P_n(x) and P_{n-1}(x) by the three-term recurrence."""
    p0, p1 = np.ones_like(x), x.copy()
    if n == 0:
        return p0, np.zeros_like(x)
    for k in range(2, n + 1):
        p0, p1 = p1, ((2 * k - 1) * x * p1 - (k - 1) * p0) / k
    return p1, p0


def gaussPoints(N):
    """Gauss-Legendre nodes/weights on [-1, 1] (ascending), symmetrised like the reference."""
    k = np.arange(1, N + 1)
    x = -np.cos(np.pi * (k - 0.25) / (N + 0.5))          # Tricomi initial guess
    for _ in range(100):
        pn, pm = _legendre(N, x)
        dpn = N * (x * pn - pm) / (x * x - 1.0)
        dx = pn / dpn
        x = x - dx
        if np.max(np.abs(dx)) < 1e-16:
            break
    pn, pm = _legendre(N, x)
    dpn = N * (x * pn - pm) / (x * x - 1.0)
    w = 2.0 / ((1.0 - x * x) * dpn * dpn)
    x = (x - x[::-1]) / 2
    w = (w + w[::-1]) / 2
    return (x, w)


def lobattoPoints(N):
    """Gauss-Lobatto-Legendre nodes/weights on [-1, 1] (ascending): roots of (1-x^2) P'_{N-1}."""
    n = N - 1
    x = -np.cos(np.pi * np.arange(N) / n)
    xi = x[1:-1].copy()
    for _ in range(100):
        pn, pm = _legendre(n, xi)
        dp = n * (xi * pn - pm) / (xi * xi - 1.0)             # P'_n
        d2p = (2.0 * xi * dp - n * (n + 1) * pn) / (1.0 - xi * xi)   # P''_n from Legendre's ODE
        dx = dp / d2p
        xi = xi - dx
        if xi.size == 0 or np.max(np.abs(dx)) < 1e-16:
            break
    x[1:-1] = xi
    pn, _ = _legendre(n, x)
    w = 2.0 / (n * (n + 1) * pn * pn)
    x = (x - x[::-1]) / 2
    w = (w + w[::-1]) / 2
    return (x, w)


def generateGaussPoints2D(gps1D, gpsWei):
    return [GaussPoint2D(r=gps1D[a], s=gps1D[b], w=gpsWei[a] * gpsWei[b])
            for a in range(len(gps1D)) for b in range(len(gps1D))]


def generateGaussPoints3D(gps1D, gpsWei):
    n = range(len(gps1D))
    return [GaussPoint3D(r=gps1D[a], s=gps1D[b], t=gps1D[c], w=gpsWei[a] * gpsWei[b] * gpsWei[c])
            for a in n for b in n for c in n]
