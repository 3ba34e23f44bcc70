"""Bjontegaard metrics (reference: BD-rate.py:17-86): BD-PSNR and BD-rate from two RD curves, cubic fit in log-rate (or
piecewise-cubic interpolation with `piecewise=1`).  Pure numpy/scipy host code -- it consumes 4-6 (rate, PSNR) points."""
import numpy as np
import scipy.interpolate

_trapz = getattr(np, "trapezoid", None) or np.trapz


def _integrals(x1, y1, x2, y2, piecewise):
    lo, hi = max(min(x1), min(x2)), min(max(x1), max(x2))
    if piecewise == 0:
        ints = []
        for x, y in ((x1, y1), (x2, y2)):
            P = np.polyint(np.polyfit(x, y, 3))
            ints.append(np.polyval(P, hi) - np.polyval(P, lo))
    else:
        samples, step = np.linspace(lo, hi, num=100, retstep=True)
        ints = []
        for x, y in ((x1, y1), (x2, y2)):
            o = np.argsort(x)
            ints.append(_trapz(scipy.interpolate.pchip_interpolate(np.asarray(x)[o], np.asarray(y)[o], samples), dx=step))
    return ints[0], ints[1], lo, hi


def BD_PSNR(R1, PSNR1, R2, PSNR2, piecewise=0):
    i1, i2, lo, hi = _integrals(np.log(R1), np.asarray(PSNR1), np.log(R2), np.asarray(PSNR2), piecewise)
    return (i2 - i1) / (hi - lo)


def BD_RATE(R1, PSNR1, R2, PSNR2, piecewise=0):
    i1, i2, lo, hi = _integrals(np.asarray(PSNR1), np.log(R1), np.asarray(PSNR2), np.log(R2), piecewise)
    return (np.exp((i2 - i1) / (hi - lo)) - 1) * 100
