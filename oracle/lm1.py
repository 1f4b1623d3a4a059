"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

MINPACK `lmder` restated for ONE unknown, step by step (qrfac, lmpar with qrsolv, the trust-region update and
the eight stopping tests), as an independent second statement of what `scipy.optimize.least_squares(method='lm')`
does inside the reference's `optimize_z_offset_by_bones_single` (reference
src/model/bone_length_based_backproj.py:38-62).  The reference's Jacobian is not the derivative of its residual
(`(z*c+d)/len` instead of `(z*c+d/2)/len`, :55-56), so the value MINPACK returns depends on its step-acceptance
history; a solver that merely converges would not reproduce it.  scipy is the pinned dependency (1.15.3 here;
`x_scale=1` -> mode 2 with diag = 1, ftol = xtol = gtol = 1e-8, factor = 100, maxfev = 100*n:
scipy/optimize/_lsq/least_squares.py call_minpack).  tests/test_heads.py checks this restatement against scipy
itself; the HIP solver (csrc/heads.hip) follows the same sequence.
"""
import numpy as np, math
EPSMCH = np.finfo(np.float64).eps
DWARF = np.finfo(np.float64).tiny

def enorm(v):
    # MINPACK enorm for mid-range components: sqrt of the sequential sum of squares
    s = 0.0
    for t in v:
        s += float(t) * float(t)
    return math.sqrt(s)

def lmder1(fn, jac, x0, ftol=1e-8, xtol=1e-8, gtol=1e-8, maxfev=100, factor=100.0, diag=1.0):
    """MINPACK lmder for ONE unknown, mode = 2 (diag given), restated step by step."""
    x = float(x0)
    fvec = np.asarray(fn(x), dtype=np.float64); nfev = 1; njev = 0
    fnorm = enorm(fvec)
    par = 0.0; it = 1; info = 0
    delta = xnorm = 0.0
    while True:
        fjac = np.asarray(jac(x), dtype=np.float64).copy(); njev += 1
        # qrfac, n = 1
        acnorm = enorm(fjac)
        ajnorm = acnorm
        if ajnorm != 0.0:
            if fjac[0] < 0.0:
                ajnorm = -ajnorm
            fjac = fjac / ajnorm
            fjac[0] += 1.0
        rdiag = -ajnorm
        if it == 1:
            xnorm = abs(diag * x)            # enorm of one element = sqrt(v*v) = |v|
            xnorm = math.sqrt((diag * x) * (diag * x))
            delta = factor * xnorm
            if delta == 0.0:
                delta = factor
        # (q^T) fvec, first component
        wa4 = fvec.copy()
        if fjac[0] != 0.0:
            s = 0.0
            for i in range(len(wa4)):
                s += fjac[i] * wa4[i]
            temp = -s / fjac[0]
            for i in range(len(wa4)):
                wa4[i] += fjac[i] * temp
        r = rdiag                           # fjac(1,1) = wa1(1)
        qtf = wa4[0]
        gnorm = 0.0
        if fnorm != 0.0 and acnorm != 0.0:
            s = r * (qtf / fnorm)
            gnorm = max(gnorm, abs(s / acnorm))
        if gnorm <= gtol:
            info = 4
            break
        while True:
            # ---- lmpar, n = 1 ----
            # Gauss-Newton direction
            if r == 0.0:
                wa1 = 0.0
            else:
                wa1 = qtf / r
            xg = wa1
            liter = 0
            wa2 = diag * xg
            dxnorm = math.sqrt(wa2 * wa2)
            fp = dxnorm - delta
            p = xg
            if fp <= 0.1 * delta:
                par_out = 0.0
            else:
                parl = 0.0
                if r != 0.0:
                    w = diag * (wa2 / dxnorm)
                    w = w / r
                    temp = math.sqrt(w * w)
                    parl = ((fp / delta) / temp) / temp
                s = r * qtf
                w = s / diag
                gn = math.sqrt(w * w)
                paru = gn / delta
                if paru == 0.0:
                    paru = DWARF / min(delta, 0.1)
                par_l = max(par, parl)
                par_l = min(par_l, paru)
                if par_l == 0.0:
                    par_l = gn / dxnorm
                while True:
                    liter += 1
                    if par_l == 0.0:
                        par_l = max(DWARF, 0.001 * paru)
                    temp = math.sqrt(par_l)
                    sd = temp * diag          # wa1 = sqrt(par)*diag
                    # qrsolv n = 1
                    rr = r; wa = qtf; qtbpj = 0.0
                    if sd != 0.0:
                        if abs(rr) < abs(sd):
                            cotan = rr / sd
                            sin = 0.5 / math.sqrt(0.25 + 0.25 * cotan * cotan)
                            cos = sin * cotan
                        else:
                            tan = sd / rr
                            cos = 0.5 / math.sqrt(0.25 + 0.25 * tan * tan)
                            sin = cos * tan
                        rr = cos * rr + sin * sd
                        temp2 = cos * wa + sin * qtbpj
                        qtbpj = -sin * wa + cos * qtbpj
                        wa = temp2
                    sdiag = rr
                    p = wa / sdiag if sdiag != 0.0 else 0.0
                    wa2 = diag * p
                    dxnorm = math.sqrt(wa2 * wa2)
                    temp = fp
                    fp = dxnorm - delta
                    if abs(fp) <= 0.1 * delta or (parl == 0.0 and fp <= temp and temp < 0.0) or liter == 10:
                        break
                    w = diag * (wa2 / dxnorm)
                    w = w / sdiag
                    temp = math.sqrt(w * w)
                    parc = ((fp / delta) / temp) / temp
                    if fp > 0.0:
                        parl = max(parl, par_l)
                    if fp < 0.0:
                        paru = min(paru, par_l)
                    par_l = max(parl, par_l + parc)
                par_out = par_l
                if liter == 0:
                    par_out = 0.0
            par = par_out
            # ---- back in lmder ----
            wa1 = -p
            x2 = x + wa1
            wa3 = diag * wa1
            pnorm = math.sqrt(wa3 * wa3)
            if it == 1:
                delta = min(delta, pnorm)
            f2 = np.asarray(fn(x2), dtype=np.float64); nfev += 1
            fnorm1 = enorm(f2)
            actred = -1.0
            if 0.1 * fnorm1 < fnorm:
                actred = 1.0 - (fnorm1 / fnorm) ** 2
            w3 = r * wa1
            temp1 = math.sqrt(w3 * w3) / fnorm
            temp2 = (math.sqrt(par) * pnorm) / fnorm
            prered = temp1 ** 2 + temp2 ** 2 / 0.5
            dirder = -(temp1 ** 2 + temp2 ** 2)
            ratio = 0.0
            if prered != 0.0:
                ratio = actred / prered
            if ratio <= 0.25:
                if actred >= 0.0:
                    temp = 0.5
                else:
                    temp = 0.5 * dirder / (dirder + 0.5 * actred)
                if 0.1 * fnorm1 >= fnorm or temp < 0.1:
                    temp = 0.1
                delta = temp * min(delta, pnorm / 0.1)
                par = par / temp
            else:
                if par == 0.0 or ratio >= 0.75:
                    delta = pnorm / 0.5
                    par = 0.5 * par
            if ratio >= 1e-4:
                x = x2
                fvec = f2
                xnorm = math.sqrt((diag * x) * (diag * x))
                fnorm = fnorm1
                it += 1
            if abs(actred) <= ftol and prered <= ftol and 0.5 * ratio <= 1.0:
                info = 1
            if delta <= xtol * xnorm:
                info = 2
            if abs(actred) <= ftol and prered <= ftol and 0.5 * ratio <= 1.0 and info == 2:
                info = 3
            if info != 0:
                break
            if nfev >= maxfev:
                info = 5
            if abs(actred) <= EPSMCH and prered <= EPSMCH and 0.5 * ratio <= 1.0:
                info = 6
            if delta <= EPSMCH * xnorm:
                info = 7
            if gnorm <= EPSMCH:
                info = 8
            if info != 0:
                break
            if ratio >= 1e-4:
                break
        if info != 0:
            break
    return x, info, nfev, njev
