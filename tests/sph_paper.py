"""An INDEPENDENT statement of the SPH equations the hydro path integrates, written from the publications and from physics - not from
hydra.c / density.c - and evaluated by brute force over all pairs with numpy.  Test infrastructure (tests/test_hydro_physics.py).

Why it exists: the reference holds no test vector for hydro_force() and hydra.c cannot be built in this image, so oracle/sph_oracle.c
(a line-by-line restatement) and csrc/sph.hip could share a misreading.  The expressions below are derived from

  [SH02]  Springel & Hernquist 2002, MNRAS 333, 649: the entropy formulation with variable smoothing lengths (grad-h factors)
  [S05]   Springel 2005, MNRAS 364, 1105 (the GADGET-2 paper): eqs (5)-(10) equations of motion, (13)-(14) signal-velocity
          viscosity, (17) Balsara limiter, section 3.1 comoving variables
  [H13]   Hopkins 2013, MNRAS 428, 2840: the pressure-entropy formulation, eqs (17)-(21) with the smoothing length tied to the
          mass density
  [P12]   Price 2012, JCP 231, 759 (arXiv:1012.1885): the B-spline kernels M4 / M5 / M6 and their 3-d normalisations

and agree with the reference's conventions only where a convention is a free choice (listed at the places concerned).

Variables (S05 section 3.1, as MP-Gadget stores them): comoving positions x, "velocities" u = a^2 dx/dt, comoving densities rho,
entropic function A with physical pressure P_phys = A rho_phys^gamma.  With r_phys = a x, rho_phys = rho a^-3, peculiar velocity
u / a and the Hubble flow H r_phys:

  * pressure force.  du/dt = a (d(a xdot)/dt + H a xdot) = -a grad_r P_phys / rho_phys = a^(-3(gamma-1)) x [the Newtonian SPH sum in
    comoving variables with P = A rho^gamma].  The power of a is carried by the kick factor (timestep.c), so HydroAccel IS the
    comoving Newtonian sum.
  * viscosity (S05 eqs 9, 13, 14 in physical variables):  w_phys = v_ij . r_ij / |r_ij| = (u_ij . x_ij + a^2 H x_ij^2) / (a |x_ij|),
    c_phys = c a^(-3(gamma-1)/2) with c^2 = gamma P / rho in comoving variables.  Writing mu = a^(3(gamma-1)/2) w_phys,
        Pi_phys = -(alpha/2) w_phys (c_i + c_j - 3 w)_phys / rho_phys,ij = a^(3 - 3(gamma-1)) x [ -(alpha/2) mu (c_i + c_j - 3 mu) / rho_ij ],
    and grad_r W_phys = a^-4 grad_x W, so du/dt|visc = a^(-3(gamma-1)) x [ -sum_j m_j Pi_ij grad_x Wbar_ij ] with the bracketed
    (comoving) Pi: the same kick factor as the pressure force.  mu = [a^(3(gamma-1)/2) / a] (u_ij . x_ij + a^2 H x_ij^2) / |x_ij|.
  * entropy (S05 eq 10):  dA/dt = (gamma-1)/2 rho_phys^(1-gamma) sum_j m_j Pi_phys v_ij . grad_r Wbar_phys.  Substituting the above,
        dA/dln a = (1/H) dA/dt = (gamma-1) / (a^2 H rho^(gamma-1)) x sum_j (1/2) m_j Pi_ij (u_ij . x_ij + a^2 H x_ij^2) / |x_ij| x Wbar'_ij.
  * Balsara limiter (S05 eq 17):  f = |div v| / (|div v| + |curl v| + 0.0001 c / h) in physical variables.  The peculiar velocity
    gradient is a^-2 times the gradient of u with respect to x, c_phys / h_phys = c a^(-3(gamma-1)/2) / (a h): multiplying through
    by a^2 leaves f = |div u| / (|div u| + |curl u| + 0.0001 (c / h) / [a^(3(gamma-1)/2) / a]).
  * the bound on the viscous force (GADGET-2 code, not in the paper: "make sure that viscous acceleration is not too large"): over
    a step dln a = 2 max(dlna_i, dlna_j) the mutual viscous deceleration of a pair, (m_i + m_j) Pi |Wbar'| a^(-3(gamma-1)) dt with
    dt = dln a / H, must not exceed half of the approach velocity |u_ij . x_ij + a^2 H x^2| / |x|:
        Pi <= (1/2) [H a^(3(gamma-1))] |w| / ((1/2)(m_i + m_j) |W'_i + W'_j| |x| dln a).

Smoothing lengths: MP-Gadget's Hsml is the SUPPORT RADIUS H of the kernel (W = 0 for r >= H); P12's h = H / support.
Free conventions taken from the reference: the neighbour set of the force loop (r < H_i or r < H_j, S05 eq 7 lets each term vanish
outside its own support, so this is the full sum); MaxSignalVel = max over neighbours of c_i + c_j (- 3 mu when approaching),
starting from c_i; the pressure-entropy form takes c^2 = gamma P / (y / A^(1/gamma)) and the mean MASS density in Pi_ij."""
import numpy as np

GAMMA = 5.0 / 3.0

# P12 eqs (6)-(8): w(q) without the normalisation sigma, q = r / h, support radius R h, 3-d sigma
_KERNELS = {
    1: (2.0, 1.0 / np.pi),            # M4 cubic spline
    2: (3.0, 1.0 / (120.0 * np.pi)),  # M6 quintic
    4: (2.5, 1.0 / (20.0 * np.pi)),   # M5 quartic
}


def _w_dw(kernel, q):
    """w(q), dw/dq of P12's B-splines"""
    pw = lambda x, n: np.where(x > 0, x, 0.0) ** n
    if kernel == 1:
        return 0.25 * pw(2 - q, 3) - pw(1 - q, 3), -0.75 * pw(2 - q, 2) + 3 * pw(1 - q, 2)
    if kernel == 2:
        return (pw(3 - q, 5) - 6 * pw(2 - q, 5) + 15 * pw(1 - q, 5),
                -5 * pw(3 - q, 4) + 30 * pw(2 - q, 4) - 75 * pw(1 - q, 4))
    if kernel == 4:
        return (pw(2.5 - q, 4) - 5 * pw(1.5 - q, 4) + 10 * pw(0.5 - q, 4),
                -4 * pw(2.5 - q, 3) + 20 * pw(1.5 - q, 3) - 40 * pw(0.5 - q, 3))
    raise ValueError(kernel)


def kernel_W(kernel, r, H):
    """W(r, H), dW/dr, dW/dH for support radius H (arrays broadcast)"""
    R, sigma = _KERNELS[kernel]
    h = H / R
    q = r / h
    w, dw = _w_dw(kernel, q)
    W = sigma / h ** 3 * w
    dWdr = sigma / h ** 4 * dw
    # dW/dH = (1/R) dW/dh,  dW/dh = -(3 W + q h dW/dr) / h
    dWdH = -(3.0 * W + r * dWdr) / H
    return W, dWdr, dWdH


def _pairs(pos, box):
    pos = np.asarray(pos, float)
    d = pos[:, None, :] - pos[None, :, :]
    d -= box * np.rint(d / box)                 # x_ij = x_i - x_j on the nearest image
    r = np.sqrt((d ** 2).sum(-1))
    return d, r


def paper_density(pos, mass, vel, A, H, box, kernel=1, formulation="density"):
    """The density loop for every particle as target (S05 eq 5, SH02 eqs 27-28, H13 eq 19), all pairs.  vel / A are the velocities and
    entropic functions the SUMS see - the predicted ones when particles carry pending kicks.  Returns a dict of per-particle fields:
    density, dhsml (SH02's f), divvel, curlvel, and for the pressure-entropy form y, dy_dH, egywtdensity, dhsmlegy."""
    pos = np.asarray(pos, float)
    N = len(pos)
    m = np.asarray(mass, float)
    u = np.asarray(vel, float)
    A = np.asarray(A, float)
    H = np.asarray(H, float)
    d, r = _pairs(pos, box)
    du = u[:, None, :] - u[None, :, :]          # u_ij
    offd = ~np.eye(N, dtype=bool)
    rs = np.where(offd, r, 1.0)
    Wi, dWi, dWHi = kernel_W(kernel, r, H[:, None])        # kernels of i's support, [i, j]
    inside = r < H[:, None]
    Wi, dWi, dWHi = Wi * inside, dWi * inside, dWHi * inside
    rho = (m[None, :] * Wi).sum(1)
    drho_dH = (m[None, :] * dWHi).sum(1)
    f_grad = 1.0 / (1.0 + H / (3.0 * rho) * drho_dH)       # SH02: f_i = [1 + (h_i / 3 rho_i) d rho_i / d h_i]^-1
    gradW_i = (dWi / rs)[:, :, None] * d * offd[:, :, None]          # grad_i W_ij(H_i)
    divu = -(m[None, :] * (du * gradW_i).sum(-1)).sum(1) / rho      # (1/rho_i) sum_j m_j (u_j - u_i) . grad_i W_ij
    curl = np.cross(du, gradW_i)
    curlu = np.linalg.norm((m[None, :, None] * curl).sum(1), axis=1) / rho
    out = dict(density=rho, dhsml=f_grad, divvel=divu, curlvel=curlu)
    if formulation != "density":
        # H13 eq 19: y_i = Pbar_i^(1/gamma) = sum_j m_j A_j^(1/gamma) W_ij(h_i)
        a_g = A ** (1.0 / GAMMA)
        y = (m[None, :] * a_g[None, :] * Wi).sum(1)
        dy_dH = (m[None, :] * a_g[None, :] * dWHi).sum(1)
        out["y"] = y
        out["egywtdensity"] = y / a_g                       # "energy weighted density" (the reference's EgyWtDensity)
        # H13 eq 18 with the smoothing length tied to rho (x~_j = m_j, y~ = rho):
        # f_ij - 1 = -(1 / A_j^(1/gamma)) (h_i / 3 rho_i) (d y_i / d h_i) [1 + (h_i / 3 rho_i) d rho_i / d h_i]^-1
        out["dhsmlegy"] = -(dy_dH * H / (3.0 * y)) * f_grad   # the reference stores this combination (times y/rho it is the bracket)
    return out


def paper_hydro(pos, mass, vel, A, H, box, F, atime=1.0, hubble=0.0, alpha=0.75, kernel=1, formulation="density", dlna=None,
                contrast_limit=None):
    """The hydro force of every particle as target from per-particle FIELDS F (what paper_density returns, or the stored / predicted values
    of particles that were not active in the density loop): density, dhsml, divvel, curlvel and, pressure-entropy, egywtdensity and dhsmlegy.
    vel / A as in paper_density.  contrast_limit (pressure-entropy; a limiter of the code, not of H13): the grad-h correction of a particle is
    scaled down where its y / (A^(1/gamma) rho) exceeds the limit (None: no limit)."""
    pos = np.asarray(pos, float)
    N = len(pos)
    m = np.asarray(mass, float)
    u = np.asarray(vel, float)
    A = np.asarray(A, float)
    H = np.asarray(H, float)
    d, r = _pairs(pos, box)
    du = u[:, None, :] - u[None, :, :]
    offd = ~np.eye(N, dtype=bool)
    rs = np.where(offd, r, 1.0)
    rho, f_grad, divu, curlu = (np.asarray(F[k], float) for k in ("density", "dhsml", "divvel", "curlvel"))
    _, dWi, _ = kernel_W(kernel, r, H[:, None])
    dWi = dWi * (r < H[:, None])
    out = {}
    if formulation == "density":
        P = A * rho ** GAMMA
        eom = rho                                   # the density of the equations of motion
        c = np.sqrt(GAMMA * P / rho)
    else:
        a_g = A ** (1.0 / GAMMA)
        eom = np.asarray(F["egywtdensity"], float)
        y = eom * a_g
        P = y ** GAMMA
        c = np.sqrt(GAMMA * P / eom)
    fac_mu = atime ** (3 * (GAMMA - 1) / 2) / atime
    hubble_a2 = hubble * atime ** 2

    Wj, dWj, _ = kernel_W(kernel, r, H[None, :])             # kernels of j's support, [i, j]
    dWj = dWj * (r < H[None, :])
    pair = offd & ((r < H[:, None]) | (r < H[None, :]))
    dWi_f = dWi                                              # W'_ij(H_i) (zero outside its own support)
    # pressure terms
    if formulation == "density":
        # S05 eq 7: -sum_j m_j [ f_i P_i/rho_i^2 grad_i W_ij(h_i) + f_j P_j/rho_j^2 grad_i W_ij(h_j) ]
        ti = (f_grad * P / rho ** 2)[:, None] * dWi_f
        tj = (f_grad * P / rho ** 2)[None, :] * dWj
    else:
        # H13 eq 21: -sum_j m_j (A_i A_j)^(1/gamma) [ f_ij Pbar_i^(1-2/gamma) grad_i W_ij(h_i) + f_ji Pbar_j^(1-2/gamma) grad_i W_ij(h_j) ]
        # with f_ij - 1 = -(1 / A_j^(1/gamma)) (h_i / 3 rho_i) (dy_i / dh_i) f_i  =  dhsmlegy_i (y_i / rho_i) / A_j^(1/gamma)
        corr = -np.asarray(F["dhsmlegy"], float) * y / rho           # (h_i / 3 rho_i) (dy_i/dh_i) [...]^-1
        if contrast_limit is not None:
            rr = eom / rho
            corr = corr * np.minimum(rr, contrast_limit) / rr
        f_ij = 1.0 - corr[:, None] / a_g[None, :]
        f_ji = 1.0 - corr[None, :] / a_g[:, None]
        pw = P ** (1.0 - 2.0 / GAMMA)
        aa = a_g[:, None] * a_g[None, :]
        ti = aa * f_ij * pw[:, None] * dWi_f
        tj = aa * f_ji * pw[None, :] * dWj
    hfc = m[None, :] * (ti + tj) / rs
    # viscosity, S05 eqs 13-14 with the Balsara factors (eq 17)
    vdotr2 = (du * d).sum(-1) + hubble_a2 * r ** 2
    mu = fac_mu * vdotr2 / rs
    appr = pair & (vdotr2 < 0)
    fbal = np.abs(divu) / (np.abs(divu) + curlu + 0.0001 * c / H / fac_mu)
    vsig = c[:, None] + c[None, :] - 3.0 * mu
    rho_ij = 0.5 * (rho[:, None] + rho[None, :])
    Pi = np.where(appr, 0.5 * alpha * vsig * (-mu) / rho_ij * 0.5 * (fbal[:, None] + fbal[None, :]), 0.0)
    dWsum = dWi_f + dWj
    if dlna is not None:
        dl = 2.0 * np.maximum(np.asarray(dlna, float)[:, None], np.asarray(dlna, float)[None, :])
        with np.errstate(divide="ignore", invalid="ignore"):
            bound = 0.5 * (hubble * atime ** (3 * (GAMMA - 1))) * vdotr2 / (0.5 * (m[:, None] + m[None, :]) * dWsum * rs * dl)
        lim = appr & (dl > 0) & (dWsum < 0)
        Pi = np.where(lim, np.minimum(Pi, bound), Pi)
    hfc_visc = 0.5 * m[None, :] * Pi * dWsum / rs                     # m_j Pi_ij Wbar'_ij / r, Wbar' = (W'_i + W'_j) / 2
    hfc = np.where(pair, hfc + hfc_visc, 0.0)
    acc = -(hfc[:, :, None] * d).sum(1)
    with np.errstate(divide="ignore", invalid="ignore"):
        dA = (0.5 * np.where(pair, hfc_visc, 0.0) * vdotr2).sum(1) * (GAMMA - 1) / (hubble_a2 * rho ** (GAMMA - 1))
    sig = np.where(pair, c[:, None] + c[None, :], 0.0)
    sig = np.where(appr, np.maximum(sig, vsig), sig)
    out.update(hydroacc=acc, dtentropy=dA, maxsignalvel=np.maximum(c, sig.max(1)), pressure=P, soundspeed=c, balsara=fbal)
    return out


def sph_paper(pos, mass, vel, A, H, box, atime=1.0, hubble=0.0, alpha=0.75, kernel=1, formulation="density", dlna=None,
              active=None):
    """All pairs, periodic minimum image.  pos [N,3], mass [N], vel = u [N,3], A [N] entropic function, H [N] support radii.
    formulation: "density" (SH02 / S05) or "pressure" (H13).  dlna [N]: the particles' steps in ln a for the bound on the viscous
    force (None: bound off).  Returns the density-loop fields and the hydro force for every particle."""
    out = paper_density(pos, mass, vel, A, H, box, kernel, formulation)
    out.update(paper_hydro(pos, mass, vel, A, H, box, out, atime, hubble, alpha, kernel, formulation, dlna))
    return out
