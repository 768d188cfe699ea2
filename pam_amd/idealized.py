"""Host-side (numpy) generators of the idealised coupler inputs the reference's standalone driver uses.

These build *inputs* (the coupler fields `density_dry, uvel, vvel, wvel, temp` + tracers and the vertical
grid) for tests and for bench.py's synthetic workloads; nothing here is on the timed path.

Reference behaviour restated:
  * L60 vertical grid recipe            utils/generate_vertical_levels_L60.py:22-48
  * constant-theta hydrostatic profile  dynamics/awfl/Dycore.h:739-748 (hydro_const_theta)
  * cos^2 ellipsoid bubble              dynamics/awfl/Dycore.h:753-766, centre/radii :1069
  * supercell sounding column           standalone/mmf_simplified/supercell_init.h:7-135,
                                        pam_core/idealized_profiles.h (supercell T / p_dry / RH / qvs)
  * column broadcast + T perturbation   pam_core/modules/broadcast_initial_gcm_column.h:8-41,
                                        pam_core/modules/perturb_temperature.h:43-61 (shape only: the
                                        reference RNG `yakl::Random` is third-party and absent, so the
                                        random stream is our own splitmix64)
  * tracer registrations (names/positive/adds_mass, in registration order)
        micro none   physics/micro/none/Microphysics.h:60
        kessler      physics/micro/kessler/Microphysics.h:69-71
        p3           physics/micro/p3/Microphysics.h:119-127
        shoc tke     physics/sgs/shoc/SGS.h:103
  * physical constants                  Dycore.h:871-876 defaults (== micro none/kessler), P3 values
                                        physics/micro/p3/Microphysics.h:73-83
"""
import numpy as np

CONSTS_DEFAULT = dict(R_d=287.0, cp_d=1003.0, R_v=461.0, cp_v=1859.0, p0=1.0e5, grav=9.81)
CONSTS_P3 = dict(R_d=287.042, cp_d=1004.64, R_v=461.505, cp_v=1859.0, p0=1.0e5, grav=9.80616)

# (name, positive, adds_mass)
TRACERS_NONE = [("water_vapor", True, True)]
TRACERS_KESSLER_SHOC = [("water_vapor", True, True), ("cloud_liquid", True, True), ("precip_liquid", True, True),
                        ("tke", True, False)]
TRACERS_P3_SHOC = [("cloud_water", True, True), ("cloud_water_num", True, False), ("rain", True, True),
                   ("rain_num", True, False), ("ice", True, True), ("ice_num", True, False),
                   ("ice_rime", True, False), ("ice_rime_vol", True, False), ("water_vapor", True, True),
                   ("tke", True, False)]


def derived_constants(c):
    """Dycore.h:883-890."""
    d = dict(c)
    d["cv_d"] = d["cp_d"] - d["R_d"]
    d["gamma_d"] = d["cp_d"] / d["cv_d"]
    d["kappa_d"] = d["R_d"] / d["cp_d"]
    d["cv_v"] = d["R_v"] - d["cp_v"]
    d["C0"] = (d["R_d"] * d["p0"] ** (-d["kappa_d"])) ** d["gamma_d"]
    return d


def l60_interfaces():
    dk_list = [12, 8, 8, 8, 8, 8, 8]
    dz_list = [100, 200, 400, 500, 1000, 2e3, 4e3]
    zint = np.zeros(sum(dk_list) + 1)
    kk = 1
    for d, dk in enumerate(dk_list):
        for _ in range(dk):
            zint[kk] = zint[kk - 1] + dz_list[d]
            kk += 1
    nm = len(zint) - 1
    for _ in range(20):
        tmp = zint.copy()
        for k in range(1, nm):
            zint[k] = 0.25 * tmp[k - 1] + 0.5 * tmp[k] + 0.25 * tmp[k + 1]
    return zint


def uniform_interfaces(nz, ztop):
    return np.arange(nz + 1, dtype=np.float64) * (ztop / nz)


def stretched_interfaces(nz, ztop, ratio=1.04):
    """Smoothly stretched grid for small non-uniform test cases."""
    dz = ratio ** np.arange(nz)
    dz *= ztop / dz.sum()
    return np.concatenate([[0.0], np.cumsum(dz)])


def hydro_const_theta(z, c):
    theta0 = 300.0
    exner = 1.0 - c["grav"] * z / (c["cp_d"] * theta0)
    p = c["p0"] * exner ** (c["cp_d"] / c["R_d"])
    rt = (p / c["C0"]) ** (1.0 / c["gamma_d"])
    return rt / theta0, theta0 + 0 * z


def sample_ellipse_cosine(amp, x, y, z, x0, y0, z0, xr, yr, zr):
    dist = np.sqrt(((x - x0) / xr) ** 2 + ((y - y0) / yr) ** 2 + ((z - z0) / zr) ** 2) * np.pi / 2.0
    return np.where(dist <= np.pi / 2.0, amp * np.cos(dist) ** 2, 0.0)


def _empty_fields(nz, ny, nx, nens, nt):
    f = {k: np.zeros((nz, ny, nx, nens)) for k in ("density_dry", "uvel", "vvel", "wvel", "temp")}
    f["tracers"] = np.zeros((nt, nz, ny, nx, nens))
    return f


def dry_bubble_fields(nens, nx, ny, nz, xlen, ylen, zint, consts=CONSTS_DEFAULT, tracers=TRACERS_NONE,
                      amp0=2.0, damp=0.1):
    """theta=300 K hydrostatic atmosphere + cos^2 bubble sampled at cell centres; bubble amplitude
    amp0 + damp*iens so ensemble members differ.  (The reference's own `thermal` init uses 9-point
    quadrature, Dycore.h:1021-1088; cell-centre sampling is an input choice, not a parity item.)"""
    c = derived_constants(consts)
    zmid = 0.5 * (zint[:-1] + zint[1:])
    dx, dy = xlen / nx, ylen / ny
    x = (np.arange(nx) + 0.5) * dx
    y = (np.arange(ny) + 0.5) * dy if ny > 1 else np.array([ylen / 2])
    Z, Y, X = np.meshgrid(zmid, y, x, indexing="ij")
    hr, ht = hydro_const_theta(Z, c)
    f = _empty_fields(nz, ny, nx, nens, len(tracers))
    for e in range(nens):
        theta = ht + sample_ellipse_cosine(amp0 + damp * e, X, Y, Z, xlen / 2, ylen / 2, 2000.0, 2000.0, 2000.0, 2000.0)
        p = c["C0"] * (hr * theta) ** c["gamma_d"]
        f["density_dry"][..., e] = hr
        f["temp"][..., e] = p / (hr * c["R_d"])
    return f


# ---- supercell sounding (supercell_init.h) ---------------------------------------------------------
def _sc_temperature(z, z_0, z_trop, z_top, T_0, T_trop, T_top):
    if z <= z_trop:
        lapse = -(T_trop - T_0) / (z_trop - z_0)
        return T_0 - lapse * (z - z_0)
    lapse = -(T_top - T_trop) / (z_top - z_trop)
    return T_trop - lapse * (z - z_trop)


def _sc_pressure_dry(z, z_0, z_trop, z_top, T_0, T_trop, T_top, p_0, R_d, grav):
    if z <= z_trop:
        lapse = -(T_trop - T_0) / (z_trop - z_0)
        T = _sc_temperature(z, z_0, z_trop, z_top, T_0, T_trop, T_top)
        return p_0 * (T / T_0) ** (grav / (R_d * lapse))
    lapse = -(T_trop - T_0) / (z_trop - z_0)
    p_trop = p_0 * (T_trop / T_0) ** (grav / (R_d * lapse))
    lapse = -(T_top - T_trop) / (z_top - z_trop)
    if lapse != 0:
        T = _sc_temperature(z, z_0, z_trop, z_top, T_0, T_trop, T_top)
        return p_trop * (T / T_trop) ** (grav / (R_d * lapse))
    return p_trop * np.exp(-grav * (z - z_trop) / (R_d * T_trop))


def _sc_relhum(z, z_0, z_trop):
    # a quadrature point can sit at z = -1e-13: the reference's pow() then gives NaN, which its std::min() turns into
    # qv = 0.014 (supercell_init.h:62-63); Python's min() does the same, so only the warning is silenced here.
    with np.errstate(invalid="ignore"):
        return 1.0 - 0.75 * np.float64(z / z_trop) ** 1.25 if z <= z_trop else 0.25


def _sc_sat_mix_dry(press, T):
    return 380.0 / press * np.exp(17.27 * (T - 273.0) / (T - 36.0))


def supercell_column(zint, consts=CONSTS_DEFAULT):
    """Returns rho_d, uvel, vvel, wvel, temp, rho_v columns (nz,)."""
    Rd, Rv, grav = consts["R_d"], consts["R_v"], consts["grav"]
    ordq = 5
    gll_pts = np.array([-0.5, -0.32732683535398857189914622812342917778, 0.0,
                        0.32732683535398857189914622812342917778, 0.5])
    gll_wts = np.array([0.05, 0.27222222222222222222, 0.35555555555555555556, 0.27222222222222222222, 0.05])
    z_0, z_trop, T_0, T_trop, T_top, p_0 = 0.0, 12000.0, 300.0, 213.0, 213.0, 100000.0
    nz = len(zint) - 1
    ztop = zint[nz]

    def qv_at(zloc):
        temp = _sc_temperature(zloc, z_0, z_trop, ztop, T_0, T_trop, T_top)
        pd = _sc_pressure_dry(zloc, z_0, z_trop, ztop, T_0, T_trop, T_top, p_0, Rd, grav)
        qvs = _sc_sat_mix_dry(pd, temp)
        relhum = _sc_relhum(zloc, z_0, z_trop)
        if relhum * qvs > 0.014:
            relhum = 0.014 / qvs
        return min(0.014, qvs * relhum), temp

    hyp = np.zeros((nz, ordq))
    hyp[0, 0] = p_0
    for k in range(nz):
        dz = zint[k + 1] - zint[k]
        cellmid = zint[k] + 0.5 * dz
        for kk in range(ordq - 1):
            ord_b = cellmid + gll_pts[kk] * dz
            ord_t = cellmid + gll_pts[kk + 1] * dz
            ord_m = 0.5 * (ord_b + ord_t)
            ord_dz = dz * (gll_pts[kk + 1] - gll_pts[kk])
            tot = 0.0
            for kkk in range(ordq):
                zloc = ord_m + ord_dz * gll_pts[kkk]
                qv, temp = qv_at(zloc)
                tot += (-(1 + qv) * grav / (Rd + qv * Rv) / temp) * gll_wts[kkk]
            tot *= dz * (gll_pts[kk + 1] - gll_pts[kk])
            hyp[k, kk + 1] = hyp[k, kk] * np.exp(tot)
            if kk == ordq - 2 and k < nz - 1:
                hyp[k + 1, 0] = hyp[k, ordq - 1]
    rho_d = np.zeros(nz); u = np.zeros(nz); T = np.zeros(nz); rho_v = np.zeros(nz)
    for k in range(nz):
        dz = zint[k + 1] - zint[k]
        zmid = 0.5 * (zint[k] + zint[k + 1])
        for kk in range(ordq):
            zloc = zmid + gll_pts[kk] * dz
            qv, temp = qv_at(zloc)
            p = hyp[k, kk]
            rd = p / (Rd + qv * Rv) / temp
            uvel = 30.0 * (zloc / 5000.0) - 15.0 if zloc < 5000.0 else 15.0
            rho_d[k] += rd * gll_wts[kk]
            u[k] += uvel * gll_wts[kk]
            T[k] += temp * gll_wts[kk]
            rho_v[k] += qv * rd * gll_wts[kk]
    return rho_d, u, np.zeros(nz), np.zeros(nz), T, rho_v


def _splitmix64(seed):
    """vectorised splitmix64 -> uniform doubles in [0,1)."""
    z = (seed.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def perturb_temperature(temp, magnitude=0.1, id0=0):
    """Shape of perturb_temperature.h:43-61 with our own RNG: lowest nz/4 levels, linear decay, then
    per-level renormalisation to the unperturbed horizontal mean."""
    nz, ny, nx, nens = temp.shape
    nl = nz // 4
    if nl == 0:
        return temp
    with np.errstate(over="ignore"):
        e, k, j, i = np.meshgrid(np.arange(nens), np.arange(nl), np.arange(ny), np.arange(nx), indexing="ij")
        seed = (((e + id0) * nl + k) * ny + j) * nx + i
        r = _splitmix64(seed.astype(np.uint64)) * 2.0 - 1.0        # (nens,nl,ny,nx)
    r = np.transpose(r, (1, 2, 3, 0))
    scaling = (nl - np.arange(nl, dtype=np.float64)) / nl
    mean1 = temp[:nl].mean(axis=(1, 2), keepdims=True)
    temp[:nl] += r * magnitude * scaling[:, None, None, None]
    mean2 = temp[:nl].mean(axis=(1, 2), keepdims=True)
    temp[:nl] = temp[:nl] * mean1 / mean2
    return temp


def supercell_fields(nens, nx, ny, nz, zint, consts=CONSTS_DEFAULT, tracers=TRACERS_NONE, magnitude=0.1, id0=0):
    rho_d, u, v, w, T, rho_v = supercell_column(zint, consts)
    f = _empty_fields(nz, ny, nx, nens, len(tracers))
    names = [t[0] for t in tracers]
    b = lambda col: np.broadcast_to(col[:, None, None, None], (nz, ny, nx, nens)).copy()
    f["density_dry"] = b(rho_d); f["uvel"] = b(u); f["vvel"] = b(v); f["wvel"] = b(w); f["temp"] = b(T)
    f["tracers"][names.index("water_vapor")] = b(rho_v)
    perturb_temperature(f["temp"], magnitude, id0)
    return f


def add_tracer_blobs(f, tracers, xlen, ylen, zint, rel=1.0e-3):
    """Fill the non-vapour tracers with smooth positive blobs surrounded by exact zeros so that the FCT
    positivity limiter and the max(0,.) clipping are exercised."""
    nt, nz, ny, nx, nens = f["tracers"].shape
    zmid = 0.5 * (zint[:-1] + zint[1:])
    dx, dy = xlen / nx, ylen / ny
    x = (np.arange(nx) + 0.5) * dx
    y = (np.arange(ny) + 0.5) * dy if ny > 1 else np.array([ylen / 2])
    Z, Y, X = np.meshgrid(zmid, y, x, indexing="ij")
    ztop = zint[-1]
    rho = f["density_dry"]
    for t, (name, _, _) in enumerate(tracers):
        if name == "water_vapor":
            continue
        x0 = xlen * (0.2 + 0.6 * ((t * 0.37) % 1.0))
        z0 = ztop * (0.05 + 0.25 * ((t * 0.61) % 1.0))
        blob = sample_ellipse_cosine(1.0, X, Y, Z, x0, ylen / 2, z0, xlen * 0.15, max(ylen * 0.15, 1.0), ztop * 0.06)
        scale = rel * (1.0 + 0.1 * t)
        if name == "tke":
            f["tracers"][t] = rho * (0.1 * blob[..., None] + 0.0)
        else:
            f["tracers"][t] = rho * scale * blob[..., None] * (1.0 + 0.01 * np.arange(nens))
    return f


def tracer_flags(tracers):
    names = [t[0] for t in tracers]
    return names, [t[1] for t in tracers], [t[2] for t in tracers], names.index("water_vapor")


def carve_dry_air(f, tracers, moist_every=4, spread=True):
    """Exact zeros in the water vapour: per member a dry slab in x, a dry layer in z and (3-D) a dry row in y, at member-dependent
    places, with every `moist_every`-th member left untouched.  Under the supercell wind the cells on both sides of every
    moist/dry edge lose more vapour per stage than they hold, so the FCT positivity limiter (Dycore.h:525-550) acts on
    water_vapor itself in every stage -- the reference limits EVERY positive tracer, vapour included (Dycore.h:533;
    physics/micro/none/Microphysics.h:61 registers it positive).  Rows of 64 members then hold limited and unlimited members
    side by side.  spread=False puts the slabs at the same place in every member (edges of one "air mass": rows are flagged
    along those edges only).  An input choice for tests and for bench.py --limiter; nothing of the reference is restated here."""
    names = [t[0] for t in tracers]
    wv = f["tracers"][names.index("water_vapor")]
    nz, ny, nx, nens = wv.shape
    for e in range(nens):
        if moist_every and e % moist_every == moist_every - 1:
            continue
        s = e if spread else 0
        i0, w = (3 * s) % nx, max(1, nx // 3)
        ii = [(i0 + d) % nx for d in range(w)]
        wv[:, :, ii, e] = 0.0
        k0 = nz // 3 + (s % 3)
        wv[k0:k0 + 2, :, :, e] = 0.0
        if ny > 1:
            wv[:, s % ny, :, e] = 0.0
    return f
