"""Deterministic synthetic range-dependent atmosphere for the 3D.RngDep tests: a grid of perturbed, vertically thinned
copies of ToyAtmo.met written as <prefix><n>.met (n = ix*ny + iy, the reference's file index, G2S_MultiDimSpline3D.cpp:154)
plus loc_x.dat / loc_y.dat.  Non-square cells (dx != dy) so that quirk Q11 matters."""
import os

import numpy as np

import harness as H

X_NODES = np.array([-1000.0, -500.0, 0.0, 500.0, 1000.0])
Y_NODES = np.array([-800.0, -400.0, 0.0, 400.0, 800.0])
THIN = 4                      # keep every 4th ToyAtmo row: 350 nodes, dz = 0.4 km


def grid_columns(thin=THIN):
    """returns z [nz], and T, u, v, rho as [nx][ny][nz] (raw .met units: K, m/s, m/s, g/cm^3)"""
    raw = np.loadtxt(H.TOYATMO)[::thin]
    z = raw[:, 0]
    nx, ny = len(X_NODES), len(Y_NODES)
    T = np.zeros((nx, ny, len(z))); u = np.zeros_like(T); v = np.zeros_like(T); rho = np.zeros_like(T)
    for i, x in enumerate(X_NODES):
        for j, y in enumerate(Y_NODES):
            bump = np.exp(-((z - 45.0) / 25.0) ** 2)
            T[i, j] = raw[:, 1] * (1.0 + 0.015 * np.sin(x / 700.0 + y / 900.0) * bump + 0.004 * np.cos(x / 450.0) * np.sin(y / 380.0))
            u[i, j] = raw[:, 2] + 12.0 * (x / 1000.0) * bump + 3.0 * np.sin(y / 500.0)
            v[i, j] = raw[:, 3] + 8.0 * (y / 800.0) * bump - 2.5 * np.cos(x / 600.0) * np.exp(-z / 60.0)
            rho[i, j] = raw[:, 4] * (1.0 + 0.01 * np.cos(x / 800.0 - y / 650.0))
    return z, T, u, v, rho, raw[:, 5]


GRID_NPZ = os.path.join(H.GOLDEN_DIR, "rngdep_grid.npz")
GRID_NPZ_FULL = os.path.join(H.GOLDEN_DIR, "rngdep_grid_1400.npz")       # thin = 1: the 5x5x1400 grid of BASELINE config 4


def _grid_npz(thin):
    assert thin in (THIN, 1)
    return GRID_NPZ if thin == THIN else GRID_NPZ_FULL


def save_grid_npz(thin=THIN):
    """(make_golden.py) evaluates the analytic perturbation once and stores the columns, so that the files written at
    test time are the same bytes on every machine"""
    z, T, u, v, rho, p = grid_columns(thin)
    np.savez_compressed(_grid_npz(thin), z=z, T=T, u=u, v=v, rho=rho, p=p, x=X_NODES, y=Y_NODES)


def load_grid_columns(thin=THIN):
    g = np.load(_grid_npz(thin))
    return g["z"], g["T"], g["u"], g["v"], g["rho"], g["p"]


def write_grid(dirpath, short_paths=True, thin=THIN):
    """writes the files; returns (prefix, locx, locy).  Paths must stay short: the reference formats file names into a
    50-byte buffer (G2S_MultiDimSpline3D.cpp:112,154)."""
    os.makedirs(dirpath, exist_ok=True)
    z, T, u, v, rho, p = load_grid_columns(thin)
    prefix = os.path.join(dirpath, "p")
    assert len(prefix) < 40 or not short_paths     # only the reference needs short names
    for i in range(len(X_NODES)):
        for j in range(len(Y_NODES)):
            n = i * len(Y_NODES) + j
            with open(f"{prefix}{n}.met", "w") as fh:
                for k in range(len(z)):
                    fh.write(f"{z[k]:.10g} {T[i, j, k]:.12g} {u[i, j, k]:.12g} {v[i, j, k]:.12g} {rho[i, j, k]:.12g} {p[k]:.10g}\n")
    locx, locy = os.path.join(dirpath, "loc_x.dat"), os.path.join(dirpath, "loc_y.dat")
    with open(locx, "w") as fh:
        fh.write("".join(f"{x:.10g}\n" for x in X_NODES))
    with open(locy, "w") as fh:
        fh.write("".join(f"{y:.10g}\n" for y in Y_NODES))
    return prefix, locx, locy


# ---- Global.RngDep: grid in latitude / longitude (degrees in the loc files), file index n = ilat*nlon + ilon
#      (G2S_GlobalMultiDimSpline3D.cpp:159); cells of 3 x 4 degrees so that lat and lon scalings differ ----
LAT_NODES = np.array([25.0, 28.0, 31.0, 34.0, 37.0])
LON_NODES = np.array([-8.0, -4.0, 0.0, 4.0, 8.0])
GRID_GLOBAL_NPZ = os.path.join(H.GOLDEN_DIR, "globalrd_grid.npz")


def grid_columns_global():
    raw = np.loadtxt(H.TOYATMO)[::THIN]
    z = raw[:, 0]
    nt, npn = len(LAT_NODES), len(LON_NODES)
    T = np.zeros((nt, npn, len(z))); u = np.zeros_like(T); v = np.zeros_like(T); rho = np.zeros_like(T)
    for i, la in enumerate(LAT_NODES):
        for j, lo in enumerate(LON_NODES):
            bump = np.exp(-((z - 45.0) / 25.0) ** 2)
            a, b = (la - 31.0) / 6.0, lo / 8.0
            T[i, j] = raw[:, 1] * (1.0 + 0.015 * np.sin(1.3 * a + 0.9 * b) * bump + 0.004 * np.cos(2.1 * a) * np.sin(2.4 * b))
            u[i, j] = raw[:, 2] + 12.0 * a * bump + 3.0 * np.sin(1.7 * b)
            v[i, j] = raw[:, 3] + 8.0 * b * bump - 2.5 * np.cos(1.6 * a) * np.exp(-z / 60.0)
            rho[i, j] = raw[:, 4] * (1.0 + 0.01 * np.cos(1.2 * a - 1.4 * b))
    return z, T, u, v, rho, raw[:, 5]


def save_grid_global_npz():
    z, T, u, v, rho, p = grid_columns_global()
    np.savez_compressed(GRID_GLOBAL_NPZ, z=z, T=T, u=u, v=v, rho=rho, p=p, lat=LAT_NODES, lon=LON_NODES)


def write_grid_global(dirpath, short_paths=True):
    """writes g<n>.met, loc_lat.dat, loc_lon.dat; returns (prefix, loclat, loclon)"""
    os.makedirs(dirpath, exist_ok=True)
    g = np.load(GRID_GLOBAL_NPZ)
    z, T, u, v, rho, p = g["z"], g["T"], g["u"], g["v"], g["rho"], g["p"]
    prefix = os.path.join(dirpath, "g")
    assert len(prefix) < 40 or not short_paths
    for i in range(len(LAT_NODES)):
        for j in range(len(LON_NODES)):
            n = i * len(LON_NODES) + j
            with open(f"{prefix}{n}.met", "w") as fh:
                for k in range(len(z)):
                    fh.write(f"{z[k]:.10g} {T[i, j, k]:.12g} {u[i, j, k]:.12g} {v[i, j, k]:.12g} {rho[i, j, k]:.12g} {p[k]:.10g}\n")
    loclat, loclon = os.path.join(dirpath, "loc_lat.dat"), os.path.join(dirpath, "loc_lon.dat")
    with open(loclat, "w") as fh:
        fh.write("".join(f"{x:.10g}\n" for x in LAT_NODES))
    with open(loclon, "w") as fh:
        fh.write("".join(f"{y:.10g}\n" for y in LON_NODES))
    return prefix, loclat, loclon
