"""CPU suite: the host side of the range-dependent sets (grid loader, differenced-coefficient table, scalar evaluator of
libgeoac_hip.so - product code, no GPU needed) against the golden probes of the compiled reference's scalar API
(c, rho, u, v at 400 random points; tests/golden/{3drd,globalrd}_small.npz)."""
import ctypes

import numpy as np
import pytest

import geoac_amd as G
import harness as H
import rngdep_data as RD

_dp = ctypes.POINTER(ctypes.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


def _load(eq, grid):
    lib = G.load_library()
    nx, ny, nz = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.geoac_grid_dims(grid[0].encode(), grid[1].encode(), grid[2].encode(), ctypes.byref(nx), ctypes.byref(ny), ctypes.byref(nz)) == 0
    nx, ny, nz = nx.value, ny.value, nz.value
    x, y, z = np.zeros(nx), np.zeros(ny), np.zeros(nz)
    F = [np.zeros((nx, ny, nz)) for _ in range(4)]
    lib.geoac_grid_load_eq.argtypes = None
    assert lib.geoac_grid_load_eq(eq, grid[0].encode(), grid[1].encode(), grid[2].encode(), b"zTuvdp", ctypes.c_double(0.0),
                                  nx, ny, nz, _p(x), _p(y), _p(z), *[_p(f) for f in F]) == 0
    lib.geoac_grid_table_size.restype = ctypes.c_size_t
    tab = np.zeros(lib.geoac_grid_table_size(nx, ny, nz))
    lib.geoac_grid_table_eq.argtypes = None
    assert lib.geoac_grid_table_eq(eq, nx, ny, nz, _p(x), _p(y), _p(z), *[_p(f) for f in F], _p(tab)) == 0
    lib.geoac_grid_eval_eq.restype = ctypes.c_double
    lib.geoac_grid_eval_eq.argtypes = [ctypes.c_int] * 4 + [_dp] * 4 + [ctypes.c_int] + [ctypes.c_double] * 3

    def ev(field, a, b, c):
        return np.array([lib.geoac_grid_eval_eq(eq, nx, ny, nz, _p(x), _p(y), _p(z), _p(tab), field, float(p), float(q), float(r))
                         for p, q, r in zip(a, b, c)])
    return ev, (x, y, z)


def test_cartesian_grid_host_evaluator_vs_reference_probes(tmp_path):
    g = np.load(f"{H.GOLDEN_DIR}/3drd_small.npz")
    ev, _ = _load(G.EQ_3D_RNGDEP, RD.write_grid(str(tmp_path), short_paths=False))
    px, py, pz, api = g["probe_x"], g["probe_y"], g["probe_z"], g["probe_api8"]
    c = np.sqrt(0.00040187 * ev(0, px, py, pz))
    assert np.abs(c / api[:, 0] - 1).max() < 1e-12
    assert np.abs(ev(3, px, py, pz) / api[:, 1] - 1).max() < 1e-12
    scale = np.abs(api[:, 2:4]).max()
    assert np.abs(ev(1, px, py, pz) - api[:, 2]).max() < 1e-12 * scale
    assert np.abs(ev(2, px, py, pz) - api[:, 3]).max() < 1e-12 * scale


def test_spherical_grid_host_evaluator_vs_reference_probes(tmp_path):
    g = np.load(f"{H.GOLDEN_DIR}/globalrd_small.npz")
    ev, (lat, lon, r) = _load(G.EQ_GLOBAL_RNGDEP, RD.write_grid_global(str(tmp_path), short_paths=False))
    assert abs(r[0] - 6370.0) < 1e-9 and abs(np.degrees(lat[0]) - 25.0) < 1e-12      # radius, radians
    pr, plat, plon, api = g["probe_r"], g["probe_lat"], g["probe_lon"], g["probe_api8"]
    c = np.sqrt(0.00040187 * ev(0, plat, plon, pr))        # table order: (lat, lon, r)
    assert np.abs(c / api[:, 0] - 1).max() < 1e-12
    assert np.abs(ev(3, plat, plon, pr) / api[:, 1] - 1).max() < 1e-12
    scale = np.abs(api[:, 2:4]).max()
    assert np.abs(ev(1, plat, plon, pr) - api[:, 2]).max() < 1e-12 * scale
    assert np.abs(ev(2, plat, plon, pr) - api[:, 3]).max() < 1e-12 * scale
