"""ctypes bindings of the TEST-ONLY checkers: the plain-C oracle (oracle/libgeoac_oracle.so) and,
where it was built (this container only), the compiled reference behind oracle/_ref/libref_*.so.
Nothing under geoac_amd/ imports this module."""
import ctypes
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
TOYATMO = os.path.join(GOLDEN_DIR, "ToyAtmo.met")

EQ_2D, EQ_3D, EQ_GLOBAL, EQ_3D_RNGDEP, EQ_GLOBAL_RNGDEP = 0, 1, 2, 3, 4
EQ_NAMES = {EQ_2D: "2d", EQ_3D: "3d", EQ_GLOBAL: "global", EQ_3D_RNGDEP: "3drd", EQ_GLOBAL_RNGDEP: "globalrd"}
REC_STRIDE = 32
SMP_STRIDE = 10
REC = dict(VALID=0, STEPS=1, BROKE=2, TTIME=3, ATTEN=4, TURN=5, INCL=6, BACKAZ=7, AMP=8, RANGE=9, JACOB=10, STATE=12)
MODE_WRITE_RAYS, MODE_WRITE_CAUSTICS = 1, 2

_dp = ctypes.POINTER(ctypes.c_double)


class FanCfg(ctypes.Structure):
    _fields_ = [("z_grnd", ctypes.c_double), ("tweak_abs", ctypes.c_double), ("freq", ctypes.c_double),
                ("vert_limit", ctypes.c_double), ("range_limit", ctypes.c_double),
                ("src", ctypes.c_double * 3), ("bounces", ctypes.c_int), ("calc_amp", ctypes.c_int),
                ("mode", ctypes.c_int), ("pad_", ctypes.c_int), ("xy_limits", ctypes.c_double * 4)]


def make_cfg(eqset, bounces=2, calc_amp=True, mode=0, src=None, z_grnd=0.0, tweak_abs=0.3, freq=0.1,
             vert_limit=float("nan"), range_limit=float("nan"), xy_limits=None):
    if src is None:
        src = (0.0, 30.0, 0.0) if eqset in (EQ_GLOBAL, EQ_GLOBAL_RNGDEP) else (0.0, 0.0, 0.0)
    if xy_limits is None:
        xy_limits = (float("nan"),) * 4
    return FanCfg(z_grnd, tweak_abs, freq, vert_limit, range_limit, (ctypes.c_double * 3)(*src),
                  bounces, 1 if calc_amp else 0, mode, 0, (ctypes.c_double * 4)(*xy_limits))


def _p(a):
    return a.ctypes.data_as(_dp)


def build_oracle():
    """(re)build the oracle library (gcc only, seconds)."""
    so = os.path.join(ORACLE_DIR, "libgeoac_oracle.so")
    src = os.path.join(ORACLE_DIR, "geoac_oracle.c")
    if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
    return so


def fan_angles(theta_min=0.5, theta_max=45.0, theta_step=0.5, phi_min=-90.0, phi_max=-90.0, phi_step=1.0):
    """launch angles exactly as the reference's `for(double x = min; x <= max; x += step)` loops
    enumerate them (repeated addition), phi outer / theta inner (GeoAcGlobal_main.cpp:241-242)."""
    th, ph = [], []
    phi = phi_min
    while phi <= phi_max:
        theta = theta_min
        while theta <= theta_max:
            th.append(theta)
            ph.append(phi)
            theta += theta_step
        phi += phi_step
    return np.array(th, dtype=np.float64), np.array(ph, dtype=np.float64)


class _FanLib:
    """shared call shapes of oracle and reference shim"""

    def _fan(self, fn, pre, cfg, theta, phi, smp_cap=0):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        phi = np.ascontiguousarray(phi, dtype=np.float64)
        n = len(theta)
        rec = np.zeros((n, cfg.bounces + 1, REC_STRIDE))
        smp = np.zeros((max(smp_cap, 1), SMP_STRIDE))
        nsmp = ctypes.c_int64(0)
        steps = fn(*pre, ctypes.byref(cfg), n, _p(theta), _p(phi), _p(rec),
                   _p(smp) if smp_cap else None, ctypes.c_int64(smp_cap), ctypes.byref(nsmp))
        return int(steps), rec, smp[:min(nsmp.value, smp_cap)], nsmp.value


class Oracle(_FanLib):
    def __init__(self, eqset, met=TOYATMO, fmt="zTuvdp"):
        self.lib = ctypes.CDLL(build_oracle())
        L = self.lib
        L.orc_create.restype = ctypes.c_void_p
        L.orc_create.argtypes = [ctypes.c_int]
        L.orc_fan.restype = ctypes.c_int64
        for f in (L.orc_destroy, L.orc_load, L.orc_load_arrays, L.orc_fan, L.orc_atmo_probe,
                  L.orc_absorption_probe, L.orc_tables, L.orc_trace_leg0, L.orc_limits, L.orc_load_grid, L.orc_grid_probe):
            f.argtypes = None
        self.eqset = eqset
        self.ctx = ctypes.c_void_p(L.orc_create(eqset))
        assert self.ctx.value, "orc_create failed"
        if met is not None:
            n = L.orc_load(self.ctx, met.encode(), fmt.encode())
            assert n > 0, f"orc_load({met}) -> {n}"
            self.n = n

    def load_grid(self, prefix, locx, locy, fmt="zTuvdp", z_grnd=0.0):
        n = self.lib.orc_load_grid(self.ctx, prefix.encode(), locx.encode(), locy.encode(), fmt.encode(), ctypes.c_double(z_grnd))
        assert n > 0, f"orc_load_grid -> {n}"
        self.n = n

    def grid_probe(self, x, y, z):
        x, y, z = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, z))
        o = np.zeros((len(x), 30)); a = np.zeros((len(x), 8))
        self.lib.orc_grid_probe(self.ctx, len(x), _p(x), _p(y), _p(z), _p(o), _p(a))
        return o, a

    def load_arrays(self, z, T, u, v, rho):
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (z, T, u, v, rho)]
        self.n = self.lib.orc_load_arrays(self.ctx, len(arrs[0]), *[_p(a) for a in arrs])
        assert self.n > 0

    def fan(self, cfg, theta, phi, smp_cap=0):
        return self._fan(self.lib.orc_fan, (self.ctx,), cfg, theta, phi, smp_cap)

    def atmo_probe(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        o = np.zeros((len(x), 9)); r = np.zeros(len(x))
        self.lib.orc_atmo_probe(self.ctx, len(x), _p(x), _p(o), _p(r))
        return o, r

    def absorption_probe(self, x, f, z_grnd=0.0, tweak=0.3):
        x = np.ascontiguousarray(x, dtype=np.float64); f = np.ascontiguousarray(f, dtype=np.float64)
        o = np.zeros(len(x))
        self.lib.orc_absorption_probe(self.ctx, len(x), _p(x), _p(f), ctypes.c_double(z_grnd), ctypes.c_double(tweak), _p(o))
        return o

    def tables(self):
        n = self.n
        a = [np.zeros(n) for _ in range(9)]
        m = self.lib.orc_tables(self.ctx, n, *[_p(t) for t in a])
        assert m == n
        return dict(zip(("x", "T", "u", "v", "rho", "sT", "su", "sv", "srho"), a))

    def set_ray_limit(self, ray_limit):
        self.lib.orc_set_ray_limit.argtypes = [ctypes.c_void_p, ctypes.c_double]
        self.lib.orc_set_ray_limit(self.ctx, float(ray_limit))

    def limits(self):
        a = ctypes.c_double(); b = ctypes.c_double()
        self.lib.orc_limits(self.ctx, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def trace_leg0(self, cfg, theta, phi, max_rows=60000):
        out = np.zeros(max_rows * 18); E = ctypes.c_int(0)
        k = self.lib.orc_trace_leg0(self.ctx, ctypes.byref(cfg), ctypes.c_double(theta), ctypes.c_double(phi),
                                    max_rows, _p(out), ctypes.byref(E))
        e = E.value; rows = min(abs(k) + 1, max_rows)
        return k, out[:rows * e].reshape(rows, e).copy()

    def __del__(self):
        try:
            self.lib.orc_destroy(self.ctx)
        except Exception:
            pass


def ref_available(eqset):
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", f"libref_{EQ_NAMES[eqset]}.so"))


class RefShim(_FanLib):
    """the compiled, unmodified reference translation units behind oracle/ref_shim_*.cpp.
    One instance per equation set per process (the reference keeps its state in globals)."""
    _loaded = {}

    def __init__(self, eqset, met=TOYATMO, fmt="zTuvdp", grid=None, z_grnd=0.0):
        """grid = (prefix, locx, locy) for the range-dependent sets"""
        path = os.path.join(ORACLE_DIR, "_ref", f"libref_{EQ_NAMES[eqset]}.so")
        key = (eqset, met if grid is None else tuple(grid) + (z_grnd,), fmt)
        if eqset in RefShim._loaded and RefShim._loaded[eqset][0] != key:
            raise RuntimeError("reference shim already loaded with another profile in this process")
        if eqset not in RefShim._loaded:
            lib = ctypes.CDLL(path)
            lib.ref_fan.restype = ctypes.c_int64
            if grid is None:
                n = lib.ref_load(met.encode(), fmt.encode())
            else:
                n = lib.ref_load_grid(grid[0].encode(), grid[1].encode(), grid[2].encode(), fmt.encode(), ctypes.c_double(z_grnd))
            assert n > 0
            RefShim._loaded[eqset] = (key, lib, n)
        _, self.lib, self.n = RefShim._loaded[eqset]
        self.eqset = eqset

    def fan(self, cfg, theta, phi, smp_cap=0):
        return self._fan(self.lib.ref_fan, (), cfg, theta, phi, smp_cap)

    def atmo_probe(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        o = np.zeros((len(x), 9)); r = np.zeros(len(x))
        self.lib.ref_atmo_probe(len(x), _p(x), _p(o), _p(r))
        return o, r

    def absorption_probe(self, x, f, z_grnd=0.0, tweak=0.3):
        x = np.ascontiguousarray(x, dtype=np.float64); f = np.ascontiguousarray(f, dtype=np.float64)
        o = np.zeros(len(x))
        self.lib.ref_absorption_probe(len(x), _p(x), _p(f), ctypes.c_double(z_grnd), ctypes.c_double(tweak), _p(o))
        return o

    def grid_probe(self, x, y, z):
        x, y, z = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, z))
        o = np.zeros((len(x), 30)); a = np.zeros((len(x), 8))
        self.lib.ref_grid_probe(len(x), _p(x), _p(y), _p(z), _p(o), _p(a))
        return o, a

    def tables(self):
        n = self.n
        a = [np.zeros(n) for _ in range(9)]
        m = self.lib.ref_tables(n, *[_p(t) for t in a])
        assert m == n
        return dict(zip(("x", "T", "u", "v", "rho", "sT", "su", "sv", "srho"), a))

    def trace_leg0(self, cfg, theta, phi, max_rows=60000):
        out = np.zeros(max_rows * 18); E = ctypes.c_int(0)
        k = self.lib.ref_trace_leg0(ctypes.byref(cfg), ctypes.c_double(theta), ctypes.c_double(phi),
                                    max_rows, _p(out), ctypes.byref(E))
        e = E.value; rows = min(abs(k) + 1, max_rows)
        return k, out[:rows * e].reshape(rows, e).copy()
