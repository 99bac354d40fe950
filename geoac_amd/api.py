"""ctypes mirror of include/geoac_hip.h + include/geoac_host.h (libgeoac_hip.so, built in-tree by
geoac_amd/csrc/Makefile or __graft_entry__.build()).  Names follow the reference's vocabulary:
rays, legs (bounces), launch angles theta/phi, arrivals."""
import ctypes
import os

import numpy as np

EQ_2D, EQ_3D, EQ_GLOBAL, EQ_3D_RNGDEP, EQ_GLOBAL_RNGDEP = 0, 1, 2, 3, 4
REC_STRIDE = 32
REC = dict(VALID=0, STEPS=1, BROKE=2, TTIME=3, ATTEN=4, TURN=5, INCL=6, BACKAZ=7, AMP=8, RANGE=9, JACOB=10, STATE=12)
MODE_WRITE_RAYS, MODE_WRITE_CAUSTICS, MODE_INTERACTIVE = 1, 2, 4
FAN_STEP_LIMIT, FAN_SUB_FALLBACK, FAN_ABS_FALLBACK = 1, 0x100, 0x200          # geoac_fan_status (include/geoac_hip.h)

_dp = ctypes.POINTER(ctypes.c_double)


class GeoAcError(RuntimeError):
    pass


class Params(ctypes.Structure):
    """geoac_params (include/geoac_hip.h): the reference's globals + *_RunProp locals."""
    _fields_ = [("ds_min", ctypes.c_double), ("ds_max", ctypes.c_double), ("ray_limit", ctypes.c_double),
                ("vert_limit", ctypes.c_double), ("range_limit", ctypes.c_double), ("z_grnd", ctypes.c_double),
                ("r_earth", ctypes.c_double), ("tweak_abs", ctypes.c_double), ("freq", ctypes.c_double),
                ("src", ctypes.c_double * 3), ("bounces", ctypes.c_int), ("calc_amp", ctypes.c_int),
                ("mode", ctypes.c_int), ("sample_stride", ctypes.c_int), ("xy_limits", ctypes.c_double * 4)]


EIG_STRIDE = 16
EIG = dict(RCVR=0, INDEX=1, BOUNCES=2, THETA=3, PHI=4, TTIME=5, CELERITY=6, AMP_DB=7, ATTEN_DB=8, INCL=9, BEARING=10, BACKAZ=11,
           AZDEV=12, NSMP=13, SMP0=14)


class EigParams(ctypes.Structure):
    """geoac_eig_params (include/geoac_eig.h)"""
    _fields_ = [("theta_min", ctypes.c_double), ("theta_max", ctypes.c_double), ("bnc_min", ctypes.c_int), ("bnc_max", ctypes.c_int),
                ("iterations", ctypes.c_int), ("azimuth_err_lim", ctypes.c_double), ("verbose", ctypes.c_int)]


def library_path():
    """the in-tree build; GEOAC_LIB names another build of the same library (A/B and diagnostic builds: tools/ab_metric.py, -DGEOAC_KSTAT)"""
    return os.environ.get("GEOAC_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgeoac_hip.so")


_lib = None


def load_library():
    """dlopen libgeoac_hip.so; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise GeoAcError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         f"or `make -C geoac_amd/csrc` (there is no CPU fallback)")
    lib = ctypes.CDLL(path)
    lib.geoac_strerror.restype = ctypes.c_char_p
    lib.geoac_last_error.restype = ctypes.c_char_p
    lib.geoac_last_error.argtypes = [ctypes.c_void_p]
    lib.geoac_version.restype = ctypes.c_char_p
    lib.geoac_grid_load.argtypes = None
    lib.geoac_grid_load_eq.argtypes = None
    lib.geoac_fan_enumerate.restype = ctypes.c_long
    lib.geoac_fan_enumerate.argtypes = [ctypes.c_double] * 6 + [ctypes.c_long, _dp, _dp]
    _lib = lib
    return lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _arr(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---------------- host set-up helpers (include/geoac_host.h) ----------------
def met_load(path, eqset, fmt="zTuvdp"):
    """read a .met profile the way Spline_Single_G2S does; returns dict x,T,u,v,rho (x = radius for Global)."""
    lib = load_library()
    rows = lib.geoac_met_rows(path.encode())
    if rows < 3:
        raise GeoAcError(f"cannot read profile {path} (rows={rows})")
    a = [np.zeros(rows) for _ in range(5)]
    n = lib.geoac_met_load(path.encode(), fmt.encode(), eqset, rows, *[_p(t) for t in a])
    if n != rows:
        raise GeoAcError(f"geoac_met_load({path}) -> {n}")
    return dict(zip(("x", "T", "u", "v", "rho"), a))


def natural_spline_slopes(x, f):
    lib = load_library()
    x, f = _arr(x), _arr(f)
    s = np.zeros(len(x))
    lib.geoac_natural_spline_slopes(len(x), _p(x), _p(f), _p(s))
    return s


def fan_enumerate(theta_min=0.5, theta_max=45.0, theta_step=0.5, phi_min=-90.0, phi_max=-90.0, phi_step=1.0):
    """launch angles of the reference's double loop (repeated addition; phi outer, theta inner)."""
    lib = load_library()
    n = lib.geoac_fan_enumerate(theta_min, theta_max, theta_step, phi_min, phi_max, phi_step, 0, None, None)
    if n < 0:
        raise GeoAcError("fan_enumerate: bad step")
    th, ph = np.zeros(n), np.zeros(n)
    lib.geoac_fan_enumerate(theta_min, theta_max, theta_step, phi_min, phi_max, phi_step, n, _p(th), _p(ph))
    return th, ph


def default_params(eqset):
    p = Params()
    rc = load_library().geoac_default_params(eqset, ctypes.byref(p))
    if rc:
        raise GeoAcError("geoac_default_params failed")
    return p


# ---------------- launch-plan options (geoac_set_option) ----------------
# Options every new FanContext / FanPool gets (key without the GEOAC_ prefix -> value); tests and A/B tools set them through options() or
# directly.  The library itself reads no environment variable (unless GEOAC_DEBUG_ENV=1).
DEFAULT_OPTIONS = {}


class options:
    """with options(S_ROWS=4096, COMPACT=0): ...  -  contexts created inside carry these launch-plan options"""

    def __init__(self, **kw):
        self.kw = {str(k): str(v) for k, v in kw.items()}

    def __enter__(self):
        self.old = dict(DEFAULT_OPTIONS)
        DEFAULT_OPTIONS.update(self.kw)
        return self

    def __exit__(self, *exc):
        DEFAULT_OPTIONS.clear()
        DEFAULT_OPTIONS.update(self.old)
        return False


def build_id(path=None):
    """geoac_build_id of the loaded library (or of the library at `path`): the hash of the sources and flags it was compiled from"""
    lib = load_library() if path is None else ctypes.CDLL(path)
    lib.geoac_build_id.restype = ctypes.c_char_p
    return lib.geoac_build_id().decode()


def has_ab_kernels():
    """True for an A/B build of the library (`make AB=1`): it also holds the diagnostic kernels the launch plan never selects (DUO, GRID_LANES=2)"""
    lib = load_library()
    return bool(lib.geoac_build_has_ab())


def option_names():
    lib = load_library()
    lib.geoac_option_names.restype = ctypes.POINTER(ctypes.c_char_p)
    p, out, i = lib.geoac_option_names(), [], 0
    while p[i]:
        out.append(p[i].decode()); i += 1
    return out


def _apply_options(lib, h, opts):
    lib.geoac_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p]
    for k, v in opts.items():
        k = str(k)
        if k.startswith("GEOAC_"):
            k = k[6:]
        rc = lib.geoac_set_option(h, k.encode(), str(v).encode())
        if rc:
            msg = lib.geoac_last_error(h)
            raise GeoAcError(f"geoac_set_option({k}={v}): {lib.geoac_strerror(rc).decode()}: {msg.decode() if msg else ''}")


# ---------------- the GPU fan context ----------------
class FanContext:
    """One geoac_ctx: a GPU, a stream, an atmosphere, a parameter set; runs fans of launch angles."""

    def __init__(self, eqset, device=0, stream=None, options=None):
        self.lib = load_library()
        self.eqset = eqset
        self._h = ctypes.c_void_p()
        rc = self.lib.geoac_create(ctypes.byref(self._h), eqset, device)
        if rc:
            raise GeoAcError(f"geoac_create: {self.lib.geoac_strerror(rc).decode()}")
        _apply_options(self.lib, self._h, dict(DEFAULT_OPTIONS, **(options or {})))
        if stream is not None:
            self._chk(self.lib.geoac_set_stream(self._h, ctypes.c_void_p(stream)))
        self.params = default_params(eqset)
        self.n_rays = 0

    def _chk(self, rc):
        if rc:
            msg = self.lib.geoac_last_error(self._h)
            raise GeoAcError(f"{self.lib.geoac_strerror(rc).decode()}: {msg.decode() if msg else ''}")

    def close(self):
        if self._h:
            self.lib.geoac_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clone(self):
        """geoac_clone: a second context on the same device sharing this one's atmosphere tables (several fans at once); valid until this context uploads another
        atmosphere or is closed - after that the clone's launches fail"""
        c = object.__new__(FanContext)
        c.lib, c.eqset, c._h = self.lib, self.eqset, ctypes.c_void_p()
        self._chk(self.lib.geoac_clone(self._h, ctypes.byref(c._h)))
        c.params = Params.from_buffer_copy(bytes(self.params))
        c.n_rays = 0
        if hasattr(self, "_grid_dims"):
            c._grid_dims = self._grid_dims
        return c

    def upload_atmo_1d(self, x, T, u, v, rho, slopes4=None):
        x, T, u, v, rho = (_arr(a) for a in (x, T, u, v, rho))
        if slopes4 is None:
            slopes4 = np.concatenate([natural_spline_slopes(x, f) for f in (T, u, v, rho)])
        slopes4 = _arr(slopes4)
        self._chk(self.lib.geoac_upload_atmo_1d(self._h, len(x), _p(x), _p(T), _p(u), _p(v), _p(rho), _p(slopes4)))

    def upload_atmo_3d(self, x, y, z, T, u, v, rho):
        """grid of profiles: fields [nx][ny][nz] (winds already tapered, km/s)"""
        x, y, z, T, u, v, rho = (_arr(a) for a in (x, y, z, T, u, v, rho))
        self._chk(self.lib.geoac_upload_atmo_3d(self._h, len(x), len(y), len(z), _p(x), _p(y), _p(z), _p(T), _p(u), _p(v), _p(rho)))
        self._grid_dims = (len(x), len(y), len(z))

    def grid_table(self):
        """the evaluation table the last upload_atmo_3d built on the device (layout of geoac_grid_table_eq)"""
        self.lib.geoac_grid_table_size.restype = ctypes.c_size_t
        n = self.lib.geoac_grid_table_size(*self._grid_dims)
        tab = np.zeros(n)
        self._chk(self.lib.geoac_grid_table_fetch(self._h, _p(tab), ctypes.c_size_t(n)))
        return tab

    def load_grid(self, prefix, locx, locy, fmt="zTuvdp", z_grnd=0.0):
        """Spline_Multi_G2S equivalent: <prefix><n>.met files + loc_x / loc_y node files"""
        nx, ny, nz = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        if self.lib.geoac_grid_dims(prefix.encode(), locx.encode(), locy.encode(), ctypes.byref(nx), ctypes.byref(ny), ctypes.byref(nz)):
            raise GeoAcError(f"cannot read grid {prefix}")
        nx, ny, nz = nx.value, ny.value, nz.value
        x, y, z = np.zeros(nx), np.zeros(ny), np.zeros(nz)
        F = [np.zeros((nx, ny, nz)) for _ in range(4)]
        rc = self.lib.geoac_grid_load_eq(self.eqset, prefix.encode(), locx.encode(), locy.encode(), fmt.encode(), ctypes.c_double(z_grnd),
                                      nx, ny, nz, _p(x), _p(y), _p(z), *[_p(f) for f in F])
        if rc:
            raise GeoAcError(f"geoac_grid_load_eq -> {rc}")
        self.upload_atmo_3d(x, y, z, *F)
        return dict(x=x, y=y, z=z, T=F[0], u=F[1], v=F[2], rho=F[3])

    def load_met(self, path, fmt="zTuvdp"):
        a = met_load(path, self.eqset, fmt)
        self.upload_atmo_1d(a["x"], a["T"], a["u"], a["v"], a["rho"])
        return a

    def set_params(self, **kw):
        for k, val in kw.items():
            if k == "src":
                self.params.src = (ctypes.c_double * 3)(*val)
            elif k == "xy_limits":
                self.params.xy_limits = (ctypes.c_double * 4)(*val)
            else:
                setattr(self.params, k, val)
        self._chk(self.lib.geoac_set_params(self._h, ctypes.byref(self.params)))

    def set_angles(self, theta_deg, phi_deg):
        th, ph = _arr(theta_deg), _arr(phi_deg)
        if th.ndim != 1 or th.shape != ph.shape:
            raise GeoAcError(f"set_angles: theta and phi must be one-dimensional arrays of one length (got shapes {th.shape} and {ph.shape})")
        self.n_rays = len(th)
        self._chk(self.lib.geoac_fan_set_angles(self._h, len(th), _p(th), _p(ph)))

    def launch(self):
        self._chk(self.lib.geoac_fan_launch(self._h))

    def fetch(self, out=None):
        """records of the last launch; `out`: caller-owned C-contiguous float64 array [n_rays][legs][32] (e.g. the numpy view of a pinned
        torch tensor) to copy into instead of a fresh array"""
        legs = self.params.bounces + 1
        if out is None:
            rec = np.empty((self.n_rays, legs, REC_STRIDE))
        else:
            rec = out
            if not (isinstance(rec, np.ndarray) and rec.dtype == np.float64 and rec.flags.c_contiguous and rec.shape == (self.n_rays, legs, REC_STRIDE)):
                raise GeoAcError(f"fetch(out=): need a C-contiguous float64 array of shape {(self.n_rays, legs, REC_STRIDE)}")
        steps = ctypes.c_uint64(0)
        self._chk(self.lib.geoac_fan_fetch(self._h, _p(rec), ctypes.byref(steps)))
        return rec, int(steps.value)

    def fetch_samples(self):
        """WriteRays / WriteCaustics rows, ordered by (ray, leg, m): [n][10] (GEOAC_SMP_* layout)"""
        n = ctypes.c_int64(0)
        self._chk(self.lib.geoac_fan_sample_count(self._h, ctypes.byref(n)))
        smp = np.zeros((max(n.value, 1), 10))
        if n.value:
            self._chk(self.lib.geoac_fan_fetch_samples(self._h, _p(smp), ctypes.c_int64(n.value)))
        return smp[:n.value]

    def records_dev(self):
        ptr = ctypes.c_void_p(); nbytes = ctypes.c_size_t()
        self._chk(self.lib.geoac_fan_records_dev(self._h, ctypes.byref(ptr), ctypes.byref(nbytes)))
        return ptr.value, nbytes.value

    def copy_records_to(self, dev_ptr):
        """async D2D copy of the record table into a caller-owned device buffer (ordered on the context's stream)"""
        self._chk(self.lib.geoac_fan_copy_records_dev(self._h, ctypes.c_void_p(dev_ptr)))

    # ---- device-function probes (include/geoac_probe.h): need one completed launch ----
    def probe_atmo_1d(self, x):
        x = _arr(x); n = len(x)
        o9 = np.zeros((n, 9)); rho = np.zeros(n)
        self._chk(self.lib.geoac_probe_atmo_1d(self._h, n, _p(x), _p(o9), _p(rho)))
        return o9, rho

    def probe_absorption(self, x, freq):
        x = _arr(x); f = _arr(freq); n = len(x)
        out = np.zeros(n)
        self._chk(self.lib.geoac_probe_absorption(self._h, n, _p(x), _p(f), _p(out)))
        return out

    def probe_absorption_table(self, x):
        """alpha from the absorption table the post-pass of the stratified sets reads (-1 where the table does not serve the point)"""
        x = _arr(x); n = len(x)
        out = np.zeros(n)
        self._chk(self.lib.geoac_probe_absorption_table(self._h, n, _p(x), _p(out)))
        return out

    def abs_table_info(self):
        """of the last launch: table entries (0 = exact evaluation everywhere), entries flagged at build time, path segments evaluated exactly"""
        e, f, n, w = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_uint64(0), ctypes.c_double(0)
        self._chk(self.lib.geoac_abs_table_info(self._h, ctypes.byref(e), ctypes.byref(f), ctypes.byref(n), ctypes.byref(w)))
        return dict(entries=e.value, flagged=f.value, fixup_segments=int(n.value), worst_rel_err=w.value)

    def fan_status(self):
        """condition flags of the last launch: FAN_STEP_LIMIT, and what the context has withdrawn from its plan (FAN_SUB_FALLBACK, FAN_ABS_FALLBACK)"""
        fl = ctypes.c_uint64(0)
        self._chk(self.lib.geoac_fan_status(self._h, ctypes.byref(fl)))
        return int(fl.value)

    def probe_grid(self, a0, a1, a2, coop=False):
        a0, a1, a2 = _arr(a0), _arr(a1), _arr(a2); n = len(a0)
        o30 = np.zeros((n, 30)); a7 = np.zeros((n, 7))
        self._chk(self.lib.geoac_probe_grid(self._h, n, _p(a0), _p(a1), _p(a2), 1 if coop else 0, _p(o30), _p(a7)))
        return o30, a7

    def total_steps(self):
        steps = ctypes.c_uint64(0)
        self._chk(self.lib.geoac_fan_fetch(self._h, None, ctypes.byref(steps)))
        return int(steps.value)

    def run(self, theta_deg, phi_deg, out=None):
        self.set_angles(theta_deg, phi_deg)
        self.launch()
        return self.fetch(out)

    def timing(self):
        ms = (ctypes.c_double * 3)(); st = (ctypes.c_uint64 * 3)()
        self._chk(self.lib.geoac_last_timing(self._h, ms, st))
        return dict(ms_total=ms[0], ms_rk4=ms[1], ms_post=ms[2], epochs=int(st[0]), path_bytes_w=int(st[1]), path_bytes_r=int(st[2]))

    # ---- eigenray searches (include/geoac_eig.h), spherical sets ----
    def _eig_collect(self, res):
        L = self.lib
        L.geoac_eig_count.restype = ctypes.c_int64
        L.geoac_eig_sample_count.restype = ctypes.c_int64
        L.geoac_eig_log.restype = ctypes.c_char_p
        L.geoac_eig_log.argtypes = [ctypes.c_void_p, ctypes.c_int]
        ne, ns = L.geoac_eig_count(res), L.geoac_eig_sample_count(res)
        eig = np.zeros((max(ne, 1), EIG_STRIDE)); smp = np.zeros((max(ns, 1), 10))
        if ne:
            self._chk(L.geoac_eig_fetch(res, _p(eig)))
        if ns:
            self._chk(L.geoac_eig_fetch_samples(res, _p(smp)))
        st = (ctypes.c_uint64 * 8)()
        L.geoac_eig_stats_ex(res, st)
        return dict(eig=eig[:ne], smp=smp[:ns], stats=dict(launches=int(st[0]), rays=int(st[1]), steps=int(st[2]), rounds=int(st[3]),
                                                              critical_steps=int(st[4]), amp_launches=int(st[5])))

    def eig_search(self, receivers, theta_min=0.5, theta_max=45.0, bnc_min=0, bnc_max=0, iterations=25, azimuth_err_lim=2.0, verbose=False):
        """GeoAc's -eig_search for every receiver [lat, lon] (degrees) around the context's source, decision rounds of all receivers
        batched into fan launches.  Returns dict(eig [n][EIG_STRIDE], smp raypath rows, stats, logs)."""
        rc_arr = _arr(receivers).reshape(-1, 2)
        ep = EigParams(theta_min, theta_max, bnc_min, bnc_max, iterations, azimuth_err_lim, 1 if verbose else 0)
        res = ctypes.c_void_p()
        self._chk(self.lib.geoac_eig_search(self._h, ctypes.byref(ep), len(rc_arr), _p(rc_arr), ctypes.byref(res)))
        out = self._eig_collect(res)
        out["logs"] = [self.lib.geoac_eig_log(res, i).decode() for i in range(len(rc_arr))]
        self.lib.geoac_eig_free.argtypes = [ctypes.c_void_p]
        self.lib.geoac_eig_free(res)
        return out

    def eig_direct(self, receivers, theta_est, phi_est, bounces=0, iterations=25, verbose=False):
        """-eig_direct: refinement from given inclination / azimuth-from-north estimates, one per receiver"""
        rc_arr = _arr(receivers).reshape(-1, 2); th = _arr(theta_est); ph = _arr(phi_est)
        ep = EigParams(0.5, 45.0, bounces, bounces, iterations, 2.0, 1 if verbose else 0)
        res = ctypes.c_void_p()
        self._chk(self.lib.geoac_eig_direct(self._h, ctypes.byref(ep), len(rc_arr), _p(rc_arr), _p(th), _p(ph), int(bounces), ctypes.byref(res)))
        out = self._eig_collect(res)
        out["logs"] = [self.lib.geoac_eig_log(res, i).decode() for i in range(len(rc_arr))]
        self.lib.geoac_eig_free.argtypes = [ctypes.c_void_p]
        self.lib.geoac_eig_free(res)
        return out


# ---------------- several GPUs from one process (include/geoac_multi.h) ----------------
class FanPool:
    """geoac_pool: one context per listed device, azimuth groups from a shared queue, records gathered into the caller's table."""

    def __init__(self, eqset, devices, options=None):
        self.lib = load_library()
        self.eqset = eqset
        self.devices = [int(d) for d in devices]
        self._h = ctypes.c_void_p()
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        rc = self.lib.geoac_pool_create(ctypes.byref(self._h), eqset, len(self.devices), arr)
        if rc:
            raise GeoAcError(f"geoac_pool_create: {self.lib.geoac_strerror(rc).decode()}")
        self.lib.geoac_pool_last_error.restype = ctypes.c_char_p
        self.lib.geoac_pool_ctx.restype = ctypes.c_void_p
        self.lib.geoac_pool_ctx.argtypes = [ctypes.c_void_p, ctypes.c_int]
        for i in range(len(self.devices)):
            _apply_options(self.lib, ctypes.c_void_p(self.lib.geoac_pool_ctx(self._h, i)), dict(DEFAULT_OPTIONS, **(options or {})))
        self.params = default_params(eqset)
        self._chk(self.lib.geoac_pool_set_params(self._h, ctypes.byref(self.params)))

    def _chk(self, rc):
        if rc:
            msg = self.lib.geoac_pool_last_error(self._h)
            raise GeoAcError(f"{self.lib.geoac_strerror(rc).decode()}: {msg.decode() if msg else ''}")

    def close(self):
        if self._h:
            self.lib.geoac_pool_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_met(self, path, fmt="zTuvdp"):
        a = met_load(path, self.eqset, fmt)
        x, T, u, v, rho = (_arr(a[k]) for k in ("x", "T", "u", "v", "rho"))
        sl = _arr(np.concatenate([natural_spline_slopes(x, f) for f in (T, u, v, rho)]))
        self._chk(self.lib.geoac_pool_upload_atmo_1d(self._h, len(x), _p(x), _p(T), _p(u), _p(v), _p(rho), _p(sl)))

    def load_grid(self, prefix, locx, locy, fmt="zTuvdp", z_grnd=0.0):
        nx, ny, nz = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        if self.lib.geoac_grid_dims(prefix.encode(), locx.encode(), locy.encode(), ctypes.byref(nx), ctypes.byref(ny), ctypes.byref(nz)):
            raise GeoAcError(f"cannot read grid {prefix}")
        nx, ny, nz = nx.value, ny.value, nz.value
        x, y, z = np.zeros(nx), np.zeros(ny), np.zeros(nz)
        F = [np.zeros((nx, ny, nz)) for _ in range(4)]
        rc = self.lib.geoac_grid_load_eq(self.eqset, prefix.encode(), locx.encode(), locy.encode(), fmt.encode(), ctypes.c_double(z_grnd),
                                      nx, ny, nz, _p(x), _p(y), _p(z), *[_p(f) for f in F])
        if rc:
            raise GeoAcError(f"geoac_grid_load_eq -> {rc}")
        self._chk(self.lib.geoac_pool_upload_atmo_3d(self._h, nx, ny, nz, _p(x), _p(y), _p(z), *[_p(f) for f in F]))

    def set_params(self, **kw):
        for k, val in kw.items():
            if k == "src":
                self.params.src = (ctypes.c_double * 3)(*val)
            elif k == "xy_limits":
                self.params.xy_limits = (ctypes.c_double * 4)(*val)
            else:
                setattr(self.params, k, val)
        self._chk(self.lib.geoac_pool_set_params(self._h, ctypes.byref(self.params)))

    def run(self, theta_deg, phi_deg, rays_per_group=0):
        th, ph = _arr(theta_deg), _arr(phi_deg)
        rec = np.zeros((len(th), self.params.bounces + 1, REC_STRIDE))
        steps = ctypes.c_uint64(0)
        self._chk(self.lib.geoac_pool_fan_run(self._h, len(th), _p(th), _p(ph), int(rays_per_group), _p(rec), ctypes.byref(steps)))
        return rec, int(steps.value)

    def shares(self):
        n = len(self.devices)
        r, s, g = ((ctypes.c_uint64 * n)() for _ in range(3))
        self._chk(self.lib.geoac_pool_last_shares(self._h, r, s, g))
        return dict(rays=list(r), steps=list(s), groups=list(g))

