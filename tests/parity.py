"""Parity rule between the HIP path and the checker (oracle / golden vectors), per BASELINE.json:
bit-exact on counts (legs, arrivals, steps), <= 1e-6 relative on floating-point fields."""
import numpy as np

from harness import REC

RTOL = 1e-6       # BASELINE.json north_star: "within 1e-6 relative on travel-time/range/amplitude"


def _rel(a, b, floor):
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


HEIGHT_COMP = {18: None, 6: None, 12: 2, 4: 2, 3: 1}       # E -> index of the height component (Cartesian sets); 2D amp-on E=6 handled by caller


def _state_scale(st_w, want, valid, hidx):
    """per-component scale of the end state over the fan.  The arrival height of the Cartesian sets is the first
    sub-ground sample, a residual of order 1e-4 km of a trajectory that spans ~100 km: its error is judged against
    the turning height, not against itself."""
    scale = np.maximum(np.abs(st_w).max(axis=0, keepdims=True), 1e-30)
    if hidx is not None:
        scale[0, hidx] = max(scale[0, hidx], np.abs(want[..., REC["TURN"]][valid]).max())
    return scale


def compare_records(got, want, E, rtol=RTOL, check_amp=True, hidx="auto"):
    """got/want: [n_rays][legs][32] record tables."""
    if hidx == "auto":
        hidx = HEIGHT_COMP.get(E)
    assert got.shape == want.shape
    # ---- integer-valued fields: exact ----
    for f in ("VALID", "STEPS", "BROKE"):
        assert np.array_equal(got[..., REC[f]], want[..., REC[f]]), f"{f} differs"
    ran = want[..., REC["STEPS"]] > 0
    valid = want[..., REC["VALID"]] > 0
    # ---- cumulative sums exist for every leg that ran ----
    for f, floor in (("TTIME", 1e-3), ("ATTEN", 1e-12)):
        r = _rel(got[..., REC[f]][ran], want[..., REC[f]][ran], floor)
        assert r.size == 0 or r.max() <= rtol, f"{f}: max rel err {r.max():.3e}"
    # ---- arrival fields ----
    for f, floor in (("TURN", 1e-3), ("RANGE", 1e-3)):
        r = _rel(got[..., REC[f]][valid], want[..., REC[f]][valid], floor)
        assert r.size == 0 or r.max() <= rtol, f"{f}: max rel err {r.max():.3e}"
    for f in ("INCL", "BACKAZ"):          # degrees: absolute 1e-6 * 180
        d = np.abs(got[..., REC[f]][valid] - want[..., REC[f]][valid])
        assert d.size == 0 or d.max() <= rtol * 180.0, f"{f}: max abs err {d.max():.3e}"
    st_g = got[..., REC["STATE"]:REC["STATE"] + E][valid]
    st_w = want[..., REC["STATE"]:REC["STATE"] + E][valid]
    if st_w.size:
        scale = _state_scale(st_w, want, valid, hidx)                          # per-component scale over the fan
        r = np.abs(st_g - st_w) / np.maximum(np.abs(st_w), 1e-3 * scale)
        assert r.max() <= rtol, f"end state: max rel err {r.max():.3e} at comp {np.unravel_index(r.argmax(), r.shape)}"
    if check_amp and E > 6:
        for f in ("AMP", "JACOB"):
            r = _rel(got[..., REC[f]][valid], want[..., REC[f]][valid], 1e-300)
            assert r.size == 0 or r.max() <= rtol, f"{f}: max rel err {r.max():.3e}"


def max_rel_errors(got, want, E, hidx="auto"):
    """diagnostic: dict of max relative errors per field (no asserts)."""
    if hidx == "auto":
        hidx = HEIGHT_COMP.get(E)
    out = {}
    ran = want[..., REC["STEPS"]] > 0
    valid = want[..., REC["VALID"]] > 0
    for f in ("VALID", "STEPS", "BROKE"):
        out[f] = int(np.sum(got[..., REC[f]] != want[..., REC[f]]))
    for f in ("TTIME", "ATTEN"):
        out[f] = float(_rel(got[..., REC[f]][ran], want[..., REC[f]][ran], 1e-300).max()) if ran.any() else 0.0
    for f in ("TURN", "RANGE", "INCL", "BACKAZ", "AMP", "JACOB"):
        out[f] = float(_rel(got[..., REC[f]][valid], want[..., REC[f]][valid], 1e-300).max()) if valid.any() else 0.0
    st_g = got[..., REC["STATE"]:REC["STATE"] + E][valid]
    st_w = want[..., REC["STATE"]:REC["STATE"] + E][valid]
    if st_w.size:
        scale = _state_scale(st_w, want, valid, hidx)
        out["STATE"] = float((np.abs(st_g - st_w) / np.maximum(np.abs(st_w), 1e-3 * scale)).max())
    return out
