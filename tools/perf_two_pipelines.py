"""config 3 as TWO fans on one GPU at once: the shallow share of the inclinations (the long rays) on one context, the rest on another, each with its own epoch pipeline,
one host thread each - does the chip stay full to the end that way?  Against the whole fan on one context.  usage: perf_two_pipelines.py [share ...]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import geoac_amd as G
import harness as H
th, ph = G.fan_enumerate(theta_min=0.25, theta_max=45.0, theta_step=0.25, phi_min=-180.0, phi_max=179.5, phi_step=0.5)
params = dict(bounces=3, calc_amp=1, mode=0, src=(0.0, 30.0, 0.0))


def make(sel, opts):
    ctx = G.FanContext(G.EQ_GLOBAL, device=0, options=opts); ctx.load_met(H.TOYATMO); ctx.set_params(**params)
    ctx.set_angles(th[sel], ph[sel]); ctx.launch()
    return ctx


whole = make(np.ones(len(th), bool), {})
ts = []
for _ in range(3):
    t0 = time.perf_counter(); whole.launch(); ts.append(time.perf_counter() - t0)
rec_ref, steps_ref = whole.fetch(); rec_ref = rec_ref.copy()
print("one context: %.1f ms per pass, %d steps -> %.3e ray-steps/s" % (min(ts) * 1e3, steps_ref, steps_ref / min(ts)), flush=True)
whole.close()
for share in ([float(a) for a in sys.argv[1:]] or [0.04, 0.07, 0.1, 0.15]):
    cut = np.sort(np.unique(th))[max(1, int(round(share * 180))) - 1]
    a_sel = th <= cut
    B = make(~a_sel, {"S_ROWS": "6904"}); A = make(a_sel, {"S_ROWS": "6904"})
    ts = []
    for _ in range(3):
        bar = threading.Barrier(3); done = []
        def run(c):
            bar.wait(); c.launch(); done.append(time.perf_counter())
        tA = threading.Thread(target=run, args=(A,)); tB = threading.Thread(target=run, args=(B,)); tA.start(); tB.start()
        bar.wait(); t0 = time.perf_counter(); tA.join(); tB.join(); ts.append(max(done) - t0)
    ra, sa = A.fetch(); rb, sb = B.fetch()
    rec = np.empty_like(rec_ref); rec[a_sel] = ra; rec[~a_sel] = rb
    print("two contexts, inclinations <= %.2f deg (%d rays) apart: %.1f ms per pass (A alone-in-company %.1f ms rk4, B %.1f), %d steps -> %.3e ray-steps/s, records bit-identical: %s"
          % (cut, int(a_sel.sum()), min(ts) * 1e3, A.timing()["ms_total"], B.timing()["ms_total"], sa + sb, (sa + sb) / min(ts), bool(np.array_equal(rec.view(np.uint64), rec_ref.view(np.uint64)))), flush=True)
    A.close(); B.close()
