"""Only where oracle/_ref exists (this container): the oracle against the LIVE compiled reference on
inputs beyond the committed goldens."""
import numpy as np
import pytest

import harness as H

EQS = [H.EQ_GLOBAL, H.EQ_3D, H.EQ_2D]


@pytest.mark.ref
@pytest.mark.parametrize("eq", EQS)
def test_random_fan_bitexact_vs_compiled_reference(eq):
    if not H.ref_available(eq):
        pytest.skip("compiled reference not present (oracle/_ref is built only where /root/reference exists)")
    rng = np.random.default_rng(7 + eq)
    th = rng.uniform(0.5, 60.0, 8)
    ph = rng.uniform(-180.0, 180.0, 8)
    O, R = H.Oracle(eq), H.RefShim(eq)
    for amp, mode in ((True, 0), (False, 1)):
        cfg = H.make_cfg(eq, bounces=1, calc_amp=amp, mode=mode)
        so, ro, _, _ = O.fan(cfg, th, ph)
        sr, rr, _, _ = R.fan(cfg, th, ph)
        assert so == sr
        assert np.array_equal(ro, rr)
