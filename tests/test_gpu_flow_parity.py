"""-m gpu: velocity-field parity of whole time steps with the IBM active -- the channel with an immersed sphere (BASELINE config 4's set-up) through
the C host mirror against the oracle's step, v / V / p field by field, and SURVEY 8(d)'s continuity bound || D V ||_inf <= 10 rtol || b ||_inf.
The same function fills bench.py's configs.flow_step.parity at 128^3."""
import pytest

from tests import flow_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,D", [(32, 8), (48, 10)])   # (bench.py runs the same check at 128^3 in every bench line: configs.flow_step.parity)
def test_channel_with_immersed_sphere_matches_the_oracle_step(n, D):
    r = flow_parity.channel_sphere(n=n, nsteps=2, diameter_cells=D)
    assert r["max_abs_v"] > 0.5 and r["rms_speed_at_markers_over_inflow_peak"] < 0.9        # a flow, and a body that slows it down
    assert r["rel_l2_diff_v"] <= 1e-6 and r["rel_l2_diff_p"] <= 1e-5, r
    assert max(r["rel_l2_diff_V"][:2]) <= 1e-6, r
    assert r["div_inf"] <= r["div_bound_10_rtol_b_inf"], r
