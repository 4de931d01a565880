"""GPU frames against the compiled ETSI REFERENCE (not the restatement) over the five BASELINE configurations, reduced size (the full-size
run: tools/ref_soak.py, recorded in profiles/): frames are compared byte for byte; where run-time libm calls flip a decision between the
device and glibc the stream's two bitstreams are decoded by the reference decoder and the ETSI mld tool must stay
within the conformance procedure's threshold of 4 (E/conformance/lc3_conformance.py:126-129)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_reference_soak_reduced():
    import ref_soak
    if not (ref_soak.have_ref() and os.path.exists(ref_soak.REF_BENCH)):
        pytest.skip("oracle/_ref did not travel")
    res = ref_soak.run(scale=0.03, verbose=False)
    tot = sum(r["channel_frames"] for r in res); diff = sum(r["frames_differ"] for r in res)
    assert tot > 10000
    assert {r["config"] for r in res} >= {"c0", "c1", "c3", "c4", "c5", "b1", "b2"}          # b1 / b2: set and switched bandwidths, band-limited input
    # Gate: ZERO differing frames (every run recorded so far: profiles/r0*_ref_soak.txt).  The one explained exception is a libm-boundary flip
    # (DESIGN.md section 4): it must have been MEASURED - the differing stream decoded twice by the reference decoder, ETSI mld <= 4
    # (E/conformance/lc3_conformance.py:126-129) - and there may be at most one such stream per configuration; without the mld tool a difference fails.
    for r in res:
        if r["frames_differ"] == 0: continue
        assert r["worst_mld"] is not None, ("frames differ and the mld tool did not travel", r)
        assert r["worst_mld"] <= 4.0 and r["streams_differ"] <= 1 and r["frames_differ"] <= 2, r
    assert diff <= 2, (diff, tot, res)
