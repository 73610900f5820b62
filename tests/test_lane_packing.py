"""Host logic of the lane-to-stream packing (vts/engine.py:_lane_groups): the part runs four hardware queues
side by side, so the discriminator lanes of a phase are packed into four streams, longest processing time first."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))


@pytest.fixture()
def engine(monkeypatch):
    for k in ("VTS_LANE_GROUPS", "VTS_LANE_GROUPS_G", "VTS_LANE_STREAMS"):
        monkeypatch.delenv(k, raising=False)
    from vts import engine as e

    return importlib.reload(e)


# the cost estimates of the headline step's discriminator updates: D1 at three scales (N = 8), D2 at three scales, the L1 terms
HEADLINE = [1.19, 0.56, 0.40, 0.835, 0.47, 0.38, 0.1]


def test_headline_lanes_pack_into_four_streams(engine):
    groups = engine._lane_groups(HEADLINE)
    assert groups == [[0], [3, 6], [1, 5], [4, 2]]
    assert sorted(i for g in groups for i in g) == list(range(7))           # every lane exactly once
    loads = [sum(HEADLINE[i] for i in g) for g in groups]
    assert max(loads) == HEADLINE[0]                                         # nothing is packed behind the heaviest lane
    assert all(HEADLINE[g[k]] >= HEADLINE[g[k + 1]] for g in groups for k in range(len(g) - 1))   # the heavier lane of a stream first


def test_few_lanes_keep_their_own_stream(engine):
    assert engine._lane_groups([0.5, 0.4, 0.3, 0.2]) == [[0], [1], [2], [3]]      # pix2pixHD: two discriminators x two scales
    assert engine._lane_groups([0.5]) == [[0]]


def test_a_heavy_extra_lane_gets_a_stream_of_its_own(engine):
    """the LPIPS workload: the perceptual terms (~90 ms) ride as the extra lane"""
    groups = engine._lane_groups(HEADLINE[:6] + [90.0])
    assert [6] in groups and groups[0] == [0]


def test_overrides(engine, monkeypatch):
    monkeypatch.setenv("VTS_LANE_GROUPS", "0|1,5|3|2,4")
    assert engine._lane_groups(HEADLINE) == engine._lane_groups(HEADLINE, "VTS_LANE_GROUPS_G")      # ignored without VTS_TUNING=1 (vts/tune.py)
    monkeypatch.setenv("VTS_TUNING", "1")
    monkeypatch.setenv("VTS_LANE_GROUPS", "0|1,5|3|2,4")
    assert engine._lane_groups(HEADLINE) == [[0], [1, 5], [3], [2, 4], [6]]       # a lane the spec does not name keeps its own stream
    assert engine._lane_groups(HEADLINE, "VTS_LANE_GROUPS_G") != [[0], [1, 5], [3], [2, 4], [6]]     # the generator step has its own variable
    monkeypatch.setenv("VTS_LANE_GROUPS", "0,1|1,2")
    with pytest.raises(ValueError):
        engine._lane_groups(HEADLINE)
    monkeypatch.delenv("VTS_LANE_GROUPS")
    monkeypatch.setenv("VTS_LANE_STREAMS", "0")
    e = importlib.reload(engine)
    assert e._lane_groups(HEADLINE) == [[i] for i in range(7)]                    # round 3: one stream per lane
