"""Host logic of the bench line (no GPU): which figure becomes `value` of a multi-rank line."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _line():
    return {"value": 2400.0, "ms_per_step": 1 / 2.4, "unit": "steps/s", "steps": 20, "energy_per_particle": -5.56, "kT_final": 0.98,
            "particle_steps_per_s": 2400.0 * 131072, "config": {"global_particles": 131072},
            "graph_variant": {"skipped": "transport 'native' is not available"},
            "graph_variant_peer": {"value": 8200.0, "ms_per_step": 1 / 8.2, "steps": 20, "energy_per_particle": -5.57, "kT": 0.985,
                                   "particles": 131072, "halo": {"transport": "peer"}}}


def test_a_verified_replay_becomes_the_value_and_the_eager_figure_stays():
    from benchlib import multirank
    d = multirank.promote_verified(_line())
    assert d["value"] == 8200.0 and d["eager"]["value"] == 2400.0 and "graph_variant_peer" in d["value_path"]
    assert abs(d["particle_steps_per_s"] - 8200.0 * 131072) < 1e-6 and abs(d["eager"]["particle_steps_per_s"] - 2400.0 * 131072) < 1e-6
    assert abs(d["ms_per_step"] - 1 / 8.2) < 1e-12


def test_a_replay_that_does_not_match_is_not_promoted():
    from benchlib import multirank
    for change in ({"kT": 1.3}, {"energy_per_particle": -4.9}, {"steps": 25}, {"value": 2000.0}, {"kT": float("nan")}):
        line = _line()
        line["graph_variant_peer"].update(change)
        d = multirank.promote_verified(copy.deepcopy(line))
        assert d["value"] == 2400.0 and "eager" not in d and d["value_path"].startswith("eager loop"), change
    line = _line()
    line["graph_variant_peer"] = {"skipped": "a phase timed out"}
    d = multirank.promote_verified(line)
    assert d["value"] == 2400.0 and "eager" not in d


def test_the_faster_of_two_verified_replays_wins():
    from benchlib import multirank
    line = _line()
    line["graph_variant"] = dict(line["graph_variant_peer"], value=9100.0, ms_per_step=1 / 9.1, halo={"transport": "native"})
    d = multirank.promote_verified(line)
    assert d["value"] == 9100.0 and "'native'" in d["value_path"]
