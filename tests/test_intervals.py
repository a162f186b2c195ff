"""CPU: the interval algebra against values produced by the reference's own
methods (oracle/make_goldens_n1.py executed TargetDiarization.py:185-548)."""
import json
import os

import pytest

from targetdiarization_amd import intervals as iv


@pytest.fixture(scope="module")
def g(gold):
    return json.load(open(os.path.join(gold, "n1_intervals.json")))


def T(x):      # json lists -> tuples, recursively, for == against tuple-producing functions
    if isinstance(x, list):
        return [T(i) for i in x] if (x and isinstance(x[0], list)) else (tuple(x) if x and not isinstance(x[0], (list, dict)) else [T(i) for i in x])
    if isinstance(x, dict):
        return {k: T(v) for k, v in x.items()}
    return x


def L(x):      # everything to plain lists for structural comparison
    if isinstance(x, (list, tuple)):
        return [L(i) for i in x]
    if isinstance(x, dict):
        return {k: L(v) for k, v in x.items()}
    return x


def test_merge_subtract(g):
    for c in g["merge"]:
        assert L(iv.merge_timeranges([tuple(r) for r in c["in"]])) == c["out"]
    for c in g["subtract"]:
        assert L(iv.subtract_timeranges([tuple(r) for r in c["base"]], [tuple(r) for r in c["sub"]])) == c["out"]
    assert iv.subtract_timeranges([(0, 10)], []) == []          # the reference's quirk


def test_iou_family(g):
    for c in g["single_iou"]:
        assert iv.calc_single_iou(c["p"], c["g"]) == pytest.approx(c["out"], abs=1e-12)
    for c in g["multi_iou"]:
        assert iv.calc_multi_iou(c["p"], c["g"], c["m"]) == pytest.approx(c["out"], abs=1e-12)
    for c in g["iou_score"]:
        got = iv.calc_iou_score(c["p"], c["g"], c.get("pw", 1.0), c.get("nw", 1.0))
        assert got == pytest.approx(c["out"], abs=1e-12)
    with pytest.raises(ValueError):
        iv.calc_multi_iou([], [(0, 1)])


def test_key_matcher_overlap_apply(g):
    for c in g["key_matcher"]:
        out = iv.sd_key_matcher(c["src"], dict(c["tgt"]))
        assert L(out) == c["out"] and list(out.keys()) == list(c["out"].keys())
    for c in g["overlap"]:
        assert L(iv.get_speaker_overlap(T(c["sd"]))) == c["out"]
    for c in g["apply_od"]:
        refined, omap = iv.apply_od_result({k: [tuple(r) for r in v] for k, v in c["sd"].items()},
                                           {k: [tuple(r) for r in v] for k, v in c["od"].items()})
        assert L(refined) == c["refined"] and L(omap) == c["omap"]
    for c in g["subtract_overlap"]:
        sd = {k: [tuple(r) for r in v] for k, v in c["sd"].items()}
        om = [[tuple(i) for i in items] for items in c["omap"]]
        assert L(iv.subtract_overlap(sd, om)) == c["single"]
        assert L(iv.subtract_overlap(sd, om, reverse_output=True)) == c["overlap"]


def test_speaker_num_and_parser(g):
    for c in g["speaker_num"]:
        assert iv.get_speaker_num(c["sd"], c["thr"]) == c["out"]
    for c in g["sd_parser"]:
        out = iv.sd_result_parser({"text": [list(r) for r in c["rows"]]}, c["single"], c["comb"])
        assert L(out) == c["out"]


def _rle(a):
    if a is None:
        return None
    out, i = [], 0
    while i < len(a):
        j = i
        while j < len(a) and a[j] == a[i]:
            j += 1
        out.append([float(a[i]), j - i]); i = j
    return out


def test_assembly_functions_match_reference_goldens():
    """combine_audio_chunks / asr_audio_parser / od_result_parser of the orchestrator against outputs
    of the reference's own methods (oracle/make_goldens_n1.py::assembly)."""
    import numpy as np
    from targetdiarization_amd.target_diarization import TargetDiarization
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "n1_assembly.json")))
    td = object.__new__(TargetDiarization)

    def mk(items):
        return [dict({k: v for k, v in it.items() if k != "alen"}, alen=it["alen"], audio=np.full(it["alen"], i + 1, dtype=np.float32))
                for i, it in enumerate(items)]
    for c in g["combine"]:
        assert _rle(td.combine_audio_chunks(mk(c["items"]), c["spk"])) == c["out"]
    for c in g["asr_parser"]:
        res, aud = td.asr_audio_parser(mk(c["items"]), c["spk"], c["flag"])
        assert _rle(aud) == c["out"]
        assert json.loads(json.dumps(res)) == c["res"]
    for c in g["od_parser"]:
        out = td.od_result_parser([tuple(r) for r in c["rows"]], {k: [tuple(x) for x in v] for k, v in c["sd"].items()}, c["single"], c["ov"])
        assert json.loads(json.dumps(out)) == c["out"]
