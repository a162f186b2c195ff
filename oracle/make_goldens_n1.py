"""Mint tests/golden/n1_intervals.json by EXECUTING the reference's own interval-algebra methods
(TargetDiarization.py:185-548) in the build container.  Test infrastructure.

TargetDiarization.py imports third-party packages that are absent here (modelscope, pyannote,
dotenv) and the sibling wrappers TargetASR / AudioProcessor (which import more).  None of them
is touched by the pure-Python methods exercised below, so empty placeholder modules are put in
sys.modules for the import to succeed and the class is instantiated WITHOUT running __init__.
"""
import json
import os
import random
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_reference_class():
    for name in ("modelscope", "pyannote", "pyannote.audio", "dotenv", "TargetASR", "AudioProcessor"):
        m = types.ModuleType(name)
        sys.modules.setdefault(name, m)
    sys.modules["pyannote.audio"].Pipeline = object
    sys.modules["dotenv"].load_dotenv = lambda *a, **k: None
    sys.modules["TargetASR"].TargetASR = object
    sys.modules["AudioProcessor"].AudioProcessor = object
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_td", "/root/reference/TargetDiarization.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return object.__new__(mod.TargetDiarization)


def rand_ranges(rng, n, span=60.0, maxlen=6.0):
    out = []
    for _ in range(n):
        s = round(rng.uniform(0, span), 3)
        out.append((s, round(s + rng.uniform(0.05, maxlen), 3)))
    return out


def main():
    ref = load_reference_class()
    rng = random.Random(7)
    cases = {"merge": [], "subtract": [], "single_iou": [], "multi_iou": [], "iou_score": [], "key_matcher": [], "overlap": [],
             "apply_od": [], "subtract_overlap": [], "speaker_num": [], "sd_parser": []}
    for _ in range(40):
        a = rand_ranges(rng, rng.randint(1, 8))
        cases["merge"].append({"in": a, "out": ref.merge_timeranges([tuple(x) for x in a])})
    for _ in range(40):
        base = ref.merge_timeranges(rand_ranges(rng, rng.randint(1, 6)))
        sub = rand_ranges(rng, rng.randint(0, 6))
        cases["subtract"].append({"base": base, "sub": sub, "out": ref.subtract_timeranges(list(base), [tuple(x) for x in sub])})
    for _ in range(40):
        p, g = rand_ranges(rng, 1)[0], rand_ranges(rng, 1, span=10)[0]
        if rng.random() < 0.3:
            p = (p[1], p[0])
        cases["single_iou"].append({"p": p, "g": g, "out": ref.calc_single_iou(list(p), list(g))})
    for _ in range(40):
        ps, gs = rand_ranges(rng, rng.randint(1, 5), span=20), rand_ranges(rng, rng.randint(1, 5), span=20)
        for m in ("pred_to_gt", "gt_to_pred", "both_mean"):
            cases["multi_iou"].append({"p": ps, "g": gs, "m": m, "out": float(ref.calc_multi_iou(ps, gs, m))})
        cases["iou_score"].append({"p": ps, "g": gs, "out": float(ref.calc_iou_score(ps, gs))})
        cases["iou_score"].append({"p": ps, "g": gs, "pw": 0.0, "nw": 1.0, "out": float(ref.calc_iou_score(ps, gs, 0.0, 1.0))})

    def rand_sd(nspk, span=40.0):
        return {str(k): ref.merge_timeranges(rand_ranges(rng, rng.randint(1, 5), span=span)) for k in range(nspk)}

    for _ in range(30):
        src, tgt = rand_sd(rng.randint(1, 3)), rand_sd(rng.randint(1, 4))
        tgt = {str(int(k) + 5): v for k, v in tgt.items()}
        cases["key_matcher"].append({"src": src, "tgt": tgt, "out": ref.sd_key_matcher(src, dict(tgt))})
    for _ in range(30):
        sd = rand_sd(rng.randint(1, 4), span=25.0)
        ov = ref.get_speaker_overlap(sd)
        cases["overlap"].append({"sd": sd, "out": ov})
        refined, omap = ref.apply_od_result({k: list(v) for k, v in sd.items()}, ov)
        cases["apply_od"].append({"sd": sd, "od": ov, "refined": refined, "omap": omap})
        cases["subtract_overlap"].append({"sd": refined, "omap": omap, "single": ref.subtract_overlap(refined, omap),
                                          "overlap": ref.subtract_overlap(refined, omap, reverse_output=True)})
        for thr in (0.0, 1.0, 3.0):
            cases["speaker_num"].append({"sd": sd, "thr": thr, "out": ref.get_speaker_num(sd, thr)})
    for _ in range(20):
        rows = [[round(rng.uniform(0, 30), 2), 0, rng.randint(0, 3)] for _ in range(rng.randint(0, 8))]
        for r in rows:
            r[1] = round(r[0] + rng.uniform(0.1, 4), 2)
        for single in (False, True):
            for comb in (False, True):
                cases["sd_parser"].append({"rows": rows, "single": single, "comb": comb,
                                           "out": ref.sd_result_parser({"text": [list(r) for r in rows]}, single, comb)})
    json.dump(cases, open(os.path.join(ROOT, "tests", "golden", "n1_intervals.json"), "w"))
    print({k: len(v) for k, v in cases.items()})
    assembly(ref, rng)


class _Seg:
    def __init__(self, s, e):
        self.start, self.end = s, e


class _Annotation:
    """the only part of pyannote's Annotation that od_result_parser touches (:233)"""
    def __init__(self, rows):
        self.rows = rows

    def __bool__(self):
        return bool(self.rows)

    def itertracks(self, yield_label=True):
        for i, (s, e, lab) in enumerate(self.rows):
            yield _Seg(s, e), i, lab


def assembly(ref, rng):
    """combine_audio_chunks (:822), asr_audio_parser (:841), od_result_parser (:230): audio clips are
    deterministic ramps (value = item index + 1) so the expected timeline is stored run-length coded."""
    import numpy as np

    def rle(a):
        if a is None:
            return None
        out, i = [], 0
        while i < len(a):
            j = i
            while j < len(a) and a[j] == a[i]:
                j += 1
            out.append([float(a[i]), j - i]); i = j
        return out

    cases = {"combine": [], "asr_parser": [], "od_parser": []}
    for _ in range(30):
        n, t, items = rng.randint(0, 7), 0.0, []
        for i in range(n):
            s = round(t + rng.uniform(-0.3, 1.0), 3) if i else round(rng.uniform(0, 1.0), 3)
            s = max(s, 0.0)
            e = round(s + rng.uniform(0.05, 1.2), 3)
            alen = int((e - s) * 16000) + rng.choice([0, 0, -7, 13])
            items.append({"speaker": str(rng.randint(0, 2)), "timerange": [s, e], "text": "", "type": "single", "alen": max(alen, 1)})
            t = e
        def mk():
            return [dict(it, audio=np.full(it["alen"], i + 1, dtype=np.float32)) for i, it in enumerate(items)]
        for spk in ("0", "1"):
            cases["combine"].append({"items": items, "spk": spk, "out": rle(ref.combine_audio_chunks(mk(), spk))})
            for flag in (True, False):
                res, aud = ref.asr_audio_parser(mk(), spk, flag)
                cases["asr_parser"].append({"items": items, "spk": spk, "flag": flag, "res": res, "out": rle(aud)})
    for _ in range(30):
        rows = []
        for _ in range(rng.randint(0, 7)):
            s = rng.uniform(0, 20)
            rows.append((s, s + rng.uniform(0.2, 5), "SPEAKER_%02d" % rng.randint(0, 2)))
        sd = {str(k): ref.merge_timeranges(rand_ranges(rng, rng.randint(1, 4), span=20.0)) for k in range(rng.randint(1, 3))}
        for single in (False, True):
            for ov in (False, True):
                for use_sd in (False, True):
                    out = ref.od_result_parser(_Annotation(rows), {k: list(v) for k, v in sd.items()} if use_sd else {}, single, ov)
                    cases["od_parser"].append({"rows": rows, "sd": sd if use_sd else {}, "single": single, "ov": ov, "out": out})
    json.dump(cases, open(os.path.join(ROOT, "tests", "golden", "n1_assembly.json"), "w"))
    print({k: len(v) for k, v in cases.items()})


if __name__ == "__main__":
    main()
