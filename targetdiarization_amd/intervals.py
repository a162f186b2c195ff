"""Interval algebra of the diarization orchestrator: the pure-Python part of
TargetDiarization.py that prepares the hot path's inputs (which segments are single-speaker,
which overlap, whose key is whose) — restated function by function, quirks included, and
pinned by tests/golden/n1_intervals.json (minted by executing the reference's own methods,
oracle/make_goldens_n1.py).  Time ranges are (start, end) tuples in seconds.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

Range = Tuple[float, float]


def merge_timeranges(timeranges: List[Range]) -> List[Range]:
    """union of ranges that touch or overlap, sorted (sorts the input in place like the reference).
    TargetDiarization.py:395-407"""
    if not timeranges:
        return []
    timeranges.sort(key=lambda r: r[0])
    out = [timeranges[0]]
    for s, e in timeranges[1:]:
        ls, le = out[-1]
        if s <= le:
            out[-1] = (ls, max(le, e))
        else:
            out.append((s, e))
    return out


def subtract_timeranges(base: Sequence[Range], sub: List[Range]) -> List[Range]:
    """base minus the union of sub.  Quirk kept: an EMPTY `sub` returns `sub` itself (i.e. []),
    not `base`.  TargetDiarization.py:410-430"""
    if not sub:
        return sub
    sub = merge_timeranges(sub)
    out = []
    for bs, be in base:
        cur = bs
        for ss, se in sub:
            if cur >= se:
                continue
            if be <= ss:
                break
            lo, hi = max(cur, ss), min(be, se)
            if lo < hi:
                if lo > cur:
                    out.append((cur, lo))
                cur = hi
        if cur < be:
            out.append((cur, be))
    return out


def calc_single_iou(pred: Sequence[float], gt: Sequence[float]) -> float:
    """IoU of two intervals over the hull (not the union of lengths).  :249-265"""
    if len(pred) != 2 or len(gt) != 2:
        raise ValueError("Length of pred_duration and gt_duration should be 2.")
    p0, p1 = (pred[0], pred[1]) if pred[0] <= pred[1] else (pred[1], pred[0])
    g0, g1 = (gt[0], gt[1]) if gt[0] <= gt[1] else (gt[1], gt[0])
    if p1 <= g0 or g1 <= p0:
        return 0.0
    return (min(p1, g1) - max(p0, g0)) / (max(p1, g1) - min(p0, g0))


def calc_multi_iou(preds, gts, method: str = "both_mean") -> float:
    """mean best-match IoU, pred->gt, gt->pred or their mean.  :268-298"""
    if len(preds) == 0 or len(gts) == 0:
        raise ValueError("Length of pred_durations and gt_durations cannot be zero.")
    p2g = [max(calc_single_iou(p, g) for g in gts) for p in preds]
    g2p = [max(calc_single_iou(p, g) for p in preds) for g in gts]
    a = sum(p2g) / len(p2g)
    b = sum(g2p) / len(g2p)
    if method == "pred_to_gt":
        return a
    if method == "gt_to_pred":
        return b
    return (a + b) / 2.0


def calc_iou_score(preds, gts, positive_weight: float = 1.0, negative_weight: float = 1.0) -> float:
    """IoU-based agreement of two segmentations with a penalty for predicted time outside the
    ground truth.  Quirk kept: every inside piece contributes iou*length_ratio AND iou.  :301-362"""
    if len(preds) == 0 or len(gts) == 0:
        raise ValueError("Length of pred_durations and gt_durations cannot be zero.")

    def dedup(ds):
        u = []
        for d in ds:
            if not any(x[0] == d[0] and x[1] == d[1] for x in u):
                u.append(d)
        return sorted(u, key=lambda x: x[0])

    inside, outside = [], []
    for g in gts:
        for p in preds:
            if p[0] >= g[0] and p[1] <= g[1]:
                inside.append(p)
                break
            elif p[0] < g[0] < p[1]:
                outside.append([p[0], g[0]])
                if g[0] < p[1] <= g[1]:
                    inside.append([g[0], p[1]])
                else:
                    inside.append([g[0], g[1]])
                    outside.append([g[1], p[1]])
                break
            elif p[0] < g[1] < p[1]:
                inside.append([p[0], g[1]])
                outside.append([g[1], p[1]])
                break
    for p in preds:
        touches = any((p[0] < g[0] < p[1]) or (p[0] < g[1] < p[1]) or (g[0] <= p[0] and p[1] <= g[1]) for g in gts)
        if not touches:
            outside.append(p)
    inside, outside = dedup(inside), dedup(outside)
    total_in = sum(d[1] - d[0] for d in inside)
    pos = 0.0
    for d in inside:
        iou = calc_multi_iou([d], gts, method="pred_to_gt")
        pos = pos + iou * ((d[1] - d[0]) / total_in)
        pos = pos + iou
    gt_sum = 0.0
    for g in gts:
        gt_sum = gt_sum + (g[1] - g[0])
    neg = 0.0
    for d in outside:
        neg = neg + (d[1] - d[0]) / gt_sum
    score = pos * positive_weight - neg * negative_weight
    if positive_weight == 0.0:
        score = abs(score)
    return max(0.0, min(score, 1.0))


def sd_key_matcher(source_sd: Dict[str, list], target_sd: Dict[str, list]) -> Dict[str, list]:
    """rename target_sd's speaker keys to the best-matching source_sd keys (greedy, in source
    order); unmatched target keys keep their name unless it is already taken.  :365-392"""
    mapping, used = {}, []
    for s_spk in source_sd:
        best, best_t = 0.0, None
        for t_spk in target_sd:
            if t_spk in used:
                continue
            sc = calc_iou_score(source_sd[s_spk], target_sd[t_spk])
            if sc > best:
                best, best_t = sc, t_spk
        if best_t:
            mapping[best_t] = s_spk
            used.append(best_t)
    if not mapping:
        return target_sd
    out = {}
    for t_spk, s_spk in mapping.items():
        out[s_spk] = target_sd[t_spk]
    for t_spk in target_sd:
        if t_spk not in mapping and t_spk not in out:
            out[t_spk] = target_sd[t_spk]
    return out


def get_speaker_overlap(result: Dict[str, list], min_overlap_sec: float = 0.4) -> Dict[str, list]:
    """pairwise overlaps >= min_overlap_sec, keyed "a-b" in key order.  :521-548"""
    out = {}
    if len(result) == 1:
        return out
    keys = list(result.keys())
    for i in range(len(keys) - 1):
        for j in range(i + 1, len(keys)):
            ov = []
            for s1, e1 in [(r[0], r[1]) for r in result[keys[i]]]:
                for s2, e2 in [(r[0], r[1]) for r in result[keys[j]]]:
                    lo, hi = max(s1, s2), min(e1, e2)
                    if lo < hi and hi - lo >= min_overlap_sec:
                        ov.append((lo, hi))
            if ov:
                out[f"{keys[i]}-{keys[j]}"] = ov
    return out


def apply_od_result(sd_result: Dict[str, list], od_result: Dict[str, list] = {}):
    """split the diarization by the overlap regions: every speaker of an "a-b" pair receives the
    overlap ranges, every speaker keeps its ranges minus ALL overlap regions; overlap_map lists,
    per overlap range, the (speaker, index) entries that equal it.  :433-470"""
    if not od_result:
        return sd_result, []
    refined: Dict[str, list] = {}
    ov_ranges: list = []
    all_ov: list = []
    for ranges in od_result.values():
        all_ov.extend(ranges)
    all_ov = merge_timeranges(all_ov)
    for pair, ranges in od_result.items():
        for spk in pair.split("-"):
            refined.setdefault(spk, []).extend(ranges)
        if ranges not in ov_ranges:          # list-in-list test of the reference: (almost) always true
            ov_ranges.extend(ranges)
    for spk, ranges in sd_result.items():
        if not ranges:
            continue
        refined.setdefault(spk, []).extend(subtract_timeranges(ranges, all_ov))
    for spk in refined:
        refined[spk].sort(key=lambda r: r[0])
    omap = []
    for ovr in ov_ranges:
        hits = [(spk, i) for spk, rs in refined.items() for i in range(len(rs)) if rs[i] == ovr]
        if hits:
            omap.append(hits)
    return refined, omap


def subtract_overlap(sd_result: Dict[str, list], overlap_map: list = [], reverse_output: bool = False) -> Dict[str, list]:
    """drop (or, reversed, keep only) the entries referenced by overlap_map.  :473-493"""
    if not overlap_map:
        return sd_result
    marked = {spk: set() for spk in sd_result}
    for items in overlap_map:
        for spk, idx in items:
            marked[spk].add(idx)
    return {spk: [r for i, r in enumerate(rs) if (i in marked[spk]) == reverse_output] for spk, rs in sd_result.items()}


def get_speaker_num(result: Dict[str, list], threshold: float = 0.0) -> int:
    """speakers with at least one range longer than `threshold` (+ the "main" speaker).  Quirk
    kept: the running maximum is never updated, so "main" = last speaker with positive total.
    :496-518"""
    if len(result) == 1 or threshold <= 0:
        return len(result)
    main = ""
    for spk, rs in result.items():
        if sum(r[1] - r[0] for r in rs) > 0:
            main = spk
    n = 0
    for spk, rs in result.items():
        if spk == main or any(r[1] - r[0] > threshold for r in rs):
            n += 1
    return n


def sd_result_parser(sd_result: dict, is_single: bool = False, combine_timerange: bool = False) -> Dict[str, list]:
    """{'text': [[start, end, label], ...]} (CAM++ pipeline output) -> {label: [(s,e), ...]}.  :185-227"""
    result: Dict[str, list] = {}
    if not sd_result or not sd_result["text"]:
        return result
    rows = sorted(sd_result["text"], key=lambda it: it[0])
    prev_label, ps, pe = "", 0.0, 0.0
    for row in rows:
        label = "0" if is_single else str(int(row[-1]))
        if combine_timerange:
            if not prev_label:
                prev_label, ps, pe = label, row[0], row[1]
                continue
            if label == prev_label:
                pe = row[1]
                continue
            s, e = ps, pe
            prev_label, ps, pe = label, row[0], row[1]
        else:
            s, e = row[0], row[1]
        result.setdefault(label, []).append((round(s, 3), round(e, 3)))
    if prev_label and prev_label not in result:
        result[prev_label] = [(round(ps, 3), round(pe, 3))]
    if is_single and result:
        result["0"] = merge_timeranges(result["0"])
    return result
