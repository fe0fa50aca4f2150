"""Set-level comparison of two detection lists: the detections of an engine that computes in
bf16 against the detections of the fp32 oracle, "matched by greedy match (same class, IoU >= 0.9) after excluding oracle
candidates with |score - 0.5| < band or pairwise |IoU - 0.45| < band (the threshold-flip band)".

Excluding single candidates is not enough once NMS runs: a pair whose IoU sits at the threshold flips one suppression,
and the box that survives may suppress a third one (a cascade).  Cascades cannot leave a connected component of the
graph  {nodes: oracle anchors scoring >= conf - band; edges: same class and IoU >= iou_thr - band}, because greedy
class-aware NMS (reference onnx_engine.cpp:837-878) only ever lets a box act on a same-class box it overlaps by more
than the threshold, in either of the two runs being compared.  So:
  * a component is AMBIGUOUS when it holds a node with |score - conf| <= band, an anchor whose two best classes score within
    band of each other (the arg-max may pick either: the anchor is then a node of both classes), an edge with
    |IoU - iou_thr| <= band, or two overlapping nodes whose scores are within band of each other (their greedy order may
    swap) and which do not suppress the same third nodes: both outcomes are legitimate there and it is skipped (and counted);
  * in every other component the two detection sets must be identical: same number, one-to-one matched with the same
    class, IoU >= min_iou and |confidence difference| <= band;
  * even inside an ambiguous component the boxes that come BEFORE everything ambiguous in greedy (confidence desc) order, in both
    runs, are decided identically (greedy NMS looks back only) and are compared exactly: a flip at the confidence threshold -- the lowest
    box of its class -- taints nothing above it;
  * every detection of the engine must land on an oracle node (same class, IoU >= min_iou): a detection from nowhere
    is an error whatever the bands.
TEST INFRASTRUCTURE (imported by tests/ only)."""
import numpy as np


def _iou_matrix(a, b):
    """IoU of centre-format boxes, a [n][4], b [m][4] (float64; the bands absorb the difference to the fp32 oracle IoU)."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 4)
    b = np.asarray(b, dtype=np.float64).reshape(-1, 4)
    ax1, ay1, ax2, ay2 = a[:, 0] - a[:, 2] / 2, a[:, 1] - a[:, 3] / 2, a[:, 0] + a[:, 2] / 2, a[:, 1] + a[:, 3] / 2
    bx1, by1, bx2, by2 = b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2
    iw = np.clip(np.minimum(ax2[:, None], bx2[None]) - np.maximum(ax1[:, None], bx1[None]), 0, None)
    ih = np.clip(np.minimum(ay2[:, None], by2[None]) - np.maximum(ay1[:, None], by1[None]), 0, None)
    inter = iw * ih
    union = (a[:, 2] * a[:, 3])[:, None] + (b[:, 2] * b[:, 3])[None] - inter
    return np.where(union > 0, inter / np.where(union > 0, union, 1), 0.0)


def _boxes(d):
    return np.stack([d["x"], d["y"], d["w"], d["h"]], 1).astype(np.float64) if len(d) else np.zeros((0, 4))


def measured_bands(ref_head, other_head, conf=0.5, iou_thr=0.45, margin=2.0, floor=2e-3, cap=2e-2):
    """The threshold-flip bands of ONE frame from the deviation actually present between two head tensors [4+nc][N] of that
    frame (the fp32 oracle's and the engine's own, or -- where the engine does not materialise its head -- the bf16-rounding
    oracle's, i.e. the noise floor of bf16 on this frame).  SURVEY 8c's band (2e-2) is the TOLERANCE; the deviation really
    present is ~5x smaller, and a set comparison that skips everything within the tolerance of a threshold compares a minority
    of the detections on a head whose scores are a smooth (Gaussian-tailed) distribution through 0.5.  A flip needs a score
    (an IoU) to be on different sides of its threshold in the two tensors, which cannot happen further from the threshold than
    the largest deviation near it.  Returns (band_score, band_iou) = margin x that largest deviation over the (class, anchor)
    entries scoring >= conf - 0.1 resp. over the same-class candidate pairs with IoU >= 0.2, clamped to [floor, cap]: never
    looser than SURVEY's band, never tighter than `floor`."""
    a = np.asarray(ref_head, dtype=np.float64)
    b = np.asarray(other_head, dtype=np.float64)
    near = a[4:] >= conf - 0.1
    ds = float(np.abs(a[4:] - b[4:])[near].max()) if near.any() else 0.0
    anc = np.nonzero(near.any(0))[0]
    di = 0.0
    if len(anc) > 1:
        cls = a[4:, anc].argmax(0)
        ia, ib = _iou_matrix(a[:4, anc].T, a[:4, anc].T), _iou_matrix(b[:4, anc].T, b[:4, anc].T)
        pair = (cls[:, None] == cls[None]) & (ia >= 0.2)
        np.fill_diagonal(pair, False)
        if pair.any():
            di = float(np.abs(ia - ib)[pair].max())
    clamp = lambda v: float(min(cap, max(floor, margin * v)))
    return clamp(ds), clamp(di)


def compare_detection_sets(oracle, ref_head, got, img_w, img_h, conf=0.5, iou_thr=0.45, band=2e-2, min_iou=0.9, band_iou=None, got_head=None, floor=1e-3):
    """oracle: tests/oracle_lib.Oracle; ref_head: fp32 oracle head tensor [4+nc][N] of the frame; got: the engine's
    detections (structured zly_det array); band: confidence tolerance of matched detections and -- without got_head -- the
    half-width of the score flip band; band_iou: half-width of the IoU flip band (default: band).
    got_head: the ENGINE's own head tensor of the frame (the very values its Detect kernel thresholded; the GPU tests assert
    separately that it is within tolerance of the oracle's).  With it, a component is ambiguous only where a threshold decision
    REALLY differs between the two tensors -- a score on different sides of conf, a different arg-max class, an IoU on
    different sides of iou_thr, a different order of two overlapping same-class boxes -- or sits within `floor` of its
    threshold; every other component is compared exactly.  Without it, everything within the bands of a threshold is skipped.
    Returns (n_compared, n_skipped, errors): detections compared exactly, oracle detections skipped inside ambiguous
    components, list of error strings (empty = parity holds)."""
    want = oracle.postprocess(ref_head, img_w, img_h, conf, iou_thr)
    errors = []
    band_iou = band if band_iou is None else band_iou
    ref_head = np.asarray(ref_head, dtype=np.float32)
    sc = ref_head[4:]
    top = sc.max(0)
    if got_head is not None:
        # node set: the best class of an anchor in EITHER tensor, when it reaches conf - floor in either
        gh = np.asarray(got_head, dtype=np.float32)
        gsc = gh[4:]
        gtop = gsc.max(0)
        ca, cg = sc.argmax(0), gsc.argmax(0)
        live = np.nonzero((top >= conf - floor) | (gtop >= conf - floor))[0]
        pairs = sorted(set((int(ca[a]), int(a)) for a in live) | set((int(cg[a]), int(a)) for a in live))
        cls_i = np.array([p[0] for p in pairs], dtype=np.int64)
        anc_i = np.array([p[1] for p in pairs], dtype=np.int64)
        s_band, i_band = floor, floor
    else:
        # nodes: (anchor, class) pairs the engine may legitimately report -- the anchor's best class, plus every class within
        # `band` of it, when they score >= conf - band (postProcess keeps the arg-max class only, onnx_engine.cpp:787-799)
        cls_i, anc_i = np.nonzero((sc >= np.maximum(top - band, conf - band)[None]) & (top >= conf - band)[None])
        s_band, i_band = band, band_iou
    per_anchor = np.bincount(anc_i, minlength=sc.shape[1]) if len(anc_i) else np.zeros(sc.shape[1], dtype=np.int64)
    nodes = np.zeros(len(anc_i), dtype=[("x", "<f8"), ("y", "<f8"), ("w", "<f8"), ("h", "<f8"), ("confidence", "<f8"), ("class_id", "<i4")])
    nodes["x"] = ref_head[0, anc_i].astype(np.float64) / img_w; nodes["y"] = ref_head[1, anc_i].astype(np.float64) / img_h
    nodes["w"] = ref_head[2, anc_i].astype(np.float64) / img_w; nodes["h"] = ref_head[3, anc_i].astype(np.float64) / img_h
    nodes["confidence"] = sc[cls_i, anc_i]; nodes["class_id"] = cls_i
    nb = _boxes(nodes)
    n = len(nodes)
    comp = np.arange(n)

    def find(i):
        while comp[i] != i:
            comp[i] = comp[comp[i]]
            i = comp[i]
        return i

    amb_node = (np.abs(nodes["confidence"] - conf) <= s_band) | (per_anchor[anc_i] > 1) if n else np.zeros(0, bool)
    amb_edges = []
    if n and got_head is not None:
        gconf = gsc[cls_i, anc_i].astype(np.float64)
        amb_node = amb_node | ((nodes["confidence"] >= conf) != (gconf >= conf))
        gb = np.stack([gh[0, anc_i] / img_w, gh[1, anc_i] / img_h, gh[2, anc_i] / img_w, gh[3, anc_i] / img_h], 1).astype(np.float64)
    if n:
        iou = _iou_matrix(nb, nb)
        same = nodes["class_id"][:, None] == nodes["class_id"][None]
        if got_head is not None:
            giou = _iou_matrix(gb, gb)
            near = same & ((iou >= iou_thr - i_band) | (giou >= iou_thr - i_band))
        else:
            near = same & (iou >= iou_thr - i_band)
        ii, jj = np.nonzero(np.triu(near, 1))
        for i, j in zip(ii, jj):
            ri, rj = find(i), find(j)
            if ri != rj:
                comp[ri] = rj
            if abs(iou[i, j] - iou_thr) <= i_band:
                amb_edges += [i, j]
            elif got_head is not None:
                if (iou[i, j] > iou_thr) != (giou[i, j] > iou_thr):
                    amb_edges += [i, j]                 # this suppression really differs between the two tensors
                elif (nodes["confidence"][i] > nodes["confidence"][j]) != (gconf[i] > gconf[j]) or nodes["confidence"][i] == nodes["confidence"][j]:
                    amb_edges += [i, j]                 # the greedy order of the two really differs
            elif abs(float(nodes["confidence"][i]) - float(nodes["confidence"][j])) <= band:
                # the greedy order of i and j may swap: harmless unless they differ in which same-class boxes they suppress
                si, sj = same[i] & (iou[i] > iou_thr), same[j] & (iou[j] > iou_thr)
                diff = si != sj
                diff[i] = diff[j] = False
                if diff.any():
                    amb_edges += [i, j]
    roots = np.array([find(i) for i in range(n)], dtype=np.int64)
    amb_all = sorted(set(int(i) for i in np.nonzero(amb_node)[0]) | set(int(i) for i in amb_edges))
    ambiguous = set(int(roots[i]) for i in amb_all)
    # Inside an ambiguous component the clean PREFIX is still compared exactly: greedy NMS decides a box from the boxes before it in
    # (confidence desc) order only, so a node that precedes -- in the oracle's order AND in the engine's -- every node involved in an
    # ambiguity of its component (a flipped node, both ends of a flipped / order-swapped edge) is kept or suppressed identically in both
    # runs.  cutoff[root] = the highest confidence any involved node has in either tensor (+ band without the engine's tensor).
    lo = nodes["confidence"].copy() if n else np.zeros(0)
    hi = nodes["confidence"].copy() if n else np.zeros(0)
    if n and got_head is not None:
        lo, hi = np.minimum(lo, gconf), np.maximum(hi, gconf)
    elif n:
        lo, hi = lo - band, hi + band
    cutoff = {}
    for i in amb_all:
        r = int(roots[i])
        cutoff[r] = max(cutoff.get(r, -1.0), float(hi[i]))
    clean = np.array([int(roots[i]) not in cutoff or lo[i] > cutoff[int(roots[i])] for i in range(n)], dtype=bool) if n else np.zeros(0, bool)

    def locate(dets, what):
        """node of every detection (its best same-class node by IoU), -1 = none"""
        out = np.full(len(dets), -1, dtype=np.int64)
        if len(dets) == 0 or n == 0:
            if len(dets):
                errors.append(f"{len(dets)} {what} detection(s) but the oracle has no candidate at all")
            return out
        m = _iou_matrix(_boxes(dets), nb)
        m = np.where(dets["class_id"][:, None] == nodes["class_id"][None], m, -1.0)
        best = m.argmax(1)
        for k in range(len(dets)):
            if m[k, best[k]] >= min_iou:
                out[k] = best[k]
            else:
                errors.append(f"{what} detection {k} (class {int(dets['class_id'][k])}, conf {float(dets['confidence'][k]):.4f}) matches no oracle "
                              f"candidate: best same-class IoU {float(m[k, best[k]]):.3f}")
        return out

    wn, gn = locate(want, "oracle"), locate(got, "engine")
    wc = np.where(wn >= 0, roots[np.maximum(wn, 0)], -1) if n else wn
    gc = np.where(gn >= 0, roots[np.maximum(gn, 0)], -1) if n else gn
    compared = skipped = 0
    for root in sorted(set(int(r) for r in roots)):
        wi, gi = np.nonzero(wc == root)[0], np.nonzero(gc == root)[0]
        if root in ambiguous:
            wk = np.array([k for k in wi if clean[wn[k]]], dtype=np.int64)
            skipped += len(wi) - len(wk)
            wi, gi = wk, np.array([k for k in gi if clean[gn[k]]], dtype=np.int64)
        compared += len(wi)
        if len(wi) != len(gi):
            errors.append(f"component {root}: oracle keeps {len(wi)} detection(s), engine {len(gi)}")
            continue
        if len(wi) == 0:
            continue
        m = _iou_matrix(_boxes(got[gi]), _boxes(want[wi]))
        used = set()
        for a in range(len(gi)):
            order = np.argsort(-m[a])
            hit = next((b for b in order if b not in used and m[a, b] >= min_iou), None)
            if hit is None:
                errors.append(f"component {root}: engine detection {int(gi[a])} has no oracle partner at IoU >= {min_iou} (best {float(m[a].max()):.3f})")
                continue
            used.add(hit)
            dc = abs(float(got["confidence"][gi[a]]) - float(want["confidence"][wi[hit]]))
            if dc > band:
                errors.append(f"component {root}: confidence differs by {dc:.4f} > {band}")
    return compared, skipped, errors
