"""Known-answer tests that pin the CPU oracle (oracle/zly_oracle.c) to the reference source, line by
line.  The reference ships no tests or golden vectors, so each KAT below is
derived from the cited lines of reference src/inference/onnx_engine.cpp (KATs 1-6)."""
import numpy as np
import pytest

from oracle_lib import DET_DTYPE


def _box(x, y, w, h):
    return np.array([x, y, w, h], dtype=np.float32)


# ---- KAT 1: calculateIoU (:881-909) ---------------------------------------------------------------
def test_iou_identical_is_one(oracle):
    assert oracle.iou(_box(.5, .5, .25, .25), _box(.5, .5, .25, .25)) == 1.0     # dyadic: exact in fp32
    assert abs(oracle.iou(_box(.5, .5, .2, .2), _box(.5, .5, .2, .2)) - 1.0) < 1e-6  # fp32 rounding of c +- w/2


def test_iou_shifted_is_one_third(oracle):
    # overlap 0.1 x 0.2 = 0.02, union 0.04 + 0.04 - 0.02 = 0.06
    got = oracle.iou(_box(.5, .5, .2, .2), _box(.6, .5, .2, .2))
    assert abs(got - 1.0 / 3.0) < 1e-6


def test_iou_disjoint_and_zero_area(oracle):
    assert oracle.iou(_box(.2, .2, .1, .1), _box(.8, .8, .1, .1)) == 0.0
    assert oracle.iou(_box(.5, .5, 0, 0), _box(.5, .5, 0, 0)) == 0.0        # union == 0 -> 0 (:904-908)


def test_iou_matches_fp32_formula(oracle):
    rng = np.random.default_rng(3)
    for _ in range(200):
        a = rng.uniform(0.05, 0.9, 4).astype(np.float32)
        b = rng.uniform(0.05, 0.9, 4).astype(np.float32)
        f = np.float32
        ax0, ay0, ax1, ay1 = a[0] - a[2] / f(2), a[1] - a[3] / f(2), a[0] + a[2] / f(2), a[1] + a[3] / f(2)
        bx0, by0, bx1, by1 = b[0] - b[2] / f(2), b[1] - b[3] / f(2), b[0] + b[2] / f(2), b[1] + b[3] / f(2)
        ox = max(f(0), min(ax1, bx1) - max(ax0, bx0))
        oy = max(f(0), min(ay1, by1) - max(ay0, by0))
        inter = f(ox * oy)
        uni = f(f(f(a[2] * a[3]) + f(b[2] * b[3])) - inter)
        want = f(inter / uni) if uni > 0 else f(0)
        assert np.float32(oracle.iou(a, b)).view(np.uint32) == want.view(np.uint32)


# ---- KAT 2: applyNMS (:837-878) --------------------------------------------------------------------
def _dets(rows):
    d = np.zeros(len(rows), dtype=DET_DTYPE)
    for i, (x, y, w, h, conf, cls) in enumerate(rows):
        d[i] = (x, y, w, h, conf, cls, 0, 0, 0)
    return d


def test_nms_keeps_pair_below_threshold(oracle):
    out = oracle.nms(_dets([(.5, .5, .2, .2, .9, 1), (.6, .5, .2, .2, .8, 1)]), 0.45)   # IoU 1/3 < 0.45
    assert len(out) == 2


def test_nms_strict_greater_keeps_iou_equal_to_threshold(oracle):
    a, b = (.5, .5, .2, .2, .9, 0), (.6, .5, .2, .2, .8, 0)
    thr = oracle.iou(_box(*a[:4]), _box(*b[:4]))
    assert len(oracle.nms(_dets([a, b]), thr)) == 2                                     # > not >= (:871)
    assert len(oracle.nms(_dets([a, b]), float(np.nextafter(np.float32(thr), np.float32(0)))) ) == 1


def test_nms_is_class_aware(oracle):
    out = oracle.nms(_dets([(.5, .5, .2, .2, .9, 0), (.5, .5, .2, .2, .8, 1)]), 0.45)   # full overlap, other class (:866)
    assert len(out) == 2


def test_nms_output_order_class_asc_conf_desc(oracle):
    rows = [(.1, .1, .05, .05, .6, 2), (.3, .3, .05, .05, .9, 0), (.5, .5, .05, .05, .7, 2), (.7, .7, .05, .05, .8, 0)]
    out = oracle.nms(_dets(rows), 0.45)
    assert list(out["class_id"]) == [0, 0, 2, 2]
    assert list(np.round(out["confidence"], 2)) == [0.9, 0.8, 0.7, 0.6]                  # (:846-851)


def test_nms_suppresses_lower_confidence(oracle):
    out = oracle.nms(_dets([(.5, .5, .2, .2, .7, 3), (.51, .5, .2, .2, .95, 3), (.9, .9, .05, .05, .6, 3)]), 0.45)
    assert len(out) == 2 and abs(out[0]["confidence"] - .95) < 1e-6 and abs(out[1]["confidence"] - .6) < 1e-6


def test_nms_single_and_empty_pass_through(oracle):
    one = _dets([(.5, .5, .2, .2, .9, 7)])
    assert len(oracle.nms(one, 0.45)) == 1                                              # (:841-843)
    assert len(oracle.nms(_dets([]), 0.45)) == 0


def test_nms_chain_is_greedy_not_transitive(oracle):
    # A suppresses B; C overlaps B but not A -> C survives (greedy semantics of :856-875)
    rows = [(.50, .5, .2, .2, .9, 0), (.56, .5, .2, .2, .8, 0), (.62, .5, .2, .2, .7, 0)]
    assert oracle.iou(_box(*rows[0][:4]), _box(*rows[1][:4])) > 0.45
    assert oracle.iou(_box(*rows[0][:4]), _box(*rows[2][:4])) < 0.45
    out = oracle.nms(_dets(rows), 0.45)
    assert [round(float(c), 1) for c in out["confidence"]] == [0.9, 0.7]


# ---- KAT 3: threshold / arg-max (:787-799) ---------------------------------------------------------
def _head(nc, cols):
    h = np.zeros((4 + nc, len(cols)), dtype=np.float32)
    for i, (box, scores) in enumerate(cols):
        h[:4, i] = box
        h[4:4 + len(scores), i] = scores
    return h


def test_decode_threshold_is_inclusive(oracle):
    h = _head(4, [((100, 100, 50, 50), (0.5, 0, 0, 0)), ((200, 100, 50, 50), (np.nextafter(np.float32(0.5), np.float32(0)), 0, 0, 0))])
    d = oracle.decode(h, 416, 416, 0.5)
    assert len(d) == 1 and d[0]["confidence"] == 0.5                                    # >= (:799)


def test_decode_all_zero_scores_dropped_even_with_zero_threshold(oracle):
    h = _head(4, [((100, 100, 50, 50), (0, 0, 0, 0))])
    assert len(oracle.decode(h, 416, 416, 0.0)) == 0                                    # max_class stays -1 (:787-788,:799)


def test_decode_first_of_equal_maxima_wins(oracle):
    h = _head(4, [((100, 100, 50, 50), (0.2, 0.8, 0.8, 0.1))])
    d = oracle.decode(h, 416, 416, 0.5)
    assert d[0]["class_id"] == 1                                                        # strict > (:792)


def test_decode_sets_track_id_zero_and_anchor_order(oracle):
    h = _head(2, [((10, 10, 5, 5), (0.9, 0)), ((20, 10, 5, 5), (0, 0)), ((30, 10, 5, 5), (0, 0.7))])
    d = oracle.decode(h, 416, 416, 0.5)
    assert list(d["class_id"]) == [0, 1] and list(d["track_id"]) == [0, 0]              # (:812)


# ---- KAT 4: preProcess index map (:649-700) --------------------------------------------------------
def test_preprocess_index_map_800x600(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (600, 800, 3), dtype=np.uint8)
    rc, out = oracle.preprocess(img, 416, 416)
    assert rc == 0
    sw, sh = np.float32(800) / np.float32(416), np.float32(600) / np.float32(416)
    assert int(np.float32(415) * sw) == 798 and int(np.float32(415) * sh) == 598        # SURVEY 8c KAT 4
    for (y, x) in [(0, 0), (415, 415), (100, 7), (207, 208)]:
        sy, sx = min(int(np.float32(y) * sh), 599), min(int(np.float32(x) * sw), 799)
        for c in range(3):
            assert out[c, y, x] == np.float32(img[sy, sx, 2 - c]) / np.float32(255.0)   # BGR->RGB (:685), /255 (:693)


def test_preprocess_identity_when_same_size(oracle):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    rc, out = oracle.preprocess(img, 64, 64)
    assert rc == 0
    want = (img[..., ::-1].astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1)
    assert np.array_equal(out, want)


def test_preprocess_wrong_byte_count_is_invalid_input(oracle):
    img = np.zeros((10, 10, 3), dtype=np.uint8)
    rc, _ = oracle.preprocess(img, 32, 32, nbytes=299)
    assert rc == 203                                                                    # INVALID_INPUT (:659-665)


def test_preprocess_upscale_clamps(oracle):
    img = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    rc, out = oracle.preprocess(img, 32, 32)
    assert rc == 0 and out[0, 31, 31] == np.float32(img[1, 2, 2]) / np.float32(255.0)


# ---- KAT 5: boxes are normalised by the REQUEST's dims (:802-805, :598) -----------------------------
def test_decode_normalises_by_request_dims(oracle):
    h = _head(1, [((208, 208, 416, 416), (0.9,))])
    d = oracle.decode(h, 800, 600, 0.5)
    assert d[0]["w"] == np.float32(416) / np.float32(800)        # 0.52, not 1.0
    assert d[0]["h"] == np.float32(416) / np.float32(600)
    assert d[0]["x"] == np.float32(208) / np.float32(800)


# ---- KAT 6: warm-up frame (:919-954) ---------------------------------------------------------------
def test_preprocess_constant_128(oracle):
    img = np.full((416, 416, 3), 128, dtype=np.uint8)
    rc, out = oracle.preprocess(img, 416, 416)
    assert rc == 0 and np.all(out == np.float32(128) / np.float32(255.0))


# ---- postProcess end to end on a seeded random head tensor -------------------------------------------
def test_postprocess_equals_decode_then_nms(oracle):
    rng = np.random.default_rng(11)
    head = np.zeros((4 + 8, 500), dtype=np.float32)
    head[0] = rng.uniform(0, 416, 500); head[1] = rng.uniform(0, 416, 500)
    head[2] = rng.uniform(10, 200, 500); head[3] = rng.uniform(10, 200, 500)
    head[4:] = rng.uniform(0, 1.0, (8, 500)).astype(np.float32) ** 3
    a = oracle.postprocess(head, 640, 480)
    b = oracle.nms(oracle.decode(head, 640, 480))
    assert len(a) == len(b) > 0 and a.tobytes() == b.tobytes()
