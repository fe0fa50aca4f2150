/*
 * zly_oracle.c -- CPU ORACLE for the detect path's pre- and post-processing.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the shipped HIP engine never links or
 * calls anything in oracle/.
 *
 * It restates, in plain C (IEEE fp32, no FMA contraction, no fast-math), the arithmetic
 * that the reference performs on the host around its ONNX-Runtime call.  Citations are
 * into /root/reference/src/inference/onnx_engine.cpp (read as text; the reference does
 * not compile, SURVEY.md F3, and nothing from it is built or copied here):
 *
 *   zlyo_preprocess   <- preProcess / preProcessZeroCopy          onnx_engine.cpp:649-700, 703-755
 *   zlyo_decode       <- postProcess (decode + threshold + norm)   onnx_engine.cpp:758-834
 *   zlyo_iou          <- calculateIoU                              onnx_engine.cpp:881-909
 *   zlyo_nms          <- applyNMS                                  onnx_engine.cpp:837-878
 *   zlyo_postprocess  <- postProcess including its NMS call        onnx_engine.cpp:822-824
 *   zlyo_det          <- Detection / BoundingBox (40-byte POD)     src/common/types.h:16-26
 *
 * Pinning: the reference holds no tests, fixtures or golden vectors for this path
 * (SURVEY.md section 4), so the oracle is pinned by known-answer tests derived line by
 * line from the cited source (tests/test_oracle_kat.py, SURVEY.md section 8c KATs 1-6).
 * The neural network between pre- and post-processing is not in the reference at all
 * (un-vendored ONNX Runtime 1.8.1 + un-vendored ultralytics export): see
 * oracle/yolov8_ref.py, whose header says "parity unpinned" for that part.
 *
 * One deliberate, documented choice: the reference sorts with std::sort, whose order for
 * equal keys is unspecified (onnx_engine.cpp:846-851).  The oracle uses a STABLE sort, so
 * equal (class, confidence) candidates keep anchor order.  The HIP path uses the same
 * total order (class asc, confidence desc, anchor index asc).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

/* Error codes used on the path (src/common/result.h:14-48). */
#define ZLYO_OK 0
#define ZLYO_INVALID_INPUT 203

/* Layout-identical to zero_latency::Detection (src/common/types.h:16-26):
 * box{x,y,width,height}@0, confidence@16, class_id@20, track_id@24, pad@28, timestamp@32. */
typedef struct {
    float x, y, w, h;
    float confidence;
    int32_t class_id;
    uint32_t track_id;
    uint32_t pad_;
    uint64_t timestamp;
} zlyo_det;

/* onnx_engine.cpp:649-700.  bgr: u8 [h][w][3] interleaved BGR.  out: fp32 [3][th][tw]
 * planar RGB in [0,1].  Stretch nearest-neighbour, no letterbox. */
int zlyo_preprocess(const uint8_t* bgr, size_t nbytes, int w, int h, int tw, int th, float* out)
{
    /* :659-665 -- byte count must be exactly w*h*3 */
    if (w <= 0 || h <= 0 || nbytes != (size_t)w * (size_t)h * 3u) return ZLYO_INVALID_INPUT;

    /* :673-674 */
    const float scale_w = (float)w / (float)tw;
    const float scale_h = (float)h / (float)th;

    for (int c = 0; c < 3; ++c) {                      /* :677 channel-outer */
        for (int y = 0; y < th; ++y) {
            int sy = (int)((float)y * scale_h);         /* :681 int(h*scale_h) */
            if (sy > h - 1) sy = h - 1;
            for (int x = 0; x < tw; ++x) {
                int sx = (int)((float)x * scale_w);     /* :682 */
                if (sx > w - 1) sx = w - 1;
                const size_t src = ((size_t)sy * (size_t)w + (size_t)sx) * 3u + (size_t)(2 - c); /* :685 */
                const size_t dst = (size_t)c * th * tw + (size_t)y * tw + (size_t)x;             /* :688 */
                out[dst] = (float)bgr[src] / 255.0f;    /* :693 */
            }
        }
    }
    return ZLYO_OK;
}

/* onnx_engine.cpp:881-909.  Boxes are {cx, cy, w, h}. */
float zlyo_iou(const float* a, const float* b)
{
    const float ax0 = a[0] - a[2] / 2, ay0 = a[1] - a[3] / 2;
    const float ax1 = a[0] + a[2] / 2, ay1 = a[1] + a[3] / 2;
    const float bx0 = b[0] - b[2] / 2, by0 = b[1] - b[3] / 2;
    const float bx1 = b[0] + b[2] / 2, by1 = b[1] + b[3] / 2;

    /* std::max(a,b) = (a<b)?b:a ; std::min(a,b) = (b<a)?b:a  (:894-895) */
    const float lo_x = (ax0 < bx0) ? bx0 : ax0, hi_x = (bx1 < ax1) ? bx1 : ax1;
    const float lo_y = (ay0 < by0) ? by0 : ay0, hi_y = (by1 < ay1) ? by1 : ay1;
    const float dx = hi_x - lo_x, dy = hi_y - lo_y;
    const float ox = (0.0f < dx) ? dx : 0.0f;               /* :894 max(0, ...) */
    const float oy = (0.0f < dy) ? dy : 0.0f;               /* :895 */
    const float inter = ox * oy;

    const float area_a = a[2] * a[3];
    const float area_b = b[2] * b[3];
    const float uni = area_a + area_b - inter;              /* :901 (left to right) */
    if (uni > 0) return inter / uni;                        /* :904-906 */
    return 0.0f;
}

/* onnx_engine.cpp:758-820 without the NMS call.  head: fp32 [4+C][N], rows 0-3 = cx,cy,w,h in
 * model-input pixels, rows 4.. = class scores.  Candidates come out in anchor order.
 * track_id = 0 (:812); timestamp is wall-clock in the reference (:813-815) and left 0 here. */
int zlyo_decode(const float* head, int num_classes, int num_boxes, int img_w, int img_h,
                float conf_thr, zlyo_det* out, int cap, int* n_out)
{
    int n = 0;
    const size_t N = (size_t)num_boxes;
    for (size_t i = 0; i < N; ++i) {
        const float cx = head[0 * N + i], cy = head[1 * N + i];
        const float bw = head[2 * N + i], bh = head[3 * N + i];
        float best = 0.0f;                                  /* :787 */
        int best_c = -1;                                    /* :788 */
        for (int j = 0; j < num_classes; ++j) {
            const float s = head[(size_t)(j + 4) * N + i];
            if (s > best) { best = s; best_c = j; }         /* :792 strict >, first max wins */
        }
        if (best >= conf_thr && best_c >= 0) {              /* :799 */
            if (n < cap) {
                zlyo_det d;
                memset(&d, 0, sizeof d);
                d.x = cx / (float)img_w;                    /* :802-805: REQUEST dims, not model dims */
                d.y = cy / (float)img_h;
                d.w = bw / (float)img_w;
                d.h = bh / (float)img_h;
                d.confidence = best;
                d.class_id = best_c;
                d.track_id = 0;
                d.timestamp = 0;
                out[n] = d;
            }
            ++n;
        }
    }
    *n_out = n;
    return ZLYO_OK;
}

/* (class asc, confidence desc) -- onnx_engine.cpp:846-851 */
static int det_before(const zlyo_det* a, const zlyo_det* b)
{
    if (a->class_id != b->class_id) return a->class_id < b->class_id;
    return a->confidence > b->confidence;
}

/* stable bottom-up merge sort */
static void stable_sort_dets(zlyo_det* d, int n)
{
    if (n < 2) return;
    zlyo_det* tmp = (zlyo_det*)malloc((size_t)n * sizeof(zlyo_det));
    zlyo_det *src = d, *dst = tmp;
    for (int width = 1; width < n; width *= 2) {
        for (int lo = 0; lo < n; lo += 2 * width) {
            int mid = lo + width < n ? lo + width : n;
            int hi = lo + 2 * width < n ? lo + 2 * width : n;
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                if (det_before(&src[j], &src[i])) dst[k++] = src[j++];
                else dst[k++] = src[i++];
            }
            while (i < mid) dst[k++] = src[i++];
            while (j < hi) dst[k++] = src[j++];
        }
        zlyo_det* t = src; src = dst; dst = t;
    }
    if (src != d) memcpy(d, src, (size_t)n * sizeof(zlyo_det));
    free(tmp);
}

/* onnx_engine.cpp:837-878.  dets is sorted in place (as the reference does); kept boxes are
 * written to out in (class asc, confidence desc) order.  Suppression uses strict > (:871). */
int zlyo_nms(zlyo_det* dets, int n, float iou_thr, zlyo_det* out, int* n_out)
{
    if (n <= 1) {                                           /* :841-843 */
        if (n == 1) out[0] = dets[0];
        *n_out = n > 0 ? n : 0;
        return ZLYO_OK;
    }
    stable_sort_dets(dets, n);
    unsigned char* removed = (unsigned char*)calloc((size_t)n, 1);
    int m = 0;
    for (int i = 0; i < n; ++i) {
        if (removed[i]) continue;
        const int cls = dets[i].class_id;
        out[m++] = dets[i];
        for (int j = i + 1; j < n; ++j) {
            if (removed[j] || dets[j].class_id != cls) continue;   /* :866 */
            if (zlyo_iou(&dets[i].x, &dets[j].x) > iou_thr) removed[j] = 1;   /* :870-873 */
        }
    }
    free(removed);
    *n_out = m;
    return ZLYO_OK;
}

/* postProcess end to end (:758-834): decode, then NMS when non-empty (:822-824).
 * out must hold num_boxes entries (the reference has no cap). */
int zlyo_postprocess(const float* head, int num_classes, int num_boxes, int img_w, int img_h,
                     float conf_thr, float iou_thr, zlyo_det* out, int* n_out)
{
    zlyo_det* cand = (zlyo_det*)malloc((size_t)(num_boxes > 0 ? num_boxes : 1) * sizeof(zlyo_det));
    int n = 0;
    zlyo_decode(head, num_classes, num_boxes, img_w, img_h, conf_thr, cand, num_boxes, &n);
    int m = 0;
    if (n > 0) zlyo_nms(cand, n, iou_thr, out, &m);
    free(cand);
    *n_out = m;
    return ZLYO_OK;
}

size_t zlyo_sizeof_det(void) { return sizeof(zlyo_det); }
