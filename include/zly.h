/*
 * zly.h -- C ABI of the MI355X-native YOLO detect engine (libzly.so).
 *
 * This is the ONLY surface that touches HIP.  Host C++ (the reference's server) binds it through
 * host/hip_inference_engine.{h,cpp}, which implements the reference's plugin interface
 * `IInferenceEngine` (reference src/inference/inference_engine.h:33-43) on top of these calls;
 * Python (tests, bench) binds it with ctypes.  Plain pointers and sizes only: no C++ or torch
 * types cross this boundary.  See INTEGRATION.md for the reference-side binding.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference):
 *
 *   zly_create / zly_destroy   OnnxInferenceEngine::initialize / shutdown + loadModel + warmupModel
 *                              src/inference/onnx_engine.cpp:67-170, 173-220, 957-1062, 919-954
 *   zly_detect                 OnnxInferenceEngine::runInference (preProcess -> Session::Run ->
 *                              postProcess/applyNMS), one frame   onnx_engine.cpp:518-646
 *   zly_detect_batch           the "dynamic batching" loop that the reference leaves as a TODO and
 *                              runs frame by frame                 onnx_engine.cpp:320-365
 *   zly_submit / zly_poll / zly_wait   OnnxInferenceEngine::submitInference + the result hand-over of its
 *                              inference thread: the asynchronous, pipelined host-to-host path
 *                                                                  onnx_engine.cpp:223-261, 315-398
 *   zly_detect_device          same path with frames already resident in HBM (no reference
 *                              counterpart; used by bench.py and the multi-GPU sharding)
 *   zly_preprocess             OnnxInferenceEngine::preProcess     onnx_engine.cpp:649-700
 *   zly_forward / zly_head_tensor   Ort::Session::Run "images" -> "output0"
 *                                                                  onnx_engine.cpp:560-586
 *   zly_postprocess            OnnxInferenceEngine::postProcess + applyNMS + calculateIoU
 *                                                                  onnx_engine.cpp:758-909
 *   zly_get_stats              OnnxInferenceEngine::getStatus counters  onnx_engine.cpp:279-312
 *   zly_det                    zero_latency::Detection (40 bytes)  src/common/types.h:16-26
 *   return codes               zero_latency::ErrorCode             src/common/result.h:14-48
 *
 * Threading: one engine handle may be used from several host threads; calls on one handle are
 * serialised internally (the reference serialises Session::Run the same way, onnx_engine.cpp:577).  Enqueue sections of DIFFERENT
 * engines of a process run concurrently; only stream capture (at zly_create, and on the first synchronous call of a new batch size) and
 * device allocation / release (zly_create, zly_destroy, the first zly_submit) are alone in the process.
 * Ownership: the caller owns every host buffer for the duration of the call; the engine owns all
 * device and pinned memory.  There is NO CPU fallback: without a usable HIP device zly_create
 * fails with ZLY_ERR_SYSTEM.
 */
#ifndef ZLY_H_
#define ZLY_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* zero_latency::ErrorCode values used on this path (result.h:14-48) */
#define ZLY_OK                  0
#define ZLY_ERR_INVALID_ARGUMENT 2
#define ZLY_ERR_NOT_INITIALIZED 3
#define ZLY_ERR_INFERENCE       200
#define ZLY_ERR_MODEL_NOT_FOUND 201
#define ZLY_ERR_MODEL_LOAD      202
#define ZLY_ERR_INVALID_INPUT   203
#define ZLY_ERR_SYSTEM          300
#define ZLY_PENDING             1   /* zly_poll: the ticket's batch has not completed yet; zly_submit_try: every ring slot is busy (neither is an error) */

#define ZLY_DTYPE_FP32 0   /* fp32 activations + exact-fp32 MFMA: verification mode */
#define ZLY_DTYPE_BF16 1   /* bf16 activations/weights, fp32 accumulate: production mode */

#define ZLY_FLAG_DUMP_LOGITS 1   /* also write the fp32 logits of the six final Detect convs (debug taps
                                   "model.22.cv2.L.2" / "model.22.cv3.L.2"); off in production */
#define ZLY_FLAG_NO_HEAD_TENSOR 4 /* production: the Detect kernel decodes in registers and does not write the fp32 [4+nc][N] head tensor
                                   (the reference's ORT output, 1.2 MB per frame); zly_head_tensor / zly_forward then return 2 */
#define ZLY_FLAG_ASYNC_NMS   8   /* zly_detect_device and the pipelined zly_submit path (batches >= 16): NMS of a call runs on an engine-owned stream beside the first kernels of the
                                   NEXT call (it is 64 latency-bound workgroups).  The slabs of a call are then complete after zly_join
                                   (stream order) or zly_sync / zly_read_slabs (host), whichever comes first. */
#define ZLY_FLAG_SINGLE_CHAIN 16 /* no side streams: the whole step is one chain of launches on one stream.  For SEVERAL engines per GPU fed alternate
                                   batches (their chains overlap each other: the idle time at every kernel boundary of one is filled by the other);
                                   each such engine owns exactly one stream, so that up to three of them fit the four hardware queues ROCm gives a process by default
                                   (leave GPU_MAX_HW_QUEUES alone: 8 measured 2x SLOWER on the host-to-host path) */
#define ZLY_FLAG_NO_FUSION   2   /* run every conv as its own kernel (no fused bottleneck pairs): every zly_debug_tap is then available */

typedef struct zly_engine zly_engine;

/* Layout-identical to zero_latency::Detection: box{x,y,width,height}@0 (centre-x, centre-y, w, h,
 * normalised by the REQUEST's width/height, onnx_engine.cpp:802-805), confidence@16, class_id@20,
 * track_id@24, timestamp@32; sizeof == 40. */
typedef struct zly_det {
    float x, y, w, h;
    float confidence;
    int32_t class_id;
    uint32_t track_id;
    uint32_t pad_;
    uint64_t timestamp;
} zly_det;

/* Fixed-size per-frame result record left in HBM by zly_detect_device and exchanged between
 * GPUs.  n_kept may exceed cap: then only the first cap detections (in the
 * reference's output order: class asc, confidence desc) are stored and ZLY_SLAB_OVERFLOW is set. */
#define ZLY_SLAB_OVERFLOW 1u
typedef struct zly_slab_header {
    int32_t n_kept;        /* detections after NMS */
    int32_t n_candidates;  /* anchors that passed the confidence threshold */
    uint32_t flags;
    uint32_t frame_tag;    /* caller-defined (bench: global frame index) */
} zly_slab_header;
/* slab bytes = sizeof(zly_slab_header) + cap * sizeof(zly_det) */

typedef struct zly_config {
    const char* weights_path;   /* ZLYW file (ServerConfig::model_path, server/config.h:306) */
    int32_t model_w, model_h;   /* detection.model_width/height, multiples of 32 (config.h:110-149) */
    float conf_thr;             /* confidence_threshold, default 0.5 (configs/server.json:7) */
    float iou_thr;              /* nms_threshold, default 0.45 (configs/server.json:8) */
    int32_t max_batch;          /* frames per zly_detect_batch / zly_detect_device call (1..65535) */
    int32_t max_dets;           /* slab capacity per frame (cap) */
    int32_t device;             /* HIP device ordinal */
    int32_t dtype;              /* ZLY_DTYPE_* */
    int32_t warmup_runs;        /* warmupModel analogue, onnx_engine.cpp:919-954 (reference: 3): that many single-frame passes, then -- when > 0 and
                                   max_batch > 1 -- the throughput path at max_batch (graphs for batch 1 and max_batch are captured and replayed here) */
    int32_t use_graph;          /* 1: replay the forward as a hipGraph per batch size */
    int32_t flags;              /* ZLY_FLAG_* */
} zly_config;

typedef struct zly_stats {
    uint64_t inference_count;
    uint64_t inference_errors;
    double total_preprocess_ms, total_forward_ms, total_postprocess_ms;   /* device time, zly_profile_ops calls only */
    double last_detect_ms;                                                /* host wall time of the last zly_detect */
    /* production sampling (the reference accumulates its three phase timers per frame, onnx_engine.cpp:530-557,605-618):
     * every 16th call of a detect path is bracketed by hipEvents -- preprocess = the preprocess / fused preprocess+stem
     * kernel, forward = everything up to and including the Detect tail (decode + threshold are fused into it),
     * postprocess = NMS.  Sums over the sampled calls; per-frame average = sampled_*_ms / sampled_frames. */
    uint64_t sampled_frames;
    double sampled_preprocess_ms, sampled_forward_ms, sampled_postprocess_ms;
    uint64_t batches;                                                     /* calls of a detect path (any entry point) */
    uint64_t graph_replays;                                               /* of them: forward replayed as a captured hipGraph ... */
    uint64_t eager_batches;                                               /* ... or launched kernel by kernel (partial batches of the pipelined path, use_graph = 0, a capture that failed twice) */
} zly_stats;

typedef struct zly_op_info {
    char name[48];
    int32_t kind;          /* 0 preprocess, 1 conv (incl. fused upsample+concat inputs), 2 sppf-pool, 4 fused detect tail (+decode), 6 nms */
    int32_t pad_;
    double flops_per_frame;   /* 2*MAC, algorithmic */
    double bytes_per_frame;   /* algorithmic: input read once + output written once + weights */
} zly_op_info;

void    zly_default_config(zly_config* cfg);
int32_t zly_create(const zly_config* cfg, zly_engine** out);
int32_t zly_destroy(zly_engine* e);
const char* zly_last_error(void);          /* thread-local message of the last failing call */
const char* zly_version(void);

/* --- whole path ------------------------------------------------------------------------------ */
/* One frame, synchronous.  bgr: u8 [h][w][3] interleaved BGR in host memory, nbytes must equal
 * w*h*3 (else ZLY_ERR_INVALID_INPUT, as onnx_engine.cpp:659-665).  Writes min(n, cap) detections
 * to out and the un-capped count to *n_out. */
int32_t zly_detect(zly_engine* e, const uint8_t* bgr, size_t nbytes, int32_t w, int32_t h,
                   zly_det* out, int32_t cap, int32_t* n_out);

/* n frames (n <= max_batch), each with its own size.  out is [n][cap]; n_out is [n]. */
int32_t zly_detect_batch(zly_engine* e, int32_t n, const uint8_t* const* bgr, const size_t* nbytes,
                         const int32_t* w, const int32_t* h, zly_det* out, int32_t cap, int32_t* n_out);

/* --- asynchronous, pipelined host-to-host path ------------------------------------------------------
 * The throughput path of a server: many host threads hand over frames, the engine batches them and overlaps the PCIe
 * transfers with compute.  Engine-owned ring of pinned staging slots (ZLY_STAGE_SLOTS, default 6; each holds one batch of
 * up to max_batch frames / ZLY_STAGE_MB megabytes, default 1.25 x max_batch model-sized frames):
 *   zly_submit   (any thread, concurrently) reserves a frame slot in the batch being filled and copies the pixels into
 *                pinned memory ON THE CALLING THREAD -- the one copy of the request the reference makes too
 *                (onnx_engine.cpp:233-235) -- then returns a ticket; it blocks only while every ring slot is busy.
 *   dispatch     an engine-owned thread closes the filling batch as soon as fewer than two batches are in flight (a lone
 *                frame is served at once, a backlog at full batches: no batching window), uploads it on a copy stream
 *                while the previous batch computes, runs the path, and downloads the slabs on a third stream.
 *   zly_wait     blocks until the ticket's batch is back in host memory and copies that frame's detections
 *                (min(n, cap) to out, un-capped count to *n_out; timestamps = completion time, onnx_engine.cpp:813-815);
 *                zly_poll is the non-blocking check (ZLY_OK = ready, ZLY_PENDING = not yet).
 * Every ticket must be consumed by exactly one zly_wait (its ring slot is recycled when all its tickets are); a failed
 * batch returns its error code from zly_wait for each of its tickets.  Wrong byte counts fail in zly_submit with
 * ZLY_ERR_INVALID_INPUT (onnx_engine.cpp:659-665) and produce no ticket. */
int32_t zly_submit(zly_engine* e, const uint8_t* bgr, size_t nbytes, int32_t w, int32_t h, uint64_t* ticket);
/* zly_submit that never blocks: ZLY_PENDING (no ticket, nothing copied) when every ring slot of this engine is busy.  A host that feeds
 * several engines (the plugin: one per GPU / engine instance) offers a frame to the next engine in turn and, only if that one is
 * back-pressured, to the others -- with the blocking call alone, submitting threads that all wait on one engine's ring starve the rest. */
int32_t zly_submit_try(zly_engine* e, const uint8_t* bgr, size_t nbytes, int32_t w, int32_t h, uint64_t* ticket);
int32_t zly_poll(zly_engine* e, uint64_t ticket);
int32_t zly_wait(zly_engine* e, uint64_t ticket, zly_det* out, int32_t cap, int32_t* n_out);

/* n frames of identical size, contiguous in DEVICE memory ([n][h][w][3] u8).  Enqueues the whole
 * path on `stream` (a hipStream_t, NULL = the engine's own stream) and returns without
 * synchronising.  d_slabs receives n slabs of zly_slab_bytes(e) each (device memory; NULL = the
 * engine's internal slab buffer, readable with zly_read_slabs).  frame_tag0 + i is stored in
 * slab i. */
int32_t zly_detect_device(zly_engine* e, int32_t n, const void* d_frames, int32_t w, int32_t h,
                          void* d_slabs, uint32_t frame_tag0, void* stream);
size_t  zly_slab_bytes(const zly_engine* e);
int32_t zly_read_slabs(zly_engine* e, int32_t n, void* host_slabs);   /* syncs the engine stream */
int32_t zly_sync(zly_engine* e);
/* Make `stream` (NULL = the engine's own) wait, on the device, for the NMS of every zly_detect_device call made so far except the
 * last `lag` ones.  lag must be 0 or 1 (lag = 1: consume call k-1's slabs right after enqueuing call k); the engine keeps the
 * completion events of its last two calls only, any other lag fails with ZLY_ERR_INVALID_ARGUMENT. */
int32_t zly_join(zly_engine* e, void* stream, int32_t lag);

/* --- stage-level entry points (parity tests) ------------------------------------------------- */
/* preProcess: out_nchw is host fp32 [3][model_h][model_w]. */
int32_t zly_preprocess(zly_engine* e, const uint8_t* bgr, size_t nbytes, int32_t w, int32_t h, float* out_nchw);
/* Session::Run: images is host fp32 [n][3][model_h][model_w] (rounded to the engine dtype on
 * upload); head_out is host fp32 [n][4+nc][N]. */
int32_t zly_forward(zly_engine* e, int32_t n, const float* images_nchw, float* head_out);
/* head tensor of frame `idx` of the most recent detect/forward call: host fp32 [4+nc][N]. */
int32_t zly_head_tensor(zly_engine* e, int32_t idx, float* head_out);
/* postProcess + applyNMS on a caller-supplied head tensor (host fp32 [4+nc][N]); any nc/N. */
int32_t zly_postprocess(zly_engine* e, const float* head, int32_t num_classes, int32_t num_boxes,
                        int32_t img_w, int32_t img_h, float conf_thr, float iou_thr,
                        zly_det* out, int32_t cap, int32_t* n_out, int32_t* n_candidates);
/* copies an intermediate conv output of frame idx to the host as fp32 [C][H][W] (debug/parity);
 * name is the conv's module path, e.g. "model.4.cv2".  *c,*h,*w receive the shape. */
int32_t zly_debug_tap(zly_engine* e, const char* name, int32_t idx, float* out, size_t cap_floats,
                      int32_t* c, int32_t* h, int32_t* w);

/* --- introspection / measurement ------------------------------------------------------------- */
int32_t zly_num_classes(const zly_engine* e);
/* 1 when the model file stores its weights as fp8 e4m3 (+ per-output-channel power-of-two exponents; BASELINE configs[4]): they are
 * dequantised once at load -- exactly representable in bf16 -- and the engine computes in bf16 as with any other file */
int32_t zly_weights_fp8(const zly_engine* e);
int32_t zly_num_anchors(const zly_engine* e);
int32_t zly_num_ops(const zly_engine* e);
int32_t zly_op_info_at(const zly_engine* e, int32_t i, zly_op_info* out);
/* Launch groups: the engine fuses ops into one kernel depending on the batch size (preprocess + model.0 + model.1; whole C2f blocks;
 * bottleneck pairs; the Detect tail of all levels; merged Detect launches on the latency path).  For op i at batch size n:
 * covered_by = the op whose launch does op i's work (== i when op i launches itself; then the other fields describe that launch):
 * n_ops it covers, their summed algorithmic flops and un-fused bytes, and bytes_fused = what has to cross the launch boundary
 * (inputs not produced inside the group + outputs read outside the group + weights) -- the byte count a fused launch's HBM
 * roofline is priced against (bench.py). */
typedef struct zly_launch_info {
    int32_t covered_by;
    int32_t n_ops;
    double flops_per_frame;
    double bytes_unfused_per_frame;
    double bytes_fused_per_frame;
    double weight_bytes;              /* part of both byte counts; per launch, not per frame */
} zly_launch_info;
int32_t zly_launch_info_at(zly_engine* e, int32_t i, int32_t n, zly_launch_info* out);
/* name and tile shape of the kernel op i launches at batch size n (the engine picks kernels per launch size) */
int32_t zly_op_kernel_name(zly_engine* e, int32_t i, int32_t n, char* out, size_t cap);
/* Runs the device path `reps` times on n resident frames with a hipEvent pair around every
 * kernel launch (no graph) and writes the mean milliseconds per op to ms_out[zly_num_ops]. */
int32_t zly_profile_ops(zly_engine* e, int32_t n, const void* d_frames, int32_t w, int32_t h,
                        int32_t reps, float* ms_out);
int32_t zly_get_stats(const zly_engine* e, zly_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* ZLY_H_ */
