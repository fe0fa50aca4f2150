/*
 * zly_gather.h -- C ABI of the in-process result-slab gather (libzly_gather.so): RCCL over xGMI between the GPUs of ONE server process.
 *
 * the reference's server is one process; with several MI355X in it (the plugin's ZLY_NUM_DEVICES mode:
 * frames go round robin to one engine per GPU) the only exchange step of the detect path is after NMS -- every GPU's fixed-size result
 * slabs (zly_slab_header + cap x zly_det per frame, include/zly.h) are all-gathered so that one GPU / one host copy sees the detections of
 * the whole global batch in frame order.  The reference has no counterpart (OnnxInferenceEngine is single-device,
 * src/inference/onnx_engine.cpp:518-646); this is north-star work behind the same plain-pointer boundary as zly.h.
 *
 * A library of its own, not part of libzly.so: it links RCCL, and a process that already carries an RCCL (PyTorch's bundled one in
 * bench.py / the tests, which gather through torch.distributed) must not get a second copy.  The one-process-per-GPU deployment uses
 * torch.distributed (zero-latency-yolo_amd/shard.py); this library is for the single-process C++ server.
 *
 *   zly_gather_create    ncclCommInitAll over the given device ordinals (one communicator per device, all owned by this process)
 *   zly_gather_all       one ncclAllGather per device inside ncclGroupStart / ncclGroupEnd: in stream order on streams[i], d_recv[i]
 *                        receives ndev * bytes_per_rank bytes, rank-major (rank r = devices[r]); d_send[i] / d_recv[i] / streams[i] live on
 *                        devices[i].  Order a stream behind its engine's NMS first (zly_join).  Returns without synchronising.
 *   zly_gather_destroy   ncclCommDestroy
 * Return codes: zero_latency::ErrorCode values as in zly.h (0 OK, 2 INVALID_ARGUMENT, 300 SYSTEM_ERROR); zly_gather_last_error() gives
 * the thread-local message of the last failing call.
 */
#ifndef ZLY_GATHER_H_
#define ZLY_GATHER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zly_gather zly_gather;

int32_t zly_gather_create(int32_t ndev, const int32_t* devices, zly_gather** out);
int32_t zly_gather_all(zly_gather* g, const void* const* d_send, void* const* d_recv, size_t bytes_per_rank, void* const* streams);
int32_t zly_gather_ndev(const zly_gather* g);
int32_t zly_gather_destroy(zly_gather* g);
const char* zly_gather_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* ZLY_GATHER_H_ */
