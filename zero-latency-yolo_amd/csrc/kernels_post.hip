// kernels_post.hip -- box decode + confidence threshold + class-aware greedy IoU NMS on gfx950,
// using wavefront ballot compaction and LDS; no MFMA (this is integer/compare work).
//
// Restates, bit for bit in fp32, the reference's host post-processing
// (reference src/inference/onnx_engine.cpp): postProcess :758-834, applyNMS :837-878,
// calculateIoU :881-909.  FP contraction is OFF for this file so that the IoU arithmetic is the
// same sequence of IEEE fp32 operations as the CPU oracle (oracle/zly_oracle.c).
#include "zly_internal.h"

#pragma clang fp contract(off)

namespace zly {

// ------------------------------------------------------------------------------------------------
// decode: one thread per anchor.  head is fp32 [n][4+nc][N]; row reads are coalesced over anchors.
//   best = 0, cls = -1; for j: if (s > best) {best = s; cls = j;}          (:787-796, strict >)
//   keep if best >= conf_thr && cls >= 0                                    (:799)
//   box = cx/img_w, cy/img_h, w/img_w, h/img_h  with the REQUEST's dims     (:802-805)
// Survivors are compacted per wave with __ballot + popcount and one atomicAdd per wave; the arrival
// order is irrelevant because the NMS kernel sorts with the anchor index as the final tie-break.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ head, int nc, int N,
                                                     const FrameDesc* __restrict__ desc, float conf_thr,
                                                     Cand* __restrict__ cand, int* __restrict__ cand_count)
{
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const float* h = head + (size_t)f * (4 + nc) * N;
    bool pass = false;
    float best = 0.0f;
    int cls = -1;
    if (i < N) {
        for (int j = 0; j < nc; ++j) {
            const float s = h[(size_t)(4 + j) * N + i];
            if (s > best) { best = s; cls = j; }
        }
        pass = (best >= conf_thr) && (cls >= 0);
    }
    const unsigned long long mask = __ballot(pass);
    if (mask == 0ull) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(&cand_count[f], __popcll(mask));
    base = __shfl(base, 0);
    if (pass) {
        const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
        if (slot >= N) return;
        const FrameDesc d = desc[f];
        Cand c;
        c.x = h[i] / (float)d.w;
        c.y = h[(size_t)N + i] / (float)d.h;
        c.w = h[(size_t)2 * N + i] / (float)d.w;
        c.h = h[(size_t)3 * N + i] / (float)d.h;
        c.conf = best;
        c.cls = cls;
        c.anchor = i;
        c.pad_ = 0;
        cand[(size_t)f * N + slot] = c;
    }
}

hipError_t launch_decode(const float* head, int nc, int N, int n, const FrameDesc* desc, float conf_thr,
                         Cand* cand, int* cand_count, hipStream_t s)
{
    dim3 grid((N + 255) / 256, n);
    hipLaunchKernelGGL(decode_kernel, grid, dim3(256), 0, s, head, nc, N, desc, conf_thr, cand, cand_count);
    return hipGetLastError();
}

// calculateIoU (:881-909); boxes are centre-x, centre-y, w, h
__device__ __forceinline__ float iou_cxcywh(const Cand& a, const Cand& b)
{
    const float ax0 = a.x - a.w / 2, ay0 = a.y - a.h / 2, ax1 = a.x + a.w / 2, ay1 = a.y + a.h / 2;
    const float bx0 = b.x - b.w / 2, by0 = b.y - b.h / 2, bx1 = b.x + b.w / 2, by1 = b.y + b.h / 2;
    const float lo_x = (ax0 < bx0) ? bx0 : ax0, hi_x = (bx1 < ax1) ? bx1 : ax1;   // std::max / std::min
    const float lo_y = (ay0 < by0) ? by0 : ay0, hi_y = (by1 < ay1) ? by1 : ay1;
    const float dx = hi_x - lo_x, dy = hi_y - lo_y;
    const float ox = (0.0f < dx) ? dx : 0.0f;
    const float oy = (0.0f < dy) ? dy : 0.0f;
    const float inter = ox * oy;
    const float area_a = a.w * a.h, area_b = b.w * b.h;
    const float uni = area_a + area_b - inter;
    return (uni > 0) ? inter / uni : 0.0f;
}

// total order of applyNMS's sort (:846-851) with the anchor index breaking exact ties
__device__ __forceinline__ bool cand_before(const Cand& a, const Cand& b)
{
    if (a.cls != b.cls) return a.cls < b.cls;
    if (a.conf != b.conf) return a.conf > b.conf;
    return a.anchor < b.anchor;
}

// ------------------------------------------------------------------------------------------------
// NMS: one workgroup (16 waves) per frame.
//   1. rank sort by (class asc, confidence desc, anchor asc): rank = #candidates ordered before.
//   2. classes are independent (:866), so each class segment of the sorted list is handed to ONE
//      wave, which runs the reference's greedy loop: for kept i, lanes test j = i+1.. in parallel
//      and set removed[j] when IoU > thr (strict, :871).
//   3. kept flags are compacted in sorted order with ballot prefix sums and written to the slab.
// n <= NMS_LDS_CAP candidates are handled in LDS; beyond that the same code runs on a global
// scratch area (the reference has no cap on candidates, so neither does this kernel).
// ------------------------------------------------------------------------------------------------
#define NMS_LDS_CAP 1024
#define NMS_THREADS 1024
#define NMS_MAX_CLASSES 1024

__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const Cand* __restrict__ cand_all, int* __restrict__ cand_count,
                                                          int N, float iou_thr, int nc, Cand* __restrict__ scratch_all,
                                                          unsigned char* __restrict__ slabs, int cap, uint32_t tag0)
{
    __shared__ Cand lds_src[NMS_LDS_CAP];
    __shared__ Cand lds_sorted[NMS_LDS_CAP];
    __shared__ int seg_start[NMS_MAX_CLASSES];
    __shared__ int seg_end[NMS_MAX_CLASSES];
    __shared__ int wave_tot[NMS_THREADS / 64];
    __shared__ int run_base;

    const int f = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int n = cand_count[f];
    if (n > N) n = N;
    __syncthreads();
    if (tid == 0) cand_count[f] = 0;       // self-cleaning: the next frame's decode appends from 0 again (no memset launch)
    const bool in_lds = n <= NMS_LDS_CAP;
    const Cand* gsrc = cand_all + (size_t)f * N;
    const Cand* src = gsrc;
    Cand* sorted = in_lds ? lds_sorted : scratch_all + (size_t)f * N;

    const size_t slab_bytes = sizeof(zly_slab_header) + (size_t)cap * sizeof(zly_det);
    zly_slab_header* hdr = reinterpret_cast<zly_slab_header*>(slabs + (size_t)f * slab_bytes);
    zly_det* dets = reinterpret_cast<zly_det*>(hdr + 1);

    for (int c = tid; c < nc; c += NMS_THREADS) { seg_start[c] = -1; seg_end[c] = -1; }
    if (in_lds) {
        for (int i = tid; i < n; i += NMS_THREADS) lds_src[i] = gsrc[i];
        src = lds_src;
    }
    if (tid == 0) run_base = 0;
    __syncthreads();

    // 1. rank sort (the removed flag lives in pad_, cleared by decode)
    for (int i = tid; i < n; i += NMS_THREADS) {
        const Cand ci = src[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += cand_before(src[j], ci) ? 1 : 0;
        sorted[rank] = ci;
    }
    if (!in_lds) __threadfence();
    __syncthreads();

    // class segment boundaries
    for (int i = tid; i < n; i += NMS_THREADS) {
        const int c = sorted[i].cls;
        if (i == 0 || sorted[i - 1].cls != c) seg_start[c] = i;
        if (i == n - 1 || sorted[i + 1].cls != c) seg_end[c] = i + 1;
    }
    __syncthreads();

    // 2. greedy suppression, one wave per class segment
    if (n > 1) {
        for (int c = wave; c < nc; c += NMS_THREADS / 64) {
            const int s = seg_start[c], e = seg_end[c];
            if (s < 0 || e - s < 2) continue;
            for (int i = s; i < e - 1; ++i) {
                if (*reinterpret_cast<volatile int*>(&sorted[i].pad_)) continue;       // wave-uniform
                const Cand bi = sorted[i];
                for (int j = i + 1 + lane; j < e; j += 64) {
                    volatile int* rj = reinterpret_cast<volatile int*>(&sorted[j].pad_);
                    if (*rj) continue;
                    if (iou_cxcywh(bi, sorted[j]) > iou_thr) *rj = 1;
                }
                if (!in_lds) __threadfence();
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (!in_lds) __threadfence();
    __syncthreads();

    // 3. ordered compaction of the survivors
    for (int i0 = 0; i0 < n; i0 += NMS_THREADS) {
        const int i = i0 + tid;
        const bool keep = (i < n) && (sorted[i].pad_ == 0);
        const unsigned long long mask = __ballot(keep);
        if (lane == 0) wave_tot[wave] = __popcll(mask);
        __syncthreads();
        int off = run_base;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (keep) {
            const int o = off + __popcll(mask & ((1ull << lane) - 1ull));
            if (o < cap) {
                const Cand c = sorted[i];
                zly_det d;
                d.x = c.x; d.y = c.y; d.w = c.w; d.h = c.h;
                d.confidence = c.conf; d.class_id = c.cls;
                d.track_id = 0; d.pad_ = 0; d.timestamp = 0;     // track_id = 0 (:812); timestamp set by the host
                dets[o] = d;
            }
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < NMS_THREADS / 64; ++w) t += wave_tot[w];
            run_base += t;
        }
        __syncthreads();
    }
    if (tid == 0) {
        hdr->n_kept = run_base;
        hdr->n_candidates = n;
        hdr->flags = run_base > cap ? ZLY_SLAB_OVERFLOW : 0u;
        hdr->frame_tag = tag0 + (uint32_t)f;
    }
}

hipError_t launch_nms(const Cand* cand, int* cand_count, int N, int n, float iou_thr, int nc,
                      Cand* scratch, void* slabs, int cap, uint32_t tag0, hipStream_t s)
{
    if (nc > NMS_MAX_CLASSES) return hipErrorInvalidValue;
    hipLaunchKernelGGL(nms_kernel, dim3(n), dim3(NMS_THREADS), 0, s, cand, cand_count, N, iou_thr, nc, scratch,
                       (unsigned char*)slabs, cap, tag0);
    return hipGetLastError();
}

}  // namespace zly
