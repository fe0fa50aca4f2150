// test_gather.cpp -- the in-process RCCL gather (libzly_gather.so) behind the engine, on the devices of THIS box (world 1 on the one-GPU
// test boxes; the same sequence serves N devices): for every device an engine detects its share of the frames (frame i -> device i % N,
// slot i / N), zly_join orders the device's stream behind the engine's NMS, zly_gather_all all-gathers the slabs, and every device's
// receive buffer must hold, rank-major, exactly the bytes zly_read_slabs returns for each engine.
//   test_gather <weights.zlyw> <report.txt>
#include "zly.h"
#include "zly_gather.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>
#include <vector>

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    std::ofstream rep(argv[2]);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { rep << "devices=0\n"; return 0; }
    if (ndev > 4) ndev = 4;
    rep << "devices=" << ndev << "\n";
    const int per = 3, W = 416, H = 416;                              // frames per device per round
    const size_t fb = (size_t)W * H * 3;
    std::vector<int32_t> devs((size_t)ndev);
    for (int d = 0; d < ndev; ++d) devs[(size_t)d] = d;
    zly_gather* g = nullptr;
    if (zly_gather_create(ndev, devs.data(), &g) != 0) { std::fprintf(stderr, "zly_gather_create: %s\n", zly_gather_last_error()); return 3; }
    std::vector<zly_engine*> eng((size_t)ndev);
    std::vector<void*> d_frames((size_t)ndev), d_slabs((size_t)ndev), d_all((size_t)ndev), streams((size_t)ndev);
    size_t sb = 0;
    std::mt19937 rng(99);
    std::vector<uint8_t> frames((size_t)ndev * per * fb);
    for (auto& b : frames) b = (uint8_t)rng();
    for (int d = 0; d < ndev; ++d) {
        zly_config c; zly_default_config(&c);
        c.weights_path = argv[1]; c.max_batch = per; c.max_dets = 64; c.device = d; c.warmup_runs = 1; c.flags = ZLY_FLAG_NO_HEAD_TENSOR;
        if (zly_create(&c, &eng[(size_t)d]) != ZLY_OK) { std::fprintf(stderr, "zly_create: %s\n", zly_last_error()); return 3; }
        sb = zly_slab_bytes(eng[(size_t)d]);
        hipSetDevice(d);
        hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); streams[(size_t)d] = s;
        hipMalloc(&d_frames[(size_t)d], per * fb); hipMalloc(&d_slabs[(size_t)d], per * sb); hipMalloc(&d_all[(size_t)d], (size_t)ndev * per * sb);
        hipMemset(d_all[(size_t)d], 0xEE, (size_t)ndev * per * sb);
        // global frame i = slot * ndev + device  ->  this device's slot-th frame
        for (int slot = 0; slot < per; ++slot)
            hipMemcpy((uint8_t*)d_frames[(size_t)d] + slot * fb, frames.data() + ((size_t)slot * ndev + d) * fb, fb, hipMemcpyHostToDevice);
    }
    for (int round = 0; round < 2; ++round) {                          // twice: the communicators are reusable
        for (int d = 0; d < ndev; ++d) {
            if (zly_detect_device(eng[(size_t)d], per, d_frames[(size_t)d], W, H, d_slabs[(size_t)d], (uint32_t)(round * 100), streams[(size_t)d]) != ZLY_OK) return 4;
            if (zly_join(eng[(size_t)d], streams[(size_t)d], 0) != ZLY_OK) return 4;
        }
        if (zly_gather_all(g, (const void* const*)d_slabs.data(), d_all.data(), per * sb, streams.data()) != 0) { std::fprintf(stderr, "%s\n", zly_gather_last_error()); return 5; }
        for (int d = 0; d < ndev; ++d) { hipSetDevice(d); hipStreamSynchronize((hipStream_t)streams[(size_t)d]); }
    }
    // expected: each engine's own slabs (device 0 .. ndev-1), rank-major
    std::vector<uint8_t> want((size_t)ndev * per * sb), got((size_t)ndev * per * sb);
    for (int d = 0; d < ndev; ++d) { hipSetDevice(d); hipMemcpy(want.data() + (size_t)d * per * sb, d_slabs[(size_t)d], per * sb, hipMemcpyDeviceToHost); }
    int equal = 1, dets = 0, tags_ok = 1;
    for (int d = 0; d < ndev; ++d) {
        hipSetDevice(d);
        hipMemcpy(got.data(), d_all[(size_t)d], got.size(), hipMemcpyDeviceToHost);
        if (std::memcmp(got.data(), want.data(), got.size()) != 0) equal = 0;
    }
    for (int r = 0; r < ndev; ++r)
        for (int slot = 0; slot < per; ++slot) {
            const zly_slab_header* h = (const zly_slab_header*)(got.data() + ((size_t)r * per + slot) * sb);
            dets += h->n_kept;
            if (h->frame_tag != (uint32_t)(100 + slot)) tags_ok = 0;
        }
    rep << "gathered_equals_engine_slabs=" << equal << "\ndetections=" << dets << "\ntags_ok=" << tags_ok << "\nbytes_per_rank=" << per * sb << "\n";
    for (int d = 0; d < ndev; ++d) zly_destroy(eng[(size_t)d]);
    zly_gather_destroy(g);
    return equal && tags_ok ? 0 : 6;
}
