// kernels_sppf.hip -- YOLOv8's SPPF block as ONE kernel (bf16): cv1 (1x1, Cin -> 128) -> three chained 5x5 max pools -> cv2 (1x1 over the concat
// [y | p1 | p2 | p3] = 512 -> Cout).  Replaces three conv / pool nodes of the reference's Ort::Session::Run (reference src/inference/onnx_engine.cpp:575-586;
// ultralytics SPPF: cv2(cat(y, m(y), m(m(y)), m(m(m(y)))))).
//
// Round 3 ran the block as three launches (weight-stationary 1x1 | sppf_pool_kernel | weight-stationary 1x1): 7.9 + 16.1 + 12.9 us at batch 64 for
// 1.8 + 1.8 + 2.7 us of attainable work, with the 512-channel concat buffer going through HBM twice.  A 13 x 13 x 128 map is 43 KB: the whole block fits
// one CU's LDS, and cv2 is a sum over its four sources, so the concat is never materialised:
//   x (one frame, HW <= 176 pixels x Cin)           -> LDS by LDS-DMA (pixel pitch Cin*2 + 32 B; aliases the pool buffers, which are not live yet)
//   cv1: 8 channel tiles x NCT pixel tiles x NK1 k-steps of v_mfma_f32_16x16x32_bf16 over 16 waves (4 tile pairs x 4 pixel groups), bias + SiLU -> TRUE
//   for s = 0..3:  acc += W2[:, s*128 .. +128) . TRUE      (TRUE = y, p1, p2, p3 in turn; this workgroup's share of cv2's output channels)
//                  p_{s+1} = pool5(p_s)                   row pass -> column pass on 16-byte channel groups, v_pk_max_i16 in a sortable domain
//   bias + SiLU -> HBM (16-byte NHWC stores)
// gfx950 has no packed bf16 max.  bf16 bit patterns compare like floats once the low 15 bits of negative values are flipped (T(x) = x ^ ((x >> 15) &
// 0x7fff) per half-word, an involution): the pools run on T(y) with v_pk_max_i16 -- two values per instruction, exact (max is a selection) -- and each
// stage's result is mapped back into TRUE for the MFMA phase.  (In f32 the pools were ~45 VALU instructions per pixel and channel group; here ~20.)
// One frame's block is 66 MFLOP: a CU needs ~7 us for it, so a frame is SPLIT over 2 or 4 workgroups by cv2 output channels; each recomputes cv1 and the
// pools (a third of the work) -- 256 workgroups at batch 64.  With a.dump (debug taps) y and the pooled maps are also written to the concat buffer.
#include "zly_internal.h"
#include "conv_device.h"

namespace zly {

typedef __attribute__((ext_vector_type(8))) short s16x8;

static constexpr int SPPF_C = 128;            // hidden width (YOLOv8n); the LDS plan below is made for it
static constexpr int SPPF_KC = SPPF_C / 32;   // k-steps per source map of cv2
static constexpr int SPPF_NW = 16;
static constexpr int SPPF_TP = SPPF_C * 2 + 32;   // TRUE pixel pitch (bytes): fragment reads as in conv1x1_ws_kernel
static constexpr int SPPF_MAXPX = 176;        // 11 pixel tiles

#ifdef ZLY_SPPF_DIAG
__device__ unsigned long long* g_sppf_diag = nullptr;            // diagnostic build only (tools/sppf_bench.hip): per-wave cycle sums of the phases
#define SPPFSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); dsum[k] += t_ - dT0; dT0 = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SPPFSTAMP(k) do { } while (0)
#endif

__device__ __forceinline__ u32x4 sortable(u32x4 v)
{
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const unsigned sg = v[i] & 0x80008000u; r[i] = v[i] ^ (sg - (sg >> 15)); }     // per half-word: negative -> flip the low 15 bits (0x8000 - 1 = 0x7fff, no borrow between the halves)
    return r;
}

template <int NK1, int NP2>      // NK1: cv1 k-steps (Cin / 32); NP2: channel-tile pairs of cv2 per workgroup (Cout / SPLIT / 32): 2 or 4
__global__ __launch_bounds__(SPPF_NW * 64) void sppf_fused_kernel(const SppfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NG2 = SPPF_NW / NP2;                 // pixel groups of the cv2 phases
    constexpr int MAXJ2 = (SPPF_MAXPX / 16 + NG2 - 1) / NG2;
    constexpr int MAXJ1 = 3;                           // cv1: 4 tile pairs x 4 pixel groups, up to 11 pixel tiles
    constexpr int NT = SPPF_NW * 64;
    constexpr int NIT = (SPPF_MAXPX * 16 + NT - 1) / NT;   // (pixel, 16-byte channel group) items per thread of a pool pass
    const int HW = a.H * a.W, NCT = (HW + 15) >> 4;
    const int XP = a.Cin * 2 + 32;
    // LDS, phase 1 (cv1):   [ x: 176 pixels x XP | w1: all of cv1's weights, in the order they lie in HBM ]
    //      phase 2 (pools): [ A | B | R ]  three maps of 176 x SPPF_TP bytes: A = the current source map as bf16 proper (what cv2's MFMAs read), B = the same map in the
    //                       sortable domain (what the pool reads), R = its row maxima (sortable)
    unsigned char* lx = smem;
    unsigned char* lw1 = smem + SPPF_MAXPX * XP;
    unsigned char* lt0 = smem;
    unsigned char* lt1 = smem + SPPF_MAXPX * SPPF_TP;
    unsigned char* lr = smem + 2 * SPPF_MAXPX * SPPF_TP;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, kq = lane >> 4;
    const int part = blockIdx.x, f = blockIdx.y;
#ifdef ZLY_SPPF_DIAG
    unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dT0 = __builtin_amdgcn_s_memtime();
    const unsigned long long dstart = dT0;
#endif

    // ---- x and cv1's weights -> LDS by LDS-DMA, every byte once per workgroup (round 4 v1 had each wave fetch its tile pair's 16 KB of weights itself:
    //      four waves per pair, 256 KB through the CU's vector memory path = most of the 7.5 k cycles this phase took).  Padding pieces, pixels beyond the
    //      map and units beyond the tile get an out-of-range offset: zeros. ----
    {
        const bf16_t* xin = static_cast<const bf16_t*>(a.x) + a.x_co;
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xin), 0, (unsigned)(((size_t)a.n * HW * a.x_cs - a.x_co) * 2), 0x00020000);
        const int upp = a.Cin >> 3, upitch = XP >> 4;
        const int ndma = (NCT * 16 * upitch + 63) >> 6;
        const float inv_upitch = 1.0f / (float)upitch;
        for (int k = wave; k < ndma; k += SPPF_NW) {
            const int u = k * 64 + lane;
            const int q = (int)(((float)u + 0.5f) * inv_upitch), pc = u - q * upitch;
            const bool ok = pc < upp && q < HW;
            const unsigned off = ok ? (unsigned)(((f * HW + q) * a.x_cs) * 2 + pc * 16) : 0x80000000u;
            if (q < NCT * 16)                              // lanes beyond x's last row write nothing (EXEC-masked): cv1's weights lie right behind it
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(lx + k * 1024), 16, off, 0, 0, 0);
        }
        const int w1_bytes = (SPPF_C / 16) * NK1 * 1024;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w1), 0, (unsigned)w1_bytes, 0x00020000);
        for (int k = wave; k < w1_bytes / 1024; k += SPPF_NW)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(lw1 + k * 1024), 16, (unsigned)(k * 1024 + lane * 16), 0, 0, 0);
    }
    const int g1 = wave & 3, pg1 = wave >> 2;
    const f32x4 b1lo = *reinterpret_cast<const f32x4*>(a.b1 + g1 * 32 + kq * 8), b1hi = *reinterpret_cast<const f32x4*>(a.b1 + g1 * 32 + kq * 8 + 4);
    // cv2: tile pair g2 of this workgroup's share, pixel group pg2
    const int g2 = wave % NP2, pg2 = wave / NP2;
    const int pair2 = part * NP2 + g2;                 // global pair index: output channels pair2 * 32 .. + 31
    const bf16_t* w2b = static_cast<const bf16_t*>(a.w2) + lane * 8;
    const int nk2 = 4 * SPPF_KC;
    auto load_w2 = [&](int s, bf16x8 (&w)[2][SPPF_KC]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int k = 0; k < SPPF_KC; ++k) w[t][k] = *reinterpret_cast<const bf16x8*>(w2b + ((size_t)(pair2 * 2 + t) * nk2 + s * SPPF_KC + k) * 512);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces have landed
    __syncthreads();
    SPPFSTAMP(0);

    // ---- cv1: 4 tile pairs x 4 pixel groups; weight and pixel fragments from LDS ----
    bf16x8 w2[2][SPPF_KC];
    bf16x8 yv[MAXJ1];
    {
        f32x4 acc[MAXJ1][2];
#pragma unroll
        for (int jj = 0; jj < MAXJ1; ++jj) { acc[jj][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[jj][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const unsigned char* wl = lw1 + (size_t)(g1 * 2) * NK1 * 1024 + lane * 16;
#pragma unroll
        for (int s = 0; s < NK1; ++s) {
            const bf16x8 wa = *reinterpret_cast<const bf16x8*>(wl + s * 1024), wb = *reinterpret_cast<const bf16x8*>(wl + (NK1 + s) * 1024);
#pragma unroll
            for (int jj = 0; jj < MAXJ1; ++jj) {
                const int j = min(pg1 + jj * 4, NCT - 1);          // a missing tile recomputes the last one; its result is dropped
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(lx + (j * 16 + p) * XP + s * 64 + kq * 16);
                acc[jj][0] = mma_step(wa, xf, acc[jj][0]);
                acc[jj][1] = mma_step(wb, xf, acc[jj][1]);
            }
        }
        load_w2(0, w2);                                    // cv2's first weights are on their way while the epilogue runs
#pragma unroll
        for (int jj = 0; jj < MAXJ1; ++jj) {
            f32x4 lo = acc[jj][0] + b1lo, hi = acc[jj][1] + b1hi;
#pragma unroll
            for (int r = 0; r < 4; ++r) { lo[r] = silu<bf16_t>(lo[r]); hi[r] = silu<bf16_t>(hi[r]); }
            yv[jj] = to_bf16x8(lo, hi);
        }
    }
    SPPFSTAMP(1);
    __syncthreads();                                       // every wave has read its last fragment of x / w1: their space becomes T0 | T1 | R
    bf16_t* cat = static_cast<bf16_t*>(a.cat);
#pragma unroll
    for (int jj = 0; jj < MAXJ1; ++jj) {
        const int j = pg1 + jj * 4, q = j * 16 + p;
        if (j >= NCT) continue;
        // y as it is (A) and in the sortable domain (B); lanes beyond the map write rows nobody pools (< 176)
        *reinterpret_cast<bf16x8*>(lt0 + q * SPPF_TP + (g1 * 32 + kq * 8) * 2) = yv[jj];
        *reinterpret_cast<u32x4*>(lt1 + q * SPPF_TP + (g1 * 32 + kq * 8) * 2) = sortable(__builtin_bit_cast(u32x4, yv[jj]));
        if (a.dump && part == 0 && q < HW) *reinterpret_cast<bf16x8*>(cat + ((size_t)(f * HW + q) * a.cat_cs + g1 * 32 + kq * 8)) = yv[jj];
    }
    __syncthreads();
    SPPFSTAMP(2);

    const float invW = 1.0f / (float)a.W;
    f32x4 acc2[MAXJ2][2];
#pragma unroll
    for (int jj = 0; jj < MAXJ2; ++jj) { acc2[jj][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[jj][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    unsigned char* const la = lt0;                          // A: bf16 proper
    unsigned char* const lb = lt1;                          // B: sortable
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        // ---- cv2 over source map s (A) ----
#pragma unroll
        for (int k = 0; k < SPPF_KC; ++k)
#pragma unroll
            for (int jj = 0; jj < MAXJ2; ++jj) {
                const int j = min(pg2 + jj * NG2, NCT - 1);
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(la + (j * 16 + p) * SPPF_TP + k * 64 + kq * 16);
                acc2[jj][0] = mma_step(w2[0][k], xf, acc2[jj][0]);
                acc2[jj][1] = mma_step(w2[1][k], xf, acc2[jj][1]);
            }
        SPPFSTAMP(3);
        if (s == 3) break;
        load_w2(s + 1, w2);                                // the next source's weights land while the pool runs
        // ---- pool stage s + 1, rows: B -> R.  No barrier in front: the phase above only reads A, and R's last readers left through the barrier
        //      that ended the previous stage.  A thread's items are all requested before the first max (one LDS round trip per pass, not three). ----
#pragma unroll 1
        for (int it0 = 0; it0 < NIT; it0 += 2) {               // two items per round: 40 registers of fragments in flight (three would spill beside w2 and the accumulators)
            s16x8 v[2][5];
            int qs[2], pcs[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int u = tid + (it0 + it) * NT;
                const int q = min(u >> 4, HW - 1), pc = u & 15;
                const int y = (int)(((float)q + 0.5f) * invW), x = q - y * a.W;
                qs[it] = ((u >> 4) < HW && it0 + it < NIT) ? q : -1; pcs[it] = pc;
                const unsigned char* row = lb + pc * 16;
#pragma unroll
                for (int d = -2; d <= 2; ++d) {
                    const int qq = (x + d >= 0 && x + d < a.W) ? q + d : q;       // outside the row: the centre again (max is idempotent)
                    v[it][d + 2] = *reinterpret_cast<const s16x8*>(row + qq * SPPF_TP);
                }
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const s16x8 m = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_elementwise_max(v[it][0], v[it][1]), __builtin_elementwise_max(v[it][3], v[it][4])), v[it][2]);
                if (qs[it] >= 0) *reinterpret_cast<s16x8*>(lr + qs[it] * SPPF_TP + pcs[it] * 16) = m;
            }
        }
        SPPFSTAMP(5);
        __syncthreads();
        SPPFSTAMP(4);
        // ---- columns: R -> B (sortable: the next stage's input) and A (mapped back: the next source of cv2).  Both are free: the row pass was B's last
        //      reader, cv2's phase above A's, and every wave has passed the barrier behind them. ----
#pragma unroll 1
        for (int it0 = 0; it0 < NIT; it0 += 2) {
            s16x8 v[2][5];
            int qs[2], pcs[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int u = tid + (it0 + it) * NT;
                const int q = min(u >> 4, HW - 1), pc = u & 15;
                const int y = (int)(((float)q + 0.5f) * invW);
                qs[it] = ((u >> 4) < HW && it0 + it < NIT) ? q : -1; pcs[it] = pc;
                const unsigned char* col = lr + pc * 16;
#pragma unroll
                for (int d = -2; d <= 2; ++d) {
                    const int qq = (y + d >= 0 && y + d < a.H) ? q + d * a.W : q;
                    v[it][d + 2] = *reinterpret_cast<const s16x8*>(col + qq * SPPF_TP);
                }
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const s16x8 m = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_elementwise_max(v[it][0], v[it][1]), __builtin_elementwise_max(v[it][3], v[it][4])), v[it][2]);
                if (qs[it] >= 0) {
                    const u32x4 tv = sortable(__builtin_bit_cast(u32x4, m));
                    *reinterpret_cast<s16x8*>(lb + qs[it] * SPPF_TP + pcs[it] * 16) = m;
                    *reinterpret_cast<u32x4*>(la + qs[it] * SPPF_TP + pcs[it] * 16) = tv;
                    if (a.dump && part == 0) *reinterpret_cast<u32x4*>(cat + ((size_t)(f * HW + qs[it]) * a.cat_cs + (s + 1) * SPPF_C + pcs[it] * 8)) = tv;
                }
            }
        }
        SPPFSTAMP(6);
        __syncthreads();
        SPPFSTAMP(7);
    }
    // ---- cv2 epilogue ----
    const int ch = pair2 * 32 + kq * 8;
    const f32x4 b2lo = *reinterpret_cast<const f32x4*>(a.b2 + ch), b2hi = *reinterpret_cast<const f32x4*>(a.b2 + ch + 4);
    bf16_t* out = static_cast<bf16_t*>(a.out);
#pragma unroll
    for (int jj = 0; jj < MAXJ2; ++jj) {
        const int j = pg2 + jj * NG2, q = j * 16 + p;
        if (j >= NCT || q >= HW) continue;
        f32x4 lo = acc2[jj][0] + b2lo, hi = acc2[jj][1] + b2hi;
#pragma unroll
        for (int r = 0; r < 4; ++r) { lo[r] = silu<bf16_t>(lo[r]); hi[r] = silu<bf16_t>(hi[r]); }
        *reinterpret_cast<bf16x8*>(out + ((size_t)(f * HW + q) * a.out_cs + a.out_co + ch)) = to_bf16x8(lo, hi);
    }
#ifdef ZLY_SPPF_DIAG
    if (lane == 0 && g_sppf_diag) {
        unsigned long long* o = g_sppf_diag + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SPPF_NW + wave) * 16;
        for (int i = 0; i < 8; ++i) o[i] = dsum[i];
        o[8] = __builtin_amdgcn_s_memtime() - dstart;
    }
#endif
}

static size_t sppf_lds_bytes(int cin, int hw)
{
    (void)hw;
    const size_t phase1 = (size_t)SPPF_MAXPX * ((size_t)cin * 2 + 32) + (size_t)(SPPF_C / 16) * (cin / 32) * 1024;      // x | cv1's weights
    const size_t phase2 = (size_t)3 * SPPF_MAXPX * SPPF_TP;                                                           // T0 | T1 | R
    return phase1 > phase2 ? phase1 : phase2;
}

typedef void (*sppf_fn)(const SppfArgs);
static sppf_fn pick_sppf(int nk1, int np2)
{
    if (nk1 == 8) return np2 == 2 ? sppf_fused_kernel<8, 2> : np2 == 4 ? sppf_fused_kernel<8, 4> : nullptr;
    return nullptr;
}

// does the fused kernel cover this block?  (YOLOv8n: 256 -> 128 -> 256 on maps of up to 176 pixels -- 13 x 13 at 416 x 416; everything else takes the
// three-launch path: at 640 x 640 three 400-pixel maps of 128 channels do not fit one CU's LDS)
bool sppf_fused_ok(int cin, int c, int cout, int H, int W)
{
    if (c != SPPF_C || cin != 256 || cout % 128 != 0 || cout > 256 || H * W > SPPF_MAXPX || H < 3 || W < 3) return false;
    return sppf_lds_bytes(cin, H * W) <= 160 * 1024;
}

// workgroups per frame (by cv2 output channels): 4 while that leaves the launch at or below one workgroup per CU, else 2
int sppf_split(int cout, int n) { return (cout / 4) % 64 == 0 && (long)n * 4 <= (long)num_cus() ? 4 : 2; }      // (one workgroup per CU: 146 KB of LDS)

hipError_t sppf_init()
{
    for (int np2 = 2; np2 <= 4; np2 += 2) {
        hipError_t r = hipFuncSetAttribute((const void*)pick_sppf(8, np2), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (r != hipSuccess) return r;
    }
    return hipSuccess;
}

hipError_t launch_sppf_fused(const SppfArgs& a, hipStream_t s)
{
    if (!sppf_fused_ok(a.Cin, a.c, a.Cout, a.H, a.W) || a.x_cs % 8 || a.x_co % 8 || a.out_cs % 8 || a.out_co % 8 || a.cat_cs % 8 || a.n < 1) return hipErrorInvalidValue;
    if ((size_t)a.n * a.H * a.W * (size_t)std::max(a.x_cs, std::max(a.out_cs, a.cat_cs)) * 2 >= ((size_t)1 << 31)) return hipErrorInvalidValue;      // 32-bit offsets
    const int split = a.split == 2 || a.split == 4 ? a.split : sppf_split(a.Cout, a.n);
    const int np2 = a.Cout / split / 32;
    sppf_fn fn = pick_sppf(a.Cin / 32, np2);
    if (!fn) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fn, dim3(split, a.n), dim3(SPPF_NW * 64), sppf_lds_bytes(a.Cin, a.H * a.W), s, a);
    return hipGetLastError();
}

}  // namespace zly
