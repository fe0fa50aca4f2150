// conv_device.h -- device helpers shared by the MFMA convolution kernels (kernels_conv.hip, kernels_pair.hip):
// fragment types, the MFMA step, the fused SiLU and the NHWC channel-group loads/stores.
#pragma once
#include "zly_internal.h"

namespace zly {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; static constexpr int EPL = 8; static constexpr int KSTEP = 32; };
template <> struct Frag<float>  { typedef f32x4  type; static constexpr int EPL = 4; static constexpr int KSTEP = 16; };

__device__ __forceinline__ f32x4 mma_step(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// 16x16x4 f32: lane l supplies A[l&15][l>>4] and B[l>>4][l&15].  Element j of the 4 floats each lane
// loaded is used by MFMA j, i.e. MFMA j sums k = {4q + j : q = 0..3}; both operands use the same
// permutation, and over j = 0..3 every k of the 16-wide step is covered exactly once.
__device__ __forceinline__ f32x4 mma_step(f32x4 a, f32x4 b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
}

template <typename T> __device__ __forceinline__ float silu(float v);
template <> __device__ __forceinline__ float silu<float>(float v) { return v / (1.0f + expf(-v)); }
// bf16 path: v_exp_f32 + v_rcp_f32 (1 ulp each), 5 VALU instead of the 16 of an IEEE divide; the result is
// rounded to bf16 (8 bits) anyway.  The epilogue was the largest VALU consumer of the LDS kernel.
template <> __device__ __forceinline__ float silu<bf16_t>(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.442695041f)); }

// fp32 -> bf16 of whole vectors: __builtin_convertvector becomes v_cvt_pk_bf16_f32 with two live operands; element-wise casts became one
// convert per value plus v_perm packing (same round-to-nearest-even, same bits)
__device__ __forceinline__ bf16x4 to_bf16x4(f32x4 v) { return __builtin_convertvector(v, bf16x4); }
__device__ __forceinline__ bf16x8 to_bf16x8(f32x4 a, f32x4 b) {
    typedef __attribute__((ext_vector_type(8))) float f32x8;
    const f32x8 ab = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_convertvector(ab, bf16x8);
}
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v) { *reinterpret_cast<bf16x4*>(p) = to_bf16x4(v); }
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 load4(const bf16_t* p) {
    bf16x4 i = *reinterpret_cast<const bf16x4*>(p);
    f32x4 o = {(float)i[0], (float)i[1], (float)i[2], (float)i[3]};
    return o;
}
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// 8 consecutive channels (two MFMA tiles of a pair): one 16-byte bf16 store / two 16-byte fp32 stores
__device__ __forceinline__ void store8(bf16_t* p, f32x4 a, f32x4 b) { *reinterpret_cast<bf16x8*>(p) = to_bf16x8(a, b); }
__device__ __forceinline__ void store8(float* p, f32x4 a, f32x4 b) { *reinterpret_cast<f32x4*>(p) = a; *reinterpret_cast<f32x4*>(p + 4) = b; }
__device__ __forceinline__ void load8(const bf16_t* p, f32x4& a, f32x4& b) {
    const bf16x8 i = *reinterpret_cast<const bf16x8*>(p);
    a = f32x4{(float)i[0], (float)i[1], (float)i[2], (float)i[3]};
    b = f32x4{(float)i[4], (float)i[5], (float)i[6], (float)i[7]};
}
__device__ __forceinline__ void load8(const float* p, f32x4& a, f32x4& b) { a = *reinterpret_cast<const f32x4*>(p); b = *reinterpret_cast<const f32x4*>(p + 4); }

// Hides a (constant) byte offset from the compiler, at no instruction: two 8-byte LDS reads a CONSTANT distance apart (the taps of a row: kx * pitch) are fused
// into one ds_read2_b64, and that instruction is served like ds_write -- in groups of 16 lanes over 32 banks, at half the bytes per clock (MI355X guide, LDS
// table) -- so a pixel pitch that is conflict-free for ds_read_b64 (two groups of 32 lanes over 64 banks: 48 B at stride 1, 40 B at stride 2) is 2-way
// conflicted there.  Round 4 PMC: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.34 on c2f_kernel<16> and 0.23 on the front kernel came from exactly these reads.
// With the tap offsets in registers of unknown value the reads stay single.  (A volatile access also prevents the fusion -- and serialises the reads: the
// front kernel's last phase 2.9 k -> 5.5 k cycles, c2f_kernel<16>'s 3x3 phases 5 k -> 10 k: measured and dropped.)
__device__ __forceinline__ int opaque_offset(int v) { asm("" : "+v"(v)); return v; }

// Output-channel map of MFMA tile `tile` (global tile index), lane group kq: with the pair permutation of
// weights.cpp (all tiles below `paired_tiles`) a lane holds channels g*32 + kq*8 + half*4 .. +3; otherwise
// tile*16 + kq*4 .. +3.
__device__ __forceinline__ int tile_channel(int tile, int kq, int paired_tiles) {
    return tile < paired_tiles ? (tile >> 1) * 32 + kq * 8 + (tile & 1) * 4 : tile * 16 + kq * 4;
}

}  // namespace zly
