// weights.h -- ZLYW model file reader and the host-side weight repacker for the MFMA conv kernel.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace zly {

struct ConvRec {
    std::string name;
    int cin = 0, cout = 0, k = 0, stride = 0, act = 0;
    std::vector<float> w;     // [cout][cin][k][k]
    std::vector<float> b;     // [cout]
};

struct ModelFile {
    int nc = 0, reg_max = 0;
    bool fp8_weights = false;     // at least one conv is stored as fp8 e4m3 (BASELINE configs[4])
    int ch[5] = {0, 0, 0, 0, 0};
    int n_c2f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<ConvRec> convs;
    const ConvRec* find(const std::string& name) const;
};

// returns 0 or a ZLY_ERR_* code; *err receives a message
int load_zlyw(const char* path, ModelFile* out, std::string* err);

uint16_t f32_to_bf16_rne(float f);
float fp8_e4m3_to_f32(uint8_t v);

// Concatenates `srcs` along cout and lays the result out as the conv kernel reads it:
//   [cout_pad/16][nk][lane = (kk/epl)*16 + row][epl] with k = (ky*ks + kx)*cin_store + ci = step*kstep + kk,
//   zero padded: every 1 KiB tile is stored in MFMA lane order.
// pair_rows: output channels are dealt to MFMA tile pairs so that a lane holds 8 consecutive channels (the conv
// kernels' 16-byte epilogue stores); the bias stays in channel order.
// cin_store is the channel count of the activation tensor the conv reads (>= cin; the stem reads an
// 8-channel tensor for its 3 input channels).  bf16 -> 2-byte elements, else fp32.
void repack_conv(const std::vector<const ConvRec*>& srcs, int cin_store, int kstep, bool bf16,
                 std::vector<uint8_t>* w_out, std::vector<float>* bias_out, int* cout_total, int* cout_pad, int* nk,
                 bool pair_rows = false, int epl_override = 0,    // epl_override: k values per lane and k-step (4 for the 16x16x16 MFMA)
                 const int* tap_slot = nullptr);                  // optional [ks*ks]: tap (ky*ks + kx) takes k positions [slot*cin_store, +cin) instead of [tap*cin_store, +cin)

}  // namespace zly
