// weights.cpp -- ZLYW reader + repacker (pure host code).  The file format is documented in
// tools/zly_model.py; it stands in for the .onnx file the reference loads in loadModel
// (reference src/inference/onnx_engine.cpp:957-1062).
#include "weights.h"
#include <algorithm>
#include "zly.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

namespace zly {

namespace {
#pragma pack(push, 1)
struct FileHeader {
    char magic[4];
    uint32_t version, nc, reg_max;
    uint32_t ch[5];
    uint32_t n_c2f[8];
    uint32_t num_convs;
};
struct FileRec {
    char name[48];
    uint32_t cin, cout, k, stride, act, wfmt;     // wfmt: 0 = fp32 weights, 1 = fp8 e4m3 + one int8 power-of-two exponent per output channel
    uint64_t w_off, b_off;
};
#pragma pack(pop)
static_assert(sizeof(FileHeader) == 4 + 4 * 17, "header layout");
static_assert(sizeof(FileRec) == 48 + 24 + 16, "record layout");
}  // namespace

const ConvRec* ModelFile::find(const std::string& name) const
{
    for (const ConvRec& c : convs)
        if (c.name == name) return &c;
    return nullptr;
}

int load_zlyw(const char* path, ModelFile* out, std::string* err)
{
    FILE* f = path ? fopen(path, "rb") : nullptr;
    if (!f) { *err = std::string("model file not found: ") + (path ? path : "(null)"); return ZLY_ERR_MODEL_NOT_FOUND; }
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> data((size_t)(size > 0 ? size : 0));
    const size_t got = data.empty() ? 0 : fread(data.data(), 1, data.size(), f);
    fclose(f);
    if (got != data.size() || data.size() < sizeof(FileHeader)) { *err = "short read / not a ZLYW file"; return ZLY_ERR_MODEL_LOAD; }
    FileHeader h;
    memcpy(&h, data.data(), sizeof h);
    if (memcmp(h.magic, "ZLYW", 4) != 0 || h.version != 1) { *err = "bad magic/version (expected ZLYW v1)"; return ZLY_ERR_MODEL_LOAD; }
    if (h.num_convs == 0 || h.num_convs > 4096 ||
        sizeof(FileHeader) + (size_t)h.num_convs * sizeof(FileRec) > data.size()) { *err = "corrupt conv table"; return ZLY_ERR_MODEL_LOAD; }
    if (h.nc < 1 || h.nc > 1024 || h.reg_max < 1 || h.reg_max > 64) { *err = "corrupt header (nc / reg_max out of range)"; return ZLY_ERR_MODEL_LOAD; }
    out->nc = (int)h.nc;
    out->reg_max = (int)h.reg_max;
    for (int i = 0; i < 5; ++i) out->ch[i] = (int)h.ch[i];
    for (int i = 0; i < 8; ++i) out->n_c2f[i] = (int)h.n_c2f[i];
    out->convs.clear();
    out->fp8_weights = false;
    out->convs.reserve(h.num_convs);
    for (uint32_t i = 0; i < h.num_convs; ++i) {
        FileRec r;
        memcpy(&r, data.data() + sizeof(FileHeader) + (size_t)i * sizeof(FileRec), sizeof r);
        ConvRec c;
        char nm[49];
        memcpy(nm, r.name, 48);
        nm[48] = 0;
        c.name = nm;
        c.cin = (int)r.cin; c.cout = (int)r.cout; c.k = (int)r.k; c.stride = (int)r.stride; c.act = (int)r.act;
        // widths are bounded before any product is formed (a crafted record must not wrap size_t), and the payload ranges
        // are checked without additions that can wrap: this file is re-read by the hot-reload watcher
        const bool dims_ok = c.cin > 0 && c.cout > 0 && c.cin <= 65536 && c.cout <= 65536 && (c.k == 1 || c.k == 3) && (c.stride == 1 || c.stride == 2);
        const size_t nw = dims_ok ? (size_t)c.cout * c.cin * c.k * c.k : 0;
        const size_t fsz = data.size();
        // payload: fp32 [cout][cin][k][k], or (wfmt 1) int8 exponent[cout] padded to 4 bytes, then e4m3 [cout][cin][k][k]
        const size_t exp_bytes = dims_ok ? ((size_t)c.cout + 3) / 4 * 4 : 0;
        const size_t wbytes = r.wfmt == 1 ? exp_bytes + nw : nw * 4;
        if (!dims_ok || r.wfmt > 1 || r.w_off > fsz || wbytes > fsz - r.w_off || r.b_off > fsz || (size_t)c.cout * 4 > fsz - r.b_off) {
            *err = "corrupt conv record: " + c.name;
            return ZLY_ERR_MODEL_LOAD;
        }
        c.w.resize(nw);
        c.b.resize((size_t)c.cout);
        if (r.wfmt == 1) {
            // dequantised here, once: w = e4m3 * 2^exp[cout].  3 mantissa bits times a power of two is exact in bf16, so the bf16
            // engine computes with exactly these values (fp8 storage, bf16 MFMA: see DESIGN.md "fp8 weights")
            const int8_t* ex = reinterpret_cast<const int8_t*>(data.data() + r.w_off);
            const uint8_t* q = data.data() + r.w_off + exp_bytes;
            const size_t per = nw / (size_t)c.cout;
            for (int co = 0; co < c.cout; ++co) {
                const float sc = ldexpf(1.0f, ex[co]);
                for (size_t i = 0; i < per; ++i) c.w[(size_t)co * per + i] = fp8_e4m3_to_f32(q[(size_t)co * per + i]) * sc;
            }
            out->fp8_weights = true;
        } else
        memcpy(c.w.data(), data.data() + r.w_off, nw * 4);
        memcpy(c.b.data(), data.data() + r.b_off, (size_t)c.cout * 4);
        out->convs.push_back(std::move(c));
    }
    return ZLY_OK;
}

// OCP fp8 e4m3 (e4m3fn: bias 7, no infinities, 0x7f / 0xff = NaN, subnormals m/8 * 2^-6) -> float
float fp8_e4m3_to_f32(uint8_t v)
{
    const int sign = v >> 7, ex = (v >> 3) & 15, man = v & 7;
    float mag;
    if (ex == 15 && man == 7) { uint32_t nan = 0x7fc00000u; memcpy(&mag, &nan, 4); return mag; }
    if (ex == 0) mag = (float)man * (1.0f / 512.0f);                   // man/8 * 2^-6
    else { uint32_t bits = ((uint32_t)(ex - 7 + 127) << 23) | ((uint32_t)man << 20); memcpy(&mag, &bits, 4); }
    return sign ? -mag : mag;
}

uint16_t f32_to_bf16_rne(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

void repack_conv(const std::vector<const ConvRec*>& srcs, int cin_store, int kstep, bool bf16,
                 std::vector<uint8_t>* w_out, std::vector<float>* bias_out, int* cout_total, int* cout_pad, int* nk,
                 bool pair_rows, int epl_override, const int* tap_slot)
{
    const int ks = srcs[0]->k, cin = srcs[0]->cin;
    int nslots = ks * ks;
    if (tap_slot) for (int t = 0; t < ks * ks; ++t) nslots = std::max(nslots, tap_slot[t] + 1);
    int ctot = 0;
    for (const ConvRec* s : srcs) ctot += s->cout;
    const int cpad = (ctot + 15) / 16 * 16;
    const int K = nslots * cin_store;
    const int nkk = (K + kstep - 1) / kstep;
    const size_t esz = bf16 ? 2 : 4;
    const int epl = epl_override > 0 ? epl_override : 16 / (int)esz;   // k values one lane feeds per k-step (one 16-byte fragment by default)
    w_out->assign((size_t)cpad * nkk * kstep * esz, 0);
    bias_out->assign((size_t)cpad, 0.0f);
    int co_base = 0;
    for (const ConvRec* s : srcs) {
        for (int co = 0; co < s->cout; ++co) {
            const int row = co_base + co;
            (*bias_out)[(size_t)row] = s->b[(size_t)co];
            // which MFMA tile / tile row computes output channel `row`.  With pair_rows, channels are dealt to tile
            // PAIRS so that MFMA lane group kq ends up with channels 8kq..8kq+3 of a 32-channel group in the even tile
            // and 8kq+4..8kq+7 in the odd tile: 8 consecutive channels per lane = one 16-byte NHWC store.
            int ct = row / 16, r = row % 16;
            if (pair_rows && (row / 32) * 32 + 32 <= cpad) {
                const int g = row / 32, w = row % 32;
                ct = 2 * g + ((w % 8) / 4);
                r = (w / 8) * 4 + (w % 4);
            }
            for (int ky = 0; ky < ks; ++ky)
                for (int kx = 0; kx < ks; ++kx)
                    for (int ci = 0; ci < cin; ++ci) {
                        const int k = (tap_slot ? tap_slot[ky * ks + kx] : ky * ks + kx) * cin_store + ci;
                        const int st = k / kstep, kk = k % kstep;
                        // 1 KiB tile in MFMA lane order: lane = (kk / epl) * 16 + r holds epl consecutive k
                        const size_t dst = ((size_t)ct * nkk + st) * 16 * kstep + ((size_t)(kk / epl) * 16 + r) * epl + kk % epl;
                        const float v = s->w[(((size_t)co * cin + ci) * ks + ky) * ks + kx];
                        if (bf16) {
                            const uint16_t h = f32_to_bf16_rne(v);
                            memcpy(w_out->data() + dst * 2, &h, 2);
                        } else {
                            memcpy(w_out->data() + dst * 4, &v, 4);
                        }
                    }
        }
        co_base += s->cout;
    }
    *cout_total = ctot;
    *cout_pad = cpad;
    *nk = nkk;
}

}  // namespace zly
