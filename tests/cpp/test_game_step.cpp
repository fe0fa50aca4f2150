// test_game_step.cpp -- CPU-only driver for host/zly_game_step.hpp.  usage: test_game_step <scenario.bin> <out.bin>
// scenario: u32 n_frames, then per frame {u8 init_now, u32 client, u8 game, u32 frame_id, u64 timestamp, u16 count, count x 40 B}
// out:      per frame {i32 error, u16 count, count x 40 B, u32 tracked_for_client, u32 next_track_id}
#include "zly_game_step.hpp"

#include <cstdio>
#include <cstring>

using namespace zero_latency;

template <typename T> static bool rd(FILE* f, T& v) { return fread(&v, sizeof v, 1, f) == 1; }
template <typename T> static void wr(FILE* f, const T& v) { fwrite(&v, sizeof v, 1, f); }

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* in = fopen(argv[1], "rb");
    FILE* out = fopen(argv[2], "wb");
    if (!in || !out) return 2;
    Cs16DetectionStep step;
    uint32_t n = 0;
    if (!rd(in, n)) return 2;
    for (uint32_t i = 0; i < n; ++i) {
        uint8_t init_now, game; uint32_t client; uint16_t count;
        GameState s;
        if (!rd(in, init_now) || !rd(in, client) || !rd(in, game) || !rd(in, s.frame_id) || !rd(in, s.timestamp) || !rd(in, count)) return 2;
        s.detections.resize(count);
        if (count && fread(s.detections.data(), sizeof(Detection), count, in) != count) return 2;
        if (init_now) step.initialize();
        auto r = step.processDetections(client, s, game);
        wr(out, static_cast<int32_t>(r.error().code));
        const uint16_t c = r.isOk() ? (uint16_t)r.value().detections.size() : 0;
        wr(out, c);
        if (c) fwrite(r.value().detections.data(), sizeof(Detection), c, out);
        wr(out, (uint32_t)step.trackedCount(client));
        wr(out, step.nextTrackId());
    }
    fclose(in); fclose(out);
    return 0;
}
