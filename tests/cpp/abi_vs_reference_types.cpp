// abi_vs_reference_types.cpp -- the one reference-held pin of the drop-in boundary: this translation unit includes the REFERENCE's own
// src/common/types.h (it compiles on its own; nothing of it is copied into this repository) next to include/zly.h and the plugin's
// compat header, and the compiler checks that zly_det / the re-declared Detection are layout-identical to zero_latency::Detection
// -- the record the reference memcpy's raw onto the wire (src/common/protocol.h:563-566).  Compiled by tests/test_abi.py in the build
// container only (`g++ -fsyntax-only -I/root/reference/src/common`); the GPU box has no /root/reference and skips it.
#include <cstddef>
#include "types.h"          // the reference's, via -I <reference>/src/common
#include "zly.h"

using RefDet = zero_latency::Detection;
using RefBox = zero_latency::BoundingBox;

static_assert(sizeof(RefDet) == sizeof(zly_det), "sizeof(Detection)");
static_assert(alignof(RefDet) == alignof(zly_det), "alignof(Detection)");
static_assert(offsetof(RefDet, box) == offsetof(zly_det, x), "box");
static_assert(offsetof(RefDet, box) + offsetof(RefBox, x) == offsetof(zly_det, x), "box.x");
static_assert(offsetof(RefDet, box) + offsetof(RefBox, y) == offsetof(zly_det, y), "box.y");
static_assert(offsetof(RefDet, box) + offsetof(RefBox, width) == offsetof(zly_det, w), "box.width");
static_assert(offsetof(RefDet, box) + offsetof(RefBox, height) == offsetof(zly_det, h), "box.height");
static_assert(offsetof(RefDet, confidence) == offsetof(zly_det, confidence), "confidence");
static_assert(offsetof(RefDet, class_id) == offsetof(zly_det, class_id), "class_id");
static_assert(offsetof(RefDet, track_id) == offsetof(zly_det, track_id), "track_id");
static_assert(offsetof(RefDet, timestamp) == offsetof(zly_det, timestamp), "timestamp");
static_assert(sizeof(((RefDet*)nullptr)->class_id) == sizeof(((zly_det*)nullptr)->class_id), "class_id width");
static_assert(sizeof(((RefDet*)nullptr)->timestamp) == sizeof(((zly_det*)nullptr)->timestamp), "timestamp width");
// GameState: the members the plugin fills (frame_id, timestamp, detections) exist with the reference's types
static_assert(sizeof(((zero_latency::GameState*)nullptr)->frame_id) == 4 && sizeof(((zero_latency::GameState*)nullptr)->timestamp) == 8, "GameState");
static_assert(sizeof(decltype(zero_latency::GameState::detections)::value_type) == sizeof(zly_det), "GameState::detections element");

int main() { return 0; }
