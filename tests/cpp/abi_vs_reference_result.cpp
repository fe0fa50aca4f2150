// abi_vs_reference_result.cpp -- the C ABI's return codes against the REFERENCE's own src/common/result.h (zero_latency::ErrorCode,
// result.h:14-48), included where it lies.  A translation unit of its own: the reference's types.h declares a second, conflicting
// `enum class ErrorCode : uint8_t` (types.h:84), one of the reasons the reference does not compile as a whole.
// Compiled by tests/test_abi.py in the build container only.
#include "result.h"         // the reference's, via -I <reference>/src/common
#include "zly.h"

// return codes of the C ABI == zero_latency::ErrorCode values (what the plugin's toErrorCode maps back)
using zero_latency::ErrorCode;
static_assert(ZLY_OK == (int)ErrorCode::OK && ZLY_ERR_INVALID_ARGUMENT == (int)ErrorCode::INVALID_ARGUMENT, "ErrorCode");
static_assert(ZLY_ERR_NOT_INITIALIZED == (int)ErrorCode::NOT_INITIALIZED && ZLY_ERR_INFERENCE == (int)ErrorCode::INFERENCE_ERROR, "ErrorCode");
static_assert(ZLY_ERR_MODEL_NOT_FOUND == (int)ErrorCode::MODEL_NOT_FOUND && ZLY_ERR_MODEL_LOAD == (int)ErrorCode::MODEL_LOAD_FAILED, "ErrorCode");
static_assert(ZLY_ERR_INVALID_INPUT == (int)ErrorCode::INVALID_INPUT && ZLY_ERR_SYSTEM == (int)ErrorCode::SYSTEM_ERROR, "ErrorCode");

int main() { return 0; }
