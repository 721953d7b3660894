// Library-level entry points of librubiks_hip.so: version, error strings, device selection and the
// host-side copies of the constant tables.
#include <stdio.h>
#include <string.h>

#include "rubiks_common.h"

using namespace rubiks;

extern "C" {

int rc_abi_version(void) { return RC_ABI_VERSION; }

const char *rc_error_string(int code) {
    static thread_local char buf[160];
    switch (code) {
        case RC_OK: return "ok";
        case RC_ERR_NULL: return "required pointer is NULL";
        case RC_ERR_ALIGN: return "pointer or stride is not 16-byte aligned";
        case RC_ERR_STRIDE: return "stride smaller than round_up(n, 16)";
        case RC_ERR_RANGE: return "size or index argument out of range";
        case RC_ERR_NODEVICE: return "no gfx950 (MI355X) device available";
        default: break;
    }
    if (code <= RC_ERR_HIP_BASE) {
        const hipError_t e = (hipError_t)(RC_ERR_HIP_BASE - code);
        snprintf(buf, sizeof(buf), "HIP error %d: %s", (int)e, hipGetErrorString(e));
        return buf;
    }
    snprintf(buf, sizeof(buf), "unknown librubiks_hip error %d", code);
    return buf;
}

int rc_init(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return RC_ERR_NODEVICE;
    RC_REQUIRE(device >= 0 && device < count, RC_ERR_RANGE);
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return hip_rc(e);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RC_ERR_NODEVICE;   // code objects are gfx950-only
    return hip_rc(hipSetDevice(device));
}

int rc_get_move_table(uint8_t *out576) {
    RC_REQUIRE(out576 != nullptr, RC_ERR_NULL);
    for (int a = 0; a < kActions; ++a)
        for (int k = 0; k < 2; ++k)
            for (int v = 0; v < kCodes; ++v) out576[(a * 2 + k) * kCodes + v] = kTables.lut[a][k][v];
    return RC_OK;
}

int rc_get_solved(int8_t *out20) {
    RC_REQUIRE(out20 != nullptr, RC_ERR_NULL);
    memcpy(out20, kTables.solved, kPlanes);
    return RC_OK;
}

}  // extern "C"
