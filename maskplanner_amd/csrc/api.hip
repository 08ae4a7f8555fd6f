// ABI version and error strings.
#include "common.h"

extern "C" int mp_abi_version(void) { return MP_ABI_VERSION; }

extern "C" const char* mp_error_string(int code)
{
    switch (code) {
        case MP_OK: return "ok";
        case MP_EINVAL: return "invalid argument (null pointer or inconsistent dimension)";
        case MP_EUNSUPPORTED: return "size outside the range the gfx950 kernels are built for";
        case MP_EWORKSPACE: return "workspace too small";
        case MP_ELAUNCH: return "HIP launch failed";
        default: return "unknown error";
    }
}
