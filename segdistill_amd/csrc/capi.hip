// capi.hip -- version / error / tunable entry points of libsegdistill_hip.so.
#include <hip/hip_runtime.h>
#include <string.h>

#include "sd_common.h"

extern "C" {

int sd_abi_version(void) { return SD_ABI_VERSION; }

const char *sd_error_string(int code) {
    switch (code) {
        case SD_OK: return "ok";
        case SD_E_NULL: return "required pointer is NULL";
        case SD_E_SHAPE: return "invalid shape";
        case SD_E_DTYPE: return "unknown dtype code";
        case SD_E_WORKSPACE: return "workspace too small or misaligned";
        case SD_E_ALIGN: return "operand pointer misaligned";
        case SD_E_UNSUPPORTED: return "unsupported configuration";
        default: break;
    }
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "unknown segdistill error";
}

int sd_set_tunable(const char *key, int value) {
    if (!key) return SD_E_NULL;
    int rc = sd::cgd_tunable(key, 1, value);
    if (rc == SD_E_UNSUPPORTED) rc = sd::cgd_up_tunable(key, 1, value);
    if (rc == SD_E_UNSUPPORTED) rc = sd::sra_tunable(key, 1, value);
    if (rc == SD_E_UNSUPPORTED) rc = sd::token_gemm_tunable(key, 1, value);
    if (rc == SD_E_UNSUPPORTED) rc = sd::ce_tunable(key, 1, value);
    if (rc == SD_E_UNSUPPORTED) rc = sd::headfuse_tunable(key, 1, value);
    if (rc == SD_E_UNSUPPORTED) rc = sd::wgrad_tn_tunable(key, 1, value);
    if (rc == SD_E_UNSUPPORTED) rc = sd::tok_gemm_bf16_tunable(key, 1, value);
    if (rc == SD_E_UNSUPPORTED) rc = sd::align_stream_tunable(key, 1, value);
    return rc;
}

int sd_get_tunable(const char *key) {
    if (!key) return SD_E_NULL;
    int rc = sd::cgd_tunable(key, 0, 0);
    if (rc == SD_E_UNSUPPORTED) rc = sd::cgd_up_tunable(key, 0, 0);
    if (rc == SD_E_UNSUPPORTED) rc = sd::sra_tunable(key, 0, 0);
    if (rc == SD_E_UNSUPPORTED) rc = sd::token_gemm_tunable(key, 0, 0);
    if (rc == SD_E_UNSUPPORTED) rc = sd::ce_tunable(key, 0, 0);
    if (rc == SD_E_UNSUPPORTED) rc = sd::headfuse_tunable(key, 0, 0);
    if (rc == SD_E_UNSUPPORTED) rc = sd::wgrad_tn_tunable(key, 0, 0);
    if (rc == SD_E_UNSUPPORTED) rc = sd::tok_gemm_bf16_tunable(key, 0, 0);
    if (rc == SD_E_UNSUPPORTED) rc = sd::align_stream_tunable(key, 0, 0);
    return rc;
}

}  // extern "C"
