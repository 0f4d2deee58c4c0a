// capi.hip -- status strings / version of libdimsum_hip.so (see include/dimsum_hip.h).
#include "common.hpp"

extern "C" const char *dimsum_status_string(int status) {
    switch (status) {
        case DIMSUM_OK: return "ok";
        case DIMSUM_ERR_NULL: return "a required pointer is NULL";
        case DIMSUM_ERR_DTYPE: return "unsupported dtype (expected float32, float16 or bfloat16)";
        case DIMSUM_ERR_SHAPE: return "unsupported shape (scan: dstate in {4, 8, 16, 32}, dim % n_groups == 0; conv: width 2..4; attention: head_dim in {24, 32, 48, 64, 72}; norm: cols <= 2048)";
        case DIMSUM_ERR_STRIDE: return "unsupported stride or alignment (innermost stride must be 1; in-tile offsets must fit 31 bits)";
        case DIMSUM_ERR_UNSUPPORTED: return "valid in the reference but out of scope here (complex A, constant B/C)";
        case DIMSUM_ERR_LAUNCH: return "HIP kernel launch failed";
        case DIMSUM_ERR_ABI: return "struct_size of a parameter struct (or of its extension) does not match this library: the caller was built against another include/dimsum_hip.h";
        default: return "unknown status";
    }
}

extern "C" int dimsum_abi_version(void) { return DIMSUM_ABI_VERSION; }
extern "C" const char *dimsum_target_arch(void) { return "gfx950"; }

// ---- measurement helpers: HIP events for the per-call timing_start_event / timing_stop_event fields of the scan parameters ----
extern "C" void *dimsum_event_create(void) {
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
extern "C" void dimsum_event_destroy(void *e) { if (e) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(e)); }
extern "C" float dimsum_event_elapsed_ms(void *start, void *stop) {
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)) != hipSuccess) return -1.f;
    return ms;
}
