// bcos_internal.h -- shared by the translation units of libbcos_hip.so (not part of the ABI).
#ifndef BCOS_INTERNAL_H
#define BCOS_INTERNAL_H
#include <hip/hip_runtime.h>

// record an error for bcos_last_error_string() and return `code`
int bcos_set_error(int code, const char* msg);
// record a HIP runtime error; returns BCOS_E_LAUNCH
int bcos_set_hip_error(const char* what, hipError_t err);


struct bcos_tapconv_geom;
struct bcos_epilogue;
// narrow-output (Cout <= 8) path, bcos_skinny.hip: 1 = handled, 0 = not applicable, < 0 = error
int bcos_try_skinny(const float* a, const float* wt, const bcos_tapconv_geom& g, const bcos_epilogue& e, int M,
                    hipStream_t stream);

#endif
