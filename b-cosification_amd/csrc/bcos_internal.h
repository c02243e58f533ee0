// bcos_internal.h -- shared by the translation units of libbcos_hip.so (not part of the ABI).
#ifndef BCOS_INTERNAL_H
#define BCOS_INTERNAL_H
#include <hip/hip_runtime.h>

// record an error for bcos_last_error_string() and return `code`
int bcos_set_error(int code, const char* msg);
// record a HIP runtime error; returns BCOS_E_LAUNCH
int bcos_set_hip_error(const char* what, hipError_t err);

#endif
