// bcos_internal.h -- shared by the translation units of libbcos_hip.so (not part of the ABI).
#ifndef BCOS_INTERNAL_H
#define BCOS_INTERNAL_H
#include <hip/hip_runtime.h>

// record an error for bcos_last_error_string() and return `code`
int bcos_set_error(int code, const char* msg);
// record a HIP runtime error; returns BCOS_E_LAUNCH
int bcos_set_hip_error(const char* what, hipError_t err);
// current value of a bcos_option (include/bcos_hip.h); `option` must be a valid enumerator
#include <stdint.h>
int64_t bcos_option(int option);


// Development builds.  Some compile-time switches of the kernels drop work for TIMING experiments (the knock-outs D_KO, H2_KO, P_KO,
// AH_KO: wrong results by design) or select code paths that were measured and never validated as the product configuration
// (D_EARLY = 0, D_A_AUX != 0).  Such a value is accepted only together with -DBCOS_DEV_BUILD, and a library built with
// BCOS_DEV_BUILD says so: bcos_version() carries BCOS_VERSION_DEV_FLAG (include/bcos_hip.h) and the Python binding refuses to load it
// unless BCOS_ALLOW_DEV_BUILD=1 (bcos_hip/lib.py).  BCOS_DEV_SWITCH(NAME, default) goes right behind the #ifndef / #define / #endif
// that gives the switch its default.
#ifdef BCOS_DEV_BUILD
#define BCOS_DEV_SWITCH(NAME, DEFAULT) static_assert(true, "")
#else
#define BCOS_DEV_SWITCH(NAME, DEFAULT) \
    static_assert((NAME) == (DEFAULT), #NAME " is a development switch (timing-only or unvalidated code path): build with -DBCOS_DEV_BUILD")
#endif

struct bcos_tapconv_geom;
struct bcos_epilogue;
// narrow-output (Cout <= 8) path, bcos_skinny.hip: 1 = handled, 0 = not applicable, < 0 = error
int bcos_try_skinny(const float* a, const float* wt, const bcos_tapconv_geom& g, const bcos_epilogue& e, int M,
                    hipStream_t stream);

// several tap sets over one input (parity classes of a strided gradient) fused into one launch, bcos_skinny.hip
int bcos_try_skinny_group(const float* a, const float* const* wts, const bcos_tapconv_geom* gs, const bcos_epilogue* es,
                          int count, hipStream_t stream);

// hipFuncAttributeMaxDynamicSharedMemorySize is sticky per kernel: raise it only when a launch needs more than any
// earlier one did (one runtime call per kernel and size class instead of one per launch; none while a launch sequence
// is being captured into a hipGraph after a warm-up pass).  `high_water` is a per-kernel static of the caller.
#include <atomic>
inline hipError_t bcos_ensure_dynamic_lds(const void* fn, size_t bytes, std::atomic<size_t>& high_water) {
    if (bytes <= high_water.load(std::memory_order_acquire)) return hipSuccess;
    hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (err == hipSuccess) {
        size_t cur = high_water.load(std::memory_order_relaxed);
        while (cur < bytes && !high_water.compare_exchange_weak(cur, bytes, std::memory_order_release)) {}
    }
    return err;
}


// GELU gate Phi(x) = 0.5 (1 + erf(x / sqrt 2)) of MyGELU (bcosify_vit.py:27-32), one definition for every kernel that evaluates it (the
// fused epilogues, the standalone gate kernel): erf by Abramowitz & Stegun 7.1.26 -- 1 - (a1 t + ... + a5 t^5) exp(-x^2), t = 1 / (1 + p |x|),
// |error| <= 1.5e-7 -- on v_rcp_f32 / v_exp_f32: ~14 vector instructions where the library erff takes ~45 (the 192 -> 768 GELU launches
// of the ViT plan were 64 us of 257 slower than the same contraction without it).  Absolute error of the gate <= 1e-7.
#ifdef __HIPCC__
__device__ __forceinline__ float bcos_gelu_gate(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float tail = poly * t * __expf(-z * z);            // 1 - erf(z), z >= 0
    const float half_tail = 0.5f * tail;
    return x >= 0.f ? 1.0f - half_tail : half_tail;
}
#endif

#endif
