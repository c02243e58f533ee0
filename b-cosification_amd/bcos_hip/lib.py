"""ctypes binding of libbcos_hip.so (the C ABI declared in include/bcos_hip.h).

This is the only place that touches the shared library.  There is deliberately NO CPU
fallback: if the library is missing, or a kernel is asked to run on a non-HIP tensor,
the call fails loudly (the CPU restatement lives in oracle/ and is test infrastructure).
"""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import torch  # noqa: F401  -- MUST precede CDLL: torch ships its own libamdhip64; loading ours first would bring a
#                      second HIP runtime (/opt/rocm) into the process and torch then finds no device

PKG_ROOT = Path(__file__).resolve().parent.parent          # b-cosification_amd/
REPO_ROOT = PKG_ROOT.parent
LIB_PATH = Path(os.environ["BCOS_HIP_LIB"]) if os.environ.get("BCOS_HIP_LIB") else PKG_ROOT / "lib" / "libbcos_hip.so"
CSRC = PKG_ROOT / "csrc"
INCLUDE = REPO_ROOT / "include"
SOURCES = ["bcos_tapconv.hip", "bcos_skinny.hip", "bcos_elementwise.hip", "bcos_vit.hip", "bcos_render.hip", "bcos_train.hip",
           "bcos_abi.hip"]

BCOS_NONE, BCOS_CONV_EPS, BCOS_LINEAR_EPS = 0, 1, 2
BCOS_EPI_NORM_ONLY = 1
BCOS_EPI_FORCE_POW = 2
BCOS_EPI_SCALE_GATE_LSB = 4
BCOS_EPI_GATE2_FROM_MUL = 8
BCOS_EPI_MUL_FROM_ACT = 16
BCOS_EPI_UNIT_NORM_W = 32
BCOS_E_NOSUP = -95
ABI_VERSION = 9
VERSION_DEV_FLAG = 0x40000000          # include/bcos_hip.h: BCOS_VERSION_DEV_FLAG
TAPCONV_PARTS = 11


class BcosHipError(RuntimeError):
    code = None                    # the library's return code where the error came from a call (check)


class TapconvGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "N", "H", "W", "C", "P", "Q", "in_sh", "in_sw", "dh0", "dw0", "dstep_h", "dstep_w",
        "TH", "TW", "OH", "OW", "out_sh", "out_sw", "out_h0", "out_w0", "Cout",
        "a_pitch", "out_pitch", "norm_pitch", "out_cgroup", "groups")]


class Epilogue(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "bias", "ch_scale", "ch_shift", "addend", "mul", "mul2", "gate2", "relu_gate",
        "out", "out2", "scale_out", "norm_out", "out_absmax", "out2_absmax", "mul_norm", "mul_csc", "mul_csh", "col_scale", "row_scale", "a_sumsq",
        "out_imgmax", "out_imgmin_c", "rowadd", "rowadd_scale")] + [
        ("bcos_mode", C.c_int32), ("relu", C.c_int32), ("b", C.c_float), ("flags", C.c_int32), ("max_out", C.c_int32), ("addend_sub", C.c_int32)]


PREP_MAX_TAPS = 49


class WeightPrepJob(C.Structure):
    """include/bcos_hip.h: bcos_weight_prep_job"""
    _fields_ = [("src", C.c_void_p), ("bank", C.c_void_p), ("image", C.c_void_p),
                ("rows", C.c_int32), ("channels", C.c_int32), ("Cp", C.c_int32), ("taps", C.c_int32),
                ("row_stride", C.c_int32), ("ch_stride", C.c_int32), ("tap_offset", C.c_int32 * PREP_MAX_TAPS), ("row_offset", C.c_int32)]


class Operands(C.Structure):
    _fields_ = [("a", C.c_void_p), ("a_absmax", C.c_void_p), ("wt", C.c_void_p), ("wt_bf16x3", C.c_void_p),
                ("wt_f16x2", C.c_void_p), ("contraction", C.c_int32), ("a_imgmax", C.c_void_p), ("a_imgmin", C.c_void_p),
                ("a_imgmin_c", C.c_void_p)]


CONTRACT_DEFAULT, CONTRACT_F32, CONTRACT_BF16X3, CONTRACT_F16X2 = 0, 1, 2, 3

# name -> (restype, argtypes); mirrors include/bcos_hip.h one to one
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
SIGNATURES = {
    "bcos_version": (C.c_int, []),
    "bcos_last_error_string": (C.c_char_p, []),
    "bcos_set_contraction_mode": (C.c_int, [_I]),
    "bcos_get_contraction_mode": (C.c_int, []),
    "bcos_set_option": (C.c_int, [_I, _L]),
    "bcos_get_option": (C.c_int, [_I, C.POINTER(C.c_int64)]),
    "bcos_tapconv": (C.c_int, [_P, _P, C.POINTER(TapconvGeom), C.POINTER(Epilogue), _P]),
    "bcos_tapconv_ops": (C.c_int, [C.POINTER(Operands), C.POINTER(TapconvGeom), C.POINTER(Epilogue), _P]),
    "bcos_tapconv_fuses_image_range": (C.c_int, [C.POINTER(Operands), C.POINTER(TapconvGeom), C.POINTER(Epilogue)]),
    "bcos_image_absmax": (C.c_int, [_P, _P, _I, _I, _P]),
    "bcos_image_absrange": (C.c_int, [_P, _P, _P, _I, _I, _P]),
    "bcos_image_absrange_c": (C.c_int, [_P, _P, _P, _I, _I, _P]),
    "bcos_split_weights_f16x2_bytes": (C.c_int, [_I, _I, C.POINTER(C.c_int64)]),
    "bcos_split_weights_f16x2": (C.c_int, [_P, _P, _I, _I, _P]),
    "bcos_split_weights_f16x2_conv": (C.c_int, [_P, _P, _I, _I, _I, _P]),
    "bcos_weight_prep_batch": (C.c_int, [_P, _I, _I, _I, _P, _P]),
    "bcos_rows_absmax": (C.c_int, [_P, _P, _L, _I, _I, _P]),
    "bcos_split_weights_bytes": (C.c_int, [_I, _I, C.POINTER(C.c_int64)]),
    "bcos_split_weights": (C.c_int, [_P, _P, _I, _I, _P]),
    "bcos_tapconv_presplit": (C.c_int, [_P, _P, _P, C.POINTER(TapconvGeom), C.POINTER(Epilogue), _P]),
    "bcos_tapconv_group": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(TapconvGeom), C.POINTER(Epilogue), _I, _P]),
    "bcos_conv2d_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P] + [_I] * 13 + [_F, _P]),
    "bcos_linear_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _P]),
    "bcos_conv2d_dgrad_s1": (C.c_int, [_P, _P, _P] + [_I] * 9 + [_P]),
    "bcos_linear_dgrad": (C.c_int, [_P, _P, _P, _L, _I, _I, _P]),
    "bcos_train_scale_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _I, _P]),
    "bcos_weight_rownorm_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I, _L, _P]),
    "bcos_maxout_scatter": (C.c_int, [_P, _P, _P, _L, _I, _I, _P]),
    "bcos_patch_norm_bwd": (C.c_int, [_P, _P, _P] + [_I] * 15 + [_P]),
    "bcos_patch_norm_bwd_add": (C.c_int, [_P, _P, _P, _P] + [_I] * 15 + [_P]),
    "bcos_conv2d_wgrad": (C.c_int, [_P, _P, _P] + [_I] * 18 + [_P]),
    "bcos_conv2d_wgrad_ws_floats": (C.c_int, [_I] * 18 + [C.POINTER(C.c_int64)]),
    "bcos_conv2d_wgrad_ordered": (C.c_int, [_P, _P, _P, _P] + [_I] * 18 + [_P]),
    "bcos_colsum": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _P]),
    "bcos_colsum_ordered": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _P]),
    "bcos_colsum_ws_floats": (C.c_int, [_L, _I, C.POINTER(C.c_int64)]),
    "bcos_colsum_ws": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _L, _I, _P]),
    "bcos_channel_axpby": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "bcos_train_scale_bwd_absmax": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _I, _P]),
    "bcos_train_scale_bwd_bn": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _I, _P]),
    "bcos_channel_affine_rows": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "bcos_bn_train_ws_floats": (C.c_int, [_L, _I, C.POINTER(C.c_int64)]),
    "bcos_bn_batch_stats": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _F, _F, _P]),
    "bcos_relu_bwd_colsums": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _P]),
    "bcos_weight_rownorm_scale": (C.c_int, [_P, _P, _P, _I, _L, _P]),
    "bcos_mul": (C.c_int, [_P, _P, _P, _L, _P]),
    "bcos_stream_copy": (C.c_int, [_P, _P, _L, _P]),
    "bcos_weight_row_invnorm": (C.c_int, [_P, _P, _P, _I, _L, _P]),
    "bcos_rows_normalize": (C.c_int, [_P, _P, _P, _L, _I, _P]),
    "bcos_cosine_grad": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "bcos_maxout_scale": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _F, _P]),
    "bcos_maxout_expand": (C.c_int, [_P, _P, _P, _L, _I, _I, _P]),
    "bcos_prep_input": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "bcos_finalize_explanation": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "bcos_contrib_map": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "bcos_avgpool2d_fwd": (C.c_int, [_P, _P] + [_I] * 9 + [_P]),
    "bcos_avgpool2d_fwd_absmax": (C.c_int, [_P, _P, _P] + [_I] * 9 + [_P]),
    "bcos_avgpool2d_bwd": (C.c_int, [_P, _P, _P, _P] + [_I] * 9 + [_P]),
    "bcos_global_avgpool_logits": (C.c_int, [_P, _P, _I, _I, _I, _F, _F, _P]),
    "bcos_head_onehot_grad": (C.c_int, [_P, _P, _P, _I, _I, _I, _F, _P]),
    "bcos_head_rank1_grad": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "bcos_head_rank1_grad_ex": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "bcos_argmax_rows": (C.c_int, [_P, _P, _P, _I, _I, _P]),
    "bcos_channel_affine": (C.c_int, [_P, _P, _P, _P, _L, _I, _I, _P]),
    "bcos_channel_affine_add": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "bcos_relu_bwd": (C.c_int, [_P, _P, _P, _L, _P]),
    "bcos_layernorm_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _F, _P]),
    "bcos_layernorm_stats": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _F, _P]),
    "bcos_layernorm_bwd_detached": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "bcos_gelu_gate": (C.c_int, [_P, _P, _P, _L, _P]),
    "bcos_groupnorm_fwd": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "bcos_layernorm_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "bcos_layernorm_bwd_add": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "bcos_gelu_bwd": (C.c_int, [_P, _P, _P, _L, _P]),
    "bcos_groupnorm_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "bcos_attention_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "bcos_groupnorm_bwd_detached": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "bcos_add_rows_bcast": (C.c_int, [_P, _P, _L, _L, _P]),
    "bcos_attention_fwd": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "bcos_attention_bwd_v": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "bcos_finalize_explanation_patches": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "bcos_render_explanations": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P]),
    "bcos_box_filter": (C.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "bcos_localisation_fractions": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
}

_lib = None


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile csrc/*.hip for gfx950 into lib/libbcos_hip.so (hipcc cross-compiles without a GPU)."""
    srcs = [CSRC / s for s in SOURCES]
    deps = srcs + [CSRC / "bcos_internal.h", INCLUDE / "bcos_hip.h"]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # the compiler flags are part of the build's identity: objects (and the library) built with other BCOS_HIPCC_FLAGS are
    # never reused -- the flags are recorded in a stamp file next to the objects and a change forces a full rebuild
    stamp = LIB_PATH.parent / "obj" / "flags.stamp"
    flag_id = " ".join([hipcc] + os.environ.get("BCOS_HIPCC_FLAGS", "").split())
    if stamp.exists() and stamp.read_text() != flag_id:
        force = True
    if LIB_PATH.exists() and not force:
        if all(LIB_PATH.stat().st_mtime >= d.stat().st_mtime for d in deps):
            return LIB_PATH
    LIB_PATH.parent.mkdir(parents=True, exist_ok=True)
    # one object per (source, part) compiled in parallel, then one link: bcos_tapconv.hip is compiled in TAPCONV_PARTS slices
    # (-DBCOS_TAPCONV_PART=k selects which kernel instantiations a slice carries) because its ~30 kernels dominate the build
    objdir = LIB_PATH.parent / "obj"
    objdir.mkdir(parents=True, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", f"-I{INCLUDE}", f"-I{CSRC}"]
    flags += os.environ.get("BCOS_HIPCC_FLAGS", "").split()
    jobs = []
    for s in srcs:
        parts = range(TAPCONV_PARTS) if s.name == "bcos_tapconv.hip" else [None]
        for k in parts:
            obj = objdir / (s.stem + ("" if k is None else f"_p{k}") + ".o")
            extra = [] if k is None else [f"-DBCOS_TAPCONV_PART={k}"]
            stale = force or not obj.exists() or any(obj.stat().st_mtime < d.stat().st_mtime for d in (s, deps[-2], deps[-1]))
            jobs.append((obj, [hipcc] + flags + extra + ["-c", str(s), "-o", str(obj)], stale))
    from concurrent.futures import ThreadPoolExecutor

    def run(job):
        obj, cmd, stale = job
        if stale:
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
        return obj
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(run, jobs))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [str(o) for o in objs] + ["-o", str(LIB_PATH)], check=True)
    stamp.write_text(flag_id)
    return LIB_PATH


def load():
    """Load the shared library (once) and type every exported symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise BcosHipError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the B-cos hot path.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    v = lib.bcos_version()
    if v & VERSION_DEV_FLAG:
        # a -DBCOS_DEV_BUILD library: the only kind in which timing knock-outs (wrong results) can be compiled in (csrc/bcos_internal.h)
        if os.environ.get("BCOS_ALLOW_DEV_BUILD") != "1":
            raise BcosHipError(f"{LIB_PATH} is a development build (bcos_version() carries BCOS_VERSION_DEV_FLAG: compiled with "
                               "-DBCOS_DEV_BUILD, possibly with timing knock-outs that compute wrong results); set BCOS_ALLOW_DEV_BUILD=1 "
                               "to load it for a timing experiment, or rebuild without BCOS_HIPCC_FLAGS")
        print(f"[bcos_hip] WARNING: {LIB_PATH} is a DEVELOPMENT build (BCOS_ALLOW_DEV_BUILD=1): results may be wrong", file=sys.stderr)
        v &= ~VERSION_DEV_FLAG
    if v != ABI_VERSION:
        raise BcosHipError(f"libbcos_hip.so ABI version {v}, bindings expect {ABI_VERSION}")
    _lib = lib
    mode = os.environ.get("BCOS_CONTRACTION", "").lower()
    if mode in _MODE_CODES:
        lib.bcos_set_contraction_mode(_MODE_CODES[mode])
    # The library itself never reads the environment (include/bcos_hip.h: bcos_set_option).  For the A/B scripts under scripts/
    # the HOST layer translates BCOS_OPT_<NAME>=<int> once, here, into option calls.
    for name in OPTIONS:
        val = os.environ.get("BCOS_OPT_" + name.upper())
        if val is not None:
            set_option(name, int(val))
    _loaded_options.update({name: get_option(name) for name in OPTIONS})
    return lib


_loaded_options = {}


def reset_options():
    """Every option back to the value it had when the library was loaded (defaults + BCOS_OPT_* of the environment)."""
    for name, val in _loaded_options.items():
        set_option(name, val)


# include/bcos_hip.h: enum bcos_option
OPTIONS = {"tail_split": 0, "d_one_wg": 1, "epi_generic": 2, "h2_loop": 3, "patch": 4, "patch_wide": 5, "h2_tile": 6, "h2_tall": 7,
           "h2_tall_min": 8, "attention_f32": 9, "split_limit": 10, "lds_min_kb": 11, "patch_levels": 13, "h2_wide_cost": 14, "wgrad_wgs": 15}


OPTION_GEN = 0           # bumped by every set_option: host-side caches of values that depend on an option (ops: workspace sizes) compare it


def set_option(name: str, value: int):
    """Process-wide development / test switch of the library (bcos_set_option); returns the previous value."""
    global OPTION_GEN
    OPTION_GEN += 1
    old = get_option(name)
    check(load().bcos_set_option(OPTIONS[name], int(value)), f"bcos_set_option({name}, {value})")
    return old


def get_option(name: str) -> int:
    out = C.c_int64(0)
    check(load().bcos_get_option(OPTIONS[name], C.byref(out)), f"bcos_get_option({name})")
    return out.value


class option:
    """`with lib.option("patch", 0): ...` -- an option changed for the duration of a block (tests, A/B scripts)."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


_MODE_NAMES = {0: "f32", 1: "bf16x3", 2: "f16x2"}
_MODE_CODES = {"f32": 0, "fp32": 0, "0": 0, "bf16x3": 1, "1": 1, "f16x2": 2, "2": 2}


def get_contraction_mode() -> str:
    """The process-wide DEFAULT arithmetic of the contraction (include/bcos_hip.h); a call may override it."""
    return _MODE_NAMES[load().bcos_get_contraction_mode()]


def set_contraction_mode(mode: str):
    """'f32' (v_mfma_f32_32x32x2_f32), 'bf16x3' (exact 3-way bf16 split, 6 products) or 'f16x2' (scaled 2-way fp16 split,
    3 products; falls back to bf16x3 for launches without operand maxima) -- include/bcos_hip.h."""
    check(load().bcos_set_contraction_mode(_MODE_CODES[mode]), "bcos_set_contraction_mode")


def check(code: int, what: str):
    if code != 0:
        msg = load().bcos_last_error_string().decode()
        err = BcosHipError(f"{what} failed with code {code}: {msg}")
        err.code = int(code)
        raise err
