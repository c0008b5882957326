"""ctypes binding of libddsp_hip.so (C ABI in include/ddsp_hip.h).

There is deliberately no fallback: if the HIP library is missing or a launch
fails, the caller gets an exception -- never a silent PyTorch/CPU path.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("DDSP_HIP_LIB", os.path.join(_DIR, "libddsp_hip.so"))  # override: A/B builds (tools/ab_bench.sh)
ABI_VERSION = 4

_lib = None


class DdspHipError(RuntimeError):
    pass


def build(force: bool = False, jobs: int = 4) -> str:
    """Compile csrc/*.hip for gfx950 into libddsp_hip.so (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_DIR, "csrc"), f"-j{jobs}"]
    if force:
        args.append("-B")
    r = subprocess.run(args, capture_output=True, text=True)
    if r.returncode != 0:
        raise DdspHipError("building libddsp_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return SO_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise DdspHipError(
            f"{SO_PATH} is missing: the DDSP hot path has no CPU fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C ddsp-pytorch_amd/csrc`.")
    L = ctypes.CDLL(SO_PATH)
    vp, i32, u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64
    L.ddsp_hip_abi_version.restype = i32
    L.ddsp_hip_abi_version.argtypes = []
    # first of all: a stale library (it is git-ignored and not rebuilt on import) must say "rebuild", not fail on a missing symbol
    if L.ddsp_hip_abi_version() != ABI_VERSION:
        raise DdspHipError(f"{SO_PATH} has ABI {L.ddsp_hip_abi_version()}, expected {ABI_VERSION}: rebuild "
                           "(`make -C ddsp-pytorch_amd/csrc` or __graft_entry__.build())")
    L.ddsp_test_hooks_enabled.restype = i32
    L.ddsp_test_hooks_enabled.argtypes = []
    L.ddsp_osc_scratch_bytes.restype = ctypes.c_size_t
    L.ddsp_osc_scratch_bytes.argtypes = [i32, i32, i32]
    L.ddsp_osc_forward.restype = i32
    L.ddsp_osc_forward.argtypes = [vp] * 8 + [i32] * 5 + [vp]
    L.ddsp_osc_forward_ex.restype = i32
    L.ddsp_osc_forward_ex.argtypes = [vp] * 8 + [i32] * 5 + [ctypes.c_uint, vp]
    L.ddsp_osc_set_path.restype = i32
    L.ddsp_osc_set_path.argtypes = [i32]
    L.ddsp_osc_clock.restype = i32
    L.ddsp_osc_clock.argtypes = [vp, i32, i32, i32, i32, i32, ctypes.POINTER(ctypes.c_double), vp]
    L.ddsp_osc_plan.restype = i32
    L.ddsp_osc_plan.argtypes = [i32] * 5 + [ctypes.POINTER(i32), i32]
    L.ddsp_noise_forward.restype = i32
    L.ddsp_noise_forward.argtypes = [vp, vp, vp, i32, i32, i32, i32, u64, u64, i32, vp]
    L.ddsp_noise_forward_counter.restype = i32
    L.ddsp_noise_forward_counter.argtypes = [vp, vp, i32, i32, i32, i32, u64, vp, i32, vp]
    L.ddsp_noise_workspace_bytes.restype = ctypes.c_size_t
    L.ddsp_noise_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.ddsp_noise_forward_ws.restype = i32
    L.ddsp_noise_forward_ws.argtypes = [vp, vp, vp, i32, i32, i32, i32, u64, u64, vp, i32, vp, ctypes.c_size_t, vp]
    L.ddsp_noise_backward_ws.restype = i32
    L.ddsp_noise_backward_ws.argtypes = [vp, vp, vp, i32, i32, i32, i32, u64, u64, vp, vp, ctypes.c_size_t, vp]
    L.ddsp_osc_set_tiling.restype = i32
    L.ddsp_osc_set_tiling.argtypes = [i32]
    L.ddsp_osc_backward_scratch_bytes.restype = ctypes.c_size_t
    L.ddsp_osc_backward_scratch_bytes.argtypes = [i32, i32, i32]
    L.ddsp_osc_backward.restype = i32
    L.ddsp_osc_backward.argtypes = [vp] * 8 + [i32] * 5 + [vp]
    L.ddsp_noise_backward.restype = i32
    L.ddsp_noise_backward.argtypes = [vp, vp, vp, i32, i32, i32, i32, u64, u64, vp]
    L.ddsp_noise_backward_counter.restype = i32
    L.ddsp_noise_backward_counter.argtypes = [vp, vp, i32, i32, i32, i32, u64, vp, vp]
    L.ddsp_noise_set_generic.restype = i32
    L.ddsp_noise_set_generic.argtypes = [i32]
    L.ddsp_profile_enable.restype = i32
    L.ddsp_profile_enable.argtypes = [i32]
    L.ddsp_noise_set_residency.restype = i32
    L.ddsp_noise_set_residency.argtypes = [i32]
    L.ddsp_noise_get_residency.restype = i32
    L.ddsp_noise_get_residency.argtypes = []
    L.ddsp_profile_select.restype = i32
    L.ddsp_profile_select.argtypes = [ctypes.c_uint]
    L.ddsp_profile_read.restype = i32
    L.ddsp_profile_read.argtypes = [ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_float), i32]
    L.ddsp_gru_scratch_bytes.restype = ctypes.c_size_t
    L.ddsp_gru_scratch_bytes.argtypes = [i32, i32]
    L.ddsp_gru_max_batch.restype = i32
    L.ddsp_gru_max_batch.argtypes = [i32, i32]
    L.ddsp_gru_forward.restype = i32
    L.ddsp_gru_forward.argtypes = [vp] * 9 + [i32] * 3 + [vp]
    L.ddsp_gru_backward.restype = i32
    L.ddsp_gru_backward.argtypes = [vp] * 11 + [i32] * 3 + [vp]
    L.ddsp_gru_forward_bf16.restype = i32
    L.ddsp_gru_forward_bf16.argtypes = [vp] * 9 + [i32] * 3 + [vp]
    L.ddsp_gru_backward_bf16.restype = i32
    L.ddsp_gru_backward_bf16.argtypes = [vp] * 11 + [i32] * 4 + [vp]
    L.ddsp_gru_set_mode.restype = i32
    L.ddsp_gru_set_mode.argtypes = [i32]
    L.ddsp_gru_set_fault_step.restype = i32
    L.ddsp_gru_set_fault_step.argtypes = [i32]
    L.ddsp_gru_status.restype = i32
    L.ddsp_gru_status.argtypes = [vp, ctypes.POINTER(i32)]
    L.ddsp_spectral_loss_scratch_bytes.restype = ctypes.c_size_t
    L.ddsp_spectral_loss_scratch_bytes.argtypes = []
    L.ddsp_spectral_loss.restype = i32
    L.ddsp_spectral_loss.argtypes = [vp, vp, vp, vp, vp, ctypes.c_long, ctypes.c_float, ctypes.c_float, vp]
    L.ddsp_scaled_sigmoid_forward.restype = i32
    L.ddsp_scaled_sigmoid_forward.argtypes = [vp, vp, ctypes.c_long, vp]
    L.ddsp_scaled_sigmoid_backward.restype = i32
    L.ddsp_scaled_sigmoid_backward.argtypes = [vp, vp, vp, ctypes.c_long, vp]
    L.ddsp_heads_sigmoid_forward.restype = i32
    L.ddsp_heads_sigmoid_forward.argtypes = [vp] * 4 + [ctypes.c_long, i32, i32, i32, i32, vp]
    L.ddsp_heads_sigmoid_backward.restype = i32
    L.ddsp_heads_sigmoid_backward.argtypes = [vp] * 5 + [ctypes.c_long, i32, i32, i32, i32, vp]
    L.ddsp_ln_lrelu_scratch_bytes.restype = ctypes.c_size_t
    L.ddsp_ln_lrelu_scratch_bytes.argtypes = [i32]
    L.ddsp_ln_lrelu_forward.restype = i32
    L.ddsp_ln_lrelu_forward.argtypes = [vp] * 6 + [ctypes.c_long, i32, ctypes.c_float, ctypes.c_float, vp]
    L.ddsp_ln_lrelu_backward.restype = i32
    L.ddsp_ln_lrelu_backward.argtypes = [vp] * 11 + [ctypes.c_long, i32, ctypes.c_float, vp]
    L.ddsp_ln_lrelu_forward_16.restype = i32
    L.ddsp_ln_lrelu_forward_16.argtypes = [vp] * 6 + [ctypes.c_long, i32, ctypes.c_float, ctypes.c_float, i32, vp]
    L.ddsp_ln_lrelu_backward_16.restype = i32
    L.ddsp_ln_lrelu_backward_16.argtypes = [vp] * 11 + [ctypes.c_long, i32, ctypes.c_float, i32, vp]
    L.ddsp_outer_ln_lrelu_scratch_bytes.restype = ctypes.c_size_t
    L.ddsp_outer_ln_lrelu_scratch_bytes.argtypes = [i32]
    L.ddsp_outer_ln_lrelu_forward.restype = i32
    L.ddsp_outer_ln_lrelu_forward.argtypes = [vp] * 8 + [ctypes.c_long, i32, ctypes.c_float, ctypes.c_float, i32, vp]
    L.ddsp_outer_ln_lrelu_backward.restype = i32
    L.ddsp_outer_ln_lrelu_backward.argtypes = [vp] * 13 + [ctypes.c_long, i32, ctypes.c_float, i32, vp]
    L.ddsp_colsum_scratch_bytes.restype = ctypes.c_size_t
    L.ddsp_colsum_scratch_bytes.argtypes = [i32]
    L.ddsp_colsum.restype = i32
    L.ddsp_colsum.argtypes = [vp, vp, vp, ctypes.c_long, i32, i32, vp]
    L.ddsp_stft_frames.restype = i32
    L.ddsp_stft_frames.argtypes = [vp, vp, vp, ctypes.c_long, ctypes.c_long, i32, i32, vp]
    L.ddsp_stft_frames_backward.restype = i32
    L.ddsp_stft_frames_backward.argtypes = [vp, vp, vp, ctypes.c_long, ctypes.c_long, i32, i32, i32, vp]
    L.ddsp_mss_scale_scratch_bytes.restype = ctypes.c_size_t
    L.ddsp_mss_scale_scratch_bytes.argtypes = []
    L.ddsp_mss_scale_supported.restype = i32
    L.ddsp_mss_scale_supported.argtypes = [i32]
    L.ddsp_mss_scale.restype = i32
    L.ddsp_mss_scale.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_long, ctypes.c_long, i32, i32, ctypes.c_float, ctypes.c_float, vp]
    L.ddsp_reverb_impulse.restype = i32
    L.ddsp_reverb_impulse.argtypes = [vp] * 5 + [i32, i32, vp]
    L.ddsp_reverb_impulse_backward.restype = i32
    L.ddsp_reverb_impulse_backward.argtypes = [vp] * 8 + [i32, i32, vp]
    L.ddsp_spectral_mul.restype = i32
    L.ddsp_spectral_mul.argtypes = [vp, vp, vp, ctypes.c_long, ctypes.c_long, vp]
    L.ddsp_spectral_mul_backward.restype = i32
    L.ddsp_spectral_mul_backward.argtypes = [vp] * 5 + [ctypes.c_long, ctypes.c_long, vp]
    L.ddsp_reverb_live_scratch_bytes.restype = ctypes.c_size_t
    L.ddsp_reverb_live_scratch_bytes.argtypes = [i32, i32]
    L.ddsp_reverb_live.restype = i32
    L.ddsp_reverb_live.argtypes = [vp] * 9 + [i32, i32, vp]
    _lib = L
    return L


EXPORTS = ("ddsp_hip_abi_version", "ddsp_test_hooks_enabled", "ddsp_osc_scratch_bytes", "ddsp_osc_forward", "ddsp_osc_forward_ex", "ddsp_osc_set_path", "ddsp_osc_plan", "ddsp_osc_clock", "ddsp_noise_forward", "ddsp_noise_forward_counter", "ddsp_noise_workspace_bytes", "ddsp_noise_forward_ws", "ddsp_noise_backward_ws", "ddsp_profile_select", "ddsp_noise_set_residency", "ddsp_noise_get_residency",
           "ddsp_osc_backward_scratch_bytes", "ddsp_osc_backward", "ddsp_noise_backward", "ddsp_noise_backward_counter",
           "ddsp_osc_set_tiling", "ddsp_noise_set_generic", "ddsp_profile_enable", "ddsp_profile_read",
           "ddsp_gru_scratch_bytes", "ddsp_gru_max_batch", "ddsp_gru_forward", "ddsp_gru_backward", "ddsp_gru_forward_bf16", "ddsp_gru_backward_bf16", "ddsp_gru_status", "ddsp_gru_set_mode", "ddsp_gru_set_fault_step",
           "ddsp_spectral_loss_scratch_bytes", "ddsp_spectral_loss", "ddsp_scaled_sigmoid_forward", "ddsp_scaled_sigmoid_backward", "ddsp_heads_sigmoid_forward", "ddsp_heads_sigmoid_backward",
           "ddsp_ln_lrelu_scratch_bytes", "ddsp_ln_lrelu_forward", "ddsp_ln_lrelu_backward", "ddsp_ln_lrelu_forward_16", "ddsp_ln_lrelu_backward_16", "ddsp_outer_ln_lrelu_scratch_bytes", "ddsp_outer_ln_lrelu_forward", "ddsp_outer_ln_lrelu_backward",
           "ddsp_colsum_scratch_bytes", "ddsp_colsum", "ddsp_stft_frames", "ddsp_stft_frames_backward", "ddsp_mss_scale_scratch_bytes", "ddsp_mss_scale_supported", "ddsp_mss_scale", "ddsp_reverb_impulse", "ddsp_reverb_impulse_backward", "ddsp_spectral_mul", "ddsp_spectral_mul_backward",
           "ddsp_reverb_live_scratch_bytes", "ddsp_reverb_live")

KERNEL_NAMES = {1: "osc_frame_totals", 2: "osc_scan", 3: "osc_frame_synth", 4: "noise_frame", 5: "noise_impulse_responses"}


def profile_enable(capacity: int, only=None) -> None:
    """`only`: kernel names (KERNEL_NAMES values) to record; None = every kernel."""
    mask = 0
    for name in only or ():
        mask |= 1 << [k for k, v in KERNEL_NAMES.items() if v == name][0]
    check(lib().ddsp_profile_select(mask), "ddsp_profile_select")
    check(lib().ddsp_profile_enable(capacity), "ddsp_profile_enable")


def profile_read(cap: int = 65536):
    """-> list of (kernel name, milliseconds) recorded since the last read."""
    ids = (ctypes.c_int * cap)()
    ms = (ctypes.c_float * cap)()
    n = lib().ddsp_profile_read(ids, ms, cap)
    return [(KERNEL_NAMES.get(ids[i], str(ids[i])), float(ms[i])) for i in range(n)]


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc == -1:
        raise DdspHipError(f"{what}: invalid argument (DDSP_EINVAL)")
    if rc == -2:
        raise DdspHipError(f"{what}: shape outside the supported range (DDSP_ERANGE)")
    if rc == -3:
        raise DdspHipError(f"{what}: test / tuning hook refused (DDSP_EPERM): set DDSP_TEST_HOOKS=1 before the library is loaded")
    raise DdspHipError(f"{what}: HIP error {rc}")


def osc_plan(B: int, T: int, H: int, hop: int, sample_rate: int) -> dict:
    """What ddsp_osc_forward launches for this shape on the current device (include/ddsp_hip.h: ddsp_osc_plan)."""
    out = (ctypes.c_int * 8)()
    check(lib().ddsp_osc_plan(B, T, H, hop, sample_rate, out, 8), "ddsp_osc_plan")
    keys = ("harmonics_per_lane", "lanes_per_row", "chunked", "chunk_samples", "chunks_per_row", "row_blocks",
            "compute_units", "workgroups_per_unit")
    return dict(zip(keys, list(out)))


def osc_clock(scratch, B: int, T: int, H: int, hop: int, sample_rate: int, stream: int = 0) -> float:
    """Shader clock (GHz) of the synth kernel of the last ddsp_osc_forward on `scratch` (a torch uint8 tensor); synchronises."""
    ghz = ctypes.c_double(0.0)
    check(lib().ddsp_osc_clock(scratch.data_ptr(), B, T, H, hop, sample_rate, ctypes.byref(ghz), stream or None), "ddsp_osc_clock")
    return float(ghz.value)
