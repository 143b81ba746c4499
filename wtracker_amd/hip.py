"""ctypes binding of libwtk_hip.so — the C ABI declared in include/wtk_hip.h.

There is NO CPU fallback: if the shared library is missing or a GPU is not visible the calls
raise.  (The oracle under oracle/ is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

# WTK_HIP_LIB: load another build of the same library instead (A/B timing of two builds in one GPU session)
_LIB_PATH = os.environ.get("WTK_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libwtk_hip.so")
_lib: Optional[C.CDLL] = None

WTK_F32 = 0
WTK_F16 = 1
WTK_F16X3 = 2
DTYPES = {"fp32": WTK_F32, "f32": WTK_F32, "float32": WTK_F32, "fp16": WTK_F16, "f16": WTK_F16, "float16": WTK_F16, "f16x3": WTK_F16X3, "fp16x3": WTK_F16X3}


class WtkError(RuntimeError):
    pass


class _MlpLayer(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("out_dim", C.c_int32), ("relu", C.c_int32), ("reserved", C.c_int32),
                ("weight", C.POINTER(C.c_float)), ("bias", C.POINTER(C.c_float))]


class _MlpDesc(C.Structure):
    _fields_ = [("device", C.c_int32), ("n_layers", C.c_int32), ("n_blocks", C.c_int32), ("layers_per_block", C.c_int32),
                ("layers", C.POINTER(_MlpLayer))]


class _ConvBlob(C.Structure):
    _fields_ = [("cout", C.c_int32), ("cin", C.c_int32), ("k", C.c_int32), ("stride", C.c_int32), ("act", C.c_int32),
                ("reserved", C.c_int32), ("weight", C.POINTER(C.c_float)), ("bias", C.POINTER(C.c_float))]


class _YoloDesc(C.Structure):
    _fields_ = [("device", C.c_int32), ("dtype", C.c_int32), ("imgsz_h", C.c_int32), ("imgsz_w", C.c_int32),
                ("max_batch", C.c_int32), ("nc", C.c_int32), ("width_mult", C.c_float), ("depth_mult", C.c_float),
                ("max_channels", C.c_int32), ("n_convs", C.c_int32), ("convs", C.POINTER(_ConvBlob))]


# every symbol include/wtk_hip.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "wtk_last_error", "wtk_abi_version", "wtk_device_count",
    "wtk_mlp_create", "wtk_mlp_destroy", "wtk_mlp_forward", "wtk_mlp_forward_host", "wtk_mlp_predict_track",
    "wtk_yolo_conv_count", "wtk_yolo_conv_info", "wtk_yolo_create", "wtk_yolo_create_planned", "wtk_yolo_plan", "wtk_yolo_status", "wtk_yolo_destroy", "wtk_yolo_predict",
    "wtk_yolo_predict_host", "wtk_yolo_debug_head", "wtk_yolo_decode_host", "wtk_yolo_workload",
    "wtk_yolo_set_profiling", "wtk_yolo_get_profile", "wtk_yolo_get_kernel_profile", "wtk_crop_views", "wtk_yolo_debug_tensor",
    "wtk_yolo_predict_views", "wtk_track_median_centers", "wtk_track_polyfit", "wtk_track_training_pairs",
    "wtk_yolo_predict_nms", "wtk_yolo_decode_nms_host",
    "wtk_comm_unique_id", "wtk_comm_create", "wtk_comm_destroy", "wtk_allgather_tracks",
    "wtk_yolo_margin_buffer", "wtk_yolo_last_margins_host",
    "wtk_recheck_select", "wtk_recheck_merge", "wtk_yolo_set_dynamic_batch", "wtk_yolo_set_side_streams",
    "wtk_release_cached_memory", "wtk_recheck_select_counted", "wtk_recheck_enqueue", "wtk_recheck_scatter",
    "wtk_hybrid_create", "wtk_hybrid_destroy", "wtk_hybrid_set_margin", "wtk_hybrid_predict", "wtk_hybrid_predict_views", "wtk_hybrid_flush",
    "wtk_hybrid_pending", "wtk_hybrid_counters", "wtk_hybrid_config", "wtk_hybrid_hold",
]


def lib_path() -> str:
    return _LIB_PATH


def _preload_torch_hip_runtime():
    """One HIP / HSA runtime per process.  PyTorch-ROCm ships its own libamdhip64.so / libhsa-runtime64.so (same sonames as
    /opt/rocm's).  Loaded AFTER `import torch`, libwtk_hip.so binds to torch's copies (soname match) and both share one runtime;
    loaded BEFORE it, /opt/rocm's copies come in first, torch later maps its own by path, and the second runtime to initialise
    finds no GPU ("No HIP GPUs are available").  So when torch is installed but not imported yet, its runtime libraries are
    mapped first (without importing torch)."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return  # fall back to the system runtime: nothing else to do here


def request_hw_queues(n: int = 8) -> bool:
    """Ask the HIP runtime for `n` hardware queues (GPU_MAX_HW_QUEUES; runtime default 4).  The detector keeps up to five streams busy per
    process (two lanes, the shared pair of side streams, the caller's default stream); the runtime multiplexes streams onto its hardware queues
    and streams that share one serialise (-3 %, and cliffs of -40 % for unlucky layouts: profiles/r02_notes.md §8).  The runtime reads the
    variable when it initialises, so call this before the first HIP call of the process (before `import torch` touches the GPU); an explicit
    value of the caller always wins.  OPT-IN on purpose: `load()` does not touch the environment, because children of the calling
    process inherit it — several processes sharing ONE device (e.g. the gloo rehearsal of the multi-rank path) must keep the runtime default
    (eight queues each: 19.5 k -> 4.9 k frames/s).  Returns True when the value was set by this call."""
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return False
    os.environ["GPU_MAX_HW_QUEUES"] = str(int(n))
    return True


_warned_queues = False


def warn_if_streams_exceed_hw_queues(n_busy_streams: int) -> bool:
    """One warning per process when a caller is about to keep more streams busy than the HIP runtime has hardware queues by default (4) and
    GPU_MAX_HW_QUEUES was not set: streams that share a queue serialise (-3 % for the bench's five streams, up to -40 % for unlucky layouts,
    profiles/r02_notes.md section 8).  `request_hw_queues()` before the first HIP call is the fix; it is opt-in because child processes inherit the
    variable.  Returns True when the warning was issued."""
    global _warned_queues
    if _warned_queues or n_busy_streams <= 4 or "GPU_MAX_HW_QUEUES" in os.environ:
        return False
    import warnings

    _warned_queues = True
    warnings.warn(f"wtracker_amd: {n_busy_streams} HIP streams will be busy at once but GPU_MAX_HW_QUEUES is unset (runtime default: 4 hardware queues; streams that "
                  "share one serialise).  Call wtracker_amd.hip.request_hw_queues() before the first HIP call of the process, or export GPU_MAX_HW_QUEUES=8.",
                  RuntimeWarning, stacklevel=3)
    return True


def load() -> C.CDLL:
    """Load libwtk_hip.so (built in-tree by `python -m wtracker_amd._build` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise WtkError(f"{_LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build()); "
                       "there is no CPU fallback")
    _preload_torch_hip_runtime()
    lib = C.CDLL(_LIB_PATH)
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    lib.wtk_last_error.restype = C.c_char_p
    lib.wtk_abi_version.restype = C.c_int
    lib.wtk_device_count.restype = C.c_int
    lib.wtk_mlp_create.argtypes = [C.POINTER(vp), C.POINTER(_MlpDesc)]
    lib.wtk_mlp_destroy.argtypes = [vp]
    lib.wtk_mlp_destroy.restype = None
    lib.wtk_mlp_forward.argtypes = [vp, vp, i32, vp, vp]
    lib.wtk_mlp_forward_host.argtypes = [vp, vp, i32, vp]
    lib.wtk_mlp_predict_track.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp, vp, vp]
    lib.wtk_yolo_conv_count.argtypes = [f32, f32, i32, i32]
    lib.wtk_yolo_conv_info.argtypes = [f32, f32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32),
                                       C.POINTER(i32), C.POINTER(i32), C.c_char_p, C.c_size_t]
    lib.wtk_yolo_create.argtypes = [C.POINTER(vp), C.POINTER(_YoloDesc)]
    lib.wtk_yolo_create_planned.argtypes = [C.POINTER(vp), C.POINTER(_YoloDesc), i32]
    lib.wtk_yolo_plan.argtypes = [vp]
    if hasattr(lib, "wtk_yolo_status") or not os.environ.get("WTK_HIP_LIB"):  # (an older build named by WTK_HIP_LIB for an A/B timing may lack it)
        lib.wtk_yolo_status.argtypes = [vp, C.POINTER(i32), i32]
    lib.wtk_yolo_destroy.argtypes = [vp]
    lib.wtk_yolo_destroy.restype = None
    lib.wtk_yolo_predict.argtypes = [vp, vp, i32, i32, i32, i32, f32, f32, i32, vp, vp, vp, vp]
    lib.wtk_yolo_predict_host.argtypes = [vp, vp, i32, i32, i32, i32, f32, f32, i32, vp, vp, vp]
    lib.wtk_yolo_debug_head.argtypes = [vp, i32, i32, vp, vp]
    lib.wtk_yolo_decode_host.argtypes = [vp, vp, vp, i32, i32, i32, f32, vp, vp, vp]
    lib.wtk_yolo_workload.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i32)]
    lib.wtk_yolo_set_profiling.argtypes = [vp, i32]
    lib.wtk_yolo_debug_tensor.argtypes = [vp, i32, i32, vp, C.c_size_t, vp]
    lib.wtk_yolo_get_profile.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.wtk_yolo_get_kernel_profile.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    lib.wtk_crop_views.argtypes = [vp, i32, i32, i32, i32, vp, i32, i32, vp, vp]
    lib.wtk_track_median_centers.argtypes = [vp, i32, i32, vp, i32, i32, i32, vp, vp, vp]
    lib.wtk_track_polyfit.argtypes = [vp, i32, i32, vp, i32, i32, vp, vp, i32, i32, C.c_double, vp, vp, vp]
    lib.wtk_track_training_pairs.argtypes = [vp, i32, i32, i32, i32, vp, i32, vp, i32, vp, vp, vp, vp]
    lib.wtk_yolo_predict_nms.argtypes = [vp, vp, i32, i32, i32, i32, f32, f32, i32, vp, vp, vp, vp, vp, vp]
    lib.wtk_yolo_decode_nms_host.argtypes = [vp, vp, vp, i32, i32, i32, f32, f32, i32, vp, vp, vp, vp, vp]
    lib.wtk_yolo_margin_buffer.argtypes = [vp, C.POINTER(vp)]
    lib.wtk_yolo_last_margins_host.argtypes = [vp, i32, vp]
    lib.wtk_recheck_select.argtypes = [vp, i32, i32, f32, vp, vp, vp]
    lib.wtk_recheck_select_counted.argtypes = [vp, i32, i32, f32, vp, vp, vp, vp]
    lib.wtk_recheck_enqueue.argtypes = [vp, i32, f32, vp, C.c_int64, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.wtk_recheck_scatter.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.wtk_yolo_set_dynamic_batch.argtypes = [vp, vp]
    lib.wtk_yolo_set_side_streams.argtypes = [vp, i32]
    lib.wtk_recheck_merge.argtypes = [vp, vp, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.wtk_comm_unique_id.argtypes = [vp, C.c_size_t]
    lib.wtk_comm_create.argtypes = [C.POINTER(vp), i32, i32, i32, vp]
    lib.wtk_comm_destroy.argtypes = [vp]
    lib.wtk_comm_destroy.restype = None
    lib.wtk_allgather_tracks.argtypes = [vp, vp, i32, vp, vp]
    lib.wtk_yolo_predict_views.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i32, i32, i32, f32, f32, i32, vp, vp, vp, vp]
    lib.wtk_hybrid_create.argtypes = [C.POINTER(vp), vp, vp, f32, i32, i32]
    lib.wtk_hybrid_destroy.argtypes = [vp]
    lib.wtk_hybrid_destroy.restype = None
    lib.wtk_hybrid_set_margin.argtypes = [vp, f32]
    lib.wtk_hybrid_predict.argtypes = [vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp]
    lib.wtk_hybrid_predict_views.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i32, i32, i32, f32, vp, vp, vp, vp]
    lib.wtk_hybrid_flush.argtypes = [vp, vp]
    lib.wtk_hybrid_pending.argtypes = [vp]
    lib.wtk_hybrid_hold.argtypes = [vp, i32]
    lib.wtk_hybrid_counters.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.wtk_hybrid_config.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(f32)]
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        raise WtkError(f"{what}: {load().wtk_last_error().decode(errors='replace')}")


def device_count() -> int:
    return int(load().wtk_device_count())


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ptr(x) -> C.c_void_p:
    """Device pointer from a torch tensor / int, host pointer from a numpy array; None -> NULL."""
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data)
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    raise TypeError(f"cannot take a pointer of {type(x)}")


def crop_views(frames_dev, N: int, H: int, W: int, Cc: int, pos_xy_dev, view_w: int, view_h: int, views_dev, stream: int = 0):
    """Device-side ViewController.camera_view for a batch: views[n] = window of frames[n] centred on pos_xy[n]
    with replicate borders.  All arguments are device tensors / pointers."""
    _check(load().wtk_crop_views(_ptr(frames_dev), N, H, W, Cc, _ptr(pos_xy_dev), view_w, view_h, _ptr(views_dev),
                                 C.c_void_p(stream)), "wtk_crop_views")


def release_cached_memory():
    """Give the device memory of destroyed detector handles (kept for reuse) back to the driver."""
    _check(load().wtk_release_cached_memory(), "wtk_release_cached_memory")


def recheck_select(margins_dev, B: int, K: int, margin: float, slots_dev, n_weak_dev=None, stream: int = 0, n_overflow_dev=None):
    """slots[k] = batch row of the k-th smallest decision margin (device int32 [K]); n_weak = min(K, rows below `margin`);
    n_overflow += the weak rows beyond the ceiling K (they get no second look); asynchronous on `stream`."""
    _check(load().wtk_recheck_select_counted(_ptr(margins_dev), B, K, margin, _ptr(slots_dev), _ptr(n_weak_dev), _ptr(n_overflow_dev), C.c_void_p(stream)),
           "wtk_recheck_select_counted")


def recheck_enqueue(margins_dev, B: int, margin: float, frames_dev, frame_bytes: int, q_frames_dev, q_cap: int, q_len_dev, q_xywh_ptrs, q_conf_ptrs, q_anchor_ptrs,
                    dst_xywh, dst_conf, dst_anchor, pos_scratch_dev, n_overflow_dev=None, stream: int = 0):
    """Append the batch rows with margin < `margin` (frame copy + output addresses) to the device-side queue of the deferred second look."""
    _check(load().wtk_recheck_enqueue(_ptr(margins_dev), B, margin, _ptr(frames_dev), frame_bytes, _ptr(q_frames_dev), q_cap, _ptr(q_len_dev), _ptr(q_xywh_ptrs),
                                      _ptr(q_conf_ptrs), _ptr(q_anchor_ptrs), _ptr(dst_xywh), _ptr(dst_conf), _ptr(dst_anchor), _ptr(pos_scratch_dev),
                                      _ptr(n_overflow_dev), C.c_void_p(stream)), "wtk_recheck_enqueue")


def recheck_scatter(q_len_dev, q_cap: int, src_xywh, src_conf, src_anchor, q_xywh_ptrs, q_conf_ptrs, q_anchor_ptrs, n_replaced_dev=None, stream: int = 0):
    """Rows 0 .. *q_len - 1 of the full-precision pass go to the addresses queued by recheck_enqueue; the queue is empty afterwards."""
    _check(load().wtk_recheck_scatter(_ptr(q_len_dev), q_cap, _ptr(src_xywh), _ptr(src_conf), _ptr(src_anchor), _ptr(q_xywh_ptrs), _ptr(q_conf_ptrs),
                                      _ptr(q_anchor_ptrs), _ptr(n_replaced_dev), C.c_void_p(stream)), "wtk_recheck_scatter")


def recheck_merge(margins_dev, slots_dev, B: int, K: int, margin: float, src_xywh, src_conf, src_anchor, dst_xywh, dst_conf=None, dst_anchor=None,
                  n_replaced_dev=None, stream: int = 0):
    """Rows slots[k] of dst_* take row k of src_* where the fast pass's margin is below `margin`."""
    _check(load().wtk_recheck_merge(_ptr(margins_dev), _ptr(slots_dev), B, K, margin, _ptr(src_xywh), _ptr(src_conf), _ptr(src_anchor), _ptr(dst_xywh),
                                    _ptr(dst_conf), _ptr(dst_anchor), _ptr(n_replaced_dev), C.c_void_p(stream)), "wtk_recheck_merge")


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """Rank 0: the RCCL rendezvous token (128 bytes) to hand to every other rank out of band."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(load().wtk_comm_unique_id(buf, COMM_ID_BYTES), "wtk_comm_unique_id")
    return bytes(buf)


class WtkComm:
    """The C ABI's own RCCL communicator (wtk_comm_*): one per rank, for callers that do not use torch.distributed."""

    def __init__(self, device: int, rank: int, world: int, unique_id: bytes):
        if len(unique_id) != COMM_ID_BYTES:
            raise WtkError("unique_id must be the 128 bytes of comm_unique_id()")
        self._h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(load().wtk_comm_create(C.byref(self._h), device, rank, world, buf), "wtk_comm_create")
        self.rank, self.world, self.device = rank, world, device

    def allgather_tracks(self, local_dev, n_local: int, all_dev, stream: int = 0):
        """all_dev[r * n_local : (r + 1) * n_local] = rank r's local_dev [n_local, 4] float32 (device tensors)."""
        _check(load().wtk_allgather_tracks(self._h, _ptr(local_dev), n_local, _ptr(all_dev), C.c_void_p(stream)), "wtk_allgather_tracks")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            load().wtk_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _track_is_f64(track_dev) -> int:
    dt = str(getattr(track_dev, "dtype", ""))
    if dt.endswith("float64"):
        return 1
    if dt.endswith("float32"):
        return 0
    raise WtkError("track must be a float32 or float64 device tensor [n_frames, 4]")


def track_median_centers(track_dev, n_frames: int, cycles_dev, n_samples: int, cycle_frame_num: int, imaging_frame_num: int, pred_dev, valid_dev,
                         stream: int = 0):
    """OptimalController's target for every cycle in `cycles_dev` (wtk_track_median_centers); pred_dev float64 [n,2]."""
    _check(load().wtk_track_median_centers(_ptr(track_dev), _track_is_f64(track_dev), n_frames, _ptr(cycles_dev), n_samples, cycle_frame_num,
                                           imaging_frame_num, _ptr(pred_dev), _ptr(valid_dev), C.c_void_p(stream)), "wtk_track_median_centers")


def track_polyfit(track_dev, n_frames: int, cycles_dev, n_samples: int, cycle_frame_num: int, sample_times: Sequence[int], weights: Sequence[float],
                  degree: int, t_eval: float, pred_dev, valid_dev, stream: int = 0):
    """PolyfitController's extrapolated head position for every cycle in `cycles_dev` (wtk_track_polyfit); pred_dev float64 [n,2]."""
    st = np.ascontiguousarray(sample_times, dtype=np.int32)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    if len(st) != len(w):
        raise WtkError("sample_times and weights differ in length")
    _check(load().wtk_track_polyfit(_ptr(track_dev), _track_is_f64(track_dev), n_frames, _ptr(cycles_dev), n_samples, cycle_frame_num, _ptr(st), _ptr(w),
                                    len(st), degree, float(t_eval), _ptr(pred_dev), _ptr(valid_dev), C.c_void_p(stream)), "wtk_track_polyfit")


def track_training_pairs(track_dev, n_frames: int, row0: int, n_rows: int, input_frames: Sequence[int], pred_frames: Sequence[int], x_dev, y_dev,
                         keep_dev, stream: int = 0):
    """NumpyDataset.create_from_config's rows row0 .. row0+n_rows-1 (wtk_track_training_pairs); the caller drops keep == 0."""
    xi = np.ascontiguousarray(input_frames, dtype=np.int32)
    yi = np.ascontiguousarray(pred_frames, dtype=np.int32)
    _check(load().wtk_track_training_pairs(_ptr(track_dev), _track_is_f64(track_dev), n_frames, row0, n_rows, _ptr(xi), len(xi), _ptr(yi), len(yi),
                                           _ptr(x_dev), _ptr(y_dev), _ptr(keep_dev), C.c_void_p(stream)), "wtk_track_training_pairs")


# -------------------------------------------------------------------------------------------------
class HipMLP:
    """Device ResMLP (wtk_mlp).  `layers` = list of (W[out,in] fp32, b[out] fp32, relu: bool), BN folded,
    in the order input, block0.l0.., ..., output."""

    def __init__(self, layers: Sequence[tuple], n_blocks: int, layers_per_block: int, device: int = 0):
        lib = load()
        self._keep = []
        arr = (_MlpLayer * len(layers))()
        for i, (w, b, relu) in enumerate(layers):
            w = np.ascontiguousarray(w, dtype=np.float32)
            b = np.ascontiguousarray(b, dtype=np.float32)
            assert w.ndim == 2 and b.shape == (w.shape[0],)
            self._keep += [w, b]
            arr[i] = _MlpLayer(w.shape[1], w.shape[0], int(bool(relu)), 0, _fptr(w), _fptr(b))
        desc = _MlpDesc(device, len(layers), n_blocks, layers_per_block, arr)
        self._h = C.c_void_p()
        _check(lib.wtk_mlp_create(C.byref(self._h), C.byref(desc)), "wtk_mlp_create")
        self.in_dim = int(layers[0][0].shape[1])
        self.out_dim = int(layers[-1][0].shape[0])
        self.device = device

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            load().wtk_mlp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:  # at interpreter shutdown the module globals (load, _lib) may already be gone
            self.close()
        except Exception:
            pass

    def forward_host(self, x: np.ndarray) -> np.ndarray:
        """x: [B, in_dim] float32 (host) -> [B, out_dim] float32 (host)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[1] == self.in_dim
        y = np.empty((x.shape[0], self.out_dim), dtype=np.float32)
        if x.shape[0] == 0:
            return y
        _check(load().wtk_mlp_forward_host(self._h, _ptr(x), x.shape[0], _ptr(y)), "wtk_mlp_forward_host")
        return y

    def forward(self, x_dev, y_dev, batch: int, stream: int = 0):
        """Device tensors (torch) / raw device pointers; asynchronous on `stream` (hipStream_t as int)."""
        _check(load().wtk_mlp_forward(self._h, _ptr(x_dev), batch, _ptr(y_dev), C.c_void_p(stream)), "wtk_mlp_forward")

    def predict_track(self, track_dev, n_frames: int, anchor_frames_dev, n_samples: int, input_frames: Sequence[int],
                      pred_dev, valid_dev=None, stream: int = 0):
        inf = np.ascontiguousarray(input_frames, dtype=np.int32)
        _check(load().wtk_mlp_predict_track(self._h, _ptr(track_dev), n_frames, _ptr(anchor_frames_dev), n_samples,
                                            _ptr(inf), len(inf), _ptr(pred_dev), _ptr(valid_dev), C.c_void_p(stream)),
               "wtk_mlp_predict_track")


# -------------------------------------------------------------------------------------------------
def yolo_conv_table(width: float, depth: float, max_channels: int, nc: int) -> list[dict]:
    """The library's own conv list (name, cout, cin, k, stride, act) — needs no GPU."""
    lib = load()
    n = lib.wtk_yolo_conv_count(width, depth, max_channels, nc)
    if n <= 0:
        raise WtkError("wtk_yolo_conv_count: bad model scale")
    out = []
    for i in range(n):
        co, ci, k, s, a = (C.c_int32() for _ in range(5))
        name = C.create_string_buffer(96)
        _check(lib.wtk_yolo_conv_info(width, depth, max_channels, nc, i, C.byref(co), C.byref(ci), C.byref(k), C.byref(s),
                                      C.byref(a), name, 96), "wtk_yolo_conv_info")
        out.append(dict(name=name.value.decode(), cout=co.value, cin=ci.value, k=k.value, stride=s.value, act=a.value))
    return out


STATUS_NONFINITE = 1
PLANS = {"auto": 0, "throughput": 1, "latency": 2}


class HipYolo:
    """Device YOLOv8 detector (wtk_yolo) for one network input size.

    weights: dict name -> (W[cout,k,k,cin] fp32 O-H-W-I, b[cout] fp32) for every conv of
    wtracker_amd.yolo_spec.conv_table(scale, nc)."""

    def __init__(self, weights: dict, imgsz: tuple[int, int], max_batch: int, dtype: str = "fp16", nc: int = 1,
                 width: float = 0.5, depth: float = 0.33, max_channels: int = 1024, device: int = 0, plan: str = "auto"):
        """plan: "auto" (latency when max_batch <= 4 and the dtype is fp32 / f16x3, else throughput; WTK_LATENCY_PLAN=0/1 overrides), "throughput" or
        "latency" (include/wtk_hip.h: wtk_yolo_create_planned) — fixed for the handle's life, so a frame's result does not depend on the batch it is in."""
        lib = load()
        if plan not in PLANS:
            raise WtkError(f"plan must be one of {sorted(PLANS)}")
        table = yolo_conv_table(width, depth, max_channels, nc)
        arr = (_ConvBlob * len(table))()
        self._keep = []
        for i, t in enumerate(table):
            if t["name"] not in weights:
                raise WtkError(f"missing weights for conv {t['name']}")
            w, b = weights[t["name"]]
            w = np.ascontiguousarray(w, dtype=np.float32)
            b = np.ascontiguousarray(b, dtype=np.float32)
            if w.shape != (t["cout"], t["k"], t["k"], t["cin"]) or b.shape != (t["cout"],):
                raise WtkError(f"conv {t['name']}: expected weight {(t['cout'], t['k'], t['k'], t['cin'])}, got {w.shape}")
            self._keep += [w, b]
            arr[i] = _ConvBlob(t["cout"], t["cin"], t["k"], t["stride"], t["act"], 0, _fptr(w), _fptr(b))
        desc = _YoloDesc(device, DTYPES[dtype], int(imgsz[0]), int(imgsz[1]), int(max_batch), nc, width, depth, max_channels,
                         len(table), arr)
        self._h = C.c_void_p()
        _check(lib.wtk_yolo_create_planned(C.byref(self._h), C.byref(desc), PLANS[plan]), "wtk_yolo_create_planned")
        self.plan = "latency" if lib.wtk_yolo_plan(self._h) == PLANS["latency"] else "throughput"
        self._keep = []  # the library copied everything to the device
        self.imgsz = (int(imgsz[0]), int(imgsz[1]))
        self.max_batch = int(max_batch)
        self.dtype = dtype
        self.nc = nc
        self.device = device
        macs, anchors = C.c_double(), C.c_int32()
        _check(lib.wtk_yolo_workload(self._h, C.byref(macs), C.byref(anchors)), "wtk_yolo_workload")
        self.macs_per_frame = macs.value
        self.anchors = anchors.value

    def close(self):
        for d in self.__dict__.pop("_dependents", []):  # native objects that borrow this handle (wtk_hybrid) go first
            d._release()
        if getattr(self, "_h", None) is not None and self._h:
            load().wtk_yolo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:  # at interpreter shutdown the module globals (load, _lib) may already be gone
            self.close()
        except Exception:
            pass

    def predict_host(self, frames: np.ndarray, conf: float = 0.1, iou: float = 0.7, max_det: int = 1):
        """frames: uint8 [B,H,W] or [B,H,W,C] (host).  Returns (xywh [B,4] f32 with NaN rows, conf [B], anchor [B])."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        if frames.ndim == 3:
            frames = frames[..., None]
        B, H, W, Cc = frames.shape
        xywh = np.empty((B, 4), dtype=np.float32)
        cf = np.empty((B,), dtype=np.float32)
        an = np.empty((B,), dtype=np.int32)
        _check(load().wtk_yolo_predict_host(self._h, _ptr(frames), B, H, W, Cc, conf, iou, max_det, _ptr(xywh), _ptr(cf),
                                            _ptr(an)), "wtk_yolo_predict_host")
        return xywh, cf, an

    def predict(self, frames_dev, B: int, H: int, W: int, Cc: int, out_xywh, out_conf=None, out_anchor=None,
                conf: float = 0.1, iou: float = 0.7, max_det: int = 1, stream: int = 0):
        """Device pointers / torch CUDA tensors; asynchronous on `stream`."""
        _check(load().wtk_yolo_predict(self._h, _ptr(frames_dev), B, H, W, Cc, conf, iou, max_det, _ptr(out_xywh),
                                       _ptr(out_conf), _ptr(out_anchor), C.c_void_p(stream)), "wtk_yolo_predict")

    def status(self, clear: bool = False) -> int:
        """Range-guard flags (include/wtk_hip.h: WTK_STATUS_NONFINITE = 1, sticky, raised by the head kernels when a head logit is inf / NaN — an
        activation left the fp16 range somewhere in the network).  No device call: valid for the work the
        caller has synchronised with."""
        f = C.c_int32()
        _check(load().wtk_yolo_status(self._h, C.byref(f), int(clear)), "wtk_yolo_status")
        return f.value

    def set_dynamic_batch(self, n_dev):
        """`n_dev`: device int32 scalar (tensor / pointer) holding the number of batch rows that matter in the following calls, or None."""
        _check(load().wtk_yolo_set_dynamic_batch(self._h, _ptr(n_dev)), "wtk_yolo_set_dynamic_batch")

    def set_side_streams(self, n: int):
        """Streams the forward pass spreads over besides the caller's: 2 (default: P3 / P4 Detect towers on a side stream each), 1 or 0."""
        _check(load().wtk_yolo_set_side_streams(self._h, n), "wtk_yolo_set_side_streams")

    def last_margins(self, B: int) -> np.ndarray:
        """Decision margins (class-logit units) of the B frames of the last max_det = 1 call: min(best - second-best anchor logit,
        |best - logit(conf)|) — small values mark frames on which a perturbation of the logits could change the result."""
        out = np.empty((B,), dtype=np.float32)
        _check(load().wtk_yolo_last_margins_host(self._h, B, _ptr(out)), "wtk_yolo_last_margins_host")
        return out

    def margin_buffer(self) -> int:
        """Device address of the handle's [max_batch] float margin buffer (for device-resident pipelines)."""
        p = C.c_void_p()
        _check(load().wtk_yolo_margin_buffer(self._h, C.byref(p)), "wtk_yolo_margin_buffer")
        return int(p.value)

    def predict_nms(self, frames_dev, B: int, H: int, W: int, Cc: int, max_det: int, out_xywh, out_conf=None, out_cls=None, out_anchor=None,
                    out_count=None, conf: float = 0.1, iou: float = 0.7, stream: int = 0):
        """General greedy NMS, up to `max_det` boxes per frame (wtk_yolo_predict_nms); outputs [B,max_det,...] device tensors."""
        _check(load().wtk_yolo_predict_nms(self._h, _ptr(frames_dev), B, H, W, Cc, conf, iou, max_det, _ptr(out_xywh), _ptr(out_conf),
                                           _ptr(out_cls), _ptr(out_anchor), _ptr(out_count), C.c_void_p(stream)), "wtk_yolo_predict_nms")

    def decode_nms_host(self, box: np.ndarray, cls: np.ndarray, H: int, W: int, max_det: int, conf: float = 0.1, iou: float = 0.7):
        """NMS on given head logits -> (xywh [B,max_det,4], conf [B,max_det], cls [B,max_det], anchor [B,max_det], count [B])."""
        box = np.ascontiguousarray(box, dtype=np.float32)
        cls = np.ascontiguousarray(cls, dtype=np.float32)
        B = box.shape[0]
        assert box.shape == (B, self.anchors, 64) and cls.shape == (B, self.anchors, self.nc)
        xywh = np.empty((B, max_det, 4), dtype=np.float32)
        cf = np.empty((B, max_det), dtype=np.float32)
        kc = np.empty((B, max_det), dtype=np.int32)
        an = np.empty((B, max_det), dtype=np.int32)
        cnt = np.empty((B,), dtype=np.int32)
        _check(load().wtk_yolo_decode_nms_host(self._h, _ptr(box), _ptr(cls), B, H, W, conf, iou, max_det, _ptr(xywh), _ptr(cf), _ptr(kc), _ptr(an),
                                               _ptr(cnt)), "wtk_yolo_decode_nms_host")
        return xywh, cf, kc, an, cnt

    def predict_views(self, frames_dev, n_frames: int, H: int, W: int, Cc: int, frame_index_dev, pos_xy_dev, B: int, view_w: int,
                      view_h: int, out_xywh, out_conf=None, out_anchor=None, conf: float = 0.1, iou: float = 0.7, max_det: int = 1,
                      stream: int = 0):
        """Detector on the (view_w x view_h) camera views of device-resident full frames (crop + letterbox fused on the
        device; wtk_yolo_predict_views).  frame_index_dev [B] int32 or None, pos_xy_dev [B,2] int32; boxes in view pixels."""
        _check(load().wtk_yolo_predict_views(self._h, _ptr(frames_dev), n_frames, H, W, Cc, _ptr(frame_index_dev), _ptr(pos_xy_dev), B,
                                             view_w, view_h, conf, iou, max_det, _ptr(out_xywh), _ptr(out_conf), _ptr(out_anchor),
                                             C.c_void_p(stream)), "wtk_yolo_predict_views")

    def debug_head(self, B: int):
        """Raw head logits of the last forward: (box [B,A,64], cls [B,A,nc]) fp32, levels concatenated."""
        lib = load()
        boxes, clss = [], []
        S_h, S_w = self.imgsz
        for lvl, s in enumerate((8, 16, 32)):
            A = (S_h // s) * (S_w // s)
            bx = np.empty((B, A, 64), dtype=np.float32)
            cl = np.empty((B, A, self.nc), dtype=np.float32)
            _check(lib.wtk_yolo_debug_head(self._h, lvl, B, _ptr(bx), _ptr(cl)), "wtk_yolo_debug_head")
            boxes.append(bx)
            clss.append(cl)
        return np.concatenate(boxes, 1), np.concatenate(clss, 1)

    def debug_tensor(self, conv_index: int, B: int) -> np.ndarray:
        """Output of conv blob `conv_index` (yolo_conv_table order) of the last forward: fp32 [B,h,w,cout]."""
        lib = load()
        shp = (C.c_int32 * 3)()
        _check(lib.wtk_yolo_debug_tensor(self._h, conv_index, B, None, 0, shp), "wtk_yolo_debug_tensor")
        out = np.empty((B, shp[0], shp[1], shp[2]), dtype=np.float32)
        _check(lib.wtk_yolo_debug_tensor(self._h, conv_index, B, _ptr(out), out.size, shp), "wtk_yolo_debug_tensor")
        return out

    def decode_host(self, box: np.ndarray, cls: np.ndarray, H: int, W: int, conf: float = 0.1):
        box = np.ascontiguousarray(box, dtype=np.float32)
        cls = np.ascontiguousarray(cls, dtype=np.float32)
        B = box.shape[0]
        assert box.shape == (B, self.anchors, 64) and cls.shape == (B, self.anchors, self.nc)
        xywh = np.empty((B, 4), dtype=np.float32)
        cf = np.empty((B,), dtype=np.float32)
        an = np.empty((B,), dtype=np.int32)
        _check(load().wtk_yolo_decode_host(self._h, _ptr(box), _ptr(cls), B, H, W, conf, _ptr(xywh), _ptr(cf), _ptr(an)),
               "wtk_yolo_decode_host")
        return xywh, cf, an

    def set_profiling(self, enabled: bool):
        _check(load().wtk_yolo_set_profiling(self._h, int(enabled)), "wtk_yolo_set_profiling")

    def get_profile(self) -> dict:
        names = ["stem", "conv", "pool", "head"]
        out = {}
        for i, nme in enumerate(names):
            ms, n = C.c_double(), C.c_int64()
            _check(load().wtk_yolo_get_profile(self._h, i, C.byref(ms), C.byref(n)), "wtk_yolo_get_profile")
            out[nme] = dict(total_ms=ms.value, launches=n.value)
        return out

    KERNELS = ["stem_mfma_kernel", "conv_igemm_kernel+conv1x1_wide_kernel", "sppf_pool_kernel", "head", "conv3x3_halo_kernel",
               "front_fused_kernel+c2f32_fused_kernel", "conv3x3_c32_kernel"]

    def get_kernel_profile(self) -> dict:
        """Per-kernel device time, launches and algorithmic FLOPs of the profiled forwards (wtk_yolo_get_kernel_profile)."""
        out = {}
        for i, nme in enumerate(self.KERNELS):
            ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
            _check(load().wtk_yolo_get_kernel_profile(self._h, i, C.byref(ms), C.byref(n), C.byref(fl)), "wtk_yolo_get_kernel_profile")
            out[nme] = dict(total_ms=ms.value, launches=n.value, flops=fl.value)
        return out
