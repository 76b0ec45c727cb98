"""ctypes binding of librsu_hip.so (the C ABI of include/rsu.h).

The library is REQUIRED: there is no CPU or eager-PyTorch fallback anywhere in this package. If the
shared object is missing or a symbol cannot be resolved, importing/using the product path raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSU_LIB_PATH") or os.path.join(_HERE, "librsu_hip.so")  # RSU_LIB_PATH: developer A/B of two builds

_vp, _i, _l, _f, _sz, _u = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_size_t, ctypes.c_uint


class RsuSrc(ctypes.Structure):
    """rsu_src_t: one (cropped) NHWC bf16 input of a virtually concatenated conv."""
    _fields_ = [("ptr", _vp), ("H", _i), ("W", _i), ("C", _i), ("oy", _i), ("ox", _i)]


class RsuWgradJob(ctypes.Structure):
    """rsu_wgrad_job_t: one layer of a grouped weight-gradient launch"""
    _fields_ = [("kind", _i), ("src", RsuSrc), ("dz", _vp), ("dw", _vp), ("db", _vp), ("Ho", _i), ("Wo", _i),
                ("Cin_total", _i), ("ci_off", _i), ("Cout", _i), ("dil", _i)]


WGRAD_CONV3X3, WGRAD_CONVT2X2, WGRAD_GROUP_MAX = 0, 1, 16


class RsuPlanRow(ctypes.Structure):
    """rsu_plan_row_t"""
    _fields_ = [(n, _i) for n in ("kind", "level", "Hin", "Win", "Cin", "Hout", "Wout", "Cout", "dilation", "nsrc")]


class RsuPlanTotals(ctypes.Structure):
    """rsu_plan_totals_t"""
    _fields_ = [("input_size", _i), ("num_params", _l), ("activation_elems", _l), ("workspace_floats", _sz)]


_PS = ctypes.POINTER(RsuSrc)
_PI = ctypes.POINTER(ctypes.c_int)

# name -> (restype, argtypes); mirrors include/rsu.h one to one
SIGNATURES = {
    "rsu_version": (ctypes.c_char_p, []),
    "rsu_last_hip_error": (_i, []),
    "rsu_set_cu_budget": (_i, [_i]),
    "rsu_get_cu_budget": (_i, []),
    "rsu_set_autotune": (_i, [_i]),
    "rsu_get_autotune": (_i, []),
    "rsu_autotune_entries": (_i, []),
    "rsu_autotune_export": (_i, [_PI, _i]),
    "rsu_autotune_import": (_i, [_PI, _i]),
    "rsu_input_size_needed": (_i, [_i, _i, _PI]),
    "rsu_packed_bytes": (_sz, [_i, _i, _PI, _i]),
    "rsu_pack_conv_fwd": (_i, [_vp, _vp, _i, _i, _i, _PI, _i, _vp]),
    "rsu_pack_conv_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rsu_pack_convT_fwd": (_i, [_vp, _vp, _i, _i, _vp]),
    "rsu_pack_convT_bwd": (_i, [_vp, _vp, _i, _i, _vp]),
    "rsu_pack_table_entry_bytes": (_sz, []),
    "rsu_pack_table_add": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _PI, _i]),
    "rsu_pack_table_finish": (_i, [_vp, _i, _PI]),
    "rsu_pack_table_run": (_i, [_vp, _i, _i, _vp]),
    "rsu_color_adjust_fwd": (_i, [_vp, _vp, _vp, _vp, _l, _f, _u, _vp]),
    "rsu_dropout_fwd": (_i, [_vp, _vp, _l, _f, _u, _vp]),
    "rsu_packed_first_bytes": (_sz, [_i]),
    "rsu_pack_conv_first": (_i, [_vp, _vp, _i, _vp]),
    "rsu_conv_first_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "rsu_color_conv_first_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "rsu_conv_first_bwd_ws_floats": (_sz, [_i]),
    "rsu_conv_first_bwd_weight": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "rsu_color_adjust_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _f, _i, _vp]),
    "rsu_head_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _vp]),
    "rsu_head_ws_floats": (_sz, [_l, _i]),
    "rsu_head_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _f, _vp]),
    "rsu_conv2d_fwd": (_i, [_PS, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "rsu_conv2d_fwd_pool": (_i, [_PS, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _u, _i, _vp]),
    "rsu_conv2d_bwd_data": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "rsu_conv_splitk_ws_floats": (_sz, []),
    "rsu_conv2d_fwd_k": (_i, [_PS, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "rsu_conv2d_fwd_pool_k": (_i, [_PS, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _u, _i, _vp, _sz, _vp]),
    "rsu_conv2d_bwd_data_k": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "rsu_conv2d_bwd_weight_ws_floats": (_sz, [_i, _i, _i]),
    "rsu_conv2d_bwd_weight": (_i, [_PS, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "rsu_conv2d_bwd_weight_update": (_i, [_PS, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _f, _f, _f, _i, _vp]),
    "rsu_wgrad_group_table_bytes": (_sz, []),
    "rsu_wgrad_group_ws_floats": (_sz, []),
    "rsu_wgrad_group_plan": (_i, [ctypes.POINTER(RsuWgradJob), _i, _vp, _i, _i, _vp]),
    "rsu_wgrad_group_run": (_i, [_vp, _vp, _vp]),
    "rsu_bias_grad_ws_floats": (_sz, [_l, _i]),
    "rsu_bias_grad": (_i, [_vp, _vp, _vp, _l, _i, _vp]),
    "rsu_maxpool2x2_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _u, _vp]),
    "rsu_maxpool2x2_fwd_code": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _u, _vp]),
    "rsu_pool_skip_relu_bwd_code": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _u, _vp]),
    "rsu_pool_skip_relu_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _u, _vp]),
    "rsu_convT2x2_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "rsu_convT2x2_bwd_data": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _i, _i, _vp]),
    "rsu_convT2x2_bwd_weight_ws_floats": (_sz, [_i, _i]),
    "rsu_convT2x2_bwd_weight": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "rsu_momentum_step": (_i, [_vp, _vp, _vp, _f, _f, _f, _l, _vp]),
    "rsu_update_table_entry_bytes": (_sz, []),
    "rsu_update_table_add_plain": (_i, [_vp, _i, _vp, _vp, _vp, _l]),
    "rsu_update_table_add": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, ctypes.POINTER(_vp), _i, _i, _PI, _i]),
    "rsu_update_table_finish": (_i, [_vp, _i, _PI]),
    "rsu_update_table_run": (_i, [_vp, _i, _i, _f, _f, _f, _vp]),
    "rsu_extract_tiles": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _l, _l, _vp]),
    "rsu_overlap_add": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _l, _l, _vp]),
    "rsu_overlap_finish": (_i, [_vp, _vp, _vp, _l, _vp]),
    "rsu_quantize_mask": (_i, [_vp, _vp, _i, _i, _i, _f, _vp]),
    "rsu_labels_for_patches": (_i, [_vp, _vp, _i, _i, _i, _f, _vp]),
    "rsu_confusion_counts": (_i, [_vp, _vp, _l, _vp, _vp]),
    "rsu_plan": (_i, [_i, _i, _i, _i, _i, ctypes.POINTER(RsuPlanRow), _i, _PI, ctypes.POINTER(RsuPlanTotals)]),
}

_lib = None


class RsuError(RuntimeError):
    pass


def lib():
    """Load librsu_hip.so once; raise loudly if it (or any declared symbol) is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RsuError("HIP extension missing: %s (build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C road_segmentation_unet_amd/csrc`); there is no fallback path" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


TUNE_OFF, TUNE_LOOKUP, TUNE_MEASURE = 0, 1, 2   # rsu.h RSU_TUNE_*
E2BIG = -7


def check(rc, what):
    if rc == E2BIG:
        raise RsuError("%s: a tensor of this call reaches 2 GiB (the kernels address tensors through 32-bit byte offsets): "
                       "use a smaller per-GPU batch" % what)
    if rc != 0:
        raise RsuError("%s failed: rc=%d (hip error %d)" % (what, rc, lib().rsu_last_hip_error()))


_TRACE = os.environ.get("RSU_TRACE", "0") == "1"


def call(name, *args):
    """Invoke an int-returning entry point and raise on a non-zero status. RSU_TRACE=1 prints every call and
    synchronises the device after it (to locate a faulting launch)."""
    if _TRACE:
        import sys
        import torch
        print("[rsu]", name, [a if isinstance(a, (int, float)) else "." for a in args], file=sys.stderr, flush=True)
        check(getattr(lib(), name)(*args), name)
        torch.cuda.synchronize()
        return
    check(getattr(lib(), name)(*args), name)


# ---- stream fork without a system-scope fence -------------------------------------------------------------------------------------
_hip = None
_fork_rings = {}   # device index -> [events, next]: HIP events belong to the device that was current when they were created


def _hip_runtime():
    """the very runtime torch has loaded (its wheel may bundle its own copy: a second instance would not know torch's streams)"""
    global _hip
    if _hip is None:
        path = "libamdhip64.so"
        try:
            with open("/proc/self/maps") as f:
                for line in f:
                    fields = line.split(None, 5)   # address perms offset dev inode pathname (the pathname may end in " (deleted)")
                    if len(fields) == 6 and "libamdhip64.so" in fields[5]:
                        cand = fields[5].strip()
                        if cand.endswith(" (deleted)"):
                            cand = cand[:-len(" (deleted)")]
                        if os.path.exists(cand):
                            path = cand
                        break
        except OSError:
            pass
        h = ctypes.CDLL(path)
        h.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        h.hipEventDestroy.argtypes = [ctypes.c_void_p]
        h.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        h.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        _hip = h
    return _hip


def hip_fork(main_stream, side_stream, device_index=None):
    """`side_stream` waits for everything issued on `main_stream` so far (both hipStream_t handles as integers), through a HIP event
    created with hipEventDisableTiming | hipEventDisableSystemFence. torch.cuda.Event releases at system scope (visible to the host and
    to other devices); a dependency between two streams of ONE device needs agent scope only, and the event's packet holds up the
    recording queue for less (unet.UNet._Side: one fork per weight-gradient launch, +0.9 % on the step). A ring of 64 events PER DEVICE
    (`device_index`: the device the two streams belong to; None = torch's current device) is re-used: an event is re-recorded long
    after its waiter has been enqueued (hipStreamWaitEvent captures the record in flight)."""
    import torch
    h = _hip_runtime()
    dev = torch.cuda.current_device() if device_index is None else int(device_index)
    ring = _fork_rings.get(dev)
    if ring is None:
        events = []
        with torch.cuda.device(dev):
            for _ in range(64):
                ev = ctypes.c_void_p()
                rc = h.hipEventCreateWithFlags(ctypes.byref(ev), 0x2 | 0x20000000)   # hipEventDisableTiming | hipEventDisableSystemFence
                if rc != 0:
                    for e in events:   # nothing half-built stays behind
                        h.hipEventDestroy(e)
                    raise RsuError("hipEventCreateWithFlags failed (%d) on device %d" % (rc, dev))
                events.append(ev)
        ring = _fork_rings[dev] = [events, 0]
    ev = ring[0][ring[1]]
    ring[1] = (ring[1] + 1) % len(ring[0])
    if h.hipEventRecord(ev, ctypes.c_void_p(main_stream)) != 0 or h.hipStreamWaitEvent(ctypes.c_void_p(side_stream), ev, 0) != 0:
        raise RsuError("hipEventRecord / hipStreamWaitEvent failed on device %d" % dev)
