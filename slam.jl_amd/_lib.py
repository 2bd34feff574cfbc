"""ctypes binding of libslamhip.so (the C ABI declared in include/slamhip.h).

The product path has NO CPU fallback: if the shared library is missing or no
HIP device is usable, loading / context creation raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLAMHIP_LIB") or os.path.join(_HERE, "libslamhip.so")     # SLAMHIP_LIB: another build of the same library (A/B timing)

f64p = C.POINTER(C.c_double)
u8p = C.POINTER(C.c_uint8)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p
dbl, cint = C.c_double, C.c_int

# name -> (restype, argtypes); kept in sync with include/slamhip.h (tests/test_abi.py checks it)
SIGNATURES = {
    "slam_ctx_create": (cint, [cint, C.POINTER(vp)]),
    "slam_ctx_create_cumask": (cint, [cint, C.POINTER(C.c_uint32), cint, C.POINTER(vp)]),
    "slam_ctx_create_priority": (cint, [cint, cint, C.POINTER(vp)]),
    "slam_ctx_destroy": (cint, [vp]),
    "slam_ctx_synchronize": (cint, [vp]),
    "slam_ctx_stream": (vp, [vp]),
    "slam_ctx_wait_for": (cint, [vp, vp]),
    "slam_event_create": (cint, [vp, C.POINTER(vp)]),
    "slam_event_record": (cint, [vp, vp]),
    "slam_ctx_wait_event": (cint, [vp, vp]),
    "slam_event_destroy": (cint, [vp]),
    "slam_event_create_timed": (cint, [vp, C.POINTER(vp)]),
    "slam_event_elapsed_ms": (cint, [vp, vp, C.POINTER(dbl)]),
    "slam_last_error": (C.c_char_p, [vp]),
    "slam_version": (C.c_char_p, []),
    "slam_prof_enable": (cint, [vp, cint]),
    "slam_prof_reset": (cint, [vp]),
    "slam_prof_get": (cint, [vp, C.c_char_p, f64p, i64p]),
    "slam_detect": (cint, [vp, f64p, cint, cint, f64p, cint, cint, cint, cint, cint, cint, dbl, dbl, i64p, cint, C.POINTER(cint)]),
    "slam_detect_pyr": (cint, [vp, vp, f64p, cint, cint, cint, cint, cint, cint, dbl, dbl, i64p, cint, C.POINTER(cint)]),
    "slam_detect_batch": (cint, [vp, vp, cint, f64p, i32p, cint, cint, cint, cint, cint, dbl, dbl, i64p, cint, i32p]),
    "slam_triangulate": (cint, [vp, f64p, f64p, f64p, f64p, f64p, f64p, f64p, cint, dbl, dbl, f64p, dbl, f64p, u8p]),
    "slam_p3p_ransac": (cint, [vp, f64p, f64p, f64p, cint, f64p, dbl, i32p, cint, f64p, f64p, u8p, C.POINTER(cint), f64p, C.POINTER(cint)]),
    "slam_five_point_ransac": (cint, [vp, f64p, f64p, f64p, f64p, cint, f64p, f64p, dbl, i32p, cint, f64p, f64p, u8p, C.POINTER(cint), f64p, C.POINTER(cint)]),
    "slam_p3p_ransac_batch": (cint, [vp, cint, i32p, f64p, f64p, f64p, f64p, dbl, i32p, cint, f64p, f64p, u8p, i32p, f64p, i32p]),
    "slam_five_point_ransac_batch": (cint, [vp, cint, i32p, f64p, f64p, f64p, f64p, f64p, f64p, dbl, i32p, cint, f64p, f64p, u8p, i32p, f64p, i32p]),
    "slam_pnp_ba_batch": (cint, [vp, cint, i32p, f64p, f64p, f64p, f64p, cint, cint, dbl, dbl, f64p, f64p, f64p, u8p, i32p]),
    "slam_describe": (cint, [vp, f64p, cint, cint, i64p, cint, i32p, cint, dbl, cint, u64p, i64p, C.POINTER(cint)]),
    "slam_pyr_create": (cint, [vp, cint, cint, cint, C.POINTER(vp)]),
    "slam_pyr_destroy": (cint, [vp]),
    "slam_pyr_update": (cint, [vp, vp, f64p, cint, dbl]),
    "slam_pyr_update_u8": (cint, [vp, vp, u8p, cint, dbl]),
    "slam_pyr_update_dev": (cint, [vp, vp, vp, cint, dbl, cint]),
    "slam_pyr_update_u8_dev": (cint, [vp, vp, vp, cint, dbl, cint]),
    "slam_frontend_create": (cint, [cint, vp, C.POINTER(vp)]),
    "slam_frontend_destroy": (cint, [vp]),
    "slam_frontend_step": (cint, [vp, u8p, u8p, f64p, cint, f64p, cint, f64p, vp, i32p, i32p]),
    "slam_frontend_flush": (cint, [vp, f64p, cint, f64p, cint, f64p, vp, i32p, i32p]),
    "slam_frontend_keypoints": (vp, [vp]),
    "slam_frontend_ctx": (vp, [vp]),
    "slam_frontend_last_error": (C.c_char_p, [vp]),
    "slam_pyr_copy": (cint, [vp, vp, vp]),
    "slam_pyr_clone": (cint, [vp, vp, C.POINTER(vp)]),
    "slam_pyr_shape": (cint, [vp, cint, C.POINTER(cint), C.POINTER(cint)]),
    "slam_pyr_levels": (cint, [vp]),
    "slam_pyr_download": (cint, [vp, vp, cint, cint, f64p]),
    "slam_fb_track": (cint, [vp, vp, vp, f64p, f64p, cint, cint, cint, cint, dbl, dbl, dbl, f64p, u8p]),
    "slam_flow_match": (cint, [vp, vp, vp, f64p, u8p, f64p, cint, cint, cint, cint, cint, dbl, dbl, dbl, f64p, u8p]),
    "slam_pyr_create_batch": (cint, [vp, cint, cint, cint, cint, C.POINTER(vp)]),
    "slam_pyr_update_batch_dev": (cint, [vp, C.POINTER(vp), C.POINTER(vp), cint, cint, dbl, cint]),
    "slam_pyr_update_batch_u8_dev": (cint, [vp, C.POINTER(vp), C.POINTER(vp), cint, cint, dbl, cint]),
    "slam_flow_match_batch": (cint, [vp, vp, vp, cint, i32p, f64p, u8p, f64p, cint, cint, cint, cint, cint, dbl, dbl, dbl, f64p, u8p]),
    "slam_flow_match_batch_kept": (cint, [vp, vp, vp, cint, i32p, f64p, u8p, f64p, cint, cint, cint, cint, cint, dbl, dbl, dbl, f64p, u8p, i32p, i32p, C.POINTER(cint), u8p]),
    "slam_kpset_create": (cint, [vp, cint, cint, C.POINTER(vp)]),
    "slam_kpset_destroy": (cint, [vp]),
    "slam_kpset_streams": (cint, [vp]),
    "slam_kpset_capacity": (cint, [vp]),
    "slam_kpset_upload": (cint, [vp, vp, cint, f64p, u8p, f64p, i64p, cint]),
    "slam_kpset_download": (cint, [vp, vp, cint, f64p, u8p, f64p, i64p, f64p, u8p, cint, C.POINTER(cint)]),
    "slam_kpset_counts": (cint, [vp, vp, i32p]),
    "slam_kpset_flow_match": (cint, [vp, vp, vp, vp, f64p, cint, cint, cint, cint, cint, dbl, dbl, dbl, cint]),
    "slam_kpset_stereo_match": (cint, [vp, vp, vp, vp, f64p, cint, cint, cint, cint, cint, dbl, dbl, dbl, dbl, cint]),
    "slam_kpset_remove": (cint, [vp, vp, vp]),
    "slam_kpset_detect": (cint, [vp, vp, vp, cint, cint, cint, cint, cint, dbl, dbl]),
    "slam_kpset_triangulate": (cint, [vp, vp, f64p, f64p, f64p, f64p, f64p, f64p, dbl, dbl, cint]),
    "slam_kpset_keyframe": (cint, [vp, vp]),
    "slam_kpset_triangulate_temporal": (cint, [vp, vp, f64p, f64p, cint, i32p, i32p, dbl, dbl, dbl, cint]),
    "slam_kpset_upload_first": (cint, [vp, vp, cint, f64p, i32p, cint, cint]),
    "slam_kpset_download_first": (cint, [vp, vp, cint, f64p, i32p, cint, C.POINTER(cint), C.POINTER(cint)]),
    "slam_kpset_upload_keyframe": (cint, [vp, vp, cint, f64p, u8p, cint]),
    "slam_kpset_download_keyframe": (cint, [vp, vp, cint, f64p, u8p, cint, C.POINTER(cint)]),
    "slam_kpset_compute_pose_5pt": (cint, [vp, vp, f64p, dbl, dbl, cint, C.c_uint64, f64p, i32p, i32p, f64p, i32p]),
    "slam_kpset_compute_pose": (cint, [vp, vp, f64p, dbl, cint, C.c_uint64, cint, cint, dbl, dbl, f64p, i32p, i32p, i32p]),
    "slam_local_ba": (cint, [vp, dbl, dbl, dbl, dbl, cint, cint, cint, f64p, u8p, f64p, i64p, i64p, u8p, cint, cint, dbl, f64p]),
    "slam_local_ba_batch": (cint, [vp, cint, f64p, i32p, i32p, i32p, f64p, u8p, f64p, i64p, i64p, u8p, cint, cint, dbl, f64p, i32p]),
    "slam_local_ba_batch_begin": (cint, [vp, cint, f64p, i32p, i32p, i32p, f64p, u8p, f64p, i64p, i64p, u8p, cint, cint, dbl, f64p, i32p]),
    "slam_local_ba_batch_end": (cint, [vp]),
    "slam_pnp_ba": (cint, [vp, dbl, dbl, dbl, dbl, f64p, f64p, f64p, cint, cint, cint, dbl, dbl, f64p, f64p, f64p, u8p, C.POINTER(cint)]),
    "slam_ba_create": (cint, [vp, dbl, dbl, dbl, dbl, cint, cint, cint, f64p, u8p, f64p, i64p, i64p, C.POINTER(vp)]),
    "slam_ba_destroy": (cint, [vp]),
    "slam_ba_reduce_len": (C.c_int64, [cint]),
    "slam_ba_build": (cint, [vp, vp, cint, dbl, vp]),
    "slam_ba_solve": (cint, [vp, vp, vp, dbl, vp]),
    "slam_ba_commit": (cint, [vp, vp, cint]),
    "slam_ba_flag_outliers": (cint, [vp, vp, dbl, dbl, C.POINTER(cint)]),
    "slam_ba_download": (cint, [vp, vp, f64p, u8p]),
    "slam_ba_plan_order": (cint, [cint, cint, cint, u8p, i64p, i64p, i32p, C.POINTER(cint)]),
    "slam_ba_halfband": (cint, [vp]),
    "slam_ba_set_halfband": (cint, [vp, cint]),
    "slam_ba_lm_begin": (cint, [vp, vp, cint, vp]),
    "slam_ba_lm_start": (cint, [vp, vp, vp, cint]),
    "slam_ba_lm_build": (cint, [vp, vp, cint, vp]),
    "slam_ba_lm_solve": (cint, [vp, vp, vp, cint, vp]),
    "slam_ba_lm_step": (cint, [vp, vp, vp, cint, cint]),
    "slam_ba_lm_state": (cint, [vp, vp, f64p]),
    "slam_comm_unique_id": (cint, [vp]),
    "slam_comm_create": (cint, [vp, cint, cint, vp, C.POINTER(vp)]),
    "slam_comm_destroy": (cint, [vp]),
    "slam_comm_size": (cint, [vp]),
    "slam_comm_rank": (cint, [vp]),
    "slam_comm_allreduce_sum": (cint, [vp, vp, vp, C.c_int64]),
    "slam_comm_allgather": (cint, [vp, vp, vp, vp, C.c_int64]),
}

_lib = None


class SlamHipError(RuntimeError):
    pass


def load():
    """Load libslamhip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SlamHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C slam.jl_amd/csrc`. There is no CPU fallback.")
        # PyTorch-ROCm bundles its own libamdhip64; two HIP runtimes in one process
        # cannot both own the device.  Load torch's first so that libslamhip.so binds
        # to the same runtime (needed anyway to exchange device pointers with torch
        # tensors in bench.py / sharded_ba.py).  Harmless when torch is absent.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def ptr(a, t=f64p):
    return None if a is None else a.ctypes.data_as(t)


class Context:
    """One per calling task (SURVEY 8b threading): owns a HIP stream + scratch."""

    def __init__(self, device=0, cu_mask=None, priority=0):
        """cu_mask: optional iterable of 0/1 per compute unit (slam_ctx_create_cumask); priority: > 0 puts the stream in the
        high-priority scheduling class with a hardware queue of its own (slam_ctx_create_priority)"""
        self.lib = load()
        h = vp()
        if cu_mask is not None:
            bits = list(cu_mask)
            nw = (len(bits) + 31) // 32
            words = (C.c_uint32 * nw)()
            for i, b in enumerate(bits):
                if b:
                    words[i // 32] |= 1 << (i % 32)
            rc = self.lib.slam_ctx_create_cumask(device, words, nw, C.byref(h))
        elif priority:
            rc = self.lib.slam_ctx_create_priority(device, int(priority), C.byref(h))
        else:
            rc = self.lib.slam_ctx_create(device, C.byref(h))
        if rc != 0:
            raise SlamHipError(f"slam_ctx_create failed ({rc}): {self.lib.slam_last_error(None).decode()}")
        self.h = h
        self.device = device

    def check(self, rc):
        if rc != 0:
            raise SlamHipError(f"libslamhip error {rc}: {self.lib.slam_last_error(self.h).decode()}")

    @property
    def stream(self):
        return self.lib.slam_ctx_stream(self.h)

    def prof_enable(self, on=True):
        self.check(self.lib.slam_prof_enable(self.h, 1 if on else 0))

    def prof_reset(self):
        self.check(self.lib.slam_prof_reset(self.h))

    def prof_get(self, name):
        """-> (total device ms, span count) of a named span since the last reset."""
        ms, n = C.c_double(), C.c_int64()
        self.check(self.lib.slam_prof_get(self.h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def wait_for(self, other):
        """Device-side: later work on this context waits for what `other` has enqueued so far."""
        self.check(self.lib.slam_ctx_wait_for(self.h, other.h))

    def record(self, event=None):
        """Marks what this context has enqueued so far; returns the Event (re-records `event` if given)."""
        ev = event or Event(self)
        self.check(self.lib.slam_event_record(self.h, ev.h))
        return ev

    def wait_event(self, event):
        """Device-side: later work on this context waits for the point `event` marks on its own context."""
        self.check(self.lib.slam_ctx_wait_event(self.h, event.h))

    def synchronize(self):
        self.check(self.lib.slam_ctx_synchronize(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.slam_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class Event:
    """slam_event: a point in a context's stream that other contexts can wait on (Context.record / wait_event)."""

    def __init__(self, ctx, timed=False):
        self.lib = ctx.lib
        h = C.c_void_p()
        ctx.check((ctx.lib.slam_event_create_timed if timed else ctx.lib.slam_event_create)(ctx.h, C.byref(h)))
        self.h = h

    def elapsed_ms(self, later):
        """milliseconds from this (timed) event to `later`, both recorded; waits for `later`"""
        ms = C.c_double(0.0)
        rc = self.lib.slam_event_elapsed_ms(self.h, later.h, C.byref(ms))
        if rc != 0:
            raise SlamHipError(f"slam_event_elapsed_ms failed ({rc}): {self.lib.slam_last_error(None).decode()}")
        return ms.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.slam_event_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
