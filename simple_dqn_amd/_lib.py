"""ctypes binding of include/sdqn.h — the only caller of libsdqn_hip.so.

Status codes are mapped the way the reference behaves: precondition / shape violations
(SDQN_ERR_ARG) become AssertionError like the reference's asserts (deepqnetwork.py:110-116,
replay_memory.py:27,38,52); HIP / RCCL failures become RuntimeError (SdqnError).
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
MT_WORDS = 625


class SdqnError(RuntimeError):
    pass


def lib_path():
    # SDQN_LIB_PATH: another build of the same library (same-box A/B of two builds in tools/exp); never set by the package or the tests
    return os.environ.get("SDQN_LIB_PATH") or os.path.join(_HERE, "libsdqn_hip.so")


class NetCfg(C.Structure):
    _fields_ = [("batch_size", C.c_int), ("history_length", C.c_int), ("screen_height", C.c_int),
                ("screen_width", C.c_int), ("num_actions", C.c_int), ("target_enabled", C.c_int),
                ("optimizer", C.c_int), ("datatype", C.c_int),
                ("discount_rate", C.c_double), ("clip_error", C.c_double), ("min_reward", C.c_double),
                ("max_reward", C.c_double), ("learning_rate", C.c_double), ("decay_rate", C.c_double),
                ("epsilon", C.c_double), ("beta_1", C.c_double), ("beta_2", C.c_double), ("loss_scale", C.c_double), ("batch_norm", C.c_double)]


_u8p, _i64p, _f32p, _u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_uint32)
_f64p = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/sdqn.h declares
SIGNATURES = {
    "sdqn_last_error": (C.c_char_p, []),
    "sdqn_version": (C.c_int, []),
    "sdqn_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "sdqn_set_device": (C.c_int, [C.c_int]),
    "sdqn_get_device": (C.c_int, [C.POINTER(C.c_int)]),
    "sdqn_device_sync": (C.c_int, []),
    "sdqn_mt_seed": (C.c_int, [_u32p, C.c_uint64]),
    "sdqn_mt_randint": (C.c_int, [_u32p, C.c_int64, C.c_int64, _i64p]),
    "sdqn_sample_indices": (C.c_int, [_u32p, _u8p, C.c_int64, C.c_int64, C.c_int, C.c_int, _i64p, _i64p]),
    "sdqn_replay_create": (C.c_int, [C.POINTER(_vp), C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sdqn_replay_destroy": (C.c_int, [_vp]),
    "sdqn_replay_host_ptrs": (C.c_int, [_vp, C.POINTER(_u8p), C.POINTER(_u8p), C.POINTER(_i64p), C.POINTER(_u8p)]),
    "sdqn_replay_minibatch_ptrs": (C.c_int, [_vp, C.POINTER(_u8p), C.POINTER(_u8p), C.POINTER(_u8p),
                                             C.POINTER(_i64p), C.POINTER(_u8p)]),
    "sdqn_replay_add": (C.c_int, [_vp, C.c_int, C.c_int64, _u8p, C.c_int]),
    "sdqn_replay_get_state": (C.c_int, [_vp, _i64p, _i64p]),
    "sdqn_replay_set_state": (C.c_int, [_vp, C.c_int64, C.c_int64]),
    "sdqn_replay_upload": (C.c_int, [_vp, C.c_int64, C.c_int64]),
    "sdqn_replay_upload_meta": (C.c_int, [_vp, C.c_int64, C.c_int64]),
    "sdqn_replay_sample": (C.c_int, [_vp, _u32p, _i64p, _i64p]),
    "sdqn_replay_gather": (C.c_int, [_vp, _i64p]),
    "sdqn_replay_minibatch_to_host": (C.c_int, [_vp]),
    "sdqn_replay_declare_minibatch_clean": (C.c_int, [_vp]),
    "sdqn_net_step_structure": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sdqn_net_tuple_counters": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sdqn_replay_minibatch_gen": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "sdqn_replay_declare_minibatch_on_device": (C.c_int, [_vp, C.c_uint64]),
    "sdqn_replay_bench_gather": (C.c_int, [_vp, _i64p, C.c_int, _f32p]),
    "sdqn_replay_bench_gather_sets": (C.c_int, [_vp, _i64p, C.c_int, C.c_int, _f32p]),
    "sdqn_net_create": (C.c_int, [C.POINTER(_vp), C.POINTER(NetCfg)]),
    "sdqn_net_destroy": (C.c_int, [_vp]),
    "sdqn_net_layer_size": (C.c_int, [_vp, C.c_int, _i64p]),
    "sdqn_net_set_weights": (C.c_int, [_vp, C.c_int, C.c_int, _f32p, C.c_int64]),
    "sdqn_net_get_weights": (C.c_int, [_vp, C.c_int, C.c_int, _f32p, C.c_int64]),
    "sdqn_net_predict": (C.c_int, [_vp, _u8p, _f32p]),
    "sdqn_net_predict_f64": (C.c_int, [_vp, _u8p, _f64p]),
    "sdqn_net_set_weights_f64": (C.c_int, [_vp, C.c_int, C.c_int, _f64p, C.c_int64]),
    "sdqn_net_get_weights_f64": (C.c_int, [_vp, C.c_int, C.c_int, _f64p, C.c_int64]),
    "sdqn_net_predict_one": (C.c_int, [_vp, _u8p, _f32p]),
    "sdqn_statebuf_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int]),
    "sdqn_statebuf_destroy": (C.c_int, [_vp]),
    "sdqn_statebuf_add": (C.c_int, [_vp, _u8p]),
    "sdqn_statebuf_reset": (C.c_int, [_vp]),
    "sdqn_statebuf_get": (C.c_int, [_vp, _u8p]),
    "sdqn_statebuf_read_device": (C.c_int, [_vp, _u8p]),
    "sdqn_net_predict_state": (C.c_int, [_vp, _vp, _f32p]),
    "sdqn_net_act_step": (C.c_int, [_vp, _vp, _vp, _u8p, C.c_int, C.c_int64, C.c_int, C.c_int]),
    "sdqn_net_act_greedy": (C.c_int, [_vp, _vp, C.POINTER(C.c_int), _f32p]),
    "sdqn_net_debug_act": (C.c_int, [_vp, _vp, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]),
    "sdqn_net_train_host": (C.c_int, [_vp, _u8p, _u8p, _i64p, _u8p, _u8p, _f32p]),
    "sdqn_net_train_replay": (C.c_int, [_vp, _vp, _i64p, _f32p]),
    "sdqn_net_train_many": (C.c_int, [_vp, _vp, _u32p, C.c_int, _f32p]),
    "sdqn_net_train_many_deferred": (C.c_int, [_vp, _vp, _u32p, C.c_int, C.POINTER(C.c_int64)]),
    "sdqn_net_cost_collect": (C.c_int, [_vp, C.c_int64, _f32p]),
    "sdqn_mt_words": (C.c_int, [C.POINTER(C.c_uint64)]),
    "sdqn_net_update_target": (C.c_int, [_vp]),
    "sdqn_net_sync": (C.c_int, [_vp]),
    "sdqn_net_apply_update": (C.c_int, [_vp, C.c_double]),
    "sdqn_net_grad_to_half": (C.c_int, [_vp, C.POINTER(C.c_uint16), C.c_int64]),
    "sdqn_net_grad_from_half": (C.c_int, [_vp, C.POINTER(C.c_uint16), C.c_int64]),
    "sdqn_net_half_payload_state": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sdqn_net_last_q": (C.c_int, [_vp, _f32p, _f32p]),
    "sdqn_net_train_iterations": (C.c_int, [_vp, _i64p]),
    "sdqn_net_overflow_steps": (C.c_int, [_vp, _i64p]),
    "sdqn_net_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "sdqn_net_set_epoch": (C.c_int, [_vp, C.c_int]),
    "sdqn_net_debug_read": (C.c_int, [_vp, C.c_char_p, _f32p, C.c_int64]),
    "sdqn_net_profile": (C.c_int, [_vp, C.c_int, C.c_int]),
    "sdqn_net_profile_count": (C.c_int, [C.POINTER(C.c_int)]),
    "sdqn_net_profile_read": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), _i64p]),
    "sdqn_net_profile_reset": (C.c_int, [_vp]),
    "sdqn_dp_unique_id": (C.c_int, [C.c_char_p, C.c_char_p]),
    "sdqn_dp_init": (C.c_int, [_vp, C.c_char_p, C.c_char_p, C.c_int, C.c_int]),
    "sdqn_dp_shutdown": (C.c_int, [_vp]),
    "sdqn_dp_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sdqn_dp_probe": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "sdqn_dp_set_overlap": (C.c_int, [_vp, C.c_int]),
    "sdqn_dp_form": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}

_lib = None


def load():
    """dlopen libsdqn_hip.so (built in-tree by __graft_entry__.build()).  No fallback: a missing
    library is an error, never a silent CPU path.

    HIP runtime note: the library needs libamdhip64.so.7.  If torch was imported first its
    bundled runtime (same soname) is already mapped and is reused; otherwise /opt/rocm's is
    loaded.  Never import torch AFTER this library in the same process (two HIP runtimes)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise SdqnError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    _runtime_env()
    lib = C.CDLL(path, mode=os.RTLD_NOW | os.RTLD_LOCAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError here == header/library mismatch
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _hip_runtime_mapped():
    try:
        with open("/proc/self/maps") as f:
            return any("libamdhip64" in line or "libhsa-runtime64" in line for line in f)
    except OSError:
        return False


def _runtime_env():
    """Two process-wide runtime knobs this library depends on.  They only take effect if they are in the environment BEFORE
    the HSA / HIP runtime initialises, so they are set here only when no runtime is mapped yet (the launchers — bench.py,
    main.py, tests/conftest — export them themselves); if a runtime is already up (torch imported first) and a knob is
    missing, that is reported instead of silently depending on import order.
      HSA_ENABLE_IPC_MODE_LEGACY=0 : the host driver only supports dmabuf IPC — without it RCCL's cross-process buffer
                                     registration fails with "hipIpcGetMemHandle: invalid argument" (multi-GPU only)
      HIP_FORCE_DEV_KERNARG=1      : kernel arguments in HBM (this stack's default; =0 costs 23 % of the step rate)"""
    want = {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "HIP_FORCE_DEV_KERNARG": "1"}
    missing = [k for k in want if k not in os.environ]
    if not missing:
        return
    if _hip_runtime_mapped():
        import warnings
        warnings.warn("simple_dqn_amd: a HIP runtime was initialised before this library with %s unset; export %s before "
                      "importing torch / HIP (data-parallel runs need the first, the step rate the second)"
                      % (", ".join(missing), " ".join("%s=%s" % (k, want[k]) for k in missing)), RuntimeWarning, stacklevel=3)
        return
    for k in missing:
        os.environ[k] = want[k]


def bind_device(args):
    """--device_id of the reference (src/main.py:52, deepqnetwork.py:29-34): the drop-in classes bind the library to
    args.device_id before their first device call; a second object asking for a different device raises."""
    dev = getattr(args, "device_id", None)
    if dev is None:
        return
    check(load().sdqn_set_device(int(dev)))


def check(rc):
    if rc == 0:
        return
    msg = (load().sdqn_last_error() or b"").decode("utf-8", "replace")
    if rc == -1:
        raise AssertionError(msg)
    raise SdqnError("libsdqn_hip status %d: %s" % (rc, msg))


_ARRAY_TYPES = {}


def ptr(arr, ctype):
    """A ctypes view of a numpy array's memory for a POINTER(ctype) parameter.  Writable arrays go through the buffer protocol
    ((ctype * n).from_buffer: 0.7 us, and the object keeps the array alive for the call); read-only ones through numpy's
    ctypes.data_as (2.3 us) — the reference-style loop passes seven arrays per step."""
    if arr.flags.writeable and arr.flags.c_contiguous and arr.size:
        key = (ctype, arr.nbytes // C.sizeof(ctype))
        t = _ARRAY_TYPES.get(key)
        if t is None:
            t = _ARRAY_TYPES[key] = ctype * key[1]
        return t.from_buffer(arr)
    return arr.ctypes.data_as(C.POINTER(ctype))


def rccl_path():
    """The RCCL this process already has mapped (torch's bundled one, if torch is loaded), else ROCm's."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "librccl" in line:
                    return line.split()[-1]
    except OSError:
        pass
    if "torch" in sys.modules:
        import torch
        p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(p):
            return p
    return "/opt/rocm/lib/librccl.so.1"
