"""Build + ctypes binding of libuwspr_hip.so (include/uwspr_hip.h).

This module is plumbing only: it compiles the HIP sources for gfx950 in-tree
and maps the C ABI one-to-one.  There is no Python or CPU implementation of the
path behind it: if the library cannot be built or loaded, or no gfx950 device
is present, calls raise UwsprError.
"""
import ctypes as C
import hashlib
import os
import shutil
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
# Experiment builds (UWSPR_EXTRA_HIPFLAGS set: -DK6_EXP=..., stamps, flag A/Bs) go to their OWN
# library file, so the product libuwspr_hip.so is never replaced by a build whose results may be
# invalid; with the variable unset the product library is (re)built from the default flags.
_EXTRA = os.environ.get("UWSPR_EXTRA_HIPFLAGS", "").replace(",", " ").split()   # (commas: for shell A/B scripts)
# (one file per flag set, so that interleaved A/B runs do not rebuild each other's library)
_EXTRA_TAG = hashlib.sha256(" ".join(_EXTRA).encode()).hexdigest()[:8] if _EXTRA else ""
LIBPATH = os.path.join(LIBDIR, "libuwspr_hip_exp_%s.so" % _EXTRA_TAG if _EXTRA else "libuwspr_hip.so")
HOSTLIB = os.path.join(LIBDIR, "libuwspr_blocks.so")

SOURCES = ["uwspr_api.hip", "k0_frontend.hip", "k1_spectrogram.hip", "k2_spectrum.hip", "k3_coarse.hip",
           "k4_tonecorr.hip", "k4_grid.hip", "k4_pair.hip", "k4_jig.hip", "k5_fold_schedule.hip", "k6_sched.hip", "pipe.hip", "dist.hip", "host_tail.cpp"]
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
            "-fno-slp-vectorize", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-Wall", "-Wno-unused-function"]

NSYM, NSLM, NK0, NIFR, NJIG = 162, 125, 26, 5, 17
HOST, DEVICE, DEVICE_FRAMES, HOST_ASYNC = 0, 1, 2, 3
LINEAR, NONLINEAR = 0, 1


class UwsprError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("uwspr status %d: %s" % (status, msg))
        self.status = status


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise UwsprError(-4, "hipcc not found")


def _compiler_id():
    """What identifies the compiler in the build stamp without starting it on every import: the resolved
    path of hipcc, its size and the ROCm release file next to it (another ROCm => another stamp => rebuild).
    "absent" when there is no compiler: a shipped library whose stamp says otherwise is then still used."""
    try:
        cc = os.path.realpath(_hipcc())
    except UwsprError:
        return "hipcc absent"
    ver = ""
    for v in (os.path.join(os.path.dirname(os.path.dirname(cc)), ".info", "version"), "/opt/rocm/.info/version"):
        if os.path.exists(v):
            ver = open(v).read().strip()
            break
    return "%s size %d rocm %s" % (cc, os.path.getsize(cc), ver)


def _digest(paths):
    """Content hash of the build inputs: what decides whether the library is current (file
    times do not survive a copy to another machine; contents do)."""
    import hashlib
    h = hashlib.sha256()
    for q in sorted(paths):
        h.update(os.path.basename(q).encode())
        with open(q, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def source_digest():
    """sha256 over the sources the product library is built from (what a profile taken with
    tools/run_profiles.sh is valid for)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    deps.append(os.path.join(_HERE, "..", "include", "uwspr_hip.h"))
    return _digest(deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 every kernel + the C ABI into lib/libuwspr_hip.so,
    then the host block mirror (gr-uwspr_amd/host) into lib/libuwspr_blocks.so.

    Safe to call from several processes at once (one rank per GPU importing the package at the same moment): the
    check and the build run under a file lock, a build goes to a temporary file that is renamed into place, and
    what decides "current" is the flags + the CONTENTS of the sources (a stamp next to the library) -- not file
    times and not the directory the tree happens to live in, neither of which survives the copy to a GPU box."""
    import fcntl
    if not os.path.isdir(LIBDIR):
        os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    deps.append(os.path.join(_HERE, "..", "include", "uwspr_hip.h"))
    extra = _EXTRA   # experiments only (separate output file, see LIBPATH)
    flags = HIPFLAGS + extra + ["-shared", "-pthread"]
    hostdir = os.path.join(_HERE, "host")
    hsrcs = [os.path.join(hostdir, f) for f in sorted(os.listdir(hostdir)) if f.endswith(".cc")] \
        if os.path.isdir(hostdir) else []
    hdeps = hsrcs + ([os.path.join(hostdir, f) for f in os.listdir(hostdir) if f.endswith(".h")] if hsrcs else [])
    hflags = ["-O2", "-std=c++17", "-fPIC", "-shared", "-Wall"]

    def atomically(cmd, out):
        tmp = "%s.tmp%d" % (out, os.getpid())
        try:
            if verbose:
                print(" ".join(cmd + ["-o", out]))
            subprocess.run(cmd + ["-o", tmp], check=True)
            os.replace(tmp, out)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)

    def current(out, want):
        stamp = out + ".cmd"
        return os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == want

    def stamp(out, want):
        tmp = "%s.cmd.tmp%d" % (out, os.getpid())
        with open(tmp, "w") as f:
            f.write(want)
        os.replace(tmp, out + ".cmd")

    cid = _compiler_id()
    want = " ".join(flags + [os.path.basename(q) for q in srcs]) + "\n" + _digest(deps) + "\n" + cid
    hwant = " ".join(hflags + [os.path.basename(q) for q in hsrcs]) + "\n" + _digest(hdeps + deps) + "\n" + cid

    def all_current():
        return current(LIBPATH, want) and (not hsrcs or bool(extra) or current(HOSTLIB, hwant))

    # the common case -- a shipped, current library -- takes no lock and writes nothing (a read-only tree imports)
    if not force and all_current():
        return LIBPATH
    if cid == "hipcc absent" and not force and os.path.exists(LIBPATH):
        return LIBPATH                       # nothing to rebuild with: the loader decides whether the file is usable
    lock = None
    try:
        lock = open(os.path.join(LIBDIR, ".build.lock"), "w")
        fcntl.flock(lock, fcntl.LOCK_EX)     # (released when the file is closed)
    except OSError as e:                     # EROFS / EACCES: build (if it must) without the lock
        import errno
        if e.errno not in (errno.EROFS, errno.EACCES, errno.EPERM):
            raise
        lock = None
    try:
        if force or not current(LIBPATH, want):
            atomically([_hipcc()] + flags + srcs + ["-ldl"], LIBPATH)
            stamp(LIBPATH, want)
        if hsrcs and not extra:
            if force or not current(HOSTLIB, hwant):
                atomically(["g++"] + hflags + ["-I" + hostdir, "-I" + os.path.join(_HERE, "..", "include")] + hsrcs +
                           ["-L" + LIBDIR, "-luwspr_hip", "-Wl,-rpath,$ORIGIN", "-lpthread"], HOSTLIB)
                stamp(HOSTLIB, hwant)
    finally:
        if lock is not None:
            lock.close()
    return LIBPATH


# ---- ABI structures --------------------------------------------------------
class Params(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("fs", "fl", "spb", "maxdrift", "maxfreqs", "halfbandwidth", "cf", "threshold")]


CAND_DTYPE = np.dtype([("freq", "<f4"), ("snr", "<f4"), ("drift", "<f4"), ("sync", "<f4"),
                       ("shift", "<i4"), ("m_type", "<i4"), ("V1", "<f8"), ("V2", "<f8"),
                       ("p1", "<i4"), ("p2", "<i4")])
HYP_DTYPE = np.dtype([("frame", "<i4"), ("m_type", "<i4"), ("f0", "<f4"), ("lag", "<i4"),
                      ("drift", "<f4"), ("p1", "<i4"), ("p2", "<i4"), ("_pad", "<i4"),
                      ("V1", "<f8"), ("V2", "<f8")])
DEMOD_DTYPE = np.dtype([("f1", "<f4"), ("drift1", "<f4"), ("sync1", "<f4"), ("shift1", "<i4"),
                        ("worth_a_try", "<i4"), ("jig_sync", "<f4", (NJIG,)),
                        ("jig_rms", "<f4", (NJIG,)), ("jig_shift", "<i4", (NJIG,)),
                        ("symbols", "u1", (NJIG, NSYM)), ("_pad", "u1", (2,))])
CALL_DTYPE = np.dtype([("frame", "<i4"), ("_p0", "<i4"), ("candidate", CAND_DTYPE),
                       ("f1", "<f4"), ("ifmin", "<i4"), ("ifmax", "<i4"), ("fstep", "<f4"),
                       ("shift1", "<i4"), ("lagmin", "<i4"), ("lagmax", "<i4"),
                       ("lagstep", "<i4"), ("drift1", "<f4"), ("symfac", "<i4"),
                       ("mode", "<i4"), ("_p1", "<i4")])
RESULT_DTYPE = np.dtype([("sync", "<f4"), ("shift1", "<i4"), ("f1", "<f4"),
                         ("symbols", "u1", (NSYM,)), ("_pad", "u1", (2,))])
assert CAND_DTYPE.itemsize == 48 and HYP_DTYPE.itemsize == 48
assert DEMOD_DTYPE.itemsize == 2980 and CALL_DTYPE.itemsize == 104 and RESULT_DTYPE.itemsize == 176


class Info(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("size", C.c_int32), ("m", C.c_int32),
                ("hpbm", C.c_int32), ("n", C.c_int32), ("finpb", C.c_int32),
                ("noiseidx", C.c_int32), ("df", C.c_float), ("min_snr", C.c_float),
                ("band_lo", C.c_int32), ("band_w", C.c_int32), ("cell_hyps", C.c_int32),
                ("off_min", C.c_int32), ("off_max", C.c_int32), ("device", C.c_int32),
                ("device_name", C.c_char * 64)]


DECODE_DTYPE = np.dtype([("frame", "<i8"), ("stream_pos", "<i8"), ("cand", "<i4"), ("npk", "<i4"),
                         ("coarse", CAND_DTYPE), ("f1", "<f4"), ("drift1", "<f4"), ("sync1", "<f4"),
                         ("shift1", "<i4"), ("worth_a_try", "<i4"), ("decoded", "<i4"), ("idt", "<i4"),
                         ("message", "i1", (7,)), ("_pad", "u1", (5,))])
assert DECODE_DTYPE.itemsize == 112


class PipeOpts(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("hop", "batch_frames", "max_per_frame", "lanes", "host_threads", "eager",
                                         "sched_form", "spare_after_us")]


class PipeStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("frames", "batches", "candidates", "decoded", "resumed", "fano_calls",
                                         "fano_timeouts")] + \
               [(k, C.c_double) for k in ("gpu_wait_s", "fano_s", "resume_s")]


K_NAMES = ("spectrogram", "spectrum", "coarse", "tonecorr", "fold", "sched")


class Prof(C.Structure):
    _fields_ = [("ms", C.c_double * 6), ("launches", C.c_int64 * 6), ("units", C.c_int64 * 6)]


# every symbol include/uwspr_hip.h declares
ABI_SYMBOLS = [
    "uwspr_ctx_create", "uwspr_ctx_destroy", "uwspr_last_error", "uwspr_status_string",
    "uwspr_get_info", "uwspr_set_stream", "uwspr_synchronize", "uwspr_frontend_batch",
    "uwspr_frontend_design", "uwspr_set_frame_stride", "uwspr_stream_open", "uwspr_stream_push", "uwspr_stream_wait_uploads",
    "uwspr_stream_take_view", "uwspr_stream_take", "uwspr_stream_reset",
    "uwspr_device_alloc", "uwspr_device_free", "uwspr_host_alloc", "uwspr_host_free", "uwspr_fdr_batch",
    "uwspr_fdr_read_spectrum", "uwspr_fdr_keep_syncgrid", "uwspr_fdr_read_syncgrid",
    "uwspr_sync_sweep", "uwspr_sync_grid", "uwspr_sync_and_demodulate_batch", "uwspr_demod_batch",
    "uwspr_pipeline_batch", "uwspr_set_tries", "uwspr_set_option", "uwspr_get_option", "uwspr_demod_resume", "uwspr_pack_slabs", "uwspr_pipeline_slabs", "uwspr_prof_enable", "uwspr_prof_read", "uwspr_prof_intervals", "uwspr_deinterleave",
    "uwspr_fano_decode", "uwspr_fano_encode", "uwspr_decode_candidate", "uwspr_host_threads", "uwspr_host_set_ranks", "uwspr_decode_batch", "uwspr_unpack_message",
    "uwspr_c2_read",
    "uwspr_dist_unique_id", "uwspr_dist_init", "uwspr_dist_gather", "uwspr_dist_finalize",
    "uwspr_pipe_open", "uwspr_pipe_close", "uwspr_pipe_last_error", "uwspr_pipe_acquire", "uwspr_pipe_commit",
    "uwspr_pipe_push", "uwspr_pipe_submit_device", "uwspr_pipe_flush", "uwspr_pipe_collect", "uwspr_pipe_get_stats",
    "uwspr_pipe_inject_failure", "uwspr_pipe_set_option",
]

_lib = None


def lib():
    """Load (building first if needed) the shared library.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    # always through build(): the command stamp + mtime check is cheap and a stale or foreign
    # library is never loaded silently.  Where the sources cannot be compiled (no hipcc on a
    # deployment box) an existing library is used as it is.
    try:
        build()
    except (UwsprError, subprocess.CalledProcessError, OSError):
        if not os.path.exists(LIBPATH):
            raise
    # One HIP/HSA runtime per process: PyTorch bundles its own libamdhip64.so.7 /
    # libhsa-runtime64 and this library is linked against /opt/rocm's (same SONAME).
    # Whichever is loaded first serves both; two live copies leave the second one
    # without a device.  Load torch's first whenever torch is going to be used.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIBPATH)
    vp, ip = C.c_void_p, C.c_int
    L.uwspr_ctx_create.argtypes = [C.POINTER(Params), ip, C.POINTER(vp)]
    L.uwspr_ctx_destroy.argtypes = [vp]
    L.uwspr_ctx_destroy.restype = None
    L.uwspr_last_error.argtypes = [vp]
    L.uwspr_last_error.restype = C.c_char_p
    L.uwspr_status_string.argtypes = [ip]
    L.uwspr_status_string.restype = C.c_char_p
    L.uwspr_get_info.argtypes = [vp, C.POINTER(Info)]
    L.uwspr_set_stream.argtypes = [vp, vp]
    L.uwspr_synchronize.argtypes = [vp]
    L.uwspr_frontend_batch.argtypes = [vp, vp, ip, ip, ip, vp]
    L.uwspr_frontend_design.argtypes = [ip, ip, vp, ip, vp]
    L.uwspr_stream_open.argtypes = [vp, ip, ip]
    L.uwspr_stream_push.argtypes = [vp, vp, ip, ip, C.POINTER(C.c_int)]
    L.uwspr_stream_take.argtypes = [vp, ip, vp, C.POINTER(vp), C.POINTER(C.c_longlong)]
    L.uwspr_stream_reset.argtypes = [vp, C.c_longlong]
    L.uwspr_set_frame_stride.argtypes = [vp, ip]
    L.uwspr_stream_wait_uploads.argtypes = [vp]
    L.uwspr_stream_take_view.argtypes = [vp, ip, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
    L.uwspr_device_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.uwspr_device_free.argtypes = [vp]
    L.uwspr_device_free.restype = None
    L.uwspr_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.uwspr_host_free.argtypes = [vp]
    L.uwspr_host_free.restype = None
    L.uwspr_fdr_batch.argtypes = [vp, vp, ip, ip, vp, vp]
    L.uwspr_fdr_read_spectrum.argtypes = [vp, ip, vp, vp, vp, vp, vp]
    L.uwspr_fdr_keep_syncgrid.argtypes = [vp, ip]
    L.uwspr_fdr_read_syncgrid.argtypes = [vp, ip, vp]
    L.uwspr_sync_sweep.argtypes = [vp, vp, ip, vp, ip, ip, vp, vp]
    L.uwspr_sync_grid.argtypes = [vp, vp, ip, ip, vp, ip, vp, ip, vp, ip, vp, vp, vp]
    L.uwspr_sync_and_demodulate_batch.argtypes = [vp, vp, ip, ip, vp, ip, vp]
    L.uwspr_demod_batch.argtypes = [vp, vp, ip, ip, vp, vp, ip, ip, vp]
    L.uwspr_pipeline_batch.argtypes = [vp, vp, ip, ip, ip, vp, vp, vp]
    L.uwspr_set_tries.argtypes = [vp, ip]
    L.uwspr_set_option.argtypes = [vp, C.c_char_p, ip]
    L.uwspr_get_option.argtypes = [vp, C.c_char_p, vp]
    L.uwspr_demod_resume.argtypes = [vp, vp, ip, ip, vp, ip, vp]
    L.uwspr_pack_slabs.argtypes = [vp, ip, ip, vp, ip]
    L.uwspr_pipeline_slabs.argtypes = [vp, ip, vp]
    L.uwspr_prof_enable.argtypes = [vp, ip]
    L.uwspr_prof_read.argtypes = [vp, C.POINTER(Prof)]
    L.uwspr_prof_intervals.argtypes = [vp, ip, vp, vp, vp, ip, C.POINTER(C.c_int)]
    L.uwspr_deinterleave.argtypes = [vp]
    L.uwspr_deinterleave.restype = None
    L.uwspr_fano_decode.argtypes = [vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_uint32), ip, C.c_uint32]
    L.uwspr_fano_encode.argtypes = [vp, vp, C.c_uint32]
    L.uwspr_decode_candidate.argtypes = [vp, vp, C.POINTER(C.c_int32)]
    L.uwspr_decode_batch.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    L.uwspr_decode_batch.restype = C.c_int
    L.uwspr_host_threads.argtypes = []
    L.uwspr_host_set_ranks.argtypes = [ip]
    L.uwspr_unpack_message.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.uwspr_c2_read.argtypes = [C.c_char_p, vp, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.uwspr_dist_unique_id.argtypes = [vp]
    L.uwspr_dist_init.argtypes = [vp, ip, ip, vp]
    L.uwspr_dist_gather.argtypes = [vp, vp, C.c_size_t, vp, ip, ip]
    L.uwspr_dist_finalize.argtypes = [vp]
    L.uwspr_pipe_open.argtypes = [C.POINTER(Params), ip, C.POINTER(PipeOpts), C.POINTER(vp)]
    L.uwspr_pipe_close.argtypes = [vp]
    L.uwspr_pipe_close.restype = None
    L.uwspr_pipe_last_error.argtypes = [vp]
    L.uwspr_pipe_last_error.restype = C.c_char_p
    L.uwspr_pipe_acquire.argtypes = [vp, ip, C.POINTER(vp)]
    L.uwspr_pipe_commit.argtypes = [vp, ip]
    L.uwspr_pipe_push.argtypes = [vp, vp, ip]
    L.uwspr_pipe_submit_device.argtypes = [vp, vp, ip, ip]
    L.uwspr_pipe_flush.argtypes = [vp]
    L.uwspr_pipe_collect.argtypes = [vp, vp, ip, ip]
    L.uwspr_pipe_get_stats.argtypes = [vp, C.POINTER(PipeStats)]
    L.uwspr_pipe_inject_failure.argtypes = [vp, C.c_longlong, ip]
    L.uwspr_pipe_set_option.argtypes = [vp, C.c_char_p, ip]
    _lib = L
    return L
