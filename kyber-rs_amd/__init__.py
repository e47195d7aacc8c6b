"""kyber-rs_amd — MI355X-native batched Ed25519 scalar-multiplication engine for kyber-rs.

This package is the thin Python harness over the C ABI declared in include/kyber_ed25519.h
(implemented by csrc/engine.hip + csrc/kernels_*.hip -> libkyber_ed25519_hip.so).  It exists for tests, bench.py and
multi-GPU bring-up over torch.distributed; the product boundary is the C ABI itself (the Rust shim in
rust/ and the C++ mirror in host/ bind the same symbols).

There is no CPU implementation here: `load_library()` raises if the HIP library has not been built,
and `Engine()` raises if no gfx950 device is usable.  (Directory name has a hyphen; import it as
`kyber_rs_amd`, the one-file alias package at the repo root.)
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libkyber_ed25519_hip.so"
LIB_PATH = os.environ.get("KYB_HIP_LIB") or os.path.join(_HERE, LIB_NAME)   # KYB_HIP_LIB: A/B builds of the same ABI
# the cross-check build (csrc/Makefile CROSSCHECK=1): the same ABI plus the alternative kernels and the options that select them — for the
# tests that compare the product's kernels with them; the product library has neither
CROSSCHECK_LIB_PATH = os.path.join(_HERE, "libkyber_ed25519_hip_crosscheck.so")
BASE_TABLE_BYTES = 335232
ABI_VERSION = 2

KYB_OK = 0
ERRORS = {-1: "KYB_E_NOT_INIT", -2: "KYB_E_BAD_ARG", -3: "KYB_E_NO_DEVICE", -4: "KYB_E_HIP", -5: "KYB_E_NOMEM", -6: "KYB_E_TRANSPORT", -7: "KYB_E_STALE"}

# every symbol include/kyber_ed25519.h declares (tests/test_abi_symbols.py checks header <-> library)
ABI_SYMBOLS = [
    "kyb_abi_version", "kyb_init", "kyb_init_no_table", "kyb_shutdown", "kyb_last_error", "kyb_device_info", "kyb_sync", "kyb_stream_release",
    "kyb_ctx_create", "kyb_ctx_destroy", "kyb_ctx_set_current", "kyb_ctx_get_current", "kyb_ctx_device",
    "kyb_group_create", "kyb_group_create_ex", "kyb_group_table_transport_note", "kyb_group_destroy", "kyb_group_size", "kyb_group_ctx", "kyb_group_table_transport",
    "kyb_group_mul_base_batch", "kyb_group_mul_batch", "kyb_group_schnorr_sign_batch", "kyb_group_verify_batch",
    "kyb_base_table_export_dev", "kyb_base_table_import_dev", "kyb_base_table_export", "kyb_base_table_import",
    "kyb_group_mul_base_batch_dev", "kyb_group_mul_batch_dev", "kyb_group_sync",
    "kyb_mul_base_batch", "kyb_mul_base_batch_dev", "kyb_mul_batch", "kyb_mul_batch_dev", "kyb_mul_public_batch",
    "kyb_add_batch", "kyb_add_batch_dev", "kyb_encode_batch", "kyb_encode_batch_dev",
    "kyb_decode_batch", "kyb_decode_batch_dev", "kyb_schnorr_sign_batch", "kyb_schnorr_sign_batch_dev",
    "kyb_eddsa_sign_batch", "kyb_eddsa_sign_batch_dev",
    "kyb_schnorr_sign_keyed_batch", "kyb_schnorr_sign_keyed_batch_dev", "kyb_eddsa_sign_keyed_batch", "kyb_eddsa_sign_keyed_batch_dev",
    "kyb_verify_batch", "kyb_verify_batch_dev", "kyb_verify_points_batch", "kyb_verify_points_batch_dev", "kyb_pubpoly_eval_batch", "kyb_pubpoly_eval_batch_dev",
    "kyb_pubpoly_eval_multi_batch", "kyb_pubpoly_eval_multi_batch_dev",
    "kyb_equal_batch", "kyb_equal_batch_dev", "kyb_point_checks_batch", "kyb_point_checks_batch_dev", "kyb_lincomb_batch", "kyb_lincomb_batch_dev", "kyb_lincomb_public_batch", "kyb_lincomb_public_batch_dev", "kyb_pripoly_eval_batch", "kyb_pripoly_eval_batch_dev", "kyb_lagrange_coeffs_batch", "kyb_lagrange_coeffs_batch_dev",
    "kyb_sum_batch", "kyb_sum_batch_dev",
    "kyb_pubpoly_eval_multi_enc_batch", "kyb_pubpoly_eval_multi_enc_batch_dev", "kyb_sum_enc_batch", "kyb_sum_enc_batch_dev",
    "kyb_dkg_verify_round_enc", "kyb_dkg_verify_round_enc_dev",
    "kyb_defer_input", "kyb_defer_input_enc", "kyb_defer_null", "kyb_defer_base", "kyb_defer_mul_base", "kyb_defer_mul", "kyb_defer_add", "kyb_defer_neg", "kyb_defer_get", "kyb_defer_equal",
    "kyb_defer_flush", "kyb_defer_mark", "kyb_defer_floor", "kyb_defer_stats",
    "kyb_host_alloc", "kyb_host_free",
    "kyb_set_option", "kyb_get_option", "kyb_profile_begin", "kyb_profile_read", "kyb_kernel_name",
    "kyb_diag_mad_peak", "kyb_diag_wave_stamps",
]
# declared inside `#ifdef KYB_CROSSCHECK` of the header: test hooks only the cross-check build exports
CROSSCHECK_ONLY_SYMBOLS = ["kyb_diag_scratch_read", "kyb_diag_coop", "kyb_diag_phase_stamps"]


def kernel_sources_id() -> str:
    """Identifies the code of the two dominant kernels (k_mul_ladder, k_mul_base64): SHA-256 over their translation units and every header of csrc/.
    bench.py only replays HBM-traffic counters from profiles/ when they were collected on THIS code (tools/summarise_profiles.py stores the id)."""
    import hashlib
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    names = sorted(f for f in os.listdir(csrc) if f.endswith(".h") or f in ("kernels_ladder.hip", "kernels_base.hip", "consts.inc"))
    h = hashlib.sha256()
    for f in names:
        h.update(f.encode() + b"\0" + open(os.path.join(csrc, f), "rb").read() + b"\0")
    return h.hexdigest()[:16]


class KyberHipError(RuntimeError):
    pass


STREAM_ENGINE = 0     # C ABI: NULL = the context's own non-blocking stream
STREAM_LEGACY = 1     # C ABI: KYB_STREAM_LEGACY = the device's null stream (hipStreamLegacy)


def _torch_current_stream(device_index):
    """handle of torch's current stream on that device as the C ABI names it, or None when torch is not in the process / has no GPU"""
    import sys
    torch = sys.modules.get("torch")
    if torch is None or device_index is None or not torch.cuda.is_available():
        return None
    return int(torch.cuda.current_stream(device_index).cuda_stream) or STREAM_LEGACY


_lib = None
_xlib = None


def load_library(crosscheck: bool = False) -> ctypes.CDLL:
    """dlopen the in-tree HIP library; never falls back to anything else.  crosscheck: the cross-check build instead (a second library in the
    process, RTLD_LOCAL: its contexts, options and error text are its own)."""
    global _lib, _xlib
    if not crosscheck and _lib is not None:
        return _lib
    if crosscheck and _xlib is not None:
        return _xlib
    path = CROSSCHECK_LIB_PATH if crosscheck else LIB_PATH
    try:
        # torch bundles its own libamdhip64.so.7 / libhsa-runtime64 (and librccl); whichever HIP runtime is loaded
        # first owns the GPU for the process, so when torch is going to share device memory with the
        # engine (tests, bench.py) it has to be imported before our library resolves its DT_NEEDED.
        # KYB_NO_TORCH=1: a process that uses host pointers only and wants the system's ROCm libraries
        # (tests/test_gpu_rccl_stub.py puts a stand-in librccl in front of them).
        if not os.environ.get("KYB_NO_TORCH"):
            import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise KyberHipError(
            f"{path} is missing: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for this engine.")
    lib = ctypes.CDLL(path)
    vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    lib.kyb_init.argtypes = [i32]
    lib.kyb_init_no_table.argtypes = [i32]
    lib.kyb_shutdown.restype = None
    lib.kyb_last_error.restype = ctypes.c_char_p
    lib.kyb_device_info.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(i32), ctypes.POINTER(sz)]
    lib.kyb_sync.argtypes = [vp]
    lib.kyb_stream_release.argtypes = [vp]
    lib.kyb_ctx_create.argtypes = [i32, i32, ctypes.POINTER(vp)]
    lib.kyb_ctx_destroy.argtypes = [vp]
    lib.kyb_ctx_set_current.argtypes = [vp]
    lib.kyb_ctx_get_current.restype = vp
    lib.kyb_ctx_device.argtypes = [vp]
    lib.kyb_group_create.argtypes = [ctypes.POINTER(i32), i32, ctypes.POINTER(vp)]
    lib.kyb_group_create_ex.argtypes = [ctypes.POINTER(i32), i32, ctypes.c_uint, ctypes.POINTER(vp)]
    lib.kyb_group_table_transport_note.argtypes = [vp]
    lib.kyb_group_table_transport_note.restype = ctypes.c_char_p
    lib.kyb_group_destroy.argtypes = [vp]
    lib.kyb_group_destroy.restype = None
    lib.kyb_group_size.argtypes = [vp]
    lib.kyb_group_ctx.argtypes = [vp, i32]
    lib.kyb_group_ctx.restype = vp
    lib.kyb_group_table_transport.argtypes = [vp]
    lib.kyb_group_table_transport.restype = ctypes.c_char_p
    lib.kyb_group_mul_base_batch.argtypes = [vp, vp, sz, vp, vp]
    lib.kyb_group_mul_batch.argtypes = [vp, vp, vp, vp, sz, vp, vp, vp]
    lib.kyb_group_schnorr_sign_batch.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    lib.kyb_group_verify_batch.argtypes = [vp, vp, vp, vp, vp, sz, i32, vp]
    lib.kyb_base_table_import.argtypes = [vp]
    lib.kyb_base_table_export_dev.argtypes = [vp, vp]
    lib.kyb_base_table_import_dev.argtypes = [vp, vp]
    lib.kyb_base_table_export.argtypes = [vp]
    lib.kyb_mul_base_batch.argtypes = [vp, sz, vp, vp]
    lib.kyb_mul_base_batch_dev.argtypes = [vp, sz, vp, vp, vp]
    lib.kyb_mul_batch.argtypes = [vp, vp, vp, sz, vp, vp, vp]
    lib.kyb_mul_batch_dev.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp]
    lib.kyb_mul_public_batch.argtypes = [vp, vp, vp, sz, vp, vp, vp]
    lib.kyb_group_mul_base_batch_dev.argtypes = [vp, vp, vp, vp, vp]
    lib.kyb_group_mul_batch_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.kyb_group_sync.argtypes = [vp]
    dp = ctypes.POINTER(ctypes.c_double)
    lib.kyb_diag_mad_peak.argtypes = [ctypes.c_double, dp, dp, dp, dp]
    lib.kyb_diag_wave_stamps.argtypes = [vp]
    if crosscheck:
        lib.kyb_diag_scratch_read.argtypes = [ctypes.c_int, vp, sz, ctypes.POINTER(ctypes.c_size_t)]
        lib.kyb_diag_coop.argtypes = [ctypes.c_int, vp, vp, vp]
        lib.kyb_diag_phase_stamps.argtypes = [vp]
    lib.kyb_add_batch.argtypes = [vp, vp, sz, vp, i32]
    lib.kyb_add_batch_dev.argtypes = [vp, vp, sz, vp, i32, vp]
    lib.kyb_encode_batch.argtypes = [vp, sz, vp]
    lib.kyb_encode_batch_dev.argtypes = [vp, sz, vp, vp]
    lib.kyb_decode_batch.argtypes = [vp, sz, vp, vp]
    lib.kyb_decode_batch_dev.argtypes = [vp, sz, vp, vp, vp]
    lib.kyb_schnorr_sign_batch.argtypes = [vp, vp, vp, vp, sz, vp]
    lib.kyb_schnorr_sign_batch_dev.argtypes = [vp, vp, vp, vp, sz, vp, vp]
    lib.kyb_eddsa_sign_batch.argtypes = [vp, vp, vp, sz, vp, vp]
    lib.kyb_schnorr_sign_keyed_batch.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    lib.kyb_schnorr_sign_keyed_batch_dev.argtypes = [vp, vp, vp, vp, vp, sz, vp, vp]
    lib.kyb_eddsa_sign_keyed_batch.argtypes = [vp, vp, vp, vp, sz, vp]
    lib.kyb_eddsa_sign_keyed_batch_dev.argtypes = [vp, vp, vp, vp, sz, vp, vp]
    lib.kyb_eddsa_sign_batch_dev.argtypes = [vp, vp, vp, sz, vp, vp, vp]
    lib.kyb_verify_batch.argtypes = [vp, vp, vp, vp, sz, i32, vp]
    lib.kyb_verify_batch_dev.argtypes = [vp, vp, vp, vp, sz, i32, vp, vp]
    lib.kyb_verify_points_batch.argtypes = [vp, vp, vp, vp, sz, i32, vp]
    lib.kyb_verify_points_batch_dev.argtypes = [vp, vp, vp, vp, sz, i32, vp, vp]
    lib.kyb_pubpoly_eval_batch.argtypes = [vp, sz, vp, sz, vp, vp]
    lib.kyb_pubpoly_eval_batch_dev.argtypes = [vp, sz, vp, sz, ctypes.c_uint32, vp, vp, vp]
    lib.kyb_pubpoly_eval_multi_batch.argtypes = [vp, sz, sz, vp, sz, vp, vp]
    lib.kyb_pubpoly_eval_multi_batch_dev.argtypes = [vp, sz, sz, vp, sz, ctypes.c_uint32, vp, vp, vp]
    lib.kyb_sum_batch.argtypes = [vp, sz, sz, vp, vp]
    lib.kyb_sum_batch_dev.argtypes = [vp, sz, sz, vp, vp, vp]
    lib.kyb_pubpoly_eval_multi_enc_batch.argtypes = [vp, sz, sz, vp, sz, vp, vp, vp]
    lib.kyb_pubpoly_eval_multi_enc_batch_dev.argtypes = [vp, sz, sz, vp, sz, ctypes.c_uint32, vp, vp, vp, vp]
    lib.kyb_sum_enc_batch.argtypes = [vp, sz, sz, i32, vp, vp, vp]
    lib.kyb_sum_enc_batch_dev.argtypes = [vp, sz, sz, i32, vp, vp, vp, vp]
    lib.kyb_dkg_verify_round_enc.argtypes = [vp, sz, sz, ctypes.c_uint32, vp, vp, vp, vp, vp]
    lib.kyb_dkg_verify_round_enc_dev.argtypes = [vp, sz, sz, vp, ctypes.c_uint32, vp, vp, vp, vp, vp, vp]
    lib.kyb_equal_batch.argtypes = [vp, vp, sz, vp]
    lib.kyb_equal_batch_dev.argtypes = [vp, vp, sz, vp, vp]
    u64, pu64 = ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)
    lib.kyb_defer_input.argtypes = [vp, pu64]
    lib.kyb_defer_input_enc.argtypes = [vp, vp, pu64]
    lib.kyb_defer_null.argtypes = [pu64]
    lib.kyb_defer_base.argtypes = [pu64]
    lib.kyb_defer_mul_base.argtypes = [vp, pu64]
    lib.kyb_defer_mul.argtypes = [vp, u64, pu64]
    lib.kyb_defer_add.argtypes = [u64, u64, i32, pu64]
    lib.kyb_defer_neg.argtypes = [u64, pu64]
    lib.kyb_defer_get.argtypes = [u64, vp, vp]
    lib.kyb_defer_equal.argtypes = [u64, u64, vp]
    lib.kyb_defer_flush.argtypes = []
    lib.kyb_defer_mark.argtypes = []
    lib.kyb_defer_mark.restype = u64
    lib.kyb_defer_floor.argtypes = [u64]
    lib.kyb_defer_stats.argtypes = [pu64, i32]
    lib.kyb_point_checks_batch.argtypes = [vp, vp, sz, vp]
    lib.kyb_point_checks_batch_dev.argtypes = [vp, vp, sz, vp, vp]
    lib.kyb_lincomb_batch.argtypes = [vp, vp, vp, i32, sz, sz, vp, vp, vp]
    lib.kyb_lincomb_batch_dev.argtypes = [vp, vp, vp, i32, sz, sz, vp, vp, vp, vp]
    lib.kyb_pripoly_eval_batch.argtypes = [vp, sz, sz, vp, sz, vp]
    lib.kyb_pripoly_eval_batch_dev.argtypes = [vp, sz, sz, vp, sz, vp, vp]
    lib.kyb_lagrange_coeffs_batch.argtypes = [vp, sz, sz, vp]
    lib.kyb_lagrange_coeffs_batch_dev.argtypes = [vp, sz, sz, vp, vp]
    lib.kyb_lincomb_public_batch.argtypes = [vp, vp, vp, i32, sz, sz, vp, vp, vp]
    lib.kyb_lincomb_public_batch_dev.argtypes = [vp, vp, vp, i32, sz, sz, vp, vp, vp, vp]
    lib.kyb_set_option.argtypes = [ctypes.c_char_p, i32]
    lib.kyb_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(i32)]
    lib.kyb_host_alloc.argtypes = [sz]
    lib.kyb_host_alloc.restype = vp
    lib.kyb_host_free.argtypes = [vp]
    lib.kyb_host_free.restype = None
    lib.kyb_profile_begin.argtypes = [i32]
    lib.kyb_profile_read.argtypes = [ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_float), i32, ctypes.POINTER(i32)]
    lib.kyb_kernel_name.argtypes = [i32]
    lib.kyb_kernel_name.restype = ctypes.c_char_p
    for name in ABI_SYMBOLS:
        if name not in ("kyb_shutdown", "kyb_last_error", "kyb_kernel_name", "kyb_host_alloc", "kyb_host_free", "kyb_ctx_get_current",
                        "kyb_group_destroy", "kyb_group_ctx", "kyb_group_table_transport", "kyb_group_table_transport_note", "kyb_defer_mark"):
            getattr(lib, name).restype = i32
    if lib.kyb_abi_version() != ABI_VERSION:
        raise KyberHipError(f"{path} implements ABI version {lib.kyb_abi_version()}, this binding expects {ABI_VERSION}: rebuild it")
    if crosscheck:
        _xlib = lib
    else:
        _lib = lib
    return lib


_last_lib = threading.local()      # the library the calling thread's last engine call went to: whose kyb_last_error explains a failure


def _check(rc: int, what: str, lib=None) -> None:
    if rc != KYB_OK:
        lib = lib or getattr(_last_lib, "lib", None) or load_library()
        msg = lib.kyb_last_error().decode(errors="replace")
        raise KyberHipError(f"{what} failed: {ERRORS.get(rc, rc)}: {msg}")


def _u8(a, width: int, name: str) -> np.ndarray:
    arr = np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a, dtype=np.uint8)
    arr = arr.reshape(-1, width)
    return arr


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _rows(a: Optional[np.ndarray], n: int, name: str) -> None:
    """every per-item array must hold exactly n records: a shorter one would make the C side read past its buffer"""
    if a is not None and a.shape[0] != n:
        raise ValueError(f"{name} holds {a.shape[0]} records, expected {n}")


class PackedMessages:
    """n messages in the C ABI's own form — one byte blob and n + 1 offsets — packed once (pack_messages) and handed to any number of
    sign / verify calls, so that a timed call does not contain the Python loop that concatenates them"""

    def __init__(self, blob: np.ndarray, off: np.ndarray):
        self.blob, self.off = blob, off

    def __len__(self):
        return self.off.shape[0] - 1

    def __getitem__(self, sl):
        """messages [a, b) of the pack (a slice with step 1): the same blob, offsets rebased"""
        a, b, step = sl.indices(len(self))
        if step != 1:
            raise ValueError("PackedMessages takes contiguous slices only")
        off = self.off[a:b + 1]
        lo, hi = int(off[0]), int(off[-1])
        return PackedMessages(np.ascontiguousarray(np.append(self.blob[lo:hi], np.uint8(0))), (off - off[0]).astype(np.uint32))


def pack_messages(msgs: Sequence[bytes]) -> PackedMessages:
    return PackedMessages(*_msg_blob(msgs, len(msgs)))


def _msg_blob(msgs, n: int):
    """concatenated messages + the n+1 offsets of the C ABI (uint32: the blob must stay below 4 GiB)"""
    if isinstance(msgs, PackedMessages):
        if len(msgs) != n:
            raise ValueError(f"{len(msgs)} messages for {n} items")
        return msgs.blob, msgs.off
    if len(msgs) != n:
        raise ValueError(f"{len(msgs)} messages for {n} items")
    total = sum(len(m) for m in msgs)
    if total >= 1 << 32:
        raise ValueError("message blob of 4 GiB or more: msg_off is uint32")
    off = np.zeros(n + 1, dtype=np.uint32)
    if n:
        off[1:] = np.cumsum([len(m) for m in msgs], dtype=np.uint64).astype(np.uint32)
    blob = np.frombuffer(b"".join(msgs) + b"\0", dtype=np.uint8).copy()
    return blob, off


class _CtxLib:
    """the ctypes library seen through one explicit context: every call first makes that context current on the calling
    thread (kyb_ctx_set_current is thread-local and costs nanoseconds)"""

    def __init__(self, lib, ctx):
        object.__setattr__(self, "_lib", lib)
        object.__setattr__(self, "_ctx", ctx)

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        lib, ctx = self._lib, self._ctx

        def call(*args):
            _last_lib.lib = lib
            lib.kyb_ctx_set_current(ctx)
            return fn(*args)
        return call


class Engine:
    """One engine = one context of the C ABI.  Engine(device) is the process's default context (kyb_init: one process per
    GPU, bench.py); Engine(device, private=True) is an additional context of its own (kyb_ctx_create) — several may live
    in one process, on the same or on different GPUs."""

    def __init__(self, device: int = 0, build_table: bool = True, private: bool = False, _ctx=None, crosscheck: bool = False):
        lib = load_library(crosscheck)
        self._raw = lib
        _last_lib.lib = lib
        self.device = device
        self.ctx = None
        if _ctx is not None:                 # a context owned by a Group
            self.ctx, self._owned = _ctx, False
            self.lib = _CtxLib(lib, ctypes.c_void_p(_ctx))
        elif private:
            h = ctypes.c_void_p()
            _check(lib.kyb_ctx_create(device, 1 if build_table else 0, ctypes.byref(h)), "kyb_ctx_create")
            self.ctx, self._owned = h.value, True
            self.lib = _CtxLib(lib, ctypes.c_void_p(h.value))
        else:
            self.lib = _CtxLib(lib, None)    # None = the default context, whatever this thread had made current before
            rc = lib.kyb_init(device) if build_table else lib.kyb_init_no_table(device)
            _check(rc, "kyb_init")

    def close(self) -> None:
        """destroy a private context (the default one goes with shutdown())"""
        if self.ctx is not None and getattr(self, "_owned", False):
            raw = self._raw
            raw.kyb_ctx_set_current(None)
            _check(raw.kyb_ctx_destroy(ctypes.c_void_p(self.ctx)), "kyb_ctx_destroy")
            self.ctx = None

    def stream_release(self, stream: int) -> None:
        _check(self.lib.kyb_stream_release(ctypes.c_void_p(stream)), "kyb_stream_release")

    # ---- info / options -------------------------------------------------------------------------
    def device_info(self):
        name = ctypes.create_string_buffer(128)
        cus, ws = ctypes.c_int(0), ctypes.c_size_t(0)
        _check(self.lib.kyb_device_info(name, 128, ctypes.byref(cus), ctypes.byref(ws)), "kyb_device_info")
        return {"name": name.value.decode(), "compute_units": cus.value, "workspace_bytes": ws.value}

    def set_option(self, key: str, value: int) -> None:
        _check(self.lib.kyb_set_option(key.encode(), int(value)), f"kyb_set_option({key})")

    def get_option(self, key: str) -> int:
        v = ctypes.c_int(0)
        _check(self.lib.kyb_get_option(key.encode(), ctypes.byref(v)), f"kyb_get_option({key})")
        return v.value

    def pinned_array(self, shape, dtype) -> np.ndarray:
        """numpy array over page-locked host memory (kyb_host_alloc); freed when the array is collected"""
        dt = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dt.itemsize
        ptr = self.lib.kyb_host_alloc(nbytes)
        if not ptr:
            _check(-5, "kyb_host_alloc")
        buf = (ctypes.c_uint8 * nbytes).from_address(ptr)
        arr = np.frombuffer(buf, dtype=dt).reshape(shape)
        lib = self.lib
        import weakref
        weakref.finalize(buf, lib.kyb_host_free, ctypes.c_void_p(ptr))
        return arr

    def mul_into(self, scalars, pts_ext, out_enc) -> None:
        """kyb_mul_batch on caller-provided arrays (e.g. pinned ones); no allocation, no copy on the Python side"""
        _check(self.lib.kyb_mul_batch(_ptr(scalars), None, _ptr(pts_ext), scalars.shape[0], _ptr(out_enc), None, None), "kyb_mul_batch")

    def mul_base_into(self, scalars, out_enc) -> None:
        _check(self.lib.kyb_mul_base_batch(_ptr(scalars), scalars.shape[0], _ptr(out_enc), None), "kyb_mul_base_batch")

    def mad_peak(self, min_ms: float = 50.0):
        """kyb_diag_mad_peak: the chip's v_mad_u64_u32 rate, clock and issue cycles, measured now (a benchmark diagnostic)"""
        r, c, y, k = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
        _check(self.lib.kyb_diag_mad_peak(float(min_ms), ctypes.byref(r), ctypes.byref(c), ctypes.byref(y), ctypes.byref(k)), "kyb_diag_mad_peak")
        return {"mads_per_s": r.value, "clock_ghz": c.value, "simd_cycles_per_mad": y.value, "kernel_ms": k.value}

    def scratch_read(self, which: int) -> bytes:
        """kyb_diag_scratch_read: the present contents of one of the context's buffers (test hook for the secret-hygiene checks; cross-check build only)"""
        size = ctypes.c_size_t(0)
        _check(self.lib.kyb_diag_scratch_read(which, None, 0, ctypes.byref(size)), "kyb_diag_scratch_read")
        if size.value == 0:
            return b""
        buf = np.empty(size.value, dtype=np.uint8)
        _check(self.lib.kyb_diag_scratch_read(which, _ptr(buf), size.value, ctypes.byref(size)), "kyb_diag_scratch_read")
        return buf.tobytes()

    def wave_stamps(self, buf) -> None:
        """kyb_diag_wave_stamps: buf = zeroed torch int64 tensor of 5 words on this engine's device, or None = off"""
        _check(self.lib.kyb_diag_wave_stamps(None if buf is None else ctypes.c_void_p(buf.data_ptr())), "kyb_diag_wave_stamps")

    def profile_begin(self, max_launches: int) -> None:
        _check(self.lib.kyb_profile_begin(max_launches), "kyb_profile_begin")

    def profile_read(self, cap: int = 4096):
        """-> list of (kernel name, milliseconds) in launch order"""
        ids = (ctypes.c_int * cap)()
        ms = (ctypes.c_float * cap)()
        cnt = ctypes.c_int(0)
        _check(self.lib.kyb_profile_read(ids, ms, cap, ctypes.byref(cnt)), "kyb_profile_read")
        return [(self.lib.kyb_kernel_name(ids[i]).decode(), float(ms[i])) for i in range(cnt.value)]

    def sync(self, stream: Optional[int] = None) -> None:
        """wait for the work queued by this engine: stream=None = both the engine's own stream and the stream the _dev methods
        launch on by default (torch's current stream of the engine's device); an explicit handle = that stream (0: the engine's own)"""
        if stream is not None:
            _check(self.lib.kyb_sync(ctypes.c_void_p(stream)), "kyb_sync")
            return
        _check(self.lib.kyb_sync(None), "kyb_sync")
        cur = _torch_current_stream(self.device)
        if cur is not None:
            _check(self.lib.kyb_sync(ctypes.c_void_p(cur)), "kyb_sync")

    def shutdown(self) -> None:
        self.lib.kyb_shutdown()

    # ---- host-buffer API (numpy in, numpy out) ---------------------------------------------------
    def mul_base(self, scalars, want_ext: bool = False, ext_only: bool = False):
        """ext_only: extended limbs only (what a trait-level Point::mul keeps); with the option ext.projective they may have Z != 1"""
        s = _u8(scalars, 32, "scalars")
        n = s.shape[0]
        enc = None if ext_only else np.empty((n, 32), dtype=np.uint8)
        ext = np.empty((n, 40), dtype=np.int32) if (want_ext or ext_only) else None
        _check(self.lib.kyb_mul_base_batch(_ptr(s), n, _ptr(enc), _ptr(ext)), "kyb_mul_base_batch")
        if ext_only:
            return ext
        return (enc, ext) if want_ext else enc

    def mul(self, scalars, pts_ext=None, pts_enc=None, want_ext: bool = False, want_ok: bool = False, ext_only: bool = False, public: bool = False):
        """public: the multipliers are declared public (kyb_mul_public_batch: short ones take a short ladder)"""
        fn = self.lib.kyb_mul_public_batch if public else self.lib.kyb_mul_batch
        if ext_only:
            s = _u8(scalars, 32, "scalars")
            px = np.ascontiguousarray(pts_ext, dtype=np.int32).reshape(-1, 40)
            _rows(px, s.shape[0], "pts_ext")
            ext = np.empty((s.shape[0], 40), dtype=np.int32)
            _check(fn(_ptr(s), None, _ptr(px), s.shape[0], None, _ptr(ext), None), "kyb_mul_batch")
            return ext
        s = _u8(scalars, 32, "scalars")
        n = s.shape[0]
        pe = None if pts_enc is None else _u8(pts_enc, 32, "pts_enc")
        px = None if pts_ext is None else np.ascontiguousarray(pts_ext, dtype=np.int32).reshape(-1, 40)
        _rows(pe, n, "pts_enc"); _rows(px, n, "pts_ext")
        enc = np.empty((n, 32), dtype=np.uint8)
        ext = np.empty((n, 40), dtype=np.int32) if want_ext else None
        ok = np.empty((n,), dtype=np.uint8) if (want_ok or pe is not None) else None
        _check(fn(_ptr(s), _ptr(pe), _ptr(px), n, _ptr(enc), _ptr(ext), _ptr(ok)), "kyb_mul_batch")
        out = [enc]
        if want_ext:
            out.append(ext)
        if want_ok:
            out.append(ok)
        return out[0] if len(out) == 1 else tuple(out)

    def add(self, a_ext, b_ext, subtract: bool = False):
        a = np.ascontiguousarray(a_ext, dtype=np.int32).reshape(-1, 40)
        b = np.ascontiguousarray(b_ext, dtype=np.int32).reshape(-1, 40)
        _rows(b, a.shape[0], "b_ext")
        out = np.empty_like(a)
        _check(self.lib.kyb_add_batch(_ptr(a), _ptr(b), a.shape[0], _ptr(out), 1 if subtract else 0), "kyb_add_batch")
        return out

    def encode(self, pts_ext):
        p = np.ascontiguousarray(pts_ext, dtype=np.int32).reshape(-1, 40)
        enc = np.empty((p.shape[0], 32), dtype=np.uint8)
        _check(self.lib.kyb_encode_batch(_ptr(p), p.shape[0], _ptr(enc)), "kyb_encode_batch")
        return enc

    def decode(self, enc):
        e = _u8(enc, 32, "enc")
        ext = np.empty((e.shape[0], 40), dtype=np.int32)
        ok = np.empty((e.shape[0],), dtype=np.uint8)
        _check(self.lib.kyb_decode_batch(_ptr(e), e.shape[0], _ptr(ext), _ptr(ok)), "kyb_decode_batch")
        return ext, ok

    def schnorr_sign(self, x, k, msgs: Sequence[bytes], pubs=None):
        """schnorr::sign with caller-supplied nonces; pubs = the stored public keys enc(x*B) (then A is not recomputed)"""
        xs, ks = _u8(x, 32, "x"), _u8(k, 32, "k")
        n = xs.shape[0]
        _rows(ks, n, "k")
        blob, off = _msg_blob(msgs, n)
        sig = np.empty((n, 64), dtype=np.uint8)
        if pubs is None:
            _check(self.lib.kyb_schnorr_sign_batch(_ptr(xs), _ptr(ks), _ptr(blob), _ptr(off), n, _ptr(sig)), "kyb_schnorr_sign_batch")
        else:
            ps = _u8(pubs, 32, "pubs")
            _rows(ps, n, "pubs")
            _check(self.lib.kyb_schnorr_sign_keyed_batch(_ptr(xs), _ptr(ps), _ptr(ks), _ptr(blob), _ptr(off), n, _ptr(sig)), "kyb_schnorr_sign_keyed_batch")
        return sig

    def eddsa_sign(self, seeds, msgs: Sequence[bytes], want_pub: bool = False, pubs=None):
        """EdDSA::sign for (seed, msg) pairs; optionally also the public keys; pubs = the public keys the key
        objects already hold (then only R is computed)"""
        sd = _u8(seeds, 32, "seeds")
        n = sd.shape[0]
        blob, off = _msg_blob(msgs, n)
        sig = np.empty((n, 64), dtype=np.uint8)
        if pubs is not None:
            ps = _u8(pubs, 32, "pubs")
            _rows(ps, n, "pubs")
            _check(self.lib.kyb_eddsa_sign_keyed_batch(_ptr(sd), _ptr(ps), _ptr(blob), _ptr(off), n, _ptr(sig)), "kyb_eddsa_sign_keyed_batch")
            return (sig, ps) if want_pub else sig
        pub = np.empty((n, 32), dtype=np.uint8) if want_pub else None
        _check(self.lib.kyb_eddsa_sign_batch(_ptr(sd), _ptr(blob), _ptr(off), n, _ptr(sig), _ptr(pub)), "kyb_eddsa_sign_batch")
        return (sig, pub) if want_pub else sig

    def verify(self, pubs, msgs: Sequence[bytes], sigs, flavor: int = 0) -> np.ndarray:
        """status per item (0 = valid); flavor 0 = eddsa::verify_with_checks order, 1 = schnorr order"""
        ps, ss = _u8(pubs, 32, "pubs"), _u8(sigs, 64, "sigs")
        n = ps.shape[0]
        _rows(ss, n, "sigs")
        blob, off = _msg_blob(msgs, n)
        st = np.empty((n,), dtype=np.uint8)
        _check(self.lib.kyb_verify_batch(_ptr(ps), _ptr(blob), _ptr(off), _ptr(ss), n, flavor, _ptr(st)), "kyb_verify_batch")
        return st

    def verify_points(self, pubs_ext, msgs: Sequence[bytes], sigs, flavor: int = 0) -> np.ndarray:
        """kyb_verify_points_batch: as verify(), the public keys given as points (n x 40 limbs, any Z) — schnorr::verify / eddsa::verify"""
        px = np.ascontiguousarray(pubs_ext, dtype=np.int32).reshape(-1, 40)
        ss = _u8(sigs, 64, "sigs")
        n = px.shape[0]
        _rows(ss, n, "sigs")
        blob, off = _msg_blob(msgs, n)
        st = np.empty((n,), dtype=np.uint8)
        _check(self.lib.kyb_verify_points_batch(_ptr(px), _ptr(blob), _ptr(off), _ptr(ss), n, flavor, _ptr(st)), "kyb_verify_points_batch")
        return st

    def verify_points_dev(self, pubs_ext, msgs, msg_off, sigs, status, flavor: int = 0, stream: Optional[int] = None) -> None:
        n = pubs_ext.numel() // 40
        _check(self.lib.kyb_verify_points_batch_dev(self._dp(pubs_ext), self._dp(msgs), self._dp(msg_off), self._dp(sigs), n, flavor, self._dp(status), self._st(stream, pubs_ext)),
               "kyb_verify_points_batch_dev")

    def verify_dev(self, pubs, msgs, msg_off, sigs, status, flavor: int = 0, stream: Optional[int] = None) -> None:
        n = pubs.numel() // 32
        _check(self.lib.kyb_verify_batch_dev(self._dp(pubs), self._dp(msgs), self._dp(msg_off), self._dp(sigs), n, flavor, self._dp(status), self._st(stream, pubs)), "kyb_verify_batch_dev")

    def pubpoly_eval(self, commits_ext, indices, want_ext: bool = False, ext_only: bool = False):
        """PubPoly::eval of one polynomial (t x 40 limbs) at every index of `indices` (x = index + 1)"""
        c = np.ascontiguousarray(commits_ext, dtype=np.int32).reshape(-1, 40)
        idx = np.ascontiguousarray(indices, dtype=np.uint32)
        n = idx.shape[0]
        enc = None if ext_only else np.empty((n, 32), dtype=np.uint8)
        ext = np.empty((n, 40), dtype=np.int32) if (want_ext or ext_only) else None
        _check(self.lib.kyb_pubpoly_eval_batch(_ptr(c), c.shape[0], _ptr(idx), n, _ptr(enc), _ptr(ext)), "kyb_pubpoly_eval_batch")
        if ext_only:
            return ext
        return (enc, ext) if want_ext else enc

    def pubpoly_eval_multi(self, commits_ext, indices, want_ext: bool = False):
        """m polynomials (m x t x 40 limbs), polynomial g evaluated at indices[g, :] -> (m, k, 32) encodings"""
        c = np.ascontiguousarray(commits_ext, dtype=np.int32)
        if c.ndim != 3 or c.shape[2] != 40:
            raise ValueError("commits_ext must have shape (m, t, 40)")
        m, t = c.shape[0], c.shape[1]
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(m, -1)
        k = idx.shape[1]
        enc = np.empty((m, k, 32), dtype=np.uint8)
        ext = np.empty((m, k, 40), dtype=np.int32) if want_ext else None
        _check(self.lib.kyb_pubpoly_eval_multi_batch(_ptr(c), t, m, _ptr(idx), k, _ptr(enc), _ptr(ext)), "kyb_pubpoly_eval_multi_batch")
        return (enc, ext) if want_ext else enc

    def pubpoly_eval_multi_enc(self, commits_enc, indices, want_ext: bool = False):
        """as pubpoly_eval_multi, the commitments as wire encodings (m, t, 32) -> (encodings (m, k, 32)[, ext], ok (m, t))"""
        c = np.ascontiguousarray(commits_enc, dtype=np.uint8)
        if c.ndim != 3 or c.shape[2] != 32:
            raise ValueError("commits_enc must have shape (m, t, 32)")
        m, t = c.shape[0], c.shape[1]
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(m, -1)
        k = idx.shape[1]
        enc = np.empty((m, k, 32), dtype=np.uint8)
        ext = np.empty((m, k, 40), dtype=np.int32) if want_ext else None
        ok = np.empty((m, t), dtype=np.uint8)
        _check(self.lib.kyb_pubpoly_eval_multi_enc_batch(_ptr(c), t, m, _ptr(idx), k, _ptr(enc), _ptr(ext), _ptr(ok)), "kyb_pubpoly_eval_multi_enc_batch")
        return (enc, ext, ok) if want_ext else (enc, ok)

    def dkg_verify_round_enc(self, commits_enc, index: int, want_sums: bool = True):
        """kyb_dkg_verify_round_enc: commits_enc (m, t, 32) as received -> (evaluations (m, 32) at `index`, column sums (t, 32) or None, ok (m, t))"""
        c = np.ascontiguousarray(commits_enc, dtype=np.uint8)
        if c.ndim != 3 or c.shape[2] != 32:
            raise ValueError("commits_enc must have shape (m, t, 32)")
        m, t = c.shape[0], c.shape[1]
        ev = np.empty((m, 32), dtype=np.uint8)
        sums = np.empty((t, 32), dtype=np.uint8) if want_sums else None
        ok = np.empty((m, t), dtype=np.uint8)
        _check(self.lib.kyb_dkg_verify_round_enc(_ptr(c), t, m, int(index), _ptr(ev), None, _ptr(sums), None, _ptr(ok)), "kyb_dkg_verify_round_enc")
        return ev, sums, ok

    def sum_points_enc(self, pts_enc, item_major: bool = False, want_ext: bool = False):
        """sums of wire encodings: pts_enc (m, t, 32) -> out[g] = sum_j pts[g, j]; item_major: pts_enc (t, m, 32) -> out[g] = sum_j pts[j, g]
        (t dealers' polynomials of m coefficients, as received).  Returns (encodings (m, 32)[, ext], ok in the shape of the input)."""
        p = np.ascontiguousarray(pts_enc, dtype=np.uint8)
        if p.ndim != 3 or p.shape[2] != 32:
            raise ValueError("pts_enc must have shape (m, t, 32) or, item_major, (t, m, 32)")
        m, t = (p.shape[1], p.shape[0]) if item_major else (p.shape[0], p.shape[1])
        enc = np.empty((m, 32), dtype=np.uint8)
        ext = np.empty((m, 40), dtype=np.int32) if want_ext else None
        ok = np.empty(p.shape[:2], dtype=np.uint8)
        _check(self.lib.kyb_sum_enc_batch(_ptr(p), m, t, int(item_major), _ptr(enc), _ptr(ext), _ptr(ok)), "kyb_sum_enc_batch")
        return (enc, ext, ok) if want_ext else (enc, ok)

    def sum_points(self, pts_ext, want_ext: bool = False, ext_only: bool = False):
        """out[g] = sum_j pts[g, j] for points of shape (m, t, 40)"""
        p = np.ascontiguousarray(pts_ext, dtype=np.int32)
        if p.ndim != 3 or p.shape[2] != 40:
            raise ValueError("pts_ext must have shape (m, t, 40)")
        m, t = p.shape[0], p.shape[1]
        enc = None if ext_only else np.empty((m, 32), dtype=np.uint8)
        ext = np.empty((m, 40), dtype=np.int32) if (want_ext or ext_only) else None
        _check(self.lib.kyb_sum_batch(_ptr(p), m, t, _ptr(enc), _ptr(ext)), "kyb_sum_batch")
        if ext_only:
            return ext
        return (enc, ext) if want_ext else enc

    def lincomb(self, scalars, pts_ext=None, pts_enc=None, want_ext: bool = False, want_ok: bool = False, public: bool = False):
        """out[g] = sum_j scalars[g, j] * P[g, j]  (points of shape (m, t, ..)) or * P[j] (points of shape (t, ..), shared
        by all groups) -- recover_commit / recover_pub_poly / PubPoly::add accumulation, poly.rs:486-507, 566-634"""
        sc = np.ascontiguousarray(scalars, dtype=np.uint8)
        if sc.ndim != 3 or sc.shape[2] != 32:
            raise ValueError("scalars must have shape (m, t, 32)")
        m, t = sc.shape[0], sc.shape[1]
        if (pts_ext is None) == (pts_enc is None):
            raise ValueError("give exactly one of pts_ext / pts_enc")
        if pts_ext is not None:
            pts = np.ascontiguousarray(pts_ext, dtype=np.int32)
            width = 40
        else:
            pts = np.ascontiguousarray(pts_enc, dtype=np.uint8)
            width = 32
        if pts.shape == (t, width):
            shared = 1
        elif pts.shape == (m, t, width):
            shared = 0
        else:
            raise ValueError(f"points must have shape ({t}, {width}) or ({m}, {t}, {width})")
        enc = np.empty((m, 32), dtype=np.uint8)
        ext = np.empty((m, 40), dtype=np.int32) if want_ext else None
        ok = np.empty((t if shared else m * t,), dtype=np.uint8) if want_ok else None
        fn = self.lib.kyb_lincomb_public_batch if public else self.lib.kyb_lincomb_batch      # public: the scalars are declared public (point tables)
        _check(fn(_ptr(sc), _ptr(pts) if pts_enc is not None else None, _ptr(pts) if pts_ext is not None else None,
                  shared, m, t, _ptr(enc), _ptr(ext), _ptr(ok)), "kyb_lincomb_batch")
        out = (enc,)
        if want_ext:
            out += (ext,)
        if want_ok:
            out += (ok,)
        return out if len(out) > 1 else enc

    def pripoly_eval(self, coeffs, indices) -> np.ndarray:
        """kyb_pripoly_eval_batch: coeffs (m, t, 32) secret polynomials (or (t, 32) for one), indices (k,) uint32 -> shares (m, k, 32) (or (k, 32)),
        PriPoly::eval at x = index + 1 (poly.rs:133-141)"""
        c = np.ascontiguousarray(coeffs, dtype=np.uint8)
        one = c.ndim == 2
        if one:
            c = c[None]
        if c.ndim != 3 or c.shape[2] != 32:
            raise ValueError("coeffs must have shape (m, t, 32) or (t, 32)")
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        out = np.empty((c.shape[0], idx.shape[0], 32), dtype=np.uint8)
        _check(self.lib.kyb_pripoly_eval_batch(_ptr(c), c.shape[0], c.shape[1], _ptr(idx), idx.shape[0], _ptr(out)), "kyb_pripoly_eval_batch")
        return out[0] if one else out

    def lagrange_coeffs(self, indices) -> np.ndarray:
        """kyb_lagrange_coeffs_batch: indices (m, t) uint32 share indices -> (m, t, 32) Lagrange coefficients at 0 (x = index + 1), mod L"""
        idx = np.ascontiguousarray(indices, dtype=np.uint32)
        if idx.ndim != 2:
            raise ValueError("indices must have shape (m, t)")
        out = np.empty(idx.shape + (32,), dtype=np.uint8)
        _check(self.lib.kyb_lagrange_coeffs_batch(_ptr(idx), idx.shape[0], idx.shape[1], _ptr(out)), "kyb_lagrange_coeffs_batch")
        return out

    def equal(self, a_ext, b_ext) -> np.ndarray:
        a = np.ascontiguousarray(a_ext, dtype=np.int32).reshape(-1, 40)
        b = np.ascontiguousarray(b_ext, dtype=np.int32).reshape(-1, 40)
        _rows(b, a.shape[0], "b_ext")
        eq = np.empty((a.shape[0],), dtype=np.uint8)
        _check(self.lib.kyb_equal_batch(_ptr(a), _ptr(b), a.shape[0], _ptr(eq)), "kyb_equal_batch")
        return eq

    # ---- deferred points (kyb_defer_*): handles are plain ints ----
    def _defer(self, fn, *args) -> int:
        h = ctypes.c_uint64(0)
        _check(fn(*args, ctypes.byref(h)), "kyb_defer_*")
        return int(h.value)

    def defer_input(self, ext) -> int:
        e = np.ascontiguousarray(ext, dtype=np.int32).reshape(40)
        return self._defer(self.lib.kyb_defer_input, _ptr(e))

    def defer_null(self) -> int:
        return self._defer(self.lib.kyb_defer_null)

    def defer_base(self) -> int:
        return self._defer(self.lib.kyb_defer_base)

    def defer_mul_base(self, scalar: bytes) -> int:
        return self._defer(self.lib.kyb_defer_mul_base, ctypes.c_char_p(bytes(scalar)))

    def defer_mul(self, scalar: bytes, p: int) -> int:
        return self._defer(self.lib.kyb_defer_mul, ctypes.c_char_p(bytes(scalar)), p)

    def defer_add(self, a: int, b: int, subtract: bool = False) -> int:
        return self._defer(self.lib.kyb_defer_add, a, b, int(subtract))

    def defer_neg(self, a: int) -> int:
        return self._defer(self.lib.kyb_defer_neg, a)

    def defer_get(self, p: int, want_ext: bool = False):
        """marshal_binary of a deferred point (and its limbs with want_ext): evaluates what it depends on"""
        enc = np.empty(32, dtype=np.uint8)
        ext = np.empty(40, dtype=np.int32) if want_ext else None
        _check(self.lib.kyb_defer_get(p, _ptr(ext) if want_ext else None, _ptr(enc)), "kyb_defer_get")
        return (enc.tobytes(), ext) if want_ext else enc.tobytes()

    def defer_get_ext(self, p: int) -> np.ndarray:
        ext = np.empty(40, dtype=np.int32)
        _check(self.lib.kyb_defer_get(p, _ptr(ext), None), "kyb_defer_get")
        return ext

    def defer_equal(self, a: int, b: int) -> bool:
        eq = np.zeros(1, dtype=np.uint8)
        _check(self.lib.kyb_defer_equal(a, b, _ptr(eq)), "kyb_defer_equal")
        return bool(eq[0])

    def defer_flush(self) -> None:
        _check(self.lib.kyb_defer_flush(), "kyb_defer_flush")

    def defer_mark(self) -> int:
        return int(self.lib.kyb_defer_mark())

    def defer_floor(self, mark: int) -> None:
        _check(self.lib.kyb_defer_floor(mark), "kyb_defer_floor")

    def defer_stats(self) -> dict:
        v = (ctypes.c_uint64 * 12)()
        _check(self.lib.kyb_defer_stats(v, 12), "kyb_defer_stats")
        return dict(zip(("nodes", "flushes", "engine_calls", "horner_fused", "sums_fused", "marshal_cache_hits", "nodes_held", "nodes_dropped",
                         "values_kept", "values_pushed_out", "kept_hits", "operands_readmitted"), (int(x) for x in v)))

    def point_checks(self, enc=None, pts_ext=None) -> np.ndarray:
        """kyb_point_checks_batch: flags per point, bit 0 = is_canonical (the reference's expression), bit 1 = has_small_order (point.rs:286-337)"""
        if (enc is None) == (pts_ext is None):
            raise ValueError("give exactly one of enc / pts_ext")
        if enc is not None:
            a = np.ascontiguousarray(enc, dtype=np.uint8).reshape(-1, 32)
            ext = None
        else:
            ext = np.ascontiguousarray(pts_ext, dtype=np.int32).reshape(-1, 40)
            a = None
        n = (a if a is not None else ext).shape[0]
        flags = np.empty((n,), dtype=np.uint8)
        _check(self.lib.kyb_point_checks_batch(_ptr(a) if a is not None else None, _ptr(ext) if ext is not None else None, n, _ptr(flags)), "kyb_point_checks_batch")
        return flags

    def base_table(self) -> np.ndarray:
        t = np.empty(BASE_TABLE_BYTES, dtype=np.uint8)
        _check(self.lib.kyb_base_table_export(_ptr(t)), "kyb_base_table_export")
        return t

    def base_table_import(self, image) -> None:
        t = np.ascontiguousarray(image, dtype=np.uint8).reshape(-1)
        if t.shape[0] != BASE_TABLE_BYTES:
            raise ValueError("table image has the wrong size")
        _check(self.lib.kyb_base_table_import(_ptr(t)), "kyb_base_table_import")

    # ---- device-pointer API (torch tensors resident in HBM; asynchronous on `stream`) -------------
    # stream=None (the default): the kernels are queued on torch's CURRENT stream of the tensors' device — the stream on which torch
    # produced the operands and will consume the results, so the call is ordered with both like any torch operation (when that is the
    # device's null stream, whose handle 0 means "the engine's own stream" in the C ABI, it is named as KYB_STREAM_LEGACY).
    # An explicit handle is passed through: 0 = the engine's own NON-BLOCKING stream, which is ordered with nothing — the caller then
    # synchronises its producers before the call and the engine (sync()) before it reads results or reuses the buffers.
    @staticmethod
    def _dp(t):
        return None if t is None else ctypes.c_void_p(t.data_ptr())

    @staticmethod
    def _st(stream, t):
        if stream is not None:
            return ctypes.c_void_p(stream)
        dev = getattr(t, "device", None)
        cur = _torch_current_stream(dev.index if getattr(dev, "type", None) == "cuda" else None) if dev is not None else None
        if cur is None:
            raise KyberHipError("a _dev call needs torch device tensors (to launch on torch's current stream) or an explicit stream handle")
        return ctypes.c_void_p(cur)

    def mul_base_dev(self, scalars, out_enc=None, out_ext=None, stream: Optional[int] = None) -> None:
        n = scalars.numel() // 32
        _check(self.lib.kyb_mul_base_batch_dev(self._dp(scalars), n, self._dp(out_enc), self._dp(out_ext), self._st(stream, scalars)), "kyb_mul_base_batch_dev")

    def mul_dev(self, scalars, pts_ext=None, pts_enc=None, out_enc=None, out_ext=None, ok=None, stream: Optional[int] = None) -> None:
        n = scalars.numel() // 32
        _check(self.lib.kyb_mul_batch_dev(self._dp(scalars), self._dp(pts_enc), self._dp(pts_ext), n, self._dp(out_enc), self._dp(out_ext), self._dp(ok), self._st(stream, scalars)), "kyb_mul_batch_dev")

    def lincomb_dev(self, scalars, m: int, t: int, pts_ext=None, pts_enc=None, shared: bool = False, out_enc=None, out_ext=None, ok=None, stream: Optional[int] = None) -> None:
        _check(self.lib.kyb_lincomb_batch_dev(self._dp(scalars), self._dp(pts_enc), self._dp(pts_ext), int(shared), m, t,
                                              self._dp(out_enc), self._dp(out_ext), self._dp(ok), self._st(stream, scalars)), "kyb_lincomb_batch_dev")

    def add_dev(self, a_ext, b_ext, out_ext, subtract: bool = False, stream: Optional[int] = None) -> None:
        n = a_ext.numel() // 40
        _check(self.lib.kyb_add_batch_dev(self._dp(a_ext), self._dp(b_ext), n, self._dp(out_ext), int(subtract), self._st(stream, a_ext)), "kyb_add_batch_dev")

    def equal_dev(self, a_ext, b_ext, eq, stream: Optional[int] = None) -> None:
        n = a_ext.numel() // 40
        _check(self.lib.kyb_equal_batch_dev(self._dp(a_ext), self._dp(b_ext), n, self._dp(eq), self._st(stream, a_ext)), "kyb_equal_batch_dev")

    def point_checks_dev(self, flags, enc=None, pts_ext=None, stream: Optional[int] = None) -> None:
        n = enc.numel() // 32 if enc is not None else pts_ext.numel() // 40
        _check(self.lib.kyb_point_checks_batch_dev(self._dp(enc), self._dp(pts_ext), n, self._dp(flags), self._st(stream, flags)), "kyb_point_checks_batch_dev")

    def encode_dev(self, pts_ext, out_enc, stream: Optional[int] = None) -> None:
        n = pts_ext.numel() // 40
        _check(self.lib.kyb_encode_batch_dev(self._dp(pts_ext), n, self._dp(out_enc), self._st(stream, pts_ext)), "kyb_encode_batch_dev")

    def decode_dev(self, enc, out_ext, ok=None, stream: Optional[int] = None) -> None:
        n = enc.numel() // 32
        _check(self.lib.kyb_decode_batch_dev(self._dp(enc), n, self._dp(out_ext), self._dp(ok), self._st(stream, enc)), "kyb_decode_batch_dev")

    def sign_dev(self, x, k, msgs, msg_off, sig, stream: Optional[int] = None, pubs=None) -> None:
        n = x.numel() // 32
        if pubs is not None:
            _check(self.lib.kyb_schnorr_sign_keyed_batch_dev(self._dp(x), self._dp(pubs), self._dp(k), self._dp(msgs), self._dp(msg_off), n, self._dp(sig), self._st(stream, x)), "kyb_schnorr_sign_keyed_batch_dev")
            return
        _check(self.lib.kyb_schnorr_sign_batch_dev(self._dp(x), self._dp(k), self._dp(msgs), self._dp(msg_off), n, self._dp(sig), self._st(stream, x)), "kyb_schnorr_sign_batch_dev")

    def base_table_export_dev(self, dst, stream: Optional[int] = None) -> None:
        _check(self.lib.kyb_base_table_export_dev(self._dp(dst), self._st(stream, dst)), "kyb_base_table_export_dev")

    def base_table_import_dev(self, src, stream: Optional[int] = None) -> None:
        _check(self.lib.kyb_base_table_import_dev(self._dp(src), self._st(stream, src)), "kyb_base_table_import_dev")


class Group:
    """kyb_group: one process driving several GPUs (one context + one host thread per device, shards [rN/G, (r+1)N/G),
    table image moved with ncclBroadcast / a host copy and validated by checksum).  `devices` may repeat a device
    (several contexts on one GPU: how the single-GPU test box exercises the sharding)."""

    REQUIRE_RCCL = 1
    RCCL_EVEN_IF_REPEATED = 2

    def __init__(self, devices: Sequence[int], flags: int = 0):
        self.lib = load_library()
        arr = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        _check(self.lib.kyb_group_create_ex(arr, len(devices), ctypes.c_uint(flags), ctypes.byref(h)), "kyb_group_create_ex", self.lib)
        self.handle = h
        self.size = self.lib.kyb_group_size(h)
        self.transport = self.lib.kyb_group_table_transport(h).decode()
        self.transport_note = self.lib.kyb_group_table_transport_note(h).decode()      # why RCCL was not used ("" when it was)

    def engine(self, rank: int) -> Engine:
        ctx = self.lib.kyb_group_ctx(self.handle, rank)
        return Engine(self.lib.kyb_ctx_device(ctypes.c_void_p(ctx)), _ctx=ctx)

    def close(self) -> None:
        if self.handle is not None:
            self.lib.kyb_group_destroy(self.handle)
            self.handle = None

    @staticmethod
    def _ptr_array(tensors):
        """per-rank device pointers (torch tensors, one per rank, each on its rank's GPU) -> void*[size], or None"""
        if tensors is None:
            return None
        return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    @staticmethod
    def _producers_done(*tensor_lists) -> None:
        """The group's calls run on every rank's OWN non-blocking stream, which is ordered with no torch stream: whatever torch has queued on
        the current stream of the shards' devices (the kernels that wrote the operands, a zero_() of an output) is waited for on the host
        before the shards are handed over.  (sync() is the other half: results are read after it.)"""
        import sys
        torch = sys.modules.get("torch")
        if torch is None:
            return
        seen = set()
        for lst in tensor_lists:
            for t in lst or ():
                dev = getattr(t, "device", None)
                if dev is not None and dev.type == "cuda" and dev.index not in seen:
                    seen.add(dev.index)
                    torch.cuda.current_stream(dev).synchronize()

    def mul_dev(self, scalars, pts_ext=None, pts_enc=None, out_enc=None, out_ext=None, ok=None) -> None:
        """kyb_group_mul_batch_dev: device-resident shards, asynchronous on every rank's own stream (sync() waits)"""
        self._producers_done(scalars, pts_ext, pts_enc, out_enc, out_ext, ok)
        n = (ctypes.c_size_t * self.size)(*[t.numel() // 32 for t in scalars])
        _check(self.lib.kyb_group_mul_batch_dev(self.handle, self._ptr_array(scalars), self._ptr_array(pts_enc), self._ptr_array(pts_ext), n,
                                                self._ptr_array(out_enc), self._ptr_array(out_ext), self._ptr_array(ok)), "kyb_group_mul_batch_dev", self.lib)

    def mul_base_dev(self, scalars, out_enc=None, out_ext=None) -> None:
        self._producers_done(scalars, out_enc, out_ext)
        n = (ctypes.c_size_t * self.size)(*[t.numel() // 32 for t in scalars])
        _check(self.lib.kyb_group_mul_base_batch_dev(self.handle, self._ptr_array(scalars), n, self._ptr_array(out_enc), self._ptr_array(out_ext)),
               "kyb_group_mul_base_batch_dev", self.lib)

    def sync(self) -> None:
        _check(self.lib.kyb_group_sync(self.handle), "kyb_group_sync", self.lib)

    def mul_base(self, scalars):
        s = _u8(scalars, 32, "scalars")
        enc = np.empty((s.shape[0], 32), dtype=np.uint8)
        _check(self.lib.kyb_group_mul_base_batch(self.handle, _ptr(s), s.shape[0], _ptr(enc), None), "kyb_group_mul_base_batch", self.lib)
        return enc

    def mul(self, scalars, pts_ext=None, pts_enc=None):
        s = _u8(scalars, 32, "scalars")
        n = s.shape[0]
        pe = None if pts_enc is None else _u8(pts_enc, 32, "pts_enc")
        px = None if pts_ext is None else np.ascontiguousarray(pts_ext, dtype=np.int32).reshape(-1, 40)
        _rows(pe, n, "pts_enc"); _rows(px, n, "pts_ext")
        enc = np.empty((n, 32), dtype=np.uint8)
        ok = np.empty((n,), dtype=np.uint8)
        _check(self.lib.kyb_group_mul_batch(self.handle, _ptr(s), _ptr(pe), _ptr(px), n, _ptr(enc), None, _ptr(ok)), "kyb_group_mul_batch", self.lib)
        return enc, ok

    def schnorr_sign(self, x, k, msgs: Sequence[bytes]):
        xs, ks = _u8(x, 32, "x"), _u8(k, 32, "k")
        n = xs.shape[0]
        _rows(ks, n, "k")
        blob, off = _msg_blob(msgs, n)
        sig = np.empty((n, 64), dtype=np.uint8)
        _check(self.lib.kyb_group_schnorr_sign_batch(self.handle, _ptr(xs), _ptr(ks), _ptr(blob), _ptr(off), n, _ptr(sig)), "kyb_group_schnorr_sign_batch", self.lib)
        return sig

    def verify(self, pubs, msgs: Sequence[bytes], sigs, flavor: int = 0):
        ps, ss = _u8(pubs, 32, "pubs"), _u8(sigs, 64, "sigs")
        n = ps.shape[0]
        _rows(ss, n, "sigs")
        blob, off = _msg_blob(msgs, n)
        st = np.empty((n,), dtype=np.uint8)
        _check(self.lib.kyb_group_verify_batch(self.handle, _ptr(ps), _ptr(blob), _ptr(off), _ptr(ss), n, flavor, _ptr(st)), "kyb_group_verify_batch", self.lib)
        return st
