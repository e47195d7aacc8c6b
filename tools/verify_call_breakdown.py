"""One synchronous verification call of 1,024 / 2,048 / 2,560 signatures: wall time and the kernels inside it, through the raw C ABI (ctypes), so
that any build of the library can be measured, also older ones:   python tools/verify_call_breakdown.py path/to/lib.so"""
import ctypes, os, sys, time
import numpy as np, torch
lib = ctypes.CDLL(sys.argv[1])
vp, sz = ctypes.c_void_p, ctypes.c_size_t
lib.kyb_init.argtypes = [ctypes.c_int]
lib.kyb_mul_base_batch_dev.argtypes = [vp, sz, vp, vp, vp]
lib.kyb_schnorr_sign_batch_dev.argtypes = [vp, vp, vp, vp, sz, vp, vp]
lib.kyb_verify_batch_dev.argtypes = [vp, vp, vp, vp, sz, ctypes.c_int, vp, vp]
lib.kyb_profile_begin.argtypes = [ctypes.c_int]
lib.kyb_profile_read.argtypes = [vp, vp, ctypes.c_int, vp]
lib.kyb_kernel_name.restype = ctypes.c_char_p; lib.kyb_kernel_name.argtypes = [ctypes.c_int]
assert lib.kyb_init(0) == 0
dev = torch.device("cuda", 0); N = 4096
rng = np.random.default_rng(11)
sc_np = rng.integers(0, 256, size=(N, 32), dtype=np.uint8); sc_np[:, 31] &= 0x0f
sc = torch.from_numpy(sc_np).to(dev); k = torch.from_numpy(np.roll(sc_np, 1, axis=0).copy()).to(dev)
msgs = torch.from_numpy(rng.integers(0, 256, size=(32 * N,), dtype=np.uint8)).to(dev)
off = torch.arange(0, 32 * (N + 1), 32, dtype=torch.int32, device=dev)
sigs = torch.empty((N, 64), dtype=torch.uint8, device=dev); pubs = torch.empty((N, 32), dtype=torch.uint8, device=dev)
out = torch.empty((N,), dtype=torch.uint8, device=dev)
p = lambda t: vp(t.data_ptr())
assert lib.kyb_schnorr_sign_batch_dev(p(sc), p(k), p(msgs), p(off), N, p(sigs), None) == 0
assert lib.kyb_mul_base_batch_dev(p(sc), N, p(pubs), None, None) == 0
torch.cuda.synchronize()
for n in (1024, 2048, 2560):
    fn = lambda: lib.kyb_verify_batch_dev(p(pubs), p(msgs), p(off), p(sigs), n, 1, p(out), None)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        a = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - a)
    lib.kyb_profile_begin(16); fn(); torch.cuda.synchronize()
    ids = (ctypes.c_int * 16)(); ms = (ctypes.c_float * 16)(); cnt = ctypes.c_int(0)
    lib.kyb_profile_read(ids, ms, 16, ctypes.byref(cnt)); lib.kyb_profile_begin(0)
    print(os.path.basename(sys.argv[1]), n, f"{sorted(ts)[15]*1e3:.3f} ms", " ".join(f"{lib.kyb_kernel_name(ids[i]).decode()}={ms[i]:.3f}" for i in range(cnt.value)), flush=True)

# the same with eight calls queued back to back (no synchronisation between them): per-kernel means
for n in (2048,):
    fn = lambda: lib.kyb_verify_batch_dev(p(pubs), p(msgs), p(off), p(sigs), n, 1, p(out), None)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    lib.kyb_profile_begin(64)
    t0 = time.perf_counter()
    for _ in range(8): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    ids = (ctypes.c_int * 64)(); ms = (ctypes.c_float * 64)(); cnt = ctypes.c_int(0)
    lib.kyb_profile_read(ids, ms, 64, ctypes.byref(cnt)); lib.kyb_profile_begin(0)
    agg = {}
    for i in range(cnt.value):
        agg.setdefault(lib.kyb_kernel_name(ids[i]).decode(), []).append(ms[i])
    print(os.path.basename(sys.argv[1]), n, f"queued back to back: {dt*1e3:.3f} ms per call;", " ".join(f"{k_}={sum(v)/len(v):.3f}" for k_, v in agg.items()), flush=True)
