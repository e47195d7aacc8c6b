"""PCIe-inclusive throughput of the host-pointer API (numpy buffers in pageable host memory)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import kyber_rs_amd, oracle_lib
eng = kyber_rs_amd.Engine(0); orc = oracle_lib.Oracle()
n = 1 << 20
rng = np.random.default_rng(1)
s = rng.integers(0, 256, (n, 32), dtype=np.uint8); s[:, 31] &= 0x0f
enc, ext = eng.mul_base(s, want_ext=True)
threads = [int(v[1:]) for v in sys.argv[1:] if v.startswith("t")] or [0]
chunk_options = [int(v) for v in sys.argv[1:] if not v.startswith("t")] or [eng.get_option("host.pipe_chunks")]
for ch, th in [(c_, t_) for c_ in chunk_options for t_ in threads]:
    eng.set_option("host.pipe_chunks", ch)
    eng.set_option("host.copy_threads", th)
    print(f"-- host.pipe_chunks {ch}, host.copy_threads {th} (0 = auto)")
    for name, fn in (("mul_base", lambda: eng.mul_base(s)), ("mul(ext in)", lambda: eng.mul(s, pts_ext=ext)), ("mul(enc in)", lambda: eng.mul(s, pts_enc=enc))):
        fn(); t = time.perf_counter(); 
        for _ in range(3): out = fn()
        dt = (time.perf_counter() - t) / 3
        print(f"{name:12s} n=2^20 host-pointer API, {ch} chunks: {dt*1e3:8.2f} ms  -> {n/dt:.3e} items/s")
eng.set_option("host.pipe_chunks", 8); eng.set_option("host.copy_threads", 0)
# the same with batch buffers in pinned memory (kyb_host_alloc)
ps = eng.pinned_array((n, 32), np.uint8); ps[:] = s
pe = eng.pinned_array((n, 40), np.int32); pe[:] = ext
po = eng.pinned_array((n, 32), np.uint8)
for name, fn in (("mul_base", lambda: eng.mul_base_into(ps, po)), ("mul(ext in)", lambda: eng.mul_into(ps, pe, po))):
    fn(); t = time.perf_counter()
    for _ in range(3): fn()
    dt = (time.perf_counter() - t) / 3
    print(f"{name:12s} n=2^20 host-pointer API, pinned buffers: {dt*1e3:8.2f} ms  -> {n/dt:.3e} items/s")
idx = rng.choice(n, 512, replace=False)
assert np.array_equal(po[idx], orc.mul_batch(s[idx], ext[idx], nthreads=8))
assert np.array_equal(eng.mul(s, pts_ext=ext)[idx], orc.mul_batch(s[idx], ext[idx], nthreads=8))
assert np.array_equal(eng.mul_base(s)[idx], orc.mul_base_batch(s[idx], nthreads=8))
print("parity ok")
