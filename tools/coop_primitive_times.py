"""Cost of the cooperative primitives for ONE wavefront on an otherwise idle chip: dependent chains of each primitive run by
the library's test hook (k_coop_selftest ops 16..23), timed as the difference between two chain lengths."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import kyber_rs_amd
import coop_model as M

eng = kyber_rs_amd.Engine(0, crosscheck=True)          # the hook lives in the cross-check build only
lib = kyber_rs_amd.load_library(crosscheck=True)
eng.device_info()
c = M.lane_consts()
A = np.ascontiguousarray(M.quad_from_ints(c, [3, 5, 7, 11]), dtype=np.uint32)


def run(op, reps):
    B = np.ascontiguousarray(M.quad_from_ints(c, [13, 17, 19, 23]), dtype=np.uint32)
    B[3] = reps
    out = np.zeros(64, np.uint32)
    best = 1e9
    for _ in range(7):
        t = time.perf_counter()
        assert lib.kyb_diag_coop(op, A.ctypes.data, B.ctypes.data, out.ctypes.data) == 0
        best = min(best, time.perf_counter() - t)
    return best


names = {16: "cmul4", 17: "csq4", 18: "cnorm(+add)", 19: "ladder step", 20: "ds_bpermute round trip", 21: "dpp move + add", 22: "mixed addition", 23: "v_mul_lo + add"}
ghz = 2.35
print("primitive, ns, cycles_at_%.2f_GHz" % ghz)
for op in sorted(names):
    n1, n2 = 2000, 10000
    t1, t2 = run(op, n1), run(op, n2)
    ns = (t2 - t1) / (n2 - n1) * 1e9
    print(f"{names[op]}, {ns:.1f}, {ns * ghz:.0f}", flush=True)
