"""The deferred-point evaluator (kyber-rs_amd/csrc/defer.inc: graph walking, Horner / sum chain recognition, batching, arena bookkeeping) compiled
for the CPU — the engine's batch entry points answered by the oracle — under AddressSanitizer and UBSan: random graphs (fusion on and off), fused
chains counted, stale handles, four threads on one arena.  tests/hostcheck/defer_host.cpp; CPU only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_deferred_evaluator_on_the_cpu_under_sanitizers(oracle):
    src = os.path.join(ROOT, "tests", "hostcheck", "defer_host.cpp")
    out = os.path.join(ROOT, "tests", "hostcheck", "_build", "defer_host")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    orc_dir = os.path.join(ROOT, "oracle", "_build")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wno-unused-function", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-o", out, src, "-L", orc_dir, "-loracle", f"-Wl,-rpath,{orc_dir}", "-lpthread"])
    r = subprocess.run([out], capture_output=True, text=True, timeout=900, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    print(r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("OK")


def test_deferred_evaluator_under_thread_sanitizer(oracle):
    """the same program under ThreadSanitizer: four threads record into and ask of one arena, a handle recorded through one context is evaluated
    through another — the arena's lock discipline after its rewrite (round 5: lock-free fast path to the context's own arena)"""
    src = os.path.join(ROOT, "tests", "hostcheck", "defer_host.cpp")
    out = os.path.join(ROOT, "tests", "hostcheck", "_build", "defer_host_tsan")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    orc_dir = os.path.join(ROOT, "oracle", "_build")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wno-unused-function", "-fsanitize=thread", "-o", out, src, "-L", orc_dir, "-loracle", f"-Wl,-rpath,{orc_dir}", "-lpthread"])
    r = subprocess.run([out], capture_output=True, text=True, timeout=900)
    print(r.stdout[-2000:], r.stderr[-3000:])
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr and r.stdout.strip().splitlines()[-1].startswith("OK")
