"""AddressSanitizer + UndefinedBehaviorSanitizer builds of the CPU-side code (the reference has no race or
memory checking of its own; GPU sanitizers are not available on this pool, so this is the CPU half): the oracle's C
restatement behind a driver that calls every entry point (oracle/san_driver.c), and the host Chain with its iterators
(tests/cpp/chain_test.cpp, including the block-prefetch thread and the InnerBenchmark-sized chain)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def test_oracle_under_address_and_ub_sanitizers():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ROOT, "oracle", "_san", "san_driver")], capture_output=True, text=True, env=ENV, timeout=600)
    assert out.returncode == 0 and "san_driver OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_host_chain_under_address_and_ub_sanitizers():
    build = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "chain_test_san")
    src = os.path.join(ROOT, "tests", "cpp", "chain_test.cpp")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include", "MCMCpp"), src, "-o", exe, "-pthread"])
    out = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=900)
    assert out.returncode == 0 and "chain_test OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
