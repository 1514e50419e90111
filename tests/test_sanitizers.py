"""AddressSanitizer + UndefinedBehaviorSanitizer over everything of this repository that runs on the host
(CPU build only -- GPU sanitizers are not available on the pool): the oracle's C
restatement, the kernel's per-lane math compiled for the host, and the N-GPU host protocol of
mojo_simdjson_amd/csrc/sharded.cpp.  One program, tests/cpp/sanitize_host.cpp, built here with
-fsanitize=address,undefined -fno-sanitize-recover=all and run on the reference's fixtures and on fuzz inputs."""
import glob
import os
import subprocess

from tests import helpers

SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-O1", "-g"]


def _build():
    out = os.path.join(helpers.ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    hip = ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]  # types only: the runtime's entry points are stubbed
    objs = []
    for cc, std, src, extra in (("gcc", "-std=c11", "oracle/stage1_oracle.c", []),
                                ("g++", "-std=c++17", "tests/lane_math_host.cpp", []),
                                ("g++", "-std=c++17", "mojo_simdjson_amd/csrc/sharded.cpp", hip),
                                ("g++", "-std=c++17", "tests/cpp/sanitize_host.cpp", hip)):
        obj = os.path.join(out, "san_" + os.path.basename(src).rsplit(".", 1)[0] + ".o")
        subprocess.check_call([cc, std, "-Wall"] + SAN + extra + ["-c", os.path.join(helpers.ROOT, src), "-o", obj])
        objs.append(obj)
    exe = os.path.join(out, "sanitize_host")
    subprocess.check_call(["g++"] + SAN + objs + ["-o", exe, "-lpthread", "-ldl"])
    return exe


def test_oracle_asan():
    """(the name oracle/Makefile has always pointed at)  The oracle, the lane math and the sharded host protocol
    under ASan + UBSan: any report aborts the program (-fno-sanitize-recover), any mismatch exits non-zero."""
    exe = _build()
    fixtures = sorted(glob.glob(os.path.join(helpers.GOLDEN, "valid", "*.json")))
    assert len(fixtures) >= 14
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for seed in (1, 2, 3):
        r = subprocess.run([exe, str(seed), "300"] + fixtures, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
        assert "sanitize_host ok" in r.stdout
