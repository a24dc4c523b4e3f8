"""The C++ host-side mirror (include/uhdr_hip.hpp: a class shaped like ultrahdr::UltraHdr over the C ABI).
CPU: it compiles with plain g++ against the public headers, links the product library, and refuses to work
without a device.  GPU: the C++ program runs every stage operator and compares with the C oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "mirror_check.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "mirror_check")
LIBDIR = os.path.join(ROOT, "libultrahdr_amd", "lib")
ORACLE = os.path.join(ROOT, "oracle")


def _build():
    if os.path.exists(BIN) and os.path.getmtime(BIN) >= max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(ROOT, "include", "uhdr_hip.hpp"))):
        return
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", ORACLE, SRC, "-o", BIN,
           "-L", LIBDIR, "-luhdr_hip", "-L", ORACLE, "-luhdr_oracle", "-L", "/opt/rocm/lib", "-lamdhip64",
           f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{ORACLE}", "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def _run(*args):
    env = dict(os.environ, LD_LIBRARY_PATH=os.pathsep.join([LIBDIR, ORACLE, "/opt/rocm/lib", os.environ.get("LD_LIBRARY_PATH", "")]))
    return subprocess.run([BIN, *args], capture_output=True, text=True, env=env, timeout=300)


def test_cpp_mirror_builds_with_plain_gxx_and_has_no_cpu_fallback():
    _build()
    r = _run("--no-gpu")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "no CPU fallback" in r.stdout or "a device is present" in r.stdout


@pytest.mark.gpu
def test_cpp_mirror_matches_the_oracle():
    _build()
    r = _run()
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ALL CHECKS PASSED" in r.stdout
