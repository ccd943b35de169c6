"""Builds and runs tests/cpp/test_reference_style.cpp: the reference's own tests re-expressed on the
C++ mirror of its Rust surface (myzkp_amd/host/myzkp.hpp), through the C ABI, on the GPU."""
import os, subprocess
import pytest
import orc

ROOT = orc.ROOT
EXE = os.path.join(ROOT, "tests", "cpp", "test_reference_style")


def build_exe():
    orc.lib()
    src = os.path.join(ROOT, "tests", "cpp", "test_reference_style.cpp")
    cmd = ["g++", "-O1", "-std=c++17", src, "-o", EXE,
           "-L" + os.path.join(ROOT, "myzkp_amd"), "-lmzk_hip", "-L" + os.path.join(ROOT, "oracle"), "-lmzk_oracle",
           "-Wl,-rpath," + os.path.join(ROOT, "myzkp_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_cpp_mirror_compiles():
    """CPU: the mirror header and the reference-style test compile and link against the ABI."""
    import myzkp_amd.build as b
    b.build()
    build_exe()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_mirror_runs_on_gpu():
    build_exe()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all reference-style tests passed" in out.stdout
