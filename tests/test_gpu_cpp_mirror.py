"""Builds and runs tests/cpp/test_reference_style.cpp: the reference's own tests re-expressed on the
C++ mirror of its Rust surface (myzkp_amd/host/myzkp.hpp), through the C ABI, on the GPU."""
import os, subprocess
import pytest
import orc

ROOT = orc.ROOT
EXE = os.path.join(ROOT, "tests", "cpp", "test_reference_style")


EXE_MULTI = os.path.join(ROOT, "tests", "cpp", "test_multi_device")


def build_exe(name="test_reference_style"):
    orc.lib()
    src = os.path.join(ROOT, "tests", "cpp", name + ".cpp")
    cmd = ["g++", "-O1", "-std=c++17", src, "-o", os.path.join(ROOT, "tests", "cpp", name),
           "-L" + os.path.join(ROOT, "myzkp_amd"), "-lmzk_hip", "-L" + os.path.join(ROOT, "oracle"), "-lmzk_oracle",
           "-Wl,-rpath," + os.path.join(ROOT, "myzkp_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_cpp_mirror_compiles():
    """CPU: the mirror header and the reference-style test compile and link against the ABI."""
    import myzkp_amd.build as b
    b.build()
    build_exe()
    build_exe("test_multi_device")
    assert os.path.exists(EXE) and os.path.exists(EXE_MULTI)


@pytest.mark.gpu
def test_cpp_mirror_runs_on_gpu():
    build_exe()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all reference-style tests passed" in out.stdout


@pytest.mark.gpu
def test_cpp_multi_device_runs_world_1_to_8():
    """Multi-GPU MSM / KZG commit through the C ABI from a plain C++ program (no Python, no torch in the process)."""
    build_exe("test_multi_device")
    # count devices without initialising the GPU in this process (torch.cuda.device_count() does not)
    import torch
    out = subprocess.run([EXE_MULTI, str(max(torch.cuda.device_count(), 1))], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "multi-device C ABI test passed" in out.stdout
