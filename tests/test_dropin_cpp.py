"""The C++14 mirror of libear's ear::dsp interfaces (libear_amd/host/ear/...), driven by a C++
program that restates the reference's own Catch2 tests (tests/cpp/test_dropin.cpp).
CPU suite: the headers compile as C++14 and link against libearhip.so, and the program fails
loudly without a GPU.  GPU suite: the program runs and every check passes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    from libear_amd import build as build_lib
    build_lib()
    exe = str(tmp_path / "test_dropin")
    libdir = os.path.join(ROOT, "libear_amd", "lib")
    cmd = ["g++", "-std=c++14", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "libear_amd", "host"), os.path.join(ROOT, "tests", "cpp", "test_dropin.cpp"),
           "-L" + libdir, "-learhip", "-ldl", "-Wl,-rpath," + libdir, "-o", exe]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    return exe


def test_mirror_headers_compile_as_cxx14_and_fail_loudly_without_gpu(tmp_path):
    exe = build(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the run is covered by the gpu test")
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode != 0
    assert "no" in res.stdout.lower() and "device" in res.stdout.lower()


@pytest.mark.gpu
def test_dropin_program_passes_on_gpu(tmp_path):
    exe = build(tmp_path)
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(res.stdout)
    assert res.returncode == 0, res.stdout
    assert " 0 failed" in res.stdout
