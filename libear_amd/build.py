"""Builds libearhip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    """the library the package loads; EARHIP_LIB names another build of it (A/B runs of kernel variants)"""
    return os.environ.get("EARHIP_LIB") or os.path.join(HERE, "lib", "libearhip.so")


def build(verbose=False):
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), "-j4"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libearhip.so failed")
    assert os.path.exists(lib_path())
    return lib_path()
