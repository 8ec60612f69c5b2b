"""CPU-side checks of the drop-in boundary: libearhip.so builds for gfx950, loads without a GPU,
exports every symbol include/earhip.h declares, and fails loudly (no CPU fallback) when asked to
do work without a device."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "earhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(earhip_[a-z0-9_]+)\s*\(", text)) - {"earhip_process_func"})


def test_header_declares_the_expected_groups():
    syms = declared_symbols()
    for must in ("earhip_interp_apply_interp", "earhip_interp_apply_constant", "earhip_conv_process",
                 "earhip_conv_crossfade_filter", "earhip_delay_process", "earhip_vbs_process",
                 "earhip_render_process_device", "earhip_fft_forward"):
        assert must in syms
    assert len(syms) >= 40


def test_library_builds_and_exports_every_declared_symbol():
    from libear_amd import build, lib_path
    build()
    lib = C.CDLL(lib_path())
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.earhip_version.restype = C.c_int
    assert lib.earhip_version() == 100


def test_only_the_c_abi_is_exported():
    from libear_amd import lib_path
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path()], stdout=subprocess.PIPE, text=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert names and all(n.startswith("earhip_") for n in names), [n for n in names if not n.startswith("earhip_")]


def test_no_device_means_loud_failure_not_fallback():
    from libear_amd import capi
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.InternalError) as e:
        capi.Context(0)
    assert "no" in str(e.value).lower()


def test_host_side_adapter_runs_without_a_gpu():
    """VariableBlockSizeAdapter is pure host logic (reference tests/variable_block_size_tests.cpp)."""
    import numpy as np
    from libear_amd import capi

    def toy(x):
        return np.stack([x[0] * 2.0, x[1] * 3.0, x[0] * 4.0, x[1] * 5.0]).astype(np.float32)

    B, sizes = 512, [0, 512, 1024, 300, 500]
    total = sum(sizes)
    x = np.random.default_rng(2).uniform(-1, 1, (2, total)).astype(np.float32)
    want = np.zeros((4, total), np.float32)
    want[:, B:] = toy(x[:, :total - B])
    ad = capi.VariableBlockSizeAdapter(B, 2, 4, toy)
    assert ad.get_delay() == B
    out = np.full((4, total), np.nan, np.float32)
    ofs = 0
    for n in sizes:
        out[:, ofs:ofs + n] = ad.process(x[:, ofs:ofs + n])
        ofs += n
    assert np.array_equal(out, want)


def test_product_never_touches_the_oracle():
    """nothing under libear_amd/ or include/ may reference oracle/ (the oracle is the checker)."""
    bad = []
    for base in ("libear_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", "Makefile", ".map")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"oracle", txt, flags=re.I):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_fft_pass_algebra_on_cpu(tmp_path):
    exe = tmp_path / "test_fft_passes"
    src = os.path.join(ROOT, "tests", "cpp", "test_fft_passes.cpp")
    subprocess.run(["g++", "-std=c++17", "-O2", src, "-o", str(exe)], check=True)
    res = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True)
    assert res.returncode == 0, res.stdout


def test_segment_search_on_cpu(tmp_path):
    """libear_amd/csrc/search.h: the guess-started search == plain upper bound (libear's find_block)"""
    exe = tmp_path / "test_search"
    src = os.path.join(ROOT, "tests", "cpp", "test_search.cpp")
    subprocess.run(["g++", "-std=c++14", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "libear_amd", "csrc"),
                    src, "-o", str(exe)], check=True)
    res = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True)
    assert res.returncode == 0, res.stdout


def test_staging_path_helpers_on_cpu(tmp_path):
    """libear_amd/csrc/host_gather.h (the host side of long host-pointer calls): the streaming-store copy == memcpy for every size
    and misalignment, the staging threads' default count within its bounds, the NUMA helpers answer or decline — under ASan + UBSan"""
    exe = tmp_path / "test_host_gather"
    src = os.path.join(ROOT, "tests", "cpp", "test_host_gather.cpp")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Werror", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I", os.path.join(ROOT, "libear_amd", "csrc"), src, "-o", str(exe), "-lpthread"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    env.pop("LD_PRELOAD", None)
    res = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    assert res.returncode == 0, res.stdout


def test_launch_plans_fit_the_bus_buffer_on_cpu(tmp_path):
    """libear_amd/csrc/curves.h: every plan plan_mix makes fits bus_samples_bound() (host code, hipcc)"""
    exe = tmp_path / "test_plan_bounds"
    src = os.path.join(ROOT, "tests", "cpp", "test_plan_bounds.cpp")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-I",
                    os.path.join(ROOT, "libear_amd", "csrc"), src, "-o", str(exe)], check=True)
    res = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True)
    assert res.returncode == 0, res.stdout


def test_the_environment_is_read_once_at_context_creation():
    """SURVEY 5 / include/earhip.h (earhip_ctx_set_option): tuning knobs are options of the context; the only getenv of
    the library is the loop in earhip_ctx_create — nothing on a process call's path reads the environment."""
    sites = []
    base = os.path.join(ROOT, "libear_amd", "csrc")
    for f in sorted(os.listdir(base)):
        if f.endswith((".h", ".hip", ".cpp")):
            for i, ln in enumerate(open(os.path.join(base, f), errors="ignore"), 1):
                if re.search(r"\bgetenv\s*\(", ln):
                    sites.append(f"{f}:{i}")
    assert len(sites) == 1 and sites[0].startswith("api_core.hip:"), sites
    text = open(os.path.join(base, "api_core.hip")).read()
    create = text[text.index("int earhip_ctx_create("):text.index("int earhip_ctx_set_option(")]
    assert "getenv" in create
