"""Evidence hygiene: what bench.py attaches to `roofline.traffic` must be traceable.  Every entry of
profiles/traffic.json names a file that is tracked by git and holds, for the kernel the entry names, the counters the
entry's bytes were computed from — with the values the entry quotes."""
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEEDED = ("TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_64B_sum",
          "TCC_EA0_WRREQ_sum")


def tracked_files():
    out = subprocess.run(["git", "ls-files"], cwd=ROOT, stdout=subprocess.PIPE, text=True, check=True).stdout
    return set(out.split())


def counters_of(path, kernel):
    found = {}
    for ln in open(path):
        if not ln.startswith("earhip::" + kernel + " "):
            continue
        parts = ln[len("earhip::" + kernel):].split()
        if len(parts) >= 3 and re.match(r"^[A-Za-z_0-9]+$", parts[0]):
            try:
                found[parts[0]] = float(parts[2])
            except ValueError:
                pass
    return found


def test_every_traffic_entry_cites_a_tracked_file_that_holds_its_counters():
    doc = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    tracked = tracked_files()
    assert doc["entries"], "traffic.json has no entries"
    for e in doc["entries"]:
        src = e["source"].split(" ")[0]
        assert src in tracked, f"{src} (cited by the {e['kernel']} / {e['scene']} entry) is not a tracked file"
        got = counters_of(os.path.join(ROOT, src), e["kernel"])
        for name in NEEDED:
            assert name in got, f"{src} has no {name} line for {e['kernel']}"
            assert abs(got[name] - e["raw_per_launch"][name]) <= 1e-6 * max(1.0, abs(got[name])), (src, name)
        rd = 128 * got["TCC_EA0_RDREQ_128B_sum"] + 64 * got["TCC_EA0_RDREQ_64B_sum"] + 32 * got["TCC_EA0_RDREQ_32B_sum"]
        wr = 64 * got["TCC_EA0_WRREQ_64B_sum"] + 32 * (got["TCC_EA0_WRREQ_sum"] - got["TCC_EA0_WRREQ_64B_sum"])
        assert abs(rd + wr - e["gain_mix_hbm_bytes_per_step"]) <= 2, (src, rd + wr, e["gain_mix_hbm_bytes_per_step"])
        # the kernel-trace pass of the same file times the same kernel
        text = open(os.path.join(ROOT, src)).read()
        assert re.search(r"earhip::" + re.escape(e["kernel"]) + r"\s+\(\d+, \d+, \d+\)\s+\d+\s+\d+\s+[\d.]+", text), \
            f"{src} has no per-launch-shape timing line for {e['kernel']}"


def test_bench_lines_of_the_round_are_tracked_and_parse():
    tracked = tracked_files()
    lines = sorted(f for f in tracked if re.match(r"profiles/r06_bench_.*\.json$", f))
    assert len(lines) >= 10, lines
    for f in lines:
        d = json.loads(open(os.path.join(ROOT, f)).read().strip().splitlines()[-1])
        assert d["value"] > 0 and d["unit"] == "Msamples/s"
        if "parity" in d and d["parity"] and "pass" in d["parity"]:
            assert d["parity"]["pass"], f


def test_default_line_of_the_round_carries_the_other_workloads():
    """profiles/r06_bench_default.json: the default `python bench.py` line as the driver runs it — headline + the
    `secondary` array (configs 2, 3, 5, the scenes that take the other gain kernels, a rank's share at 8 GPUs), every entry
    parity-gated on its own timed buffer"""
    d = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_default.json")).read().strip().splitlines()[-1])
    assert d["config"]["baseline_config"] == "C4" and d["parity"]["pass"]
    assert d["parity"]["gpu_rel_rms_vs_float64"] <= d["parity"]["cpu_rel_rms_vs_float64"]
    sec = {s["workload"]: s for s in d["secondary"]}
    assert len(sec) >= 8, sorted(sec)
    for name, s in sec.items():
        assert "error" not in s, (name, s.get("error"))
        assert s["value"] > 0 and s["parity"]["pass"] and s["parity"]["max_channel_rel_rms_vs_cpu"] <= 1e-6, name
