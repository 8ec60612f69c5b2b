#!/usr/bin/env python3
"""tools/make_traffic_json.py — rebuild the entries of profiles/traffic.json from the tracked rocprofv3 summaries
(tools/profile_passes.sh / tools/traffic_passes.sh outputs copied to profiles/).  tests/test_profiles.py checks the
result against the same files."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NEEDED = ("TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_64B_sum",
          "TCC_EA0_WRREQ_sum", "FETCH_SIZE", "WRITE_SIZE")
OPTIONAL = ("TCP_TCC_READ_REQ_sum",)
# (summary file, kernel, shape: objects, blocks, block size, channels, buses, scene, gain kernel id, tile)
ENTRIES = [
    ("r06_final_rocprofv3_summary.txt", "k_gain_mix_h2<3, 8, 2>", 1024, 1024, 512, 24, 2, "dense", 3, 512),
    ("r06_adm_scene_rocprofv3_summary.txt", "k_gain_mix_p2<3, 8, true>", 1024, 1024, 512, 24, 2, "adm", 4, 512),
    ("r06_moving_scene_rocprofv3_summary.txt", "k_gain_mix_hg<3, 8>", 1024, 1024, 512, 24, 2, "moving", 5, 512),
    ("r06_bursty_moving_scene_rocprofv3_summary.txt", "k_gain_mix_hg<3, 8>", 1024, 1024, 512, 24, 2, "bursty-moving", 5, 512),
    ("r06_traffic_C2_summary.txt", "k_gain_mix_h2<1, 8, 2>", 64, 8192, 512, 10, 1, "dense", 3, 512),
    ("r06_traffic_C3_summary.txt", "k_gain_mix_h2<3, 8, 2>", 256, 4096, 512, 24, 2, "dense", 3, 512),
    ("r06_traffic_C5_summary.txt", "k_gain_mix_h2<3, 8, 2>", 528, 512, 1024, 24, 2, "dense", 3, 512),
]


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


def algorithmic(m, n, b, k, t, m_static=0):
    import bench
    g = bench.algorithmic_bytes(m, n, b, k, m_static)
    return int((g[0] if isinstance(g, (tuple, list)) else g) * t)


def main():
    path = os.path.join(ROOT, "profiles", "traffic.json")
    doc = json.load(open(path))
    entries = []
    for src, kernel, m, t, b, n, k, scene, gk, tile in ENTRIES:
        c = counters_of(os.path.join(ROOT, "profiles", src), kernel)
        missing = [x for x in NEEDED if x not in c]
        if missing:
            raise SystemExit(f"{src}: no {missing} for {kernel}")
        rd = 128 * c["TCC_EA0_RDREQ_128B_sum"] + 64 * c["TCC_EA0_RDREQ_64B_sum"] + 32 * c["TCC_EA0_RDREQ_32B_sum"]
        wr = 64 * c["TCC_EA0_WRREQ_64B_sum"] + 32 * (c["TCC_EA0_WRREQ_sum"] - c["TCC_EA0_WRREQ_64B_sum"])
        hoa = 16 if m == 528 else 0  # (config 5: 512 objects + a 16-channel bed through a constant matrix)
        alg = algorithmic(m - hoa, n, b, k, t, hoa)
        raw = {x: c[x] for x in NEEDED}
        raw.update({x: c[x] for x in OPTIONAL if x in c})
        entries.append({
            "objects": m, "blocks": t, "block_size": b, "channels": n, "buses": k, "scene": scene, "gain_kernel": gk,
            "tile": tile, "kernel": kernel,
            "source": f"profiles/{src} (TCC_EA0_RDREQ_{{sum,32B,64B,128B}}, TCC_EA0_WRREQ_{{sum,64B}}, FETCH_SIZE, WRITE_SIZE)",
            "raw_per_launch": raw, "read_bytes": int(round(rd)), "write_bytes": int(round(wr)),
            "gain_mix_hbm_bytes_per_step": int(round(rd + wr)), "algorithmic_bytes_per_step": alg,
            "remark": f"{(rd + wr) / alg:.2f} x the algorithmic bytes; FETCH_SIZE x 1024 x 2 = {c['FETCH_SIZE'] * 2048 / 1e9:.3f} GB "
                      f"(the counter tallies 128-B requests at 64 B), WRITE_SIZE x 1024 = {c['WRITE_SIZE'] * 1024 / 1e6:.1f} MB"})
    doc["entries"] = entries
    json.dump(doc, open(path, "w"), indent=1)
    open(path, "a").write("\n")
    for e in entries:
        print(e["kernel"], e["scene"], e["objects"], e["remark"])


if __name__ == "__main__":
    main()
