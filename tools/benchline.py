#!/usr/bin/env python3
"""print the figures of bench.py lines: tools/benchline.py file.json ...   (or lines on stdin)"""
import json
import sys


def show(label, text):
    for ln in text.splitlines():
        ln = ln.strip()
        if not ln.startswith("{"):
            continue
        d = json.loads(ln)
        print(label, d["value"], d["ms_per_step"], d.get("kernels_ms"), "frac", d.get("roofline", {}).get("frac"), "par",
              d.get("parity", {}).get("max_channel_rel_rms_vs_cpu"), "blk", d.get("block_mode", {}).get("ms_per_block"))


if len(sys.argv) == 1:
    show("", sys.stdin.read())
for f in sys.argv[1:]:
    try:
        show(f.split("/")[-1], open(f).read())
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
        try:
            print(open(f.replace(".json", ".err")).read()[-1500:])
        except OSError:
            pass
