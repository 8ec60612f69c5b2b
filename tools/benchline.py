#!/usr/bin/env python3
"""print the figures of bench.py lines: tools/benchline.py file.json ..."""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("kernels_ms"), "frac", d["roofline"]["frac"], "par",
              d.get("parity", {}).get("max_channel_rel_rms_vs_cpu"), "blk", d.get("block_mode", {}).get("ms_per_block"))
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
        try:
            print(open(f.replace(".json", ".err")).read()[-1500:])
        except OSError:
            pass
