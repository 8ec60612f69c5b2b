import sys,json
for line in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    line=line.strip()
    if line.startswith("{"):
        d=json.loads(line); print(d["value"], d["ms_per_step"], d["kernels_ms"], d["whole_path"].get("gain_fp32_equivalent_tflops"), d.get("parity",{}).get("rel_rms_vs_cpu"))
