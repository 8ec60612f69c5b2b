"""Register / scratch / LDS use of the kernels of one translation unit, as the compiler reports them:
    python tools/kernel_resources.py libear_amd/csrc/api_render.hip [filter] [-DNAME=VALUE ...]
(hipcc --save-temps into a temporary directory; reads the .amdgpu_metadata of the gfx950 assembly)"""
import os, re, subprocess, sys, tempfile
src = os.path.abspath(sys.argv[1])
filt = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
defs = [a for a in sys.argv[2:] if a.startswith("-")]
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-x", "hip", "-c", src,
                    "-o", "a.o", "--save-temps"] + defs, cwd=d, check=True, stderr=subprocess.DEVNULL)
    asm = next(f for f in os.listdir(d) if f.endswith("gfx950.s"))
    text = open(os.path.join(d, asm)).read()
meta = text[text.index(".amdgpu_metadata"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = get("name")
    demangled = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip().split("(")[0]
    if filt and filt not in demangled:
        continue
    print(f"{demangled[:70]:70s} vgpr {get('vgpr_count'):>4s} spill {get('vgpr_spill_count'):>3s} sgpr {get('sgpr_count'):>4s} sspill {get('sgpr_spill_count'):>3s} "
          f"scratch {get('private_segment_fixed_size'):>5s} lds {get('group_segment_fixed_size'):>6s}")
