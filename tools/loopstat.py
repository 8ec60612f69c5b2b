#!/usr/bin/env python3
"""tools/loopstat.py FILE.s START END — instruction classes per basic block (blocks with MFMAs) of an assembly range"""
import re, sys, collections
lines = open(sys.argv[1]).read().splitlines()[int(sys.argv[2]):int(sys.argv[3])]
blocks, cur = [], ["entry", collections.Counter()]
for ln in lines:
    t = ln.strip()
    if re.match(r"^\.LBB\S+:", t):
        blocks.append(cur); cur = [t.split(":")[0], collections.Counter()]; continue
    if not t or t.startswith((";", ".", "//")): continue
    op = t.split()[0]
    if op.startswith("v_mfma"): k = "mfma"
    elif op.startswith("v_pk_"): k = "vpk"
    elif op.startswith("v_cvt"): k = "vcvt"
    elif op.startswith("v_"): k = "valu"
    elif op.startswith("ds_"): k = "ds"
    elif op.startswith(("global_load", "buffer_load")): k = "vmld"
    elif op.startswith(("global_store", "buffer_store", "scratch")): k = "vmst/scr"
    elif op.startswith("s_waitcnt"): k = "wait"
    elif op.startswith("s_barrier"): k = "barrier"
    elif op.startswith("s_load"): k = "sld"
    elif op.startswith("s_"): k = "salu"
    else: k = "other"
    cur[1][k] += 1
    cur[1]["op:" + op] += 1
blocks.append(cur)
for name, c in blocks:
    if c["mfma"] >= 8:
        tot = sum(v for k, v in c.items() if not k.startswith("op:"))
        print(name, "total", tot, {k: v for k, v in c.items() if not k.startswith("op:")})
        if len(sys.argv) > 4:
            print("   ", sorted(((v, k[3:]) for k, v in c.items() if k.startswith("op:")), reverse=True)[:40])
