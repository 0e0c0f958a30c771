"""Sketch of a kernel's memory behaviour, block by block: L = global load, D = LDS-DMA load, S = global store, A = global atomic, M = MFMA,
wN = s_waitcnt vmcnt(N); blocks without any of them are dropped.  Shows at a glance whether the loads of a tile are issued up front or
sunk to their uses, and where the compiler falls back to vmcnt(0) (behind control-flow merges).
    python tools/asm_mem_sketch.py <file.s> <kernel name substring> [max chars per block]"""
import re, sys
s = open(sys.argv[1]).read()
width = int(sys.argv[3]) if len(sys.argv) > 3 else 220
for m in re.finditer(r"^(\S*" + re.escape(sys.argv[2]) + r"\S*):", s, re.M):
    k = m.group(1)
    body = s[m.start():s.index(".end_amdhsa_kernel", m.start())]
    print("==", k[:120])
    cur, out = "", []
    for l in body.split("\n"):
        t = l.strip()
        if not t or t.startswith((";", ".loc", ".cfi", ".p2")):
            continue
        if t.endswith(":") or t.startswith(".LBB"):
            if re.search(r"[LDSAMw]", cur.split(": ", 1)[-1]):
                out.append(cur[:width])
            cur = t.split(":")[0] + ": "
            continue
        if t.startswith("."):
            continue
        op = t.split()[0]
        if op.startswith("global_load_lds"): cur += "D"
        elif op.startswith("global_load"): cur += "L"
        elif op.startswith("global_store"): cur += "S"
        elif op.startswith("global_atomic"): cur += "A"
        elif op.startswith("v_mfma"): cur += "M"
        elif op.startswith("s_waitcnt") and "vmcnt" in t: cur += " w" + re.search(r"vmcnt\((\d+)\)", t).group(1) + " "
    if re.search(r"[LDSAMw]", cur.split(": ", 1)[-1]):
        out.append(cur[:width])
    print("\n".join(out))
