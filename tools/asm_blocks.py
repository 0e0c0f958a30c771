"""Instruction mix per basic block of one kernel in a saved .s file.
    python tools/asm_blocks.py <file.s> <kernel name substring> [min instructions]"""
import re, sys
s = open(sys.argv[1]).read()
m = re.search(r"^(\S*" + re.escape(sys.argv[2]) + r"\S*):", s, re.M)
i = m.start()
j = s.index(".end_amdhsa_kernel", i)
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 25
blk, stats, order = "entry", {}, ["entry"]
stats[blk] = dict(n=0, spill=0, rl=0, ds=0, gl=0, valu=0, mfma=0, salu=0, wait=0)
for ln in s[i:j].split("\n"):
    mm = re.match(r"^(\.LBB\d+_\d+):", ln)
    if mm:
        blk = mm.group(1); order.append(blk)
        stats[blk] = dict(n=0, spill=0, rl=0, ds=0, gl=0, valu=0, mfma=0, salu=0, wait=0)
        continue
    t = ln.split()
    if not ln.startswith("\t") or not t or t[0].startswith((";", ".")):
        continue
    st, op = stats[blk], t[0]
    st["n"] += 1
    if "v_writelane" in op: st["spill"] += 1
    elif "v_readlane" in op: st["rl"] += 1
    elif op.startswith("ds_"): st["ds"] += 1
    elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): st["gl"] += 1
    elif op.startswith("v_mfma"): st["mfma"] += 1
    elif op.startswith("v_"): st["valu"] += 1
    elif op.startswith("s_waitcnt"): st["wait"] += 1
    elif op.startswith("s_"): st["salu"] += 1
print(m.group(1))
for b in order:
    if stats[b]["n"] >= mn:
        print(f"  {b:12s}", " ".join(f"{k}={v}" for k, v in stats[b].items()))
