"""Where do the scratch (spill) instructions of a kernel sit?  Per basic block of the saved assembly: number of instructions, MFMAs,
LDS-DMA loads, calls and scratch loads / stores -- tells spills on a cold path (around a call) from spills in the hot loop.
    python tools/asm_scratch_map.py <file.s> <kernel name substring>"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2]
for m in re.finditer(r"^(_ZN3bts\w+):\s*; @", s, re.M):
    name = m.group(1)
    if flt not in name:
        continue
    body = s[m.end():]
    body = body[:body.index(".Lfunc_end")]
    blocks, cur = [], ["entry", []]
    for l in body.split("\n"):
        t = l.strip()
        lab = re.match(r"^(\.LBB\d+_\d+):", t)
        if lab:
            blocks.append(cur)
            cur = [lab.group(1), []]
        elif t and not t.startswith(";") and not t.startswith("."):
            cur[1].append(t)
    blocks.append(cur)
    print(name)
    tot = 0
    for lab, ins in blocks:
        sc = sum("scratch_" in i for i in ins)
        tot += sc
        if len(ins) > 150 or sc:
            print(f"  {lab:12s} instr {len(ins):5d}  mfma {sum('v_mfma' in i for i in ins):3d}  lds-dma {sum('global_load_lds' in i for i in ins):3d}  "
                  f"call {sum('s_swappc' in i for i in ins):2d}  scratch ld {sum('scratch_load' in i for i in ins):4d} st {sum('scratch_store' in i for i in ins):4d}  "
                  f"flat {sum(i.startswith('flat_') for i in ins):3d}  readlane {sum('v_readlane' in i for i in ins):4d} writelane {sum('v_writelane' in i for i in ins):4d}")
    print("  total scratch instructions:", tot)
