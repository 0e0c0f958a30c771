"""Scans gfx950 assembly (hipcc -save-temps *.s) for MFMA instructions whose destination tuple overlaps their A or B source
registers.  hipcc (ROCm 7.2) allocates such overlaps when srcC is the inline constant 0 (a fresh accumulator); on MI355X the result
is then wrong in the last-written lanes, timing-dependently (found with the rotated render kernel: sigma of lanes 48-63 differed
run to run).   usage: python tools/check_mfma_overlap.py file.s [...]"""
import re, sys

def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return m.group(1), {int(m.group(2))}
    return None, set()

bad = 0
for path in sys.argv[1:]:
    kernel = "?"
    for ln, line in enumerate(open(path), 1):
        if line.startswith("_Z") and line.rstrip().endswith(":") or re.match(r"^_Z\w+:", line):
            kernel = line.split(":")[0]
        s = line.strip()
        if not s.startswith("v_mfma"):
            continue
        ops = [t for t in re.split(r",\s*", s.split(None, 1)[1])]
        if len(ops) < 4:
            continue
        dk, d = regs(ops[0]); ak, a = regs(ops[1]); bk, b = regs(ops[2])
        if (dk == ak and d & a) or (dk == bk and d & b):
            bad += 1
            print(f"{path}:{ln}: [{kernel[:60]}] {s}")
print(f"{bad} MFMA(s) with dst overlapping srcA/srcB")
sys.exit(1 if bad else 0)
