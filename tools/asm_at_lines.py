"""Print the instructions of a kernel whose OUTERMOST call site (in the kernel body) falls into a source line range.
    python tools/asm_at_lines.py <file.s> <kernel substring> <first> <last> [file substring]
Needs a -gline-tables-only build (tools/eval_kernel_static.sh)."""
import re, sys
path, kern, lo, hi = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
s = open(path).read()
m = re.search(r"^(\S*" + re.escape(kern) + r"\S*):", s, re.M)
body = s[m.start():s.index(".end_amdhsa_kernel", m.start())]
cur = None
n = 0
for ln in body.split("\n"):
    t = ln.strip()
    mm = re.match(r"\.loc\s+(\d+)\s+(\d+)\s+(\d+)(.*)", t)
    if mm:
        cur = int(mm.group(2))
        # "inlined_at" chains are not in .loc; the outermost statement is approximated by the line itself when file == kernel header
        continue
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    if cur is not None and lo <= cur <= hi:
        print(f"{cur:5d}  {t.split(';')[0].rstrip()}")
        n += 1
print(n, "instructions")
