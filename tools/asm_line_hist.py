"""Which source functions do the instructions of a kernel come from?  Static attribution from a line-table build:
    python -m behindthescenes_amd.build --tag lines -gline-tables-only
    python tools/asm_line_hist.py /tmp/bts_render_obj_lines/bts_fwd_proj-hip-amdgcn-amd-amdhsa-gfx950.s render_kernel_pILi64ELi64ELi0ELi1ELb1ELb1ELb0 [block label]
Every instruction carries the .loc of the innermost inlined function and the chain of call sites up to the kernel body; instructions
are counted (VALU / MFMA / LDS / VMEM / SALU) per innermost FUNCTION (looked up in the source by line) and per top-level statement of
the kernel body.  With a block label (e.g. the hot loop's main block from tools/asm_blocks.py) only that basic block is counted.
Static counts: a block that runs once per ray counts once, whatever its trip count."""
import collections, os, re, sys

path, kern = sys.argv[1], sys.argv[2]
only_block = sys.argv[3] if len(sys.argv) > 3 else None      # a block label, or "loop" = every block of the persistent loop
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(path).read()
m = re.search(r"^(\S*" + re.escape(kern) + r"\S*):", s, re.M)
body = s[m.start():s.index(".end_amdhsa_kernel", m.start())]

_fn_cache = {}
def functions_of(rel):
    if rel not in _fn_cache:
        out = []
        try:
            lines = open(os.path.join(ROOT, rel)).read().split("\n")
        except OSError:
            lines = []
        for i, l in enumerate(lines, 1):
            if not re.match(r"^\s*(?:static\s+|inline\s+)*(?:__device__|__global__|__host__)", l):
                continue
            l2 = re.sub(r"__launch_bounds__\([^)]*\)|__attribute__\(\([^)]*\)\)|__forceinline__", " ", l)
            mm = re.search(r"\b([A-Za-z_]\w*)\s*\(", l2)
            if mm and mm.group(1) not in ("if", "for", "while", "explicit"):
                out.append((i, mm.group(1)))
            elif re.search(r"explicit\s+(\w+)\s*\(", l2):
                out.append((i, re.search(r"explicit\s+(\w+)\s*\(", l2).group(1)))
        _fn_cache[rel] = out
    return _fn_cache[rel]

def fn_at(rel, line):
    best = "?"
    for i, name in functions_of(rel):
        if i <= line:
            best = name
        else:
            break
    return best

def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_readlane", "v_writelane")): return "lane"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"

inner = collections.defaultdict(collections.Counter)
outer = collections.defaultdict(collections.Counter)
cur_inner, cur_outer, blk = ("?", 0), ("?", 0), "entry"
loop_blocks = None
if only_block == "loop":
    # The persistent loop = the Depth-1 loop with the most instructions.  LLVM lays a loop out contiguously but annotates only some of
    # its blocks ("in Loop: Header=BBx_y Depth=1", inner loops "Parent Loop BBx_y Depth=1" / "Header=<inner> Depth=2"): take everything
    # from the header to the last annotated member, in file order.
    labels, parent = [], {}
    for i, ln in enumerate(body.split("\n")):
        mb = re.match(r"^\.L(BB\d+_\d+):(.*)", ln)
        if mb:
            labels.append((i, mb.group(1), mb.group(2)))
            mp = re.search(r"Parent Loop (BB\d+_\d+) Depth=1", mb.group(2))
            if mp:
                parent[mb.group(1)] = mp.group(1)
    span = {}
    for i, name, c in labels:
        hdr = None
        if "Loop Header: Depth=1" in c:
            hdr = name
        mh = re.search(r"Header=(BB\d+_\d+) Depth=(\d)", c)
        if mh:
            hdr = mh.group(1) if mh.group(2) == "1" else parent.get(mh.group(1))
        if name in parent:
            hdr = parent[name]
        if hdr:
            lo, hi = span.get(hdr, (i, i))
            span[hdr] = (min(lo, i), max(hi, i))
    best = max(span, key=lambda h: span[h][1] - span[h][0])
    lo, hi = span[best]
    nxt = [i for i, _, _ in labels if i > hi]
    hi_end = nxt[0] if nxt else 10 ** 9
    loop_blocks = {".L" + name for i, name, _ in labels if lo <= i < hi_end}
    print(f"persistent loop .L{best}: {len(loop_blocks)} blocks (assembly lines {lo} .. {hi_end})")
for ln in body.split("\n"):
    mb = re.match(r"^(\.LBB\d+_\d+):", ln)
    if mb:
        blk = mb.group(1)
        continue
    t = ln.strip()
    if t.startswith(".loc"):
        c = t.split(";", 1)[1] if ";" in t else ""
        locs = re.findall(r"([\w/\.]+\.(?:h|hip)):(\d+):\d+", c)
        if locs:
            cur_inner = (locs[0][0], int(locs[0][1]))
            cur_outer = (locs[-1][0], int(locs[-1][1]))
        continue
    if not ln.startswith("\t") or not t or t[0] in ".;":
        continue
    if loop_blocks is not None:
        if blk not in loop_blocks:
            continue
    elif only_block and blk != only_block:
        continue
    k = kind(t.split()[0])
    inner[fn_at(*cur_inner)][k] += 1
    outer[(os.path.basename(cur_outer[0]), cur_outer[1] // 10 * 10)][k] += 1

def show(title, table, key_fmt, top=28):
    print(title)
    rows = sorted(table.items(), key=lambda kv: -kv[1]["valu"])[:top]
    for k, c in rows:
        print(f"  {key_fmt(k):46s} valu {c['valu']:5d}  mfma {c['mfma']:3d}  lds {c['lds']:4d}  vmem {c['vmem']:4d}  salu {c['salu']:4d}  lane {c['lane']:4d}  wait {c['wait']:3d}")
    tot = collections.Counter()
    for c in table.values():
        tot.update(c)
    print(f"  {'TOTAL':46s} valu {tot['valu']:5d}  mfma {tot['mfma']:3d}  lds {tot['lds']:4d}  vmem {tot['vmem']:4d}  salu {tot['salu']:4d}  lane {tot['lane']:4d}  wait {tot['wait']:3d}")

show(f"{m.group(1)}{' block ' + only_block if only_block else ''}: instructions by innermost source function", inner, lambda k: k)
show("by statement of the outermost function (line / 10 * 10)", outer, lambda k: f"{k[0]}:{k[1]}")
