"""Register / LDS / spill table of every kernel in the saved assembly of a build (default: /tmp/bts_render_obj).
    python tools/kernel_stats.py [obj_dir] [name filter]"""
import glob, os, re, sys
d = sys.argv[1] if len(sys.argv) > 1 else "/tmp/bts_render_obj"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
print(f"{'kernel':70s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>7s} {'sspill':>6s} {'vspill':>6s}")
for f in sorted(glob.glob(os.path.join(d, "*gfx950.s"))):
    s = open(f).read()
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:", s, re.S):
        b = m.group(0)
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)
        name = g("name")
        if flt not in name:
            continue
        short = re.sub(r"^_ZN3bts\d+", "", name)
        print(f"{short[:70]:70s} {g('vgpr_count'):>5s} {g('agpr_count'):>5s} {g('sgpr_count'):>5s} {g('group_segment_fixed_size'):>7s} "
              f"{g('private_segment_fixed_size'):>7s} {g('sgpr_spill_count'):>6s} {g('vgpr_spill_count'):>6s}")
