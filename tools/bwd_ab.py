"""A/B timing of differently built libraries on bts_render_bwd within ONE gpurun call: every library runs tools/bwd_probe.py in its own
subprocess (BTS_RENDER_LIB), interleaved over passes.  Used with the timing-ablation builds of the row backward (-DBTS_ABL_B1 .. B6).
    python tools/bwd_ab.py <shape> <K> default b1 b2 ...      (names under behindthescenes_amd/variants/, or "default")"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shape, K, names = sys.argv[1], sys.argv[2], sys.argv[3:]
res = {a: [] for a in names}
for _ in range(int(os.environ.get("LIB_AB_PASSES", "2"))):
    for a in names:
        lib = os.path.join(ROOT, "behindthescenes_amd", "libbts_render.so" if a == "default" else f"variants/libbts_{a}.so")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bwd_probe.py"), "7", shape, K], env=dict(os.environ, BTS_RENDER_LIB=lib, BTS_ALLOW_LIB_OVERRIDE="1"),
                           capture_output=True, text=True)
        m = re.findall(r"colours from the forward: median ([0-9.]+) ms", r.stdout)
        if r.returncode or not m:
            print(a, "FAILED", r.stderr[-300:])
            continue
        res[a].append(float(m[-1]))
for a, v in res.items():
    if v:
        print(f"{a:12s} median {sorted(v)[len(v) // 2]:.4f} ms   {v}")
