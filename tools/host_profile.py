"""Where does the HOST spend a training step?  cProfile over `bench.py --workload <w>` (default kitti_raw, the host-sensitive one):
    python tools/host_profile.py [workload] [steps]
prints the functions by own time, per step."""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
w = sys.argv[1] if len(sys.argv) > 1 else "kitti_raw"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
sys.argv = ["bench.py", "--workload", w, "--no-cpu-baseline", "--steps", str(steps), "--warmup", "5"]
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats(45)
out = s.getvalue()
print(f"(times are totals over {steps} + 5 steps and set-up; divide the step-loop entries by {steps + 5})")
print("\n".join(l[:200] for l in out.split("\n")[:75]))
