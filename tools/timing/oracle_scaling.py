"""How the CPU oracle's pieces scale on this box's cores (the 2^24 unstructured-point test spends its time here):
python tools/timing/oracle_scaling.py [LOG=22]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import orc
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << lg
print("os.cpu_count() =", os.cpu_count(), " sched_getaffinity =", len(os.sched_getaffinity(0)))
for th in (16, 64, 256):
    t0 = time.perf_counter(); s = orc.synth_vector(orc.FR, 1, n, th); t1 = time.perf_counter()
    p = orc.synth_points(2, n, th); t2 = time.perf_counter()
    r = orc.msm_fast(s, p, th); t3 = time.perf_counter()
    print("2^%d, %3d threads: synth scalars %.2f s, synth points %.2f s, msm_fast %.2f s" % (lg, th, t1 - t0, t2 - t1, t3 - t2), flush=True)
