"""Kernel timeline of ONE event from a rocprofv3 --kernel-trace run.
Run:   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python scripts/timeline.py run blob64|blob1024|block8
Report: python scripts/timeline.py report gpurun_out/tl
The run makes 6 events one at a time (host synchronised between) and prints the host wall time of each; the report takes the kernels
of the last event (everything after the last host gap > 200 us) and prints start / end relative to the first kernel, and the
sum of the gaps on the critical path (time in which no kernel of the event was running)."""
import os, sys, time, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    from surtr_amd import engine as E, scenes as S
    what = sys.argv[2]
    eng = E.Engine(0)
    if what.startswith("blob"):
        n = int(what[4:]); sc = S.blob_scene(n); cb, ce = 0, n
    else:
        sc = S.torus_scene(4096); cb, ce = 1024, 1536
    sc["convex"], _ = S.ach_convex(eng, sc["mesh"]["pos"])
    eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
    for _ in range(6):
        time.sleep(0.002)
        t0 = time.perf_counter(); c = eng.fracture_event(cb, ce); print("event %.3f ms, %d fragments" % ((time.perf_counter() - t0) * 1e3, c.n_frag), flush=True)
    eng.close()
else:
    fs = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(fs[-1]))]
    rows.sort()
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - max(r[1] for r in rows[:i][-40:]) > 200000: cut = i
    ev = rows[cut:]
    t0 = ev[0][0]
    busy_end, gaps = ev[0][0], 0
    for s, e, k in ev:
        g = max(0, s - busy_end)
        gaps += g
        print("%8.1f %8.1f  %6.1f us  %s%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, k, ("   <- gap %.1f us" % (g / 1e3)) if g > 0 else ""))
        busy_end = max(busy_end, e)
    print("event: %d kernels, first start to last end %.1f us, of which no kernel running %.1f us" % (len(ev), (busy_end - t0) / 1e3, gaps / 1e3))
