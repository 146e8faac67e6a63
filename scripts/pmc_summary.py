"""Summarises the rocprofv3 --pmc passes of scripts/pmc.sh (gpurun_out/pmc/*) per kernel -> JSON, and writes
profiles/traffic.json for bench.py: HBM-side bytes per launch of the dominant kernel, tied to the build they were measured on.
Usage: python scripts/pmc_summary.py gpurun_out/pmc profiles/rNN_pmc_summary.json [fetch_calibration.json]"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
root, out = sys.argv[1], sys.argv[2]
calib = json.load(open(sys.argv[3])) if len(sys.argv) > 3 and os.path.exists(sys.argv[3]) else None
res = {}
for name in ["fetch", "write", "tcc", "sq1", "sq2"]:
    fs = sorted(glob.glob(os.path.join(root, name, "*", "*counter_collection.csv")), key=os.path.getmtime)
    if not fs:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        for c, v in d.items():
            res.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k in sorted(res):
    if k.startswith("k_"):
        d = res[k]
        print("%-18s FETCH %10.0f KB  WRITE %10.0f KB  wave cycles %.3g  wait_any %.3g" % (k, d.get("FETCH_SIZE", 0), d.get("WRITE_SIZE", 0), d.get("SQ_WAVE_CYCLES", 0), d.get("SQ_WAIT_ANY", 0)))
import bench
# the kernel that clips the Meshes: k_clip_pairs_main (split arrangement, round 4), k_clip_pairs_wave, or k_clip_pairs
dom = max(("k_clip_pairs_main", "k_clip_pairs_wave", "k_clip_pairs"), key=lambda k: res.get(k, {}).get("SQ_WAVE_CYCLES", 0))
prep_name = "k_prep_pairs_sorted" if "k_prep_pairs_sorted" in res else "k_prep_pairs"
clip, prep = res.get(dom, {}), res.get(prep_name, {})
# FETCH_SIZE correction: the guide's x2 holds for 16-B-per-lane streams; this kernel reads 16-B words of the images (x2) and
# 2..12-B gathers.  With a calibration file the factor measured for its narrowest common access (dword gathers: bytes counted
# per 64-B line touched) bounds the read side from above; without one the uncorrected figure is a lower bound.
factor = 2.0
note = "FETCH_SIZE x2 (the guide's gfx950 correction for wide streaming reads; the kernel's image loads are 16-B words)"
if calib:
    note += "; calibration on this box: dwordx4 %.2f, dword %.2f, 3 x dword stride 12 %.2f of the true bytes; random dword gathers are counted at %.2f of the 64-B lines they touch" % (
        calib["k16_dwordx4_per_lane"], calib["k4_dword_per_lane"], calib["k12_three_dwords_stride12"], calib["g4_random_dword_vs_64B_lines"])
rec = {"build_id": bench.kernel_build_id(), "kernel": dom,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/pmc.sh + scripts/pmc_summary.py) on `python bench.py --steps 2 --warmup 1`, MI355X; per-kernel means in " + os.path.basename(out),
       "k_clip_pairs_FETCH_SIZE_KB": clip.get("FETCH_SIZE"), "k_clip_pairs_WRITE_SIZE_KB": clip.get("WRITE_SIZE"),
       "fetch_correction": factor, "note": note,
       "k_clip_pairs_hbm_bytes_per_launch": int((clip.get("FETCH_SIZE", 0) * factor + clip.get("WRITE_SIZE", 0)) * 1024),
       "k_prep_pairs_hbm_bytes_per_launch": int((prep.get("FETCH_SIZE", 0) * factor + prep.get("WRITE_SIZE", 0)) * 1024),
       "prep_kernel": prep_name,
       # the kernels that run beside the dominant one on other streams (split arrangement): general clipper for irregular pairs / large bands
       "beside": {k: int((res[k].get("FETCH_SIZE", 0) * factor + res[k].get("WRITE_SIZE", 0)) * 1024) for k in ("k_clip_pairs_catch", "k_clip_pairs_big") if k in res}}
for k in ("TCC_HIT_sum", "TCC_MISS_sum", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
          "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
    if k in clip:
        rec[k] = clip[k]
if calib:
    rec["fetch_calibration"] = calib
json.dump(rec, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print("profiles/traffic.json: build", rec["build_id"], "k_clip_pairs", rec["k_clip_pairs_hbm_bytes_per_launch"], "B per launch")
