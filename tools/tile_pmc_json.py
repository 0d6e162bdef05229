"""gpurun_out/pmc_tile{1,2} (tools/collect_tile_pmc.sh) -> profiles/r02_k_tile_pmc.json: per-launch averages of k_tile's SQ counters."""
import csv, glob, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"kernel": "k_tile<false> (one launch = 1280x800x256, 64 000 workgroups of 256 threads)",
       "source": "rocprofv3 --pmc, two passes over tools/bench_thrseg.py 1280 800 256 synth (tools/collect_tile_pmc.sh); per-launch averages "
                 "(SQ_* cycle counters are quad-cycles summed over waves)"}
for k in (1, 2):
    acc, launches = {}, {}
    files = glob.glob(os.path.join(root, "gpurun_out", f"pmc_tile{k}", "**", "*counter_collection.csv"), recursive=True)
    for f in sorted(files, key=os.path.getmtime)[-1:]: # the newest pass only: gpurun merges into whatever earlier runs left behind
        for r in csv.DictReader(open(f)):
            if "k_tile" not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            launches.setdefault(r["Counter_Name"], set()).add(r["Dispatch_Id"])
    out[f"pass{k}"] = {c: round(v / max(1, len(launches[c]))) for c, v in sorted(acc.items())}
p1, p2 = out["pass1"], out["pass2"]
waves = 64000 * 4
out["per_wave"] = {"valu_instructions": round(p2["SQ_INSTS_VALU"] / waves), "salu_instructions": round(p2["SQ_INSTS_SALU"] / waves),
                   "lds_instructions": round(p1["SQ_INSTS_LDS"] / waves)}
out["wait_any_fraction_of_wave_cycles"] = round(p1["SQ_WAIT_ANY"] / p1["SQ_WAVE_CYCLES"], 3)
out["lds_bank_conflict_fraction_of_lds_active"] = round(p2["SQ_LDS_BANK_CONFLICT"] / max(1, p2["SQ_LDS_IDX_ACTIVE"]), 3)
px = 1280 * 800 * 256
out["valu_lane_slots_per_pixel"] = round(p2["SQ_INSTS_VALU"] * 64 / px, 1)
out["salu_instructions_per_pixel_x64"] = round(p2["SQ_INSTS_SALU"] * 64 / px, 1)
json.dump(out, open(os.path.join(root, "profiles", sys.argv[1] if len(sys.argv) > 1 else "r02_k_tile_pmc.json"), "w"), indent=1)
print(json.dumps(out["per_wave"]), out["wait_any_fraction_of_wave_cycles"], out["lds_bank_conflict_fraction_of_lds_active"])
