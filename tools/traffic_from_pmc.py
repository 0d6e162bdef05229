#!/usr/bin/env python3
"""profiles/traffic_latest.json from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately as
MI355X_MICROARCH.md §HBM prescribes).  Counter values are KiB-free kilobytes (1 unit = 1024 B per rocprofv3's derived
definition on this image: TCC_EA0_RDREQ*64/1024); FETCH_SIZE is doubled (gfx950 correction for wide streaming reads).
Bytes are summed over the threshold+segment kernels and divided by the number of batch launches (= k_tile dispatches).

usage: traffic_from_pmc.py <fetch_dir> <write_dir> [out.json]"""
import csv, glob, json, os, re, sys

KERNELS = ("k_tile", "k_merge", "k_roots", "k_roots_a", "k_roots_b")


def per_launch(d, counter):
    tot, launches = 0.0, 0
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(p)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            m = re.search(r"::(k_[a-z_]+)", name)
            base = m.group(1) if m else ""
            if base in KERNELS:
                tot += float(row["Counter_Value"]) * 1024.0
                launches += base == "k_tile"
    if launches == 0:
        raise SystemExit(f"no {counter} rows for {KERNELS} under {d}")
    return tot / launches, launches


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(__file__), "..", "profiles", "traffic_latest.json")
    f, nf = per_launch(fd, "FETCH_SIZE")
    w, nw = per_launch(wd, "WRITE_SIZE")
    rec = {"hbm_bytes_per_launch": 2.0 * f + w, "fetch_bytes_corrected": 2.0 * f, "write_bytes": w,
           "launches": [nf, nw],
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE doubled per MI355X_MICROARCH.md; "
                     "kernels k_tile+k_merge+k_roots; bench.py --steps 2 --warmup 1 --no-cpu-baseline",
           "round": 1}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
