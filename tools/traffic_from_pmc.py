#!/usr/bin/env python3
"""profiles/traffic_latest.json from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately as
MI355X_MICROARCH.md §HBM prescribes).  Counter values are KiB-free kilobytes (1 unit = 1024 B per rocprofv3's derived
definition on this image: TCC_EA0_RDREQ*64/1024); FETCH_SIZE is doubled (gfx950 correction for wide streaming reads).
Bytes are summed over the threshold+segment kernels and divided by the number of batch launches (= k_tile dispatches).

usage: traffic_from_pmc.py <fetch_dir> <write_dir> [out.json]"""
import csv, glob, hashlib, json, os, re, sys

KERNELS = ("k_tile", "k_fmerge")


def per_launch(d, counter):
    """Bytes of the threshold+segment kernels per BATCH launch.  bench.py also makes one small call (8 frames: the point count behind
    its `gradient` line), whose k_tile / k_fmerge dispatches must neither add bytes nor count as launches: only dispatches with the
    largest grid of their kernel are taken."""
    rows = []
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(p)):
            if row["Counter_Name"] != counter:
                continue
            m = re.search(r"::(k_[a-z_]+)", row["Kernel_Name"])
            base = m.group(1) if m else ""
            if base in KERNELS:
                rows.append((base, int(row["Grid_Size"]), float(row["Counter_Value"]) * 1024.0))
    full = {k: max((g for b, g, _ in rows if b == k), default=0) for k in KERNELS}
    tot = sum(v for b, g, v in rows if g == full[b])
    launches = sum(1 for b, g, _ in rows if b == "k_tile" and g == full[b])
    if launches == 0:
        raise SystemExit(f"no {counter} rows for {KERNELS} under {d}")
    return tot / launches, launches


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(__file__), "..", "profiles", "traffic_latest.json")
    f, nf = per_launch(fd, "FETCH_SIZE")
    w, nw = per_launch(wd, "WRITE_SIZE")
    import subprocess
    try:
        commit = subprocess.check_output(["git", "-C", os.path.dirname(os.path.abspath(__file__)), "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        commit = None
    rec = {"hbm_bytes_per_launch": 2.0 * f + w, "fetch_bytes_corrected": 2.0 * f, "write_bytes": w,
           "launches": [nf, nw],
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes: tools/collect_profiles.sh); kernels k_tile + k_fmerge; "
                     "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras",
           "note": "FETCH_SIZE is doubled for BOTH kernels (MI355X_MICROARCH.md: gfx950 tallies 128-byte requests at 64 bytes). That rule is "
                   "calibrated for k_tile's reads (16 bytes per lane, streaming); k_fmerge reads 2-byte ring entries and 8-byte list entries, "
                   "for which the counter is uncalibrated, so its share (a few per cent of the total) may be over-counted by up to 2x.",
           "round": int(sys.argv[4]) if len(sys.argv) > 4 else 3, "commit": commit,
           # bench.py compares this with the k_ccl.hip it runs on and flags the figure as stale when the kernels have changed since
           "k_ccl_sha16": hashlib.sha256(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "chalkydri_amd", "csrc", "k_ccl.hip"), "rb").read()).hexdigest()[:16]}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
