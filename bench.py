#!/usr/bin/env python3
"""bench.py — frames/sec of the AprilTag detect + pose hot path on MI355X, with the HBM roofline of the
threshold+segment stage and the CPU oracle timed beside it.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one pass of the whole path (threshold -> segment -> clusters -> quad fit -> decode -> glue -> SQPnP -> 64-byte pose
records) over one batch of synthetic frames that is already resident in HBM.  Workload at every N: BASELINE.json
configs[1] — 1280x800 mono8, batch 256 per GPU, tag36h11, 6 field tags per frame rendered from 3-D scenes on the SURVEY
§8d background (ramp +-24, uniform noise +-3).  N > 1 shards one camera stream per GPU (weak scaling) and ends every step
with ONE RCCL all_gather of the 64-byte records; there is no other collective.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)
ALG_BYTES_PER_PX = 7.0  # threshold 1R+1W, segment 1R+4W


def physical_cores():
    """Distinct (package, core) pairs of /proc/cpuinfo (hardware threads / SMT siblings counted once); 0 when unknown."""
    seen, phys, core = set(), None, None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
    except OSError:
        return 0
    return len(seen)


def cpu_share():
    """CPUs this process may use at once: the affinity mask, cut down by the cgroup's CPU quota (v2 cpu.max, v1 cfs_quota_us)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(frames, gyro, task, cfg, budget_s=10.0):
    """Oracle (oracle/libck_oracle.so, kind 'port') on the host cores, on a bounded sample of the same frames: the C driver
    oracle/bench_threads.c runs ora_process_frame on T POSIX threads (frames handed out by an atomic counter, the threads' malloc
    arenas reused from frame to frame) for T in {1, 16, physical cores, nproc}, each capped at the CPUs the box grants this process; `value` is the best of them, `cores` its T."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import subprocess
    import pyoracle
    L0 = pyoracle.lib()                        # the checker build (-O2), shipped with the repository
    # SURVEY §8d asks for the CPU stand-in at -O3 -march=native: built HERE, on the host that is being timed (oracle/Makefile:
    # native; -ffp-contract=off stays, so it is the same function — checked below on frames of this batch, and on the goldens
    # by tests/test_oracle_detector.py).  If the box has no compiler the -O2 build is timed and the line says so.
    flags, L = "-O2 -ffp-contract=off -fno-fast-math (checker build: no compiler on this host for the native one)", L0
    try:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        L = C.CDLL(os.path.join(ROOT, "oracle", "libck_oracle_native.so"))
        flags = "-O3 -march=native -ffp-contract=off -fno-fast-math (gcc, built on this host)"
    except Exception as e:  # noqa: BLE001
        print(f"bench.py: native oracle build failed ({e}); timing the -O2 build", file=sys.stderr)
    L.ora_bench_process.restype = C.c_double
    L.ora_bench_process.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.POINTER(C.c_int)]
    nf = min(len(frames), 64)
    fr = np.ascontiguousarray(frames[:nf])
    gy = np.ascontiguousarray(np.asarray(gyro[:nf], np.float64))
    h, w = fr.shape[1:]

    def run(threads, n_work):
        nv = C.c_int(0)
        dt = L.ora_bench_process(fr.ctypes.data, w, h, w, w * h, nf, n_work, C.addressof(cfg), C.addressof(task._pp), gy.ctypes.data, threads,
                                 C.byref(nv))
        if dt <= 0:
            raise RuntimeError("ora_bench_process failed")
        return dt, nv.value

    nproc = max(1, os.cpu_count() or 1)
    phys = physical_cores() or nproc
    share = cpu_share() or nproc                # CPUs this process may actually use (affinity mask, cgroup quota)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    same = None
    if L is not L0:                            # the timed build returns the checker build's bytes on frames of this batch
        from chalkydri_amd import _abi as A
        same = True
        for i in range(min(3, nf)):
            recs = []
            for lib_ in (L0, L):
                out, v = A.VisionMeasurement(), C.c_int(0)
                lib_.ora_process_frame(C.c_void_p(fr[i].ctypes.data), w, h, w, C.byref(cfg), C.byref(task._pp), C.c_double(float(gy[i])), 1, C.byref(out), C.byref(v))
                recs.append((bytes(out), v.value))
            same = same and recs[0] == recs[1]
        if not same:
            raise RuntimeError("the -O3 -march=native oracle differs from the -O2 checker build")
    run(1, 1)                                  # (first touch of the allocator arenas and of the frames)
    dt1, _ = run(1, 4)
    per = dt1 / 4                              # CPU seconds per frame, one thread
    # T = 1, 16, the physical cores and the hardware threads — as far as the box lets this process have them: beyond its CPU
    # share (a one-GPU box of the pool grants 16 of the host's 256 hardware threads) more threads only take turns on the same
    # CPUs (round 2's 256-thread figure BELOW its 16-thread one was that, not the allocator)
    counts = sorted({1, min(16, share), min(phys, share), min(nproc, share)})
    each = budget_s / len(counts)              # wall seconds each thread count may take
    rates, sample = {}, []
    for t in counts:
        n_work = max(t, min(4096, int(each * t / max(per, 1e-4)) // t * t))   # a whole number of frames per thread
        dt, nvalid = run(t, n_work)
        rates[t] = n_work / dt
        sample.append(f"T={t}: {n_work} frames in {dt:.2f} s")
    best = max(rates, key=rates.get)
    return {"value": round(rates[best], 2), "unit": "frames/s", "cores": best, "kind": "port",
            "by_threads": {str(t): round(r, 2) for t, r in rates.items()}, "single_thread_value": round(rates[1], 2),
            "physical_cores": phys, "nproc": nproc, "cpu_share": share, "cpu_model": model,
            "oracle_flags": flags, "native_build_equals_checker_on_sample": same,
            "sample": f"the same {w}x{h} workload ({nf} distinct frames, cycled) through oracle/ (C restatement of the path; the reference's "
                      f"Rust path cannot be built here), detect+pose, POSIX threads over frames (oracle/bench_threads.c); " + "; ".join(sample)}


def visible_gpus():
    """GPUs of this host counted WITHOUT touching HIP (the parent of the ranks must not initialise the GPU before it starts
    them): KFD topology nodes that have SIMDs (CPU nodes have simd_count 0), narrowed by ROCR/HIP_VISIBLE_DEVICES when set;
    falls back to the DRM render nodes."""
    import glob
    n = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(path):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        except (OSError, ValueError):
            pass
    if n == 0:
        n = len(glob.glob("/dev/dri/renderD*"))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run (one process per GPU,
    RCCL over xGMI) and hand back its exit code.  The parent never touches HIP (devices are counted from sysfs) and never
    execs: the ranks are spawned."""
    import socket
    import subprocess
    have = visible_gpus()
    if have < n_gpus:
        print(f"bench.py: --gpus {n_gpus} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--tags", type=int, default=6)
    ap.add_argument("--noise", type=int, default=3)
    ap.add_argument("--decimate", type=int, default=1)
    ap.add_argument("--unique", type=int, default=0, help="distinct frames rendered per stream (0 = every frame of the batch is distinct)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the two extra measurements reported under \"also\"")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as tdist
    from chalkydri_amd import default_config, dist, scenes
    from chalkydri_amd.apriltags import AprilTags

    rank, local_rank, world = dist.env_rank()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} HIP device(s) are visible")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init("nccl")

    w, h, n = args.width, args.height, args.batch
    if args.unique <= 0 or args.unique > n:
        args.unique = n
    frames, gyro, layout, calib, r2c = scenes.bench_stream(2, n, w, h, args.tags, stream=rank, unique=args.unique, noise_amp=args.noise)
    task = AprilTags(w, h, layout, calib, r2c, cam_id=rank, max_batch=n, device=local_rank, quad_decimate=args.decimate)
    task.detector.upload(frames)                      # inputs resident in HBM before the timed region
    d_gyro = torch.from_numpy(np.ascontiguousarray(gyro)).to(dev)
    d_has = torch.ones(n, dtype=torch.uint8, device=dev)
    d_rec = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
    d_valid = torch.zeros(n, dtype=torch.int32, device=dev)
    gathered = None
    thr_ms = []
    # The one collective of the path: the C ABI's ck_gather_poses (ncclAllGather of n x 64 bytes on the handle's stream).  If
    # its communicator cannot be made on some rank, EVERY rank uses the Python mirror's torch.distributed all_gather instead
    # (same bytes, same layout) and the JSON line says which one ran.
    comm, gather_kind, gather_lib = None, "none (1 GPU)", None
    force_comm = os.environ.get("CK_BENCH_FORCE_COMM") == "1"   # (tests: the C ABI's collective inside this loop on ONE rank)
    if world > 1 or force_comm:
        try:
            comm = dist.PoseComm(task.detector, rank, world, dev)
        except Exception as e:  # noqa: BLE001
            print(f"[rank {rank}] ck_comm_create failed ({e}); falling back to torch.distributed", file=sys.stderr)
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
        if world > 1:
            tdist.all_reduce(ok, op=tdist.ReduceOp.MIN)
        if not int(ok.item()):
            if comm is not None:
                comm.close()
            comm = None
        gather_kind = "ck_gather_poses: RCCL ncclAllGather of 64-byte records (C ABI)" if comm is not None else \
            "torch.distributed all_gather_into_tensor of 64-byte records (RCCL; C-ABI communicator unavailable)"
        d_all = torch.zeros((world * n, 64), dtype=torch.uint8, device=dev)
        # Two users of RCCL live in this process at N > 1: torch's process group and the C ABI's communicator (dlopen by soname).
        # Which file each resolved to is printed per rank and carried in the JSON line (they should be one mapped library).
        mapped = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln})
        gather_lib = {"ck_comm": comm.library() if comm is not None else None, "mapped_librccl": mapped}
        print(f"[rank {rank}] librccl: ck_comm -> {gather_lib['ck_comm']}; mapped in this process: {mapped}", file=sys.stderr)

    def step(record=False):
        nonlocal gathered
        task.process_uploaded_into(n, d_gyro.data_ptr(), d_has.data_ptr(), d_rec.data_ptr(), d_valid.data_ptr())
        if record:
            thr_ms.append(task.detector.stage_ms()["threshold"])
        if comm is not None:
            comm.gather(n, out_ptr=d_all.data_ptr(), sync=False)
            gathered = d_all
        else:
            gathered = dist.gather_records(d_rec, world)

    for _ in range(args.warmup):
        step()
    if world > 1:
        tdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(record=True)
    if comm is not None:
        comm.sync()                                    # the handle's stream is not torch's current stream
    torch.cuda.synchronize()
    if world > 1:
        tdist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        tdist.all_reduce(tmax, op=tdist.ReduceOp.MAX)
    dt = float(tmax.item())

    # the collective alone, after the timed region: enqueue + completion of one gather of this batch's records, wall clock
    gather_us = None
    if comm is not None:
        comm.sync()
        ts = []
        for _ in range(20):
            t1 = time.perf_counter()
            comm.gather(n, out_ptr=d_all.data_ptr(), sync=False)
            comm.sync()
            ts.append((time.perf_counter() - t1) * 1e6)
        gather_us = {"median": round(float(np.median(ts)), 1), "min": round(float(np.min(ts)), 1), "bytes_per_rank": n * 64,
                     "what": "ck_gather_poses(sync=0) + ck_comm_sync, host wall clock, 20 repeats after the timed region"}
    if rank == 0:
        stage = task.detector.stage_ms()
        recs = dist.records_to_numpy(gathered)
        valid_frac = float(np.count_nonzero(recs["tag_count"] > 0)) / len(recs)
        ms_thr = float(np.mean(thr_ms))
        achieved = ALG_BYTES_PER_PX * w * h * n / (ms_thr * 1e-3) / 1e9
        # attainable ceiling next to the spec peak: a 1 GiB device-to-device copy, read + write bytes
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
        # HBM bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs of this same command:
        # tools/collect_profiles.sh), which cannot run inside this process; the file names the round and commit it was taken at
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_src = {k: tj.get(k) for k in ("round", "commit", "note") if tj.get(k) is not None}
                # the figure belongs to the kernels it was measured on: the file carries the hash of their source
                import hashlib
                now = hashlib.sha256(open(os.path.join(ROOT, "chalkydri_amd", "csrc", "k_ccl.hip"), "rb").read()).hexdigest()[:16]
                if tj.get("k_ccl_sha16") != now:
                    traffic_src["stale"] = True
                    print(f"bench.py: profiles/traffic_latest.json was measured on another k_ccl.hip ({tj.get('k_ccl_sha16')} != {now}): "
                          "re-run tools/collect_profiles.sh", file=sys.stderr)
            except Exception:
                traffic = None
        out = {
            "metric": "frames/sec (detect+pose) on 1280x800 batch; HBM GB/s vs roofline",
            "value": round(world * n * args.steps / dt, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 (threshold/segment/clusters: integer; quad fit/decode/SQPnP: f64)", "data": "synthetic",
            "config": {"workload": f"{w}x{h} mono8 batch={n}/GPU, tag36h11, {args.tags} field tags/frame (3-D scenes), background ramp+-24 "
                                   f"noise+-{args.noise}, quad_decimate={args.decimate}, detect+pose, one stream per GPU",
                       "frames_with_pose": round(valid_frac, 4), "post_segment_streams": int(os.environ.get("CK_STREAMS", "1")), "gather": gather_kind, "gather_us": gather_us, "gather_lib": gather_lib},
            "roofline": {"bound": "hbm", "kernel": "threshold+segment (k_tile + k_fmerge)", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         # what `achieved` is: SURVEY §8d's ALGORITHMIC bytes (7 per pixel: 1 R + 1 W threshold, 1 R + 4 W segment) over the
                         # measured time — an accounting figure, not bytes on the bus: the stage writes 16-bit tile-local label words and
                         # never re-reads the threshold map, so it moves fewer (`traffic`, PMC); `real_hbm_GBps` = traffic / time
                         "achieved_is": "algorithmic bytes (7 B/px, SURVEY 8d) / measured time; the kernels move `traffic` bytes",
                         "real_hbm_GBps": (round(traffic / (ms_thr * 1e-3) / 1e9, 1) if traffic else None),
                         "real_hbm_frac_of_peak": (round(traffic / (ms_thr * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None),
                         "traffic": traffic, "traffic_source": traffic_src, "measured_copy_GBps": round(copy_gbps, 1),
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_PX * w * h * n, "avg_launch_ms": round(ms_thr, 4),
                         "launch_ms_p10_median_p90": [round(float(np.percentile(thr_ms, q)), 4) for q in (10, 50, 90)]},
            "stage_ms_last_step": {k: round(v, 3) for k, v in stage.items()},
        }
        if world == 1 and not args.no_extras:   # (--no-extras: the profiled runs, whose kernel averages must be over whole-batch launches only)
            # "Gradient extraction (next stage) = 5 B/px read (label + thresh) + compacted output, reported separately":
            # the clusters stage (k_emit -> k_scan -> k_scatter: boundary points between adjacent black / white components, keyed by the
            # component pair, compacted per cluster).  Algorithmic bytes = 5 per pixel read + 4 bytes per point written, read and
            # written again (temporary run, scatter in, final place); points counted on a sample of the batch's frames.
            ns = min(8, n)
            try:
                pts = float(np.mean([len(p_) for _, p_ in task.detector.clusters(frames[:ns])]))   # (per frame: cluster records, point records)
            except Exception as e:  # noqa: BLE001
                print(f"bench.py: point count for the gradient line failed: {e}", file=sys.stderr)
                pts = None
            if pts is not None:
                gbytes = 5.0 * w * h * n + 12.0 * pts * n
                gms = stage["clusters"]
                out["gradient"] = {"kernel": "clusters stage (k_emit + k_scan + k_scatter)", "bound": "hbm", "bytes": gbytes, "points_per_frame": round(pts, 1),
                                   "points_counted_on": f"{ns} frames of the batch", "ms": round(gms, 4),
                                   "achieved": round(gbytes / (gms * 1e-3) / 1e9, 1), "unit": "GB/s", "frac": round(gbytes / (gms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                   "bytes_is": "algorithmic: 5 B/px read (1 thresholded + 4 label, SURVEY 8d) + 12 B per boundary point"}
        if world == 1 and not args.no_extras:
            # Not part of the contract fields: the same live run at (a) the detector's library-default quad_decimate = 2, which is
            # what the reference actually runs (it never changes detector defaults, crates/apriltags/src/lib.rs:258-262), and
            # (b) threshold+segment on a low-noise background (noise +-1 plus the ramp stays under min_white_black_diff = 5, so the background
            # is "no contrast" instead of binary noise) — the regime of a well-exposed camera frame.
            also = {}
            if args.decimate == 1:
                task2 = AprilTags(w, h, layout, calib, r2c, cam_id=rank, max_batch=n, device=local_rank, quad_decimate=2)
                task2.detector.upload(frames)
                run2 = lambda: task2.process_uploaded_into(n, d_gyro.data_ptr(), d_has.data_ptr(), d_rec.data_ptr(), d_valid.data_ptr())  # noqa: E731
                run2()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for _ in range(3):
                    run2()
                torch.cuda.synchronize()
                t2 = (time.perf_counter() - t2) / 3
                also["quad_decimate_2"] = {"value": round(n / t2, 1), "unit": "frames/s", "ms_per_step": round(t2 * 1e3, 3),
                                           "frames_with_pose": round(float(np.count_nonzero(d_valid.cpu().numpy())) / n, 4)}
                task2.detector.close()
            quiet = scenes.bench_stream(2, n, w, h, args.tags, stream=rank, unique=min(16, n), noise_amp=1)[0]
            task.detector.upload(quiet)
            ms_q = task.detector.time_threshold_segment(n, 5)
            gb_q = ALG_BYTES_PER_PX * w * h * n / (ms_q * 1e-3) / 1e9
            also["threshold_segment_low_noise"] = {"noise": 1, "avg_launch_ms": round(ms_q, 4), "achieved": round(gb_q, 1), "unit": "GB/s",
                                                   "frac": round(gb_q / HBM_PEAK_GBPS, 4)}
            if (w, h, n) == (1280, 800, 256) and args.decimate == 1:
                # BASELINE configs 3 and 5 at full size, one warm call and one timed call each (they fit one GPU: the handles take
                # about 100 and 125 GB), so that the driver's own run carries the figures DESIGN.md quotes.  Frames are 8 distinct
                # ones per config, repeated; the headline line above stays on configs[1].
                if comm is None:
                    task.detector.close()      # (its 22 GB are not needed any more; the CPU baseline below uses the task's parameters only)
                for name, cw, ch, cn, ctags, fams in (("config3_1920x1080x512_30tags_pose", 1920, 1080, 512, 30, ("tag36h11",)),
                                                       ("config5_2448x2048x256_20tags_mixed_families", 2448, 2048, 256, 20, ("tag16h5", "tag36h11"))):
                    try:
                        if len(fams) == 1:   # detect + pose, as the headline
                            cf, cg, clay, ccal, cr2c = scenes.bench_stream(3, cn, cw, ch, ctags, stream=rank, unique=8, noise_amp=args.noise)
                            ct = AprilTags(cw, ch, clay, ccal, cr2c, cam_id=rank, max_batch=cn, device=local_rank, quad_decimate=1)
                            cdet = ct.detector
                            cdet.upload(cf)
                            cgy = torch.from_numpy(np.ascontiguousarray(cg)).to(dev)
                            chas = torch.ones(cn, dtype=torch.uint8, device=dev)
                            crec = torch.zeros((cn, 64), dtype=torch.uint8, device=dev)
                            cval = torch.zeros(cn, dtype=torch.int32, device=dev)
                            call = lambda: ct.process_uploaded_into(cn, cgy.data_ptr(), chas.data_ptr(), crec.data_ptr(), cval.data_ptr())  # noqa: E731
                            what = "detect+pose"
                        else:                # mixed families: detection only (the field layout of the pose glue is tag36h11's)
                            from chalkydri_amd import synth
                            from chalkydri_amd.detector import AprilTagDetector
                            f8, _ = synth.render_batch(55, 8, cw, ch, ctags, fams, family_mode=1)
                            cdet = AprilTagDetector(cw, ch, max_batch=cn, families=fams, bits_corrected=1, device=local_rank)
                            cdet.upload(np.concatenate([f8] * (cn // 8)))
                            from chalkydri_amd import _abi as A
                            cd_, cc_, cs_ = (A.Detection * (cn * 64))(), (C.c_int32 * cn)(), (C.c_uint32 * cn)()
                            call = lambda: cdet._L.ck_detect_uploaded(cdet._h, cn, cd_, 64, cc_, cs_)  # noqa: E731  (the C entry point itself: no Python result objects)
                            what = "detect (ids + corners to the host)"
                        call()
                        torch.cuda.synchronize()
                        t3 = time.perf_counter()
                        call()
                        torch.cuda.synchronize()
                        t3 = time.perf_counter() - t3
                        st3 = cdet.stage_ms()
                        gb3 = ALG_BYTES_PER_PX * cw * ch * cn / (st3["threshold"] * 1e-3) / 1e9
                        also[name] = {"what": what + ", one timed call after one warm call, 8 distinct frames repeated", "value": round(cn / t3, 1), "unit": "frames/s",
                                      "ms_per_call": round(t3 * 1e3, 2), "threshold_segment_ms": round(st3["threshold"], 3),
                                      "threshold_segment_frac": round(gb3 / HBM_PEAK_GBPS, 4), "stage_ms": {k: round(v, 3) for k, v in st3.items()}}
                        cdet.close()
                    except Exception as e:  # noqa: BLE001
                        also[name] = {"error": str(e)[:300]}
            out["also"] = also
        if world == 1 and not args.no_cpu_baseline:
            cfg = default_config(w, h, quad_decimate=args.decimate)
            out["cpu_baseline"] = cpu_baseline(frames[:args.unique], gyro[:args.unique], task, cfg)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
