#!/usr/bin/env python3
"""Collects, ON THE GPU BOX, everything bench.py's roofline objects quote, at the bench shape
(32768 scenes per step, 512-scene render sequences), and writes it under gpurun_out/<round>/:

  kernel_stats.csv          rocprofv3 --kernel-trace --stats of the default `python bench.py`
  bench_under_rocprof.json  the JSON line printed by that very run (its HIP-event durations must agree with the CSV)
  counters.json             per kernel: HBM bytes per launch / per scene (FETCH_SIZE and WRITE_SIZE from two separate
                            --pmc passes, KiB units, corrected by the factors measured on a known-byte kernel in the
                            same passes), and for k_settle the SQ counters: VALU instructions per scene, active lanes
                            (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU), VALU duty of a wave

usage (through gpurun):  python tools/collect_counters.py gpurun_out/r02
Every rocprofv3 call combines --pmc only with --kernel-trace (the pool refuses / crashes on other mixes)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = ["python", os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"]
PMC_SHAPE = ["--steps", "1", "--warmup", "0", "--settle-streams", "1", "--render-streams", "1"]   # one step: every kernel of the path once, serialised
N_CAL = 1 << 31


def kname(n):
    m = re.search(r"(k_\w+)", n)
    return m.group(1) if m else None


def run(cmd, **kw):
    print("+ " + " ".join(cmd), flush=True)
    return subprocess.run(cmd, **kw)


def pmc(out, tag, counters, cmd):
    d = os.path.join(out, "raw_" + tag)
    shutil.rmtree(d, ignore_errors=True)
    run(["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "--"] + cmd,
        stdout=subprocess.DEVNULL, check=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    ids = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = kname(r["Kernel_Name"]) or ("cal" if "elementwise" in r["Kernel_Name"] else None)
        if not k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ids[k].add(r["Dispatch_Id"])
    return {k: {c: v / len(ids[k]) for c, v in cs.items()} for k, cs in acc.items()}, {k: len(v) for k, v in ids.items()}


def main():
    out = os.path.abspath(sys.argv[1])
    os.makedirs(out, exist_ok=True)
    batch, chunk = 32768, 512     # bench.py defaults
    # 1. per-kernel durations of the default command + the bench line under the profiler
    d = os.path.join(out, "raw_stats")
    shutil.rmtree(d, ignore_errors=True)
    r = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + BENCH,
            stdout=subprocess.PIPE, text=True, check=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    open(os.path.join(out, "bench_under_rocprof.json"), "w").write(line + "\n")
    st = glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)[0]
    shutil.copy(st, os.path.join(out, "kernel_stats.csv"))
    stats = {kname(r_["Name"]) or r_["Name"][:40]: r_ for r_ in csv.DictReader(open(st))}
    # 2. HBM traffic: two passes, each also over the calibration kernel
    cal_cmd = ["python", os.path.join(ROOT, "tools", "pmc_calibrate.py")]
    fetch, n_f = pmc(out, "fetch", ["FETCH_SIZE"], BENCH + PMC_SHAPE)
    write, _ = pmc(out, "write", ["WRITE_SIZE"], BENCH + PMC_SHAPE)
    cal_f, _ = pmc(out, "cal_fetch", ["FETCH_SIZE"], cal_cmd)
    cal_w, _ = pmc(out, "cal_write", ["WRITE_SIZE"], cal_cmd)
    f_fac = N_CAL / (cal_f["cal"]["FETCH_SIZE"] * 1024.0)
    w_fac = N_CAL / (cal_w["cal"]["WRITE_SIZE"] * 1024.0)
    # 3. SQ counters of the settle kernel (8 SQ slots per pass)
    sq, n_sq = pmc(out, "sq", ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES",
                               "SQ_INSTS_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"], BENCH + PMC_SHAPE)
    # LDS pipe and residency (the north star names both): bank conflicts against the cycles the LDS is busy, waves per SIMD =
    # SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / the SIMDs; a pass of its own (8 SQ slots per pass)
    try:
        lds, _ = pmc(out, "lds", ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
                                  "SQ_INST_LEVEL_LDS"], BENCH + PMC_SHAPE)
    except Exception as e:      # a counter this build of rocprofv3 does not know: say so, keep the rest
        print("LDS counter pass failed: %r" % (e,), flush=True)
        lds = {}
    # the fetch factor once more on a 4 B / lane kernel (the guide's factor 2.0 is quoted for 16 B / lane streams)
    cal4_cmd = ["python", os.path.join(ROOT, "tools", "pmc_calibrate.py"), "4"]
    try:
        cal4_f, _ = pmc(out, "cal4_fetch", ["FETCH_SIZE"], cal4_cmd)
        f_fac4 = N_CAL / (cal4_f["cal"]["FETCH_SIZE"] * 1024.0)
    except Exception as e:
        print("4 B / lane calibration failed: %r" % (e,), flush=True)
        f_fac4 = None
    sq_cal, _ = pmc(out, "sq_cal", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES"], cal_cmd)
    full = sq_cal["cal"]["SQ_THREAD_CYCLES_VALU"] / sq_cal["cal"]["SQ_ACTIVE_INST_VALU"]   # the ratio of a kernel with all 64 lanes on
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if k == "cal":
            continue
        n = batch if (k in ("k_settle", "k_synth_stage", "k_synth_place") or k.startswith("k_w_")) else chunk
        fb = fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0
        wb = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
        kernels[k] = {"scenes_per_launch": n, "dispatches_in_pass": n_f.get(k, 0),
                      "fetch_bytes_raw": fb, "write_bytes_raw": wb,
                      "fetch_bytes": fb * f_fac, "write_bytes": wb * w_fac,
                      "hbm_bytes_per_launch": fb * f_fac + wb * w_fac,
                      "hbm_bytes_per_scene": (fb * f_fac + wb * w_fac) / n}
        if k in stats:
            kernels[k]["avg_ms_default_run"] = float(stats[k]["AverageNs"]) / 1e6
            kernels[k]["calls_default_run"] = int(stats[k]["Calls"])
            kernels[k]["pct_gpu_time_default_run"] = float(stats[k]["Percentage"])
    for k, c in sq.items():      # SQ counters of every kernel (averages per launch)
        if k == "cal" or not c.get("SQ_ACTIVE_INST_VALU"):
            continue
        n = batch if (k in ("k_settle", "k_synth_stage", "k_synth_place") or k.startswith("k_w_")) else chunk
        kernels.setdefault(k, {})
        kernels[k].update({
            "sq": c,
            "valu_insts_per_scene_launch": c["SQ_INSTS_VALU"] / n,
            "active_lanes": 64.0 * (c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]) / full,
            "valu_duty_of_a_wave": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
            "wait_any_share": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in c else None,
            "wait_inst_share": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in c else None,
        })
    for k, c in lds.items():
        if k == "cal" or k not in kernels:
            continue
        kernels[k]["lds"] = {
            "counters": c,
            "bank_conflict_share_of_lds_cycles": (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None,
            # SQ_WAVE_CYCLES counts 4-cycle quads per resident wave, SQ_BUSY_CYCLES the quads the SQ is busy (summed over the XCDs' SQs):
            # their ratio is the average number of waves resident while the kernel runs; / 1024 SIMDs
            "waves_resident_avg": (c["SQ_WAVE_CYCLES"] / c["SQ_BUSY_CYCLES"]) if c.get("SQ_BUSY_CYCLES") else None,
        }
    notes = {"active_lanes": "64 x (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU of the kernel) / (the same ratio of an elementwise kernel with all "
                             "64 lanes on, measured in the same session: %.1f)" % full}
    sys.path.insert(0, ROOT)
    import bench

    res = {
        "kernel_source_sha": bench.kernel_source_sha(),     # bench.py quotes per-instruction figures only for THESE sources
        "commands": {
            "stats": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline",
            "pmc": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ_...> --kernel-trace --output-format csv -- python bench.py --no-cpu-baseline " + " ".join(PMC_SHAPE),
            "calibration": "same two --pmc passes over python tools/pmc_calibrate.py (2 GiB read + 2 GiB written per dispatch, 16 B per lane)",
        },
        "shape": {"scenes_per_settle_launch": batch, "scenes_per_render_launch": chunk},
        "units": "FETCH_SIZE / WRITE_SIZE are KiB; bytes = counter x 1024 x the factor measured on the known-byte kernel",
        "calibration": {"fetch_factor": f_fac, "write_factor": w_fac, "fetch_factor_4B_per_lane": f_fac4,
                        "expected": "fetch factor 2.0 for 16 B/lane streaming reads (MI355X_MICROARCH.md HBM section), write factor ~1.0; "
                                    "the 16 B / lane factor is the one applied (the kernels' bulk reads are dwordx4)"},
        "notes": notes,
        "kernels": kernels,
    }
    json.dump(res, open(os.path.join(out, "counters.json"), "w"), indent=1)
    for k, v in kernels.items():
        print("%-18s %8.2f MB/scene HBM  avg %s ms" % (k, v.get("hbm_bytes_per_scene", 0) / 1e6, v.get("avg_ms_default_run")))
    print("calibration factors: fetch %.3f write %.3f" % (f_fac, w_fac))
    for dname in glob.glob(os.path.join(out, "raw_*")):
        shutil.rmtree(dname, ignore_errors=True)     # keep gpurun_out small: the summaries are what is committed


if __name__ == "__main__":
    main()
