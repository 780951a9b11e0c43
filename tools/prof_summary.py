#!/usr/bin/env python3
"""prof_summary.py -- condenses rocprofv3 output (gpurun_out/prof/{stats,fetch,write}) into the small
files committed under profiles/: the --stats kernel table as-is, and a per-kernel PMC summary with
the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB; FETCH_SIZE reads exactly 1/2 of a wide coalesced stream on gfx950, so the
read side is reported both raw and doubled.

usage: tools/prof_summary.py <prof_dir> <profiles_dir> <tag> [workload]
"""
import collections
import csv
import json
import os
import shutil
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]      # (template arguments dropped: k_fused_pool<true> -> k_fused_pool)
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return {k: {"launches": v[0], "avg_KiB_per_launch": v[1] / v[0]} for k, v in agg.items()}


def per_kernel_all(path):
    """{kernel: {counter: average per launch}} for every counter of one pass"""
    if not os.path.exists(path):
        return {}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[k].add(r.get("Dispatch_Id"))
    return {k: dict({c: v / max(1, len(seen[k])) for c, v in d.items()}, launches=len(seen[k])) for k, d in agg.items() if k.startswith("rdx::")}


def main():
    prof, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    workload = sys.argv[4] if len(sys.argv) > 4 else "sample1"
    os.makedirs(out, exist_ok=True)
    st = os.path.join(prof, "stats", "r1_kernel_stats.csv")
    if os.path.exists(st):
        shutil.copy(st, os.path.join(out, "%s_kernel_stats.csv" % tag))
    fetch = per_kernel(os.path.join(prof, "fetch", "r1_counter_collection.csv"), "FETCH_SIZE")
    write = per_kernel(os.path.join(prof, "write", "r1_counter_collection.csv"), "WRITE_SIZE")
    summ = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("rdx::"):
            continue
        f = fetch.get(k, {}).get("avg_KiB_per_launch", 0.0)
        w = write.get(k, {}).get("avg_KiB_per_launch", 0.0)
        summ[k] = {
            "launches_fetch_pass": fetch.get(k, {}).get("launches", 0),
            "launches_write_pass": write.get(k, {}).get("launches", 0),
            "FETCH_SIZE_KiB_per_launch_raw": round(f, 1),
            "WRITE_SIZE_KiB_per_launch": round(w, 1),
            # calibration for this code's access shapes: profiles/r03_fetch_calibrate.log (64-B record gathers: counter at face
            # value; wide coalesced streams: x 2).  Traversal kernels gather, stream kernels (generate / shade / accumulate) stream.
            "read_bytes_per_launch_gather_calibration_x1": int(f * 1024),
            "read_bytes_per_launch_stream_calibration_x2": int(f * 1024 * 2),
            "write_bytes_per_launch": int(w * 1024),
            "bytes_per_launch_gather_calibration": int(f * 1024 + w * 1024),
            "bytes_per_launch_stream_calibration": int(f * 1024 * 2 + w * 1024),
        }
    with open(os.path.join(out, "%s_pmc_hbm.json" % tag), "w") as fjs:
        json.dump({"workload": workload, "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; KiB units; FETCH_SIZE counts wide coalesced "
                           "streams at half their bytes (guide) and per-lane 64-B record gathers at face value (tools/fetch_calibrate.py): both given",
                   "kernels": summ}, fjs, indent=1)
    # instruction-issue and cache counters (separate passes), averages per launch
    extra = {}
    for name in ("issue", "cache"):
        f = os.path.join(prof, name, "r1_counter_collection.csv")
        d = per_kernel_all(f)
        if d:
            for k, v in d.items():
                if "SQ_INSTS_VALU" in v and v["SQ_INSTS_VALU"]:
                    v["active_lanes_per_valu_inst"] = v.get("SQ_THREAD_CYCLES_VALU", 0.0) / v["SQ_INSTS_VALU"]
                    v["salu_per_valu"] = v.get("SQ_INSTS_SALU", 0.0) / v["SQ_INSTS_VALU"]
                    if v.get("GRBM_GUI_ACTIVE"):
                        v["valu_issue_frac_of_pass_cycles"] = v["SQ_INSTS_VALU"] * 2.0 / (1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0)
                    if v.get("SQ_WAVE_CYCLES"):
                        v["wave_cycles_parked_frac"] = v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"]
                if v.get("TCC_HIT_sum") is not None and (v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)):
                    v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
                if v.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
                    v["l1_hit_rate"] = 1.0 - v.get("TCP_TCC_READ_REQ_sum", 0.0) / v["TCP_TOTAL_CACHE_ACCESSES_sum"]
            extra[name] = d
    if extra:
        with open(os.path.join(out, "%s_pmc_issue.json" % tag), "w") as fjs:
            json.dump({"workload": workload, "note": "rocprofv3 --pmc, one pass per group, averages per launch; valu_issue_frac = "
                       "SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) of the PMC pass itself",
                       "passes": extra}, fjs, indent=1)
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    main()
