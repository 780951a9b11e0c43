#!/usr/bin/env python3
"""fetch_calibrate.py -- what does rocprofv3's FETCH_SIZE report for the access shape of the traversal kernels?

/opt/skills/guides/MI355X_MICROARCH.md calibrates the counter for wide coalesced reads only (gfx950: it reports HALF their bytes)
and says "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  The pool engine's
reads are per-lane gathers of 64-byte wide nodes and 48-byte triangle records.  This tool runs librdx's probe kernel
(csrc/kernels.hip k_probe_gather: one record per lane at a uniformly random record index, known count) under
`rocprofv3 --pmc FETCH_SIZE` for tables that live in HBM (8 GiB), in the Infinity Cache (64 MiB) and in L2 (2 MiB), next to
the guide's own streaming case, and prints requested bytes / reported bytes per case:

    python tools/fetch_calibrate.py            # parent: runs the passes, prints the table (json with --json)
"""
import csv, glob, json, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [  # name, record bytes, table bytes, reads
    ("stream16_8GiB", 16, 8 << 30, 1 << 28),
    ("gather64_8GiB", 64, 8 << 30, 1 << 26),
    ("gather48_6GiB", 48, 6 << 30, 1 << 26),
    ("gather64_64MiB", 64, 64 << 20, 1 << 26),
    ("gather48_48MiB", 48, 48 << 20, 1 << 26),
    ("gather64_2MiB", 64, 2 << 20, 1 << 26),
]


def child(name):
    import ctypes
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import rrt_amd  # noqa: F401
    from radiance_ray_tracing_amd import _lib, rd
    rd.Platform.GetPlatform(0)
    L = _lib.lib()
    L.rdx_debug_gather_probe.restype = ctypes.c_float
    L.rdx_debug_gather_probe.argtypes = [ctypes.c_uint32, ctypes.c_ulonglong, ctypes.c_uint32, ctypes.c_uint32]
    for n, rec, table, reads in CASES:
        if n == name:
            ms = L.rdx_debug_gather_probe(rec, table, reads, 3)
            print("probe %s: %.3f ms" % (n, ms), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    out = {}
    for name, rec, table, reads in CASES:
        tmp = tempfile.mkdtemp(prefix="rdx_cal_", dir="/tmp")
        try:
            subprocess.run([prof, "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "p", "--",
                            "python3", os.path.abspath(__file__), "--child", name], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                           timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            f = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
            rows = [r for r in csv.DictReader(open(f[0])) if "k_probe_gather" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
            # one row per dispatch (3 repetitions): the last one (warm TLB)
            per = {}
            for r in rows:
                per.setdefault(r.get("Dispatch_Id") or r.get("Correlation_Id"), 0.0)
                per[r.get("Dispatch_Id") or r.get("Correlation_Id")] += float(r["Counter_Value"])
            last = per[sorted(per, key=lambda k: int(k))[-1]]
            requested = rec * reads
            out[name] = {"record_bytes": rec, "table_bytes": table, "reads": reads, "requested_bytes": requested,
                         "FETCH_SIZE_KiB": last, "reported_bytes": last * 1024.0, "requested_over_reported": round(requested / (last * 1024.0), 4) if last else None}
        except Exception as e:
            out[name] = {"error": str(e)[:200]}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    if "--json" in sys.argv:
        print(json.dumps(out, indent=1))
    else:
        for k, v in out.items():
            print(k, v)


if __name__ == "__main__":
    main()
