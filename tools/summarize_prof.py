"""Condense the rocprofv3 CSV output of tools/profile_bench.sh into one JSON: per-kernel launch durations, counter sums,
and the derived figures bench.py / DESIGN.md quote for the dominant kernel of the run.

HBM bytes follow /opt/skills/guides (MI355X_MICROARCH.md, HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are in KiB (x 1024);
on gfx950 FETCH_SIZE reports exactly half of a wide coalesced stream (16 B per lane), so it is DOUBLED for the kernels whose
reads are of that shape and left as reported (flagged) otherwise; WRITE_SIZE is uncalibrated and taken as reported."""
import csv, glob, json, os, sys
from collections import defaultdict

root, cfg, cmd = sys.argv[1], int(sys.argv[2]), (sys.argv[3] if len(sys.argv) > 3 else "")
out = {"command": cmd}

# kernels whose global reads are 16-byte-per-lane streams (the guide's calibrated case)
WIDE_READ_KERNELS = ("nuts_gauss_", "gemm_step_kernel")      # per-lane rows moved 16 bytes per instruction (gemm_step_kernel: its tiles; the epilogue reads 8 bytes per lane)


def rows(path):
    with open(path, newline="") as f:
        yield from csv.DictReader(f)


stats = []
for p in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in rows(p):
        stats.append({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
out["kernel_stats"] = sorted(stats, key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))[:6]

for group in ("fetch", "write", "sq"):
    agg = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for p in glob.glob(os.path.join(root, group, "**", "*counter_collection.csv"), recursive=True):
        for r in rows(p):
            name = r.get("Kernel_Name") or r.get("Kernel Name") or "?"
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[name].add(r.get("Dispatch_Id"))
    out[group] = {k[:120]: {"launches": len(launches[k]), "counters_sum_over_launches": dict(v)}
                  for k, v in agg.items() if "mi::" in k}

if out["kernel_stats"]:
    top = out["kernel_stats"][0]
    name = top["Name"]
    key = name[:120]
    d = {"kernel": name, "launches": int(top["Calls"]), "kernel_ms_avg": float(top["AverageNs"]) / 1e6}

    def per_launch(group, counter):
        g = out.get(group, {}).get(key)
        return None if not g or counter not in g["counters_sum_over_launches"] else g["counters_sum_over_launches"][counter] / g["launches"]

    f, w = per_launch("fetch", "FETCH_SIZE"), per_launch("write", "WRITE_SIZE")
    wide = any(k in name for k in WIDE_READ_KERNELS)
    if f is not None and w is not None:
        d["fetch_factor"] = 2.0 if wide else 1.0
        d["hbm_read_bytes_per_launch"] = f * 1024.0 * d["fetch_factor"]
        d["hbm_write_bytes_per_launch"] = w * 1024.0
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]
        d["hbm_TBps"] = d["hbm_bytes_per_launch"] / (d["kernel_ms_avg"] * 1e-3) / 1e12
    busy, gui = per_launch("sq", "SQ_VALU_MFMA_BUSY_CYCLES"), per_launch("sq", "GRBM_GUI_ACTIVE")
    if busy is not None and gui:
        d["mfma_busy_frac"] = busy / (gui / 8.0 * 1024.0)       # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
    v = per_launch("sq", "SQ_INSTS_VALU")
    if v is not None:
        d["valu_insts_per_launch"] = v
    wa = per_launch("sq", "SQ_WAIT_INST_ANY")
    if wa is not None:
        d["wait_inst_any_per_launch"] = wa
    # the combined-pipe budget: fp64 VALU and fp64 MFMA issue from the same pipe of a SIMD, so what a
    # kernel's instruction mix allows is  [ 4 cycles x VALU wave-instructions + MFMA busy cycles ] / [ SIMDs x kernel cycles ]  (a wave64 VALU
    # instruction occupies its SIMD for 4 cycles at least; every VALU instruction is priced at that minimum, so this is a LOWER bound of the
    # pipe's occupancy)
    if v is not None and busy is not None and gui:
        simd_cycles = gui / 8.0 * 1024.0
        d["pipe_budget"] = {"valu_issue_frac": 4.0 * v / simd_cycles, "mfma_busy_frac": busy / simd_cycles,
                            "pipe_frac": (4.0 * v + busy) / simd_cycles,
                            "what": "(4 x SQ_INSTS_VALU + SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), per launch"}
    d["note"] = ("FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE x2 only for 16-B-per-lane read streams (guide's gfx950 correction), else as "
                 "reported; counters summed over launches / launch count; one rocprofv3 run per counter group, --kernel-trace only")
    out["derived"] = d
    try:        # the workload this profile is of: what bench.py matches before quoting `traffic`
        line = json.load(open(os.path.join(root, "bench_line.json")))
        c = line["config"]
        out["workload_key"] = [cfg, c["chains_per_gpu"], c["d"], c["n_burnin_draws"] + c["n_keep_draws"]]
        out["bench_line"] = {k: line[k] for k in ("value", "ms_per_step", "roofline") if k in line}
    except (OSError, ValueError, KeyError):
        pass
print(json.dumps(out, indent=1))
