#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs (one directory per pass) into one JSON.

    python3 profiles/pmc_summary.py OUT.json PASS_DIR [PASS_DIR ...]

Each PASS_DIR is the `-d` directory of one `rocprofv3 --pmc ... -- python3 bench.py ...`
run (profiles/collect_pmc.sh writes them under gpurun_out/).  Counters are summed over all
dispatches of a kernel and divided by the number of dispatches (= per launch).  HBM bytes
follow MI355X_MICROARCH.md's gfx950 rule: FETCH_SIZE and WRITE_SIZE are in KiB and
FETCH_SIZE reports half of a streaming read, so bytes = 2 * FETCH_SIZE * 1024 +
WRITE_SIZE * 1024.
"""
import csv
import glob
import json
import os
import sys


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    kernels = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = {}
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = row["Kernel_Name"]
                    e = per.setdefault(k, {"disp": set(), "sum": {}, "ns": {}})
                    e["disp"].add(row["Dispatch_Id"])
                    e["sum"][row["Counter_Name"]] = e["sum"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    e["ns"][row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                    e["meta"] = {"VGPR": row["VGPR_Count"], "AGPR": row["Accum_VGPR_Count"], "SGPR": row["SGPR_Count"],
                                 "LDS": row["LDS_Block_Size"], "scratch": row["Scratch_Size"],
                                 "workgroup": row["Workgroup_Size"], "grid": row["Grid_Size"]}
            for k, e in per.items():
                n = len(e["disp"])
                kk = kernels.setdefault(k, {"launches_per_pass": n, "duration_ms": {}})
                kk.update(e["meta"])
                kk["duration_ms"][os.path.basename(os.path.normpath(d))] = sum(e["ns"].values()) / n / 1e6
                for c, v in e["sum"].items():
                    kk[c] = v / n
    for k, e in kernels.items():
        der = {}
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = int(2 * e["FETCH_SIZE"] * 1024 + e["WRITE_SIZE"] * 1024)
        wc = e.get("SQ_WAVE_CYCLES")
        if wc:
            for name, c in (("wait_frac_of_wave_cycles", "SQ_WAIT_ANY"), ("issue_stall_frac_of_wave_cycles", "SQ_WAIT_INST_ANY"),
                            ("active_frac_of_wave_cycles", "SQ_ACTIVE_INST_ANY")):
                if c in e:
                    der[name] = e[c] / wc
        if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_LDS_IDX_ACTIVE"):
            der["lds_conflict_frac"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e and e["TCC_HIT_sum"] + e["TCC_MISS_sum"] > 0:
            der["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
            if e["duration_ms"]:
                ms = sum(e["duration_ms"].values()) / len(e["duration_ms"])
                # 128-byte L2 requests per second (reads and writes) -- an upper bound on bytes moved
                der["l2_req_128B_TBps"] = e.get("TCC_REQ_sum", 0.0) * 128 / (ms * 1e-3) / 1e12
        if "SQ_INSTS_VALU" in e and e["duration_ms"]:
            ms = sum(e["duration_ms"].values()) / len(e["duration_ms"])
            der["valu_issue_util"] = e["SQ_INSTS_VALU"] / (ms * 1e-3) / 1024 / 1.2e9  # per SIMD, 2 cycles per wave64 op at 2.4 GHz
        if "GRBM_GUI_ACTIVE" in e and e["duration_ms"]:
            ms = sum(e["duration_ms"].values()) / len(e["duration_ms"])
            der["clock_GHz"] = e["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e6)  # the counter sums the 8 XCDs
        if der:
            e["derived"] = der
    with open(out_path, "w") as fh:
        json.dump({"kernels": kernels}, fh, indent=1)
    print(json.dumps({k: v.get("derived", {}) for k, v in kernels.items()}, indent=1))


if __name__ == "__main__":
    main()
