#!/usr/bin/env python3
"""SQ issue counters per launch from rocprofv3 --pmc passes -> profiles/<name>.json (bench.py reads the search kernel's
wave occupancy from it).

    python tools/pmc_issue.py <out.json> <note> <pmc_dir> [<pmc_dir> ...]

Each <pmc_dir> is the output of one `rocprofv3 --pmc <counters> --output-format csv -d <pmc_dir> -- python3 bench.py ...`
pass (counters in their own runs, never with a trace: MI355X_MICROARCH.md, rocprofv3 PMC slots: SQ 8 per pass, GRBM 2).
Values are summed over the counter's instances (whole GPU) and averaged over the dispatches of each kernel.

Units (same guide, cycle constants): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count QUAD-cycles per wave;
GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the cycles the dispatch kept the chip busy.  Derived here:

    kernel_cycles   = GRBM_GUI_ACTIVE / 8
    resident_waves  = 4 * SQ_WAVE_CYCLES / kernel_cycles            (mean waves resident while the kernel runs)
    wave_occupancy  = resident_waves / 8192                          (256 CUs x 32 wave slots)
    wait_any_share  = SQ_WAIT_ANY / SQ_WAVE_CYCLES                    (wave-cycles parked in s_waitcnt / s_barrier)
    issue_stall     = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
    valu_per_wave   = SQ_INSTS_VALU / SQ_WAVES
    mfma_busy_share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * kernel_cycles)
"""
import csv, glob, json, os, sys
from collections import defaultdict

KERNELS = ["softattn_rt_kernel", "softattn_kernel", "maxpath_pipelined_kernel", "expand_kernel", "conv_gemm_kernel", "conv1d_prepared_kernel",
           "conv_split_kernel", "conv_narrow_fused_kernel", "conv_narrow_kernel", "mobo_chain_map_kernel"]


def collect(dirs):
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))     # kernel -> counter -> dispatch -> value
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                for k in KERNELS:
                    if k in r["Kernel_Name"]:
                        acc[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
                        break
    return {k: {c: sum(v.values()) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    out, note, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    res = {"note": note + "  " + " ".join(__doc__.split("\n\n")[2].split()), "kernels": {}}
    for k, c in sorted(collect(dirs).items()):
        e = {n: int(round(v)) for n, v in sorted(c.items())}
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if cyc and wc:
            e["kernel_cycles"] = int(round(cyc))
            e["resident_waves"] = round(4.0 * wc / cyc, 1)
            e["wave_occupancy"] = round(4.0 * wc / cyc / 8192.0, 4)
        if wc and "SQ_WAIT_ANY" in c:
            e["wait_any_share"] = round(c["SQ_WAIT_ANY"] / wc, 4)
        if wc and "SQ_WAIT_INST_ANY" in c:
            e["issue_stall_share"] = round(c["SQ_WAIT_INST_ANY"] / wc, 4)
        if c.get("SQ_WAVES") and "SQ_INSTS_VALU" in c:
            e["valu_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
        if cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            e["mfma_busy_share"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4)
        if c.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in c:
            e["lds_bank_conflict_share"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
        res["kernels"][k] = e
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main()
