#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate passes, as
MI355X_MICROARCH.md prescribes) -> profiles/<name>.json, the file bench.py's roofline.traffic reads.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [note]

Counter units are KB (x1024 B), summed over the counter's instances, averaged per dispatch of each kernel.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the 128-B requests of a wide coalesced
stream as 64 B, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is taken as is."""
import csv, glob, json, os, sys
from collections import defaultdict

KERNELS = {"softattn_rt_kernel": "softattn_rt_kernel", "softattn_kernel": "softattn_kernel", "maxpath_pipelined_kernel": "maxpath_pipelined_kernel",
           "expand_kernel": "expand_kernel",
           "mobo_forward_kernel": "mobo_forward_kernel"}


def per_kernel(d, counter):
    acc = defaultdict(lambda: defaultdict(float))          # kernel -> dispatch -> value
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for key, pat in KERNELS.items():
                if pat in r["Kernel_Name"]:
                    acc[key][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in acc.items() if v}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    res = {"note": (note + " " if note else "") + __doc__.split("\n\n")[2].replace("\n", " "), "kernels": {}}
    for k in sorted(set(fe) | set(wr)):
        f, w = fe.get(k, 0.0), wr.get(k, 0.0)
        rd = int(round(2 * f * 1024))
        wb = int(round(w * 1024))
        res["kernels"][k] = {"FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1), "hbm_read_bytes_corrected": rd,
                             "hbm_write_bytes": wb, "hbm_bytes": rd + wb}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main()
