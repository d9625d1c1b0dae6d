#!/usr/bin/env python3
"""Registers, spills, scratch and occupancy of every kernel in libaligner_amd.so, from the compiler's resource report the
build keeps beside each object (aligner_amd/lib/obj/*.resources.txt, `-Rpass-analysis=kernel-resource-usage`).

    python tools/kernel_resources.py [out.txt]        (default: stdout)
"""
import glob, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "aligner_amd", "lib", "obj")
FIELDS = ["VGPRs", "AGPRs", "TotalSGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]",
          "LDS Size [bytes/block]"]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
        return out.split("\n")[:len(names)]
    except OSError:
        return names


def parse(path):
    """[(mangled name, {field: int})] in file order."""
    kernels, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = (m.group(1), {})
            kernels.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z \[\]/]*): (\d+) \[-Rpass-analysis", line)
        if m and cur is not None:
            cur[1][m.group(1).strip()] = int(m.group(2))
    return kernels


def table():
    rows = []
    for f in sorted(glob.glob(os.path.join(OBJ, "*.resources.txt"))):
        ks = parse(f)
        names = demangle([k for k, _ in ks])
        for (mangled, v), name in zip(ks, names):
            if name and name != mangled:
                name = re.sub(r"^void ", "", name)
                name = re.sub(r"\(aligner::\w+Params\)$|\(.*\)$", "", name).replace("aligner::", "")
            else:
                name = mangled
            rows.append((os.path.basename(f).replace(".resources.txt", ""), name, v))
    return rows


def main():
    rows = table()
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    print("# kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage, gfx950): VGPR | AGPR | SGPR | VGPR spills | SGPR spills | scratch B/lane | waves/SIMD", file=out)
    for src, name, v in rows:
        print("%-16s %-88s %4d %4d %4d %5d %5d %6d %3d" % (src, name[:88], v.get("VGPRs", -1), v.get("AGPRs", 0), v.get("TotalSGPRs", -1),
              v.get("VGPRs Spill", 0), v.get("SGPRs Spill", 0), v.get("ScratchSize [bytes/lane]", 0), v.get("Occupancy [waves/SIMD]", 0)), file=out)
    bad = [(s, n, v) for s, n, v in rows if v.get("ScratchSize [bytes/lane]", 0) or v.get("VGPRs Spill", 0)]
    print("# kernels with scratch or VGPR spills: %d of %d" % (len(bad), len(rows)), file=out)
    for s, n, v in bad:
        print("#   %s %s: %d VGPRs spilled, %d B/lane" % (s, n[:100], v.get("VGPRs Spill", 0), v.get("ScratchSize [bytes/lane]", 0)), file=out)


if __name__ == "__main__":
    main()
