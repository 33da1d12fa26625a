#!/usr/bin/env python3
"""tools/kres.py remarks.txt [remarks2.txt]: per-kernel resource table from `hipcc -Rpass-analysis=kernel-resource-usage` output
(VGPRs, scratch bytes per lane, SGPR / VGPR spills, waves per SIMD, LDS); with two files, old -> new."""
import re
import sys


def parse(f):
    out, cur = {}, None
    keys = {"VGPRs": "vgpr", "ScratchSize [bytes/lane]": "scratch", "SGPRs Spill": "sspill", "VGPRs Spill": "vspill",
            "Occupancy [waves/SIMD]": "occ", "LDS Size [bytes/block]": "lds", "TotalSGPRs": "sgpr"}
    for l in open(f, errors="ignore"):
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        for k, short in keys.items():
            m = re.search(r"remark:\s+" + re.escape(k) + r": (\d+)", l)
            if m and cur:
                out[cur][short] = int(m.group(1))
    return out


def main():
    tabs = [parse(f) for f in sys.argv[1:3]]
    pat = sys.argv[3] if len(sys.argv) > 3 else "trace_kernel|debug_render|trace_debug|seed_seg"
    for k in tabs[-1]:
        if not re.search(pat, k):
            continue
        name = re.sub(r"N2hr.*", "", k)
        if len(tabs) == 2:
            print("%-44s %s\n%44s -> %s" % (name, tabs[0].get(k), "", tabs[1][k]))
        else:
            print("%-44s %s" % (name, tabs[0][k]))


if __name__ == "__main__":
    main()
