#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: one line per kernel."""
import re
import sys


def main(path):
    txt = open(path).read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    for b in blocks:
        name = b.split("\n")[0].strip()

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return m.group(1) if m else "?"

        m = re.search(r"neptune_(\w+?)I", name)
        kind = m.group(1) if m else name[:40]
        body = re.search(r"builtin\d+(\w+?)E", name)
        ints = ",".join(re.findall(r"Li(\d+)E", name))
        flags = "".join(re.findall(r"Lb([01])E", name))
        print("%-16s %-9s ints=%-14s b=%-3s VGPR=%-4s SGPR=%-4s scratch=%-4s occ=%s" % (
            kind, body.group(1) if body else "", ints, flags, g("VGPRs"), g("SGPRs"),
            g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))


if __name__ == "__main__":
    main(sys.argv[1])
