#!/usr/bin/env python3
"""Instruction mix of the MFMA-carrying loops of one kernel in a hipcc -S listing.
usage: hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only x.hip -o x.s ; tools/isa_loop.py x.s <mangled-name substring>"""
import re
import sys
from collections import Counter


def main():
    s = open(sys.argv[1]).read()
    pat = sys.argv[2]
    rx = re.compile(r"^(_Z\S*" + re.escape(pat) + r"\S*):[^\n]*\n(.*?)^\.Lfunc_end", re.S | re.M)
    for m in rx.finditer(s):
        name, body = m.group(1), m.group(2)
        lines = [l.split(";")[0].strip() for l in body.split("\n")]
        lines = [l for l in lines if l and not (l.startswith(".") and not l.endswith(":"))]
        labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
        loops = []
        for i, l in enumerate(lines):
            mm = re.match(r"s_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)", l)
            if mm:
                t = mm.group(1) or mm.group(2)
                if t in labels and labels[t] < i:
                    loops.append((labels[t], i))
        print("==", name[:100])
        for a, b in loops:
            blk = [l for l in lines[a:b + 1] if not l.endswith(":")]
            c = Counter(l.split()[0] for l in blk)
            nm = sum(v for k, v in c.items() if k.startswith("v_mfma"))
            if nm == 0:
                continue
            valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
            ds = sum(v for k, v in c.items() if k.startswith("ds_"))
            sal = sum(v for k, v in c.items() if k.startswith("s_"))
            print(f" loop [{a},{b}] {len(blk)} instrs: mfma {nm}, valu {valu}, ds {ds}, salu {sal}")
            print("   ", dict(c.most_common(30)))


if __name__ == "__main__":
    main()
