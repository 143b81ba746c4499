#!/usr/bin/env python3
"""List the COMPILER-inserted `s_waitcnt vmcnt(0)` instructions that sit inside a loop of a kernel (device asm from
`hipcc --cuda-device-only -S`).  The hand-scheduled kernels keep LDS-DMA requests in flight across their tap / step
barriers with counted waits written as inline asm; a vmcnt(0) the compiler adds inside the loop (e.g. in front of the first
use of a register an ordinary load before the loop produced) drains exactly those requests on every trip.

usage: asm_loop_waits.py file.s [name-filter]"""
import re
import sys


def kernels(text):
    for m in re.finditer(r"^(_Z\S+):\s*; @\S+\n", text, re.M):
        end = text.find(".Lfunc_end", m.end())
        yield m.group(1), text[m.end():end].split("\n")


def main():
    text = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, lines in kernels(text):
        if flt not in name:
            continue
        # loop membership from the compiler's own block comments ("=>This Inner Loop Header", "in Loop: Header=BBx_y"); a block without a
        # label of its own continues the previous one
        cur = None
        member = []
        for l in lines:
            if l.startswith(".LBB"):
                m = re.search(r"Header=(BB\w+)", l)
                if "Loop Header" in l:
                    cur = l.split(":")[0][2:]
                elif m:
                    cur = m.group(1)
                else:
                    cur = None
            member.append(cur)
        main = set()
        for hdr in set(x for x in member if x):
            seg = [l for l, mem in zip(lines, member) if mem == hdr]
            if any("v_mfma" in x for x in seg) and any(" lds" in x and "load" in x for x in seg):
                main.add(hdr)
        in_asm = False
        hits = []
        for k, l in enumerate(lines):
            if "#ASMSTART" in l:
                in_asm = True
            elif "#ASMEND" in l:
                in_asm = False
            elif not in_asm and re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l) and member[k] in main:
                hits.append((k, l.strip(), member[k]))
        n_mfma = sum("v_mfma" in l for l in lines)
        if hits:
            print(f"{name[:150]}  (mfma {n_mfma})")
            for k, l, d in hits:
                print(f"    line {k}: {l}  [loop {d}]")


if __name__ == "__main__":
    main()
