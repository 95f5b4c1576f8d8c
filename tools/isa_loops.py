"""Static look at the loops of one kernel in hipcc's gfx950 assembly (-S): for every loop (a label that a later branch jumps back
to) its instruction counts - MFMA, LDS-DMA, scratch (spill) accesses, `s_waitcnt vmcnt(0)` (compiler-inserted drains of the
vector-memory queue; inline-asm waits appear between ;;#ASMSTART / ;;#ASMEND and are reported separately).

  python tools/isa_loops.py file.s kernel_name_substring [...]
"""
import re
import sys


def kernel_bodies(text):
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        out[m.group(1)] = m.group(2).split("\n")
    return out


def loops(lines):
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    res = []
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            res.append((labels[m.group(1)], i))
    # merge back-edges to the same header: the loop is header .. last back-edge
    by_head = {}
    for h, e in res:
        by_head[h] = max(by_head.get(h, h), e)
    return sorted(by_head.items())


def stats(lines, lo, hi):
    body = lines[lo:hi + 1]
    in_asm, own, comp0, compn = False, [], 0, []
    for l in body:
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
        if m:
            n = int(m.group(1))
            if in_asm:
                own.append(n)
            elif n == 0:
                comp0 += 1
            else:
                compn.append(n)
    return dict(lines=len(body), mfma=sum("v_mfma" in l for l in body), lds_dma=sum(" lds" in l and "buffer_load" in l for l in body),
                scratch=sum("scratch_" in l for l in body), compiler_vmcnt0=comp0, compiler_vmcnt=sorted(set(compn)), own_vmcnt=sorted(set(own)),
                barriers=sum("s_barrier" in l for l in body))


def report(path, pats):
    text = open(path).read()
    rows = []
    for name, lines in kernel_bodies(text).items():
        if pats and not any(p in name for p in pats):
            continue
        total = stats(lines, 0, len(lines) - 1)
        ls = loops(lines)
        rows.append((name, total, [(lo, hi, stats(lines, lo, hi)) for lo, hi in ls]))
    return rows


if __name__ == "__main__":
    for name, total, ls in report(sys.argv[1], sys.argv[2:]):
        print(name, "whole kernel:", total)
        for lo, hi, st in ls:
            print("   loop lines %d-%d: %s" % (lo, hi, st))
