#!/usr/bin/env python3
"""Static check of a hand-scheduled kernel's compiler output (hipcc -save-temps .s file).

The MFMAs of these kernels are inline asm, so the compiler neither knows their latency nor that the
fragment ring (named accumulator registers) carries data between statements.  Two things it may
then do silently break results:
  * touch the VGPR destination of an inline-asm MFMA (a copy when it splits a live range, say) before
    the MFMA has written it -- an error when a compiler-generated instruction reads or writes such
    registers before any further MFMA has been issued (the next MFMA issues when the pipe is free,
    which is when the first one has finished its passes), counted as a note when one MFMA lies
    between.  MFMAs the compiler emitted itself (builtins) are its own business: it knows their latency;
  * move data through the ring registers -- flagged for any compiler-generated instruction that
    names a register at or above --ring-base;
  * touch the VGPR destination of an inline-asm LDS read before a wait: the read returns at once and
    the data lands later; the kernels wait (s_waitcnt lgkmcnt, also inline asm) before THEIR use, but
    a copy the compiler makes in between (to park the value in another register) takes what was in
    the register before -- flagged for any compiler-generated instruction that names such a register
    between the read and the next s_waitcnt with an lgkmcnt field.
Exit status 1 when anything is flagged.  usage: check_mfma_hazards.py file.s [--kernel substring] [--ring-base 224]"""
import re
import sys


def regs(text, file):
    out = set()
    for m in re.finditer(r"\b%s\[(\d+):(\d+)\]|\b%s(\d+)\b" % (file, file), text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check(path, kernel_filter, ring_base):
    src = open(path).read()
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", src, re.M)]
    bad = 0
    for n, (st, name) in enumerate(starts):
        if kernel_filter not in name:
            continue
        en = starts[n + 1][0] if n + 1 < len(starts) else len(src)
        lines = src[st:en].split("\n")
        in_asm = False
        rows = []  # (text, from_asm)
        for l in lines:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            rows.append((t.split(";")[0].strip(), in_asm))
        pending = []  # [regs, mfmas_seen, text]
        landing = []  # [regs, text] of asm LDS reads into VGPRs that no wait has covered yet
        notes = 0
        for t, from_asm in rows:
            if t.startswith("s_waitcnt") and "lgkmcnt" in t:
                landing = []
            if not from_asm:
                touched_l = regs(t, "v")
                for q in landing:
                    if touched_l & q[0]:
                        print(f"{name}: '{t}' touches the destination of '{q[1]}' before any wait for it")
                        bad += 1
            elif t.startswith("ds_read") or t.startswith("ds_bpermute"):
                d = regs(t.split(None, 1)[1].split(",")[0], "v")
                if d:
                    landing.append([d, t])
            if not from_asm:
                if ring_base is not None and any(r >= ring_base for r in regs(t, "a")):
                    print(f"{name}: compiler-generated instruction touches the ring: {t}")
                    bad += 1
                touched = regs(t, "v")
                for p in pending:
                    if touched & p[0]:
                        if p[1] == 0:
                            print(f"{name}: '{t}' touches the destination of '{p[2][:70]}...' right behind it")
                            bad += 1
                        else:
                            notes += 1
            if t.startswith("v_mfma"):
                for p in pending:
                    p[1] += 1
                pending = [p for p in pending if p[1] < 2]
                dst = t.split(None, 1)[1].split(",")[0]
                d = regs(dst, "v")
                if d and from_asm:
                    pending.append([d, 0, t])
        if notes:
            print(f"{name}: {notes} compiler-generated access(es) to an asm MFMA's destination one MFMA later (note)")
    return bad


if __name__ == "__main__":
    args = sys.argv[1:]
    path = args[0]
    kf = args[args.index("--kernel") + 1] if "--kernel" in args else ""
    rb = int(args[args.index("--ring-base") + 1]) if "--ring-base" in args else None
    n = check(path, kf, rb)
    print(f"{path}: {n} finding(s)")
    sys.exit(1 if n else 0)
