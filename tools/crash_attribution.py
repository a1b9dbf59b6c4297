"""Name the libraries behind the unsymbolised frames of a glog crash dump (rocprofv3 --pmc SIGSEGVs of round 2).

The crashed processes are gone, but (a) every dump contains glibc's signal trampoline (__restore_rt, libc + 0x42520 on
this image), which gives the crashed process's libc base, and (b) the libraries that are mapped at program start lie at
fixed distances from libc for one command line on one image -- the three crashes show identical distances.  So a
/proc/self/maps of the SAME command (bench.py under rocprofv3's LD_PRELOADs; SVDQ_DEBUG_MAPS=<file> makes bench.py write
it) translates `frame - libc_base_then` into library + offset.

usage: crash_attribution.py <maps file> <crash log> [<crash log> ...]"""
import re
import sys

RESTORE_RT = 0x42520


def load_maps(path):
    ent = []
    for line in open(path):
        f = line.split()
        a, b = (int(x, 16) for x in f[0].split("-"))
        ent.append((a, b, f[1], int(f[2], 16), f[5] if len(f) >= 6 else "[anon]"))
    libc = min(e[0] for e in ent if "libc.so.6" in e[4])
    return ent, libc


def where(ent, libc, off):
    a = libc + off
    for s, e, perm, fo, name in ent:
        if s <= a < e:
            return f"{name.split('/')[-1]} + {a - s + fo:#x} ({perm})"
    return "not mapped in the reference layout"


def main():
    ent, libc = load_maps(sys.argv[1])
    for log in sys.argv[2:]:
        text = open(log, errors="replace").read()
        frames = [int(x, 16) for x in re.findall(r"^\s+@\s+(0x[0-9a-f]+)", text, re.M)]
        fault = re.search(r"SIGSEGV \(@(0x[0-9a-f]+)\)", text)
        rr = [f for f in frames if (f & 0xfff) == (RESTORE_RT & 0xfff)]
        if not rr:
            print(f"{log}: no signal trampoline frame found")
            continue
        base = rr[0] - RESTORE_RT
        print(f"{log}: libc base of the crashed process {base:#x}")
        for f in frames:
            w = where(ent, libc, f - base)
            # libraries dlopen'ed late (torch, MIOpen ...) land at run-dependent distances: only frames that fall on an
            # executable mapping of the reference layout are attributions; the rest are printed as they are
            print(f"    {f:#016x}  libc{f - base:+#x}  {w if '(r-xp)' in w else '(late-loaded library or the interpreter: see the symbolised frames of the dump)'}")
        if fault:
            fa = int(fault.group(1), 16)
            print(f"    fault address {fa:#x}: 1 MiB aligned = {fa % (1 << 20) == 0}; {where(ent, libc, fa - base)}")


if __name__ == "__main__":
    main()
