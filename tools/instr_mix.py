#!/usr/bin/env python3
"""Development aid (no GPU): static instruction mix of one kernel in a `hipcc --cuda-device-only -S` listing.
    tools/instr_mix.py listing.s _ZN3ssg11step_kernelILi8ELi256ELb1ELb0ELb0EE"""
import collections, re, sys
txt = open(sys.argv[1]).read()
name = sys.argv[2]
i = txt.index("\n" + name)
j = txt.index("s_endpgm", i)
lines = [l.strip() for l in txt[i:j].split("\n")[1:] if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
c = collections.Counter(l.split()[0] for l in lines)
valu = sum(v for k, v in c.items() if k.startswith("v_"))
print("static instructions %d (VALU %d): v_fma_f64 %d v_mul_f64 %d v_add_f64 %d ds_* %d s_waitcnt %d" % (
    len(lines), valu, c["v_fma_f64"], c["v_mul_f64"], c["v_add_f64"], sum(v for k, v in c.items() if k.startswith("ds_")), c["s_waitcnt"]))
