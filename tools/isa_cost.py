#!/usr/bin/env python3
"""Static price list of a kernel's vector instructions on gfx950, from tools/probes/valu_rate_probe.hip (round 4): clocks one SIMD
needs per wave-instruction with its waves saturating it.  FAST (2.6): v_mov / add / sub / and / or / xor / not / lshrrev / ashrrev /
bitop3 / cndmask_e32 / f32 mul, fma — when no operand is an SGPR; everything else 4.2 (shifts LEFT, compares, bfe, ffb*, bcnt, min/max,
24-bit multiplies, every other three-operand VOP3, SDWA, DPP, packed, readlane, cndmask_e64, and ANY instruction with an SGPR source).
usage: isa_cost.py file.s kernel_substring   -> totals per region between s_barrier instructions"""
import re, sys
FAST = {"v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32",
        "v_ashrrev_i32", "v_bitop3_b32", "v_cndmask_b32", "v_mul_f32", "v_fma_f32", "v_add_f32", "v_add_u16", "v_sub_u16", "v_mov_b64"}
def cost(line):
    m = re.match(r"\s*(v_[a-z0-9_]+)\s*(.*)", line)
    if not m: return None
    op, rest = m.group(1), m.group(2)
    rest = rest.split(";")[0]
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    enc = op[len(base):]
    ops = [o.strip() for o in rest.split(",")]
    srcs = ops[1:]
    sg = any(re.match(r"(s\d+|s\[\d+:\d+\]|vcc|exec|vcc_lo|vcc_hi|exec_lo|exec_hi|m0|ttmp)", o) for o in srcs)
    if base == "v_cndmask_b32" and enc in ("", "_e32"):   # the implicit vcc of the e32 form is free
        sg = any(re.match(r"(s\d+|s\[\d+:\d+\])", o) for o in srcs[:2])
    fast = base in FAST and enc in ("", "_e32") and not sg
    if base in ("v_bitop3_b32", "v_mov_b64") and not sg: fast = True
    return (2.6 if fast else 4.2), base, sg
def main():
    path, sub = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sub in l and l.rstrip().endswith(":") or (l.startswith("_Z") and sub in l and ":" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    region, tot, n, nsg, slow_ops = 0, {}, {}, {}, {}
    for l in lines[start:end]:
        if "s_barrier" in l: region += 1
        c = cost(l)
        if c is None: continue
        tot[region] = tot.get(region, 0) + c[0]; n[region] = n.get(region, 0) + 1
        if c[2]: nsg[region] = nsg.get(region, 0) + 1
        if c[0] > 3: slow_ops[c[1] + ("+sgpr" if c[2] else "")] = slow_ops.get(c[1] + ("+sgpr" if c[2] else ""), 0) + 1
    for r in sorted(tot): print(f"region {r:2d}: {n[r]:5d} vector instructions, {tot[r]:8.0f} clocks static, {nsg.get(r,0):4d} with an SGPR source")
    print("total", sum(n.values()), "instructions,", round(sum(tot.values())), "clocks,", sum(nsg.values()), "with SGPR sources")
    print("slow, by kind:", sorted(slow_ops.items(), key=lambda kv: -kv[1])[:40])
main()
