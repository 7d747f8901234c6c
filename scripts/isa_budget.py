#!/usr/bin/env python3
"""Static issue budget of one kernel from its gfx950 assembly: every basic block with its
instructions sorted by the unit that executes them and priced in cycles (the constants of
/opt/skills/guides/MI355X_MICROARCH.md: a wave64 VALU instruction occupies its SIMD-32 for 2
cycles, an f64 one for 4, transcendental 4; LDS-array cycles per ds_ instruction from the LDS
table; the scalar unit one instruction per cycle).

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o span.s sequali_amd/csrc/sq_span.hip
    python scripts/isa_budget.py span.s 'k_spanILi5ELb1ELb0ELi3ELb1' [--min 40]

prints, per block of at least --min instructions: its label, line span, and VALU / SALU /
LDS / VMEM / other counts and cycles.  DESIGN.md 5.0 "the headline's issue budget" uses it."""
import re
import sys
from collections import Counter

LDS_CYCLES = {  # LDS-array cycles per wave-instruction, conflict-free (guide, LDS table)
    "ds_read_b32": 2, "ds_read_b64": 2, "ds_read_b128": 4, "ds_read_b96": 8, "ds_read2_b32": 4, "ds_read2_b64": 8,
    "ds_read_u8": 2, "ds_read_u16": 2, "ds_read_i8": 2, "ds_read_u8_d16": 2, "ds_read_u8_d16_hi": 2,
    "ds_read_b64_tr_b8": 2, "ds_read_b64_tr_b16": 2,
    "ds_write_b8": 4, "ds_write_b16": 4, "ds_write_b32": 4, "ds_write_b64": 6, "ds_write2_b32": 6, "ds_write_b96": 10,
    "ds_write_b128": 13, "ds_write2_b64": 13, "ds_write_addtid_b32": 2,
    "ds_bpermute_b32": 4, "ds_permute_b32": 4, "ds_swizzle_b32": 2,
}
F64 = re.compile(r"^v_(add|mul|fma|max|min|ldexp|frexp|cvt_f64|cvt_.*_f64|trunc|floor|ceil|rndne|fract|div_|rcp|rsq|sqrt)_?.*f64")
TRANS = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_f(16|32)")


def classify(op: str):
    """(unit, cycles the unit is occupied by one wave-instruction)"""
    if op.startswith("v_"):
        if "f64" in op or op in ("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_mad_u64_u32", "v_mad_i64_i32",
                                  "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32"):
            return "valu", 4      # quarter rate on the SIMD-32's 32-bit lanes: f64 and the full-width integer multiplies
        if TRANS.match(op):
            return "valu", 4
        if op.startswith("v_mfma"):
            return "mfma", 16
        return "valu", 2
    if op.startswith("ds_"):
        base = op
        if base.startswith("ds_add") or base.startswith("ds_max") or base.startswith("ds_min") or base.startswith("ds_or") or base.startswith("ds_and"):
            return "lds", 4
        return "lds", LDS_CYCLES.get(base, 4)
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", 4
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_store") or op.startswith("s_memtime") or op.startswith("s_dcache"):
        return "smem", 1
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier") or op.startswith("s_sleep") or op.startswith("s_setprio"):
        return "wait", 1
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_endpgm") or op.startswith("s_setpc") or op.startswith("s_swappc"):
        return "branch", 1
    if op.startswith("s_"):
        return "salu", 1
    return "other", 1


def blocks_of(lines):
    """[(label, first line, [ops])] -- a block ends at a label or behind a branch"""
    out, cur, label, first = [], [], "entry", 0
    for i, ln in enumerate(lines):
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"^(\.LBB\w+):", t)
            if m:
                if cur:
                    out.append((label, first, cur))
                cur, label, first = [], m.group(1), i
            continue
        op = t.split()[0]
        if not re.match(r"^[sv]_|^ds_|^global_|^buffer_|^flat_|^scratch_", op):
            continue
        cur.append((op, t))
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm")):
            out.append((label, first, cur))
            cur, label, first = [], label + "+", i + 1
    if cur:
        out.append((label, first, cur))
    return out


def budget(ops):
    n, cyc = Counter(), Counter()
    for op, _ in ops:
        unit, c = classify(op)
        n[unit] += 1
        cyc[unit] += c
    return n, cyc


def main():
    path, pattern = sys.argv[1], sys.argv[2]
    at_least = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 40
    detail = sys.argv[sys.argv.index("--detail") + 1] if "--detail" in sys.argv else None
    text = open(path).read().split("\n")
    start = next(i for i, ln in enumerate(text) if re.match(r"^_Z\w*" + re.escape(pattern) + r"\w*:", ln))
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    print(text[start].split(":")[0], "lines", start + 1, "-", end + 1)
    total_n, total_c = Counter(), Counter()
    for label, first, ops in blocks_of(text[start:end + 1]):
        n, c = budget(ops)
        total_n += n
        total_c += c
        if len(ops) >= at_least or (detail and label == detail):
            print(f"{label:16s} @{start + first + 1:7d} {len(ops):5d} instr | " + " ".join(
                f"{u} {n[u]}/{c[u]}c" for u in ("valu", "salu", "lds", "vmem", "smem", "wait", "branch") if n[u]))
        if detail and label == detail:
            hist = Counter(op for op, _ in ops)
            for op, k in hist.most_common():
                print(f"      {k:4d} {op}  ({classify(op)[0]} {classify(op)[1]}c)")
    print("whole kernel:", dict(total_n))


if __name__ == "__main__":
    main()
