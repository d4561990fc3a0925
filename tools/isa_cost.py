"""Static issue-cycle estimate of a range of basic blocks in a hipcc -S listing, with the per-instruction costs
measured by tools/valu_issue_bench (profiles/r02_valu_issue_rates.txt): wave64 VALU instructions hold their SIMD
for 2 cycles (mov, and, or, xor, not, add, sub, lshr, ashr, bitop3, fp32 add/mul/fma) or 4 (everything else:
lshl, min/max, ffbh, bfe, mul, perm, every 3-operand integer VOP3, v_pk_*, SDWA / DPP forms, v_cmp, v_cndmask, readlane).

    python tools/isa_cost.py file.s kernel_substr [first_label last_label]
"""
import re
import sys

TWO = {"v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
       "v_lshrrev_b32", "v_ashrrev_i32", "v_bitop3_b32", "v_add_f32", "v_mul_f32", "v_fma_f32", "v_sub_f32", "v_nop",
       "v_add_i32", "v_sub_i32", "v_xnor_b32"}


def cost(op, line):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.endswith("_sdwa") or op.endswith("_dpp") or "row_" in line or "quad_perm" in line or " clamp" in line:
        return 4
    return 2 if base in TWO else 4


def main():
    s = open(sys.argv[1]).read()
    key = sys.argv[2]
    i = s.index(key)
    i = s.index(":\n", i)
    k = s[i:]
    k = k[:k.index(".Lfunc_end")]
    lines = k.split("\n")
    first = sys.argv[3] if len(sys.argv) > 3 else None
    last = sys.argv[4] if len(sys.argv) > 4 else None
    on = first is None
    tot = {"valu_n": 0, "valu_cyc": 0, "salu": 0, "lds": 0, "vmem": 0, "smem": 0, "nop": 0}
    hist = {}
    for l in lines:
        t = l.split(";")[0].strip()
        if not t:
            continue
        if t.endswith(":"):
            lab = t[:-1]
            if first and lab == first:
                on = True
            elif last and lab == last:
                on = False
            continue
        if not on or t.startswith("."):
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            c = cost(op, t)
            tot["valu_n"] += 1
            tot["valu_cyc"] += c
            hist[op] = hist.get(op, 0) + 1
        elif op.startswith("s_nop"):
            tot["nop"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer"):
            tot["smem"] += 1
        elif op.startswith("s_"):
            tot["salu"] += 1
        elif op.startswith("ds_"):
            tot["lds"] += 1
            hist[op] = hist.get(op, 0) + 1
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
            tot["vmem"] += 1
    print(tot)
    for op, n in sorted(hist.items(), key=lambda kv: -kv[1]):
        print("  %-28s %4d  (%d cyc each)" % (op, n, cost(op, "") if op.startswith("v_") else 0))


if __name__ == "__main__":
    main()
