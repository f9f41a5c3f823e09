#!/usr/bin/env python3
"""Remove the wait states hipcc's hazard recogniser puts after the multiply-add chains of fp29_mac.hpp.

The field multiplications of the point kernels issue their v_mad_u64_u32 runs as inline-asm statements (the compiler
would otherwise re-associate the column sums, fp29.hpp).  LLVM treats an inline-asm statement as a possible producer of a
"destination-select forwarding" hazard (gfx940+: an SDWA / op_sel write followed by a read of the same VGPR needs one
wait state) and conservatively puts `s_nop 0` in front of every instruction that reads a VGPR the statement wrote -- i.e.
after every run, ~60 per multiplication.  Our statements contain v_mad_u64_u32 only, which has no such hazard: plain
VALU -> VALU dependencies are interlocked by the hardware.  Measured on MI355X (tools/nonop): a multiplication is 19 %
faster without them, results identical.

Only the pattern  v_mad_u64_u32 ; s_nop 0 ; {v_mad_u64_u32 | v_lshrrev_b64 | v_mul_lo_u32 | v_and_b32}  is touched; every
other s_nop (transcendental results, VCC / SGPR / lane-select hazards, memory operations) stays.

    strip_hazard_nops.py in.s out.s
"""
import sys

PRODUCER = "v_mad_u64_u32"
CONSUMERS = {"v_mad_u64_u32", "v_lshrrev_b64", "v_mul_lo_u32", "v_and_b32_e32"}


def opcode(line):
    t = line.strip()
    if not t or t[0] in ";." or t.endswith(":"):
        return None
    return t.split()[0]


def main():
    lines = open(sys.argv[1]).read().split("\n")
    ops = [opcode(l) for l in lines]
    real = [i for i, o in enumerate(ops) if o is not None]
    pos = {i: k for k, i in enumerate(real)}
    removed = kept = 0
    out = list(lines)
    for i in real:
        if lines[i].strip() != "s_nop 0":
            continue
        k = pos[i]
        prev_op = ops[real[k - 1]] if k > 0 else None
        next_op = ops[real[k + 1]] if k + 1 < len(real) else None
        # the three instructions must be consecutive in one basic block: no label between them
        lo, hi = real[k - 1] if k > 0 else i, real[k + 1] if k + 1 < len(real) else i
        if any(lines[j].strip().endswith(":") for j in range(lo, hi + 1)):
            kept += 1
            continue
        if prev_op == PRODUCER and next_op in CONSUMERS:
            out[i] = "\t; s_nop 0 removed (inline-asm multiply-add run: no hazard)"
            removed += 1
        else:
            kept += 1
    open(sys.argv[2], "w").write("\n".join(out))
    print(f"strip_hazard_nops: {removed} removed, {kept} kept", file=sys.stderr)


if __name__ == "__main__":
    main()
