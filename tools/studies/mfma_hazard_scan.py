"""Static scan of a gfx950 ISA listing (hipcc -S --cuda-device-only) for the software-managed hazard "XDL (MFMA) write of D -> VALU
read / write of the same registers": the hardware does not interlock it, the compiler's hazard recogniser must keep N independent
issue states between the two (passes + 3 on gfx950: 4-pass 16x16x32 -> 7, 8-pass 32x32x16 -> 11, 16-pass fp32 32x32x2 -> 19 -- LLVM
GCNHazardRecognizer::checkMAIVALUHazards; MI355X guide: "8-pass XDL: 12 states").  For every v_mfma the script walks forward inside the kernel body until the first instruction that touches a
destination register (skipping MFMAs that take D whole as their C operand: the accumulate chain needs 0) and records the number of
states in between, separately for packed-fp32 consumers (v_pk_*_f32) and all others.

Purpose: fine_match.hip's run-to-run deviations vanished when the SLP vectoriser's v_pk_* instructions did.  If the
compiler under-counted the wait states for PACKED consumers of an MFMA result, that would be the mechanism (a stale accumulator read
whose outcome depends on what the co-resident wave does to the matrix pipe's timing) -- and this scan would show packed consumers
closer to their MFMA than the rule allows.

usage: python tools/studies/mfma_hazard_scan.py file.s [file2.s ...]"""
import re
import sys

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(operand_text):
    out = set()
    for m in REG.finditer(operand_text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), k) for k in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def passes(op):
    # gfx950: a pass is 4 cycles.  32x32x16 f16/bf16 = 32 cycles = 8 passes, 16x16x32 f16/bf16 = 16 cycles = 4 passes;
    # fp32-input 32x32x2 = 16 passes, 16x16x4 = 8 passes
    if "32x32x16" in op:
        return 8
    if "16x16x32" in op:
        return 4
    if "32x32x2" in op:
        return 16
    if "16x16x4" in op:
        return 8
    return 16 if "32x32" in op else 8


def scan(path):
    lines = [l.split("//")[0].split(";")[0].strip() for l in open(path)]
    ins = [(i, l) for i, l in enumerate(lines) if l and not l.startswith(".") and not l.endswith(":") and not l.startswith("#")]
    stats = {}
    for k, (ln, text) in enumerate(ins):
        op = text.split()[0]
        if not op.startswith("v_mfma"):
            continue
        ops_ = text[len(op):].split(",")
        dst = regs(ops_[0])
        need = passes(op) + 3                    # LLVM GFX940_XDL_N_PassWriteVgprVALUMemExpReadWaitStates: passes + 2, + 1 on gfx950
        states = 0
        for ln2, t2 in ins[k + 1:k + 400]:
            op2 = t2.split()[0]
            if op2 in ("s_endpgm",) or op2.startswith("s_branch") or op2.startswith("s_setpc"):
                break                            # (a conditional branch falls through: the loop exit path is scanned)
            if op2 == "s_nop":
                states += int(t2.split()[1], 0) + 1
                continue
            touched = regs(t2[len(op2):]) & dst
            if touched:
                if op2.startswith("v_mfma"):
                    o2 = t2[len(op2):].split(",")
                    if regs(o2[0]) == dst and len(o2) > 3 and regs(o2[3]) == dst:
                        break                    # accumulate chain (same D as C): no software wait needed; a new producer starts
                kind = "packed" if re.match(r"v_pk_\w+_f32", op2) else ("mfma" if op2.startswith("v_mfma") else
                                                                       "valu" if op2.startswith("v_") else "other")
                key = (op.replace("v_mfma_f32_", ""), kind)
                s = stats.setdefault(key, {"n": 0, "min": 10 ** 9, "need": need, "short": []})
                s["n"] += 1
                s["min"] = min(s["min"], states)
                if states < need and kind in ("packed", "valu", "other"):
                    s["short"].append((ln + 1, ln2 + 1, states, op2))
                break
            states += 1
    return stats


for path in sys.argv[1:]:
    print(f"== {path}")
    st = scan(path)
    for (shape, kind), s in sorted(st.items()):
        print(f"  {shape:12s} first consumer {kind:7s}: {s['n']:4d} MFMAs, min states in between {s['min']:4d} (rule: >= {s['need']})"
              f"{'   SHORT: ' + str(s['short'][:6]) if s['short'] else ''}")
