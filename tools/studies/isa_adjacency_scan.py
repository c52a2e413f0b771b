import re, sys, collections
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
def regs(t):
    out=set()
    for m in REG.finditer(t):
        if m.group(1): out.add((m.group(1),int(m.group(2))))
        else: out.update((m.group(3),k) for k in range(int(m.group(4)),int(m.group(5))+1))
    return out
TRANS=("v_exp_f32","v_log_f32","v_rcp_f32","v_rsq_f32","v_sqrt_f32","v_sin_f32","v_cos_f32","v_rcp_iflag")
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if re.match(r"v_pk_\w+_f32",op): return "pk32"
    if op.startswith("v_pk_"): return "pk16"
    if op.startswith(TRANS): return "trans"
    if "dpp" in op or op.startswith("v_permlane") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"): return "xlane"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_","global_","scratch_","flat_")): return "vmem"
    if op.startswith("v_"): return "valu"
    return "other"
for path in sys.argv[1:]:
    lines=[l.split("//")[0].split(";")[0].strip() for l in open(path)]
    ins=[l for l in lines if l and not l.startswith(".") and not l.endswith(":") and not l.startswith("#")]
    hist=collections.Counter()
    last={}   # reg -> (idx, class)
    for i,t in enumerate(ins):
        op=t.split()[0]
        c=cls(op)
        ops_=t[len(op):]
        parts=ops_.split(",")
        if op.startswith(("v_","ds_read","buffer_load","global_load")) and not op.startswith("v_cmp"):
            dst=regs(parts[0]); src=regs(",".join(parts[1:]))
            if op.startswith("v_mfma") or "%0" in t: pass
        else:
            dst=set(); src=regs(ops_)
        if "dpp" in t: c="xlane" if c=="valu" else c
        for r in src:
            if r in last:
                j,pc=last[r]
                d=i-j
                if d<=2 and pc in ("trans","mfma","xlane","pk32") or (c=="pk32" and d<=2):
                    hist[(pc,c,d)]+=1
        for r in dst: last[r]=(i,c)
    print("==",path)
    for k,v in sorted(hist.items()): print("  producer %-6s -> consumer %-6s distance %d : %d"%(k+(v,)))
