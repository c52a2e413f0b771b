"""Assembly-level bisect of the fine_match SLP ("packed fp32") build -- VERDICT r05 "what's weak" #2 / "next" #3.

csrc/fine_match.hip built WITH the SLP vectoriser gives run-to-run different `std` outputs once two of its workgroups share a CU;
built without (-fno-slp-vectorize) it is bit-reproducible.  r04 / r05 changed the SOURCE build (13 variants) and scanned the ISA; that
always changes register allocation and schedule together with the opcodes.  This tool changes ONLY opcodes: it takes the device
assembly of the failing build, rewrites chosen v_pk_{mul,add,fma}_f32 instructions of the <128, split> kernel into their two scalar
halves IN PLACE (same registers, same order of everything else), assembles, links and runs the reproducibility check -- so a variant
differs from the failing build by exactly the instructions named.  Modes:

    python tools/studies/fine_bisect.py auto      # on the GPU box: none / all, then delta-debugging down to a minimal culprit set,
                                                  # then nop-padding experiments around the culprits
    python tools/studies/fine_bisect.py build <spec> <out.so>      # CPU: build one variant (spec: none | all | 3,7-12,40 ...)
    python tools/studies/fine_bisect.py list      # CPU: the packed instructions of the kernel with their index and context

Build flow per variant (what hipcc does, with the device assembly swapped): clang -x assembler -> lld -> clang-offload-bundler ->
objcopy --update-section .hip_fatbin of a host object compiled once -> a small .so (fine_match + capi objects only).
"""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "detectorfreesfm_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-unused-function",
         f"-I{ROOT}/include", f"-I{CSRC}"]
KERNEL = "_ZN12_GLOBAL__N_117fine_match_kernelILi128ELb1EEEvNS_8FineArgsE"
TMPV = 255                       # scratch VGPR for the one form whose halves read each other's destination

WORK = os.environ.get("FINE_BISECT_DIR") or os.path.join(tempfile.gettempdir(), "fine_bisect")


def run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if r.returncode:
        raise RuntimeError(f"{' '.join(cmd)}\n{r.stdout}")
    return r.stdout


def prepare(extra_flags=()):
    """Device assembly of the SLP build, the host object (compiled once) and capi.o."""
    os.makedirs(WORK, exist_ok=True)
    asm = os.path.join(WORK, "fine_packed.s")
    if not os.path.exists(asm):
        run([HIPCC, *FLAGS, *extra_flags, "-S", "--cuda-device-only", os.path.join(CSRC, "fine_match.hip"), "-o", asm])
    host = os.path.join(WORK, "fine_host.o")
    if not os.path.exists(host):
        fb = build_fatbin(open(asm).read(), "base")
        run([HIPCC, *FLAGS, "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c",
             os.path.join(CSRC, "fine_match.hip"), "-o", host])
    capi = os.path.join(WORK, "capi.o")
    if not os.path.exists(capi):
        run([HIPCC, *FLAGS, "-c", os.path.join(CSRC, "capi.hip"), "-o", capi])
    return asm, host, capi


def build_fatbin(text, tag):
    s = os.path.join(WORK, f"{tag}.s")
    open(s, "w").write(text)
    o, out, fb = (os.path.join(WORK, f"{tag}.{e}") for e in ("o", "out", "hipfb"))
    run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
    run([f"{LLVM}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", out, o])
    run([f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096",
         "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", f"-input={out}", f"-output={fb}"])
    return fb


# ---- the rewrite ----------------------------------------------------------------------------------------------------------
PK = re.compile(r"^\s*v_pk_(mul|add|fma)_f32\s+(.*)$")
MOD = re.compile(r"(op_sel|op_sel_hi|neg_lo|neg_hi):\[([0-9,]+)\]")


def _half(opnd, sel):
    """Register / constant of half `sel` (0 lo, 1 hi) of a packed source operand."""
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", opnd)
    if m:
        return f"{m.group(1)}{int(m.group(2)) + sel}"
    return opnd                   # inline constant: the same value in both halves


def scalarize(line):
    """The two scalar VOP3 instructions (plus a move when needed) computing what one v_pk_*_f32 line computes."""
    m = PK.match(line)
    op, rest = m.group(1), m.group(2).split(";")[0].strip()
    mods = {k: [int(x) for x in v.split(",")] for k, v in MOD.findall(rest)}
    rest = MOD.sub("", rest).strip()
    opnds = [o.strip() for o in rest.split(",") if o.strip()]
    dst, srcs = opnds[0], opnds[1:]
    n = len(srcs)
    sel = {0: mods.get("op_sel", [0] * n), 1: mods.get("op_sel_hi", [1] * n)}
    neg = {0: mods.get("neg_lo", [0] * n), 1: mods.get("neg_hi", [0] * n)}
    d = re.fullmatch(r"v\[(\d+):(\d+)\]", dst)
    dreg = {0: f"v{int(d.group(1))}", 1: f"v{int(d.group(1)) + 1}"}
    mnem = {"mul": "v_mul_f32_e64", "add": "v_add_f32_e64", "fma": "v_fma_f32"}[op]
    use = {h: [_half(s, sel[h][i]) for i, s in enumerate(srcs)] for h in (0, 1)}
    txt = {h: ", ".join(("-" if neg[h][i] else "") + use[h][i] for i in range(n)) for h in (0, 1)}
    lo_first_ok = dreg[0] not in use[1]
    hi_first_ok = dreg[1] not in use[0]
    if lo_first_ok:
        return [f"\t{mnem} {dreg[0]}, {txt[0]}", f"\t{mnem} {dreg[1]}, {txt[1]}"]
    if hi_first_ok:
        return [f"\t{mnem} {dreg[1]}, {txt[1]}", f"\t{mnem} {dreg[0]}, {txt[0]}"]
    return [f"\t{mnem} v{TMPV}, {txt[0]}", f"\t{mnem} {dreg[1]}, {txt[1]}", f"\tv_mov_b32_e32 {dreg[0]}, v{TMPV}"]


def kernel_span(lines):
    beg = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(beg, len(lines)) if "s_endpgm" in lines[i])
    return beg, end


def packed_lines(lines):
    beg, end = kernel_span(lines)
    return [i for i in range(beg, end) if PK.match(lines[i])]


def parse_spec(spec, n):
    if spec == "none":
        return set()
    if spec == "all":
        return set(range(n))
    out = set()
    for part in spec.split(","):
        if "-" in part:
            a, b = part.split("-")
            out.update(range(int(a), int(b) + 1))
        else:
            out.add(int(part))
    return out


def variant_text(asm_text, scalar_set, nops=None, swap_after=(), replace=None):
    """`scalar_set`: indices (among the kernel's packed instructions) to rewrite; `nops`: {index: (before, after)} s_nop counts;
    `swap_after`: indices k whose two FOLLOWING instructions change places; `replace`: {index: [lines]} literal replacement."""
    lines = asm_text.split("\n")
    pk = packed_lines(lines)
    nops = nops or {}
    replace = replace or {}
    used_tmp = False
    for k in swap_after:          # on the untouched text; the ORDER of the packed instructions does not change
        i = pk[k]
        lines[i + 1], lines[i + 2] = lines[i + 2], lines[i + 1]
    pk = packed_lines(lines)
    for k in sorted(range(len(pk)), reverse=True):
        i = pk[k]
        new = [lines[i]]
        if k in replace:
            new = list(replace[k])
        elif k in scalar_set:
            new = scalarize(lines[i])
            used_tmp |= len(new) == 3
        b, a = nops.get(k, (0, 0))
        new = [f"\ts_nop {b - 1}"] * (1 if b else 0) + new + [f"\ts_nop {a - 1}"] * (1 if a else 0)
        lines[i:i + 1] = new
    text = "\n".join(lines)
    if used_tmp:                  # the kernel descriptor must cover the scratch register (still two waves per SIMD: <= 256)
        text = re.sub(r"(\.amdhsa_next_free_vgpr\s+)(\S+)", lambda m: m.group(1) + "256", text)
        text = re.sub(r"(\.vgpr_count:\s+)(\d+)", lambda m: m.group(1) + "256", text)
    return text


def build_variant(scalar_set, out_so, nops=None, tag="v", swap_after=(), replace=None):
    asm, host, capi = prepare()
    fb = build_fatbin(variant_text(open(asm).read(), scalar_set, nops, swap_after, replace), tag)
    obj = os.path.join(WORK, f"{tag}_host.o")
    run([f"{LLVM}/llvm-objcopy", "--update-section", f".hip_fatbin={fb}", host, obj])
    run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", obj, capi, "-o", out_so])
    return out_so


# ---- the check (GPU) ------------------------------------------------------------------------------------------------------
class Checker:
    """Runs dfsfm_fine_match_split of a variant library `runs` times on BASELINE configs[2]'s shape (2000 tracks x 4 views, W 15,
    C 128; correlated windows as tools/fine_determinism.py) and counts runs whose outputs differ in any bit from the first."""

    def __init__(self, T=2000, Vq=4, W=15, C=128):
        import torch
        self.torch = torch
        dev = "cuda:0"
        g = torch.Generator().manual_seed(3)
        ref = torch.randn((T, W * W, C), generator=g)
        qry = 0.7 * ref[:, None] + torch.randn((T, Vq, W * W, C), generator=g)

        def split(x):
            hi = x.half()
            lo = ((x - hi.float()) * 2048.0).half()
            return hi.to(dev).contiguous(), lo.to(dev).contiguous()
        self.rh, self.rl = split(ref)
        self.qh, self.ql = split(qry)
        self.mask = torch.ones((T, Vq), dtype=torch.uint8, device=dev)
        self.mov = torch.ones((T,), dtype=torch.uint8, device=dev)
        self.T, self.Vq, self.W, self.C = T, Vq, W, C
        self.dev = dev

    def outputs(self, lib):
        torch = self.torch
        T, Vq = self.T, self.Vq
        best = torch.empty((T,), dtype=torch.int32, device=self.dev)
        left = torch.empty((T, 2), device=self.dev)
        coords = torch.empty((T, Vq, 2), device=self.dev)
        std = torch.empty((T, Vq), device=self.dev)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        rc = lib.dfsfm_fine_match_split(p(self.rh), p(self.rl), p(self.qh), p(self.ql), p(self.mask), p(self.mov), T, Vq, self.W, 7, self.C,
                                        None, None, None, None, ctypes.c_int64(0), ctypes.c_int64(0), p(best), p(left), p(coords), p(std),
                                        None, None, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc:
            raise RuntimeError(f"dfsfm_fine_match_split -> {rc}")
        return best, coords, std

    def check(self, so, runs=30):
        torch = self.torch
        lib = ctypes.CDLL(so)
        lib.dfsfm_fine_match_split.restype = ctypes.c_int
        first = self.outputs(lib)
        torch.cuda.synchronize()
        bad_runs, n_std, n_coord, n_best = 0, 0, 0, 0
        for _ in range(runs):
            o = self.outputs(lib)
            db = int((o[0] != first[0]).sum())
            dc = int((o[1].view(torch.int32) != first[1].view(torch.int32)).any(-1).sum())
            ds = int((o[2].view(torch.int32) != first[2].view(torch.int32)).sum())
            bad_runs += 1 if (db or dc or ds) else 0
            n_best += db
            n_coord += dc
            n_std += ds
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            self.outputs(lib)
        b.record()
        torch.cuda.synchronize()
        return dict(bad_runs=bad_runs, runs=runs, std=n_std, coords=n_coord, best=n_best, ms=a.elapsed_time(b) / 10)


def fmt(s):
    s = sorted(s)
    out, i = [], 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[j] + 1:
            j += 1
        out.append(str(s[i]) if i == j else f"{s[i]}-{s[j]}")
        i = j + 1
    return ",".join(out) or "none"


def auto():
    asm, _, _ = prepare()
    lines = open(asm).read().split("\n")
    pk = packed_lines(lines)
    n = len(pk)
    print(f"# fine_match<128, split>, SLP build: {n} packed-fp32 instructions in the kernel", flush=True)
    ck = Checker()
    cache = {}
    counter = [0]

    def test(S, nops=None, label=""):
        key = (frozenset(S), tuple(sorted((nops or {}).items())))
        if key in cache:
            return cache[key]
        counter[0] += 1
        so = build_variant(S, os.path.join(WORK, f"lib_t{counter[0]}.so"), nops, tag=f"t{counter[0]}")
        r = ck.check(so)
        ok = r["bad_runs"] == 0
        cache[key] = ok
        print(f"test {counter[0]:3d} {label:28s} scalarised [{fmt(S)}] ({len(S)}/{n})" + (f" nops {nops}" if nops else "") +
              f": {'REPRODUCIBLE' if ok else 'DEVIATES'} ({r['bad_runs']}/{r['runs']} runs; std {r['std']}, coords {r['coords']}, best "
              f"{r['best']} entries) {r['ms']:.3f} ms", flush=True)
        return ok

    none_ok = test(set(), label="as compiled")
    all_ok = test(set(range(n)), label="every packed op scalarised")
    if none_ok:
        print("# the SLP build is reproducible on this box: nothing to bisect")
        return
    if not all_ok:
        print("# scalarising EVERY packed instruction in place does not restore reproducibility: the opcodes are not the cause;\n"
              "# what differs from the -fno-slp-vectorize build is then only register allocation / schedule")
        return
    # delta debugging on the set that must be scalarised: drop chunks whose packed form turns out harmless
    S = set(range(n))
    chunk = max(1, n // 2)
    while chunk >= 1:
        order = sorted(S)
        i = 0
        progressed = False
        while i < len(order):
            part = set(order[i:i + chunk])
            cand = S - part
            if part and test(cand, label=f"keep packed {len(part)} @ {min(part)}"):
                S = cand
                progressed = True
            i += chunk
        if chunk == 1 and not progressed:
            break
        chunk = chunk // 2 if chunk > 1 else (1 if progressed else 0)
        if chunk == 0:
            break
    print(f"# minimal set that must be scalarised for reproducibility: [{fmt(S)}]", flush=True)
    for k in sorted(S):
        i = pk[k]
        print(f"## packed op #{k} (asm line {i + 1}) and its neighbourhood:")
        for j in range(max(0, i - 6), min(len(lines), i + 7)):
            print(("  >> " if j == i else "     ") + lines[j].rstrip())
    # does padding cure it?  nops before / after each culprit in the otherwise untouched packed build
    for pad in ((1, 0), (0, 1), (4, 0), (0, 4), (8, 8)):
        test(set(), nops={k: pad for k in S}, label=f"packed + s_nop {pad} at culprits")
    # each culprit alone scalarised (is every one of them needed?)
    if len(S) > 1:
        for k in sorted(S):
            test({k}, label=f"only #{k} scalarised")


def focus():
    """Second session: the strong culprit of the `auto` run (packed op #25 of the r06 build: v_pk_mul_f32 v[152:153], v[202:203],
    v[152:153] op_sel:[0,1], followed by a ds_read2_b32 that reloads v[202:203] and then by the consumer of v[152:153]) kept packed
    with EVERYTHING else scalarised, and what cures it."""
    asm, _, _ = prepare()
    lines = open(asm).read().split("\n")
    pk = packed_lines(lines)
    n = len(pk)
    K = int(os.environ.get("FINE_FOCUS", "25"))
    print(f"# focus on packed op #{K}: {lines[pk[K]].strip()}", flush=True)
    for j in range(pk[K] - 5, pk[K] + 5):
        print(("  >> " if j == pk[K] else "     ") + lines[j].rstrip())
    ck = Checker()
    allbut = lambda *keep: set(range(n)) - set(keep)
    m = PK.match(lines[pk[K]])
    exps = [
        ("all scalarised but #K", allbut(K), {}, (), None),
        ("... + s_nop 0 after #K", allbut(K), {K: (0, 1)}, (), None),
        ("... + s_nop 0 before consumer", allbut(K), {K + 1: (1, 0)}, (), None),
        ("... + s_nop 0 before #K", allbut(K), {K: (1, 0)}, (), None),
        ("... + s_nop 3 before #K", allbut(K), {K: (4, 0)}, (), None),
        ("... + s_nop 3 after #K", allbut(K), {K: (0, 4)}, (), None),
        ("... + s_nop 3 before consumer", allbut(K), {K + 1: (4, 0)}, (), None),
        ("... + s_nop 7 x (before, after, consumer)", allbut(K), {K: (8, 8), K + 1: (8, 0)}, (), None),
        ("... ds_read moved behind the consumer", allbut(K), {}, (K,), None),
        ("... ds_read behind the consumer, s_nop 0 after #K", allbut(K), {K: (0, 1)}, (K,), None),
        ("#K and its consumer packed", allbut(K, K + 1), {}, (), None),
        ("#K-1 and #K packed", allbut(K - 1, K), {}, (), None),
        ("all packed but #K", {K}, {}, (), None),
    ]
    if "op_sel:[0,1]" in lines[pk[K]] and "op_sel_hi" not in lines[pk[K]]:
        # the same product without the op_sel modifier: broadcast src1.hi into src1.lo first (dst == src1 here, so lo is dead)
        d = re.search(r"v_pk_mul_f32 v\[(\d+):(\d+)\], (v\[\d+:\d+\]), v\[(\d+):(\d+)\]", lines[pk[K]])
        if d and d.group(1) == d.group(4):
            lo, hi, src0 = int(d.group(1)), int(d.group(2)), d.group(3)
            exps.append(("#K without op_sel (v_mov lo, hi first)", allbut(K), {}, (),
                         {K: [f"\tv_mov_b32_e32 v{lo}, v{hi}", f"\tv_pk_mul_f32 v[{lo}:{hi}], {src0}, v[{lo}:{hi}]"]}))
            exps.append(("#K without op_sel + s_nop between", allbut(K), {}, (),
                         {K: [f"\tv_mov_b32_e32 v{lo}, v{hi}", "\ts_nop 0", f"\tv_pk_mul_f32 v[{lo}:{hi}], {src0}, v[{lo}:{hi}]"]}))
    runs = int(os.environ.get("FINE_RUNS", "60"))
    for i, (label, S, nops, swap, repl) in enumerate(exps):
        so = build_variant(S, os.path.join(WORK, f"lib_f{i}.so"), nops, tag=f"f{i}", swap_after=swap, replace={K: repl[K]} if repl else None)
        r = ck.check(so, runs=runs)
        print(f"focus {i:2d} {label:44s}: {'REPRODUCIBLE' if r['bad_runs'] == 0 else 'DEVIATES'} ({r['bad_runs']}/{r['runs']} runs; std "
              f"{r['std']}, coords {r['coords']}, best {r['best']} entries) {r['ms']:.3f} ms", flush=True)


def hazardous(line):
    """A packed fp32 instruction whose LOW lane takes the HIGH half of src1 / src2 (op_sel bit set behind position 0): the form the
    reproducer (tools/ubench/pk_opsel_inplace.hip) shows reading 0 while an MFMA of the same wave is in flight."""
    m = re.search(r"op_sel:\[([0-9,]+)\]", line)
    return bool(PK.match(line) and m and any(int(x) for x in m.group(1).split(",")[1:]))


def opsel():
    """Third session: the SLP build with ONLY the hazardous form rewritten (everything else stays packed) must be bit-reproducible."""
    asm, _, _ = prepare()
    lines = open(asm).read().split("\n")
    pk = packed_lines(lines)
    haz = {k for k, i in enumerate(pk) if hazardous(lines[i])}
    print(f"# {len(pk)} packed-fp32 instructions, {len(haz)} with op_sel on src1/src2: " + "; ".join(f"#{k} {lines[pk[k]].strip()}" for k in sorted(haz)), flush=True)
    ck = Checker()
    runs = int(os.environ.get("FINE_RUNS", "300"))
    for label, S in (("as compiled", set()), ("only the op_sel (src1/src2) forms scalarised", haz),
                     ("everything BUT those forms scalarised", set(range(len(pk))) - haz)):
        so = build_variant(S, os.path.join(WORK, f"lib_o{len(S)}.so"), tag=f"o{len(S)}")
        r = ck.check(so, runs=runs)
        print(f"opsel: {label:46s}: {'REPRODUCIBLE' if r['bad_runs'] == 0 else 'DEVIATES'} ({r['bad_runs']}/{r['runs']} runs; std {r['std']}, "
              f"coords {r['coords']}, best {r['best']} entries) {r['ms']:.3f} ms", flush=True)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "list"
    if mode == "auto":
        auto()
    elif mode == "opsel":
        opsel()
    elif mode == "focus":
        focus()
    elif mode == "build":
        asm, _, _ = prepare()
        n = len(packed_lines(open(asm).read().split("\n")))
        print(build_variant(parse_spec(sys.argv[2], n), sys.argv[3]))
    elif mode == "check":
        ck = Checker()
        for so in sys.argv[2:]:
            print(so, ck.check(so))
    else:
        asm, _, _ = prepare()
        lines = open(asm).read().split("\n")
        for k, i in enumerate(packed_lines(lines)):
            print(f"#{k:3d} line {i + 1:5d}: {lines[i].strip():90s} -> {' ; '.join(x.strip() for x in scalarize(lines[i]))}")


if __name__ == "__main__":
    main()
