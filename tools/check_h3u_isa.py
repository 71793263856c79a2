"""conv_h3u_kernel's loader waves issue their prefetch loads from inline asm and order their use with hand-placed s_waitcnt (wunet_h3u.h):
hipcc neither counts those loads nor protects their destination registers.  This checks, in the ISA hipcc generated for THIS build, the one
thing the scheme depends on: between a prefetch load and the wait that covers it (the SECOND stage barrier `s_waitcnt vmcnt(10) lgkmcnt(0)` -
or `vmcnt(18)` behind a conversion that copied the operand to HBM: its 8 counted stores are younger than the loads as well -
after it - a stage's barrier lets only the 10 loads issued in that stage stay outstanding, so the loads of the stage before have landed; they
are consumed in the stage that follows) NO instruction reads or writes the load's destination registers - no compiler-inserted copy, no re-use
as a temporary.  The steady loop is walked cyclically (three unrolled stages, rotating register sets).  (The per-item barrier of the
statistics hand-over is written `s_waitcnt lgkmcnt(0) vmcnt(10)` and not counted; the vmcnt(0) forms are the stricter alternative paths.)

    python tools/check_h3u_isa.py            # compiles csrc/h3u_inst.cpp to ISA with hipcc and checks every instantiation
Exit code 0 = clean.  Run by tests/test_abi.py (CPU: hipcc cross-compiles) so a compiler or source change that breaks the assumption fails the suite.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wave-u-net-for-speech-enhancement_amd", "csrc")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def kernels(asm):
    cur, name = None, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\d+conv_h3u_kernel\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                yield name, cur
                cur = None


def check(name, lines):
    ins = []                                   # (text, label or None); instructions of inline-asm blocks carry the prefix "asm "
    in_asm = False
    for ln in lines:
        if "#ASMSTART" in ln:
            in_asm = True
        elif "#ASMEND" in ln:
            in_asm = False
        t = ln.split(";")[0].strip()
        if not t or t.startswith("#"):
            continue
        m = re.match(r"^(\.\w+):$", t)
        if m:
            ins.append((None, m.group(1)))
            continue
        if t.startswith("."):
            continue
        ins.append((("asm " + t) if in_asm else t, None))
    labels = {lab: i for i, (t, lab) in enumerate(ins) if lab}
    is_load = lambda t: t is not None and re.match(r"asm global_load_dword(x4)? v", t)
    # (vmcnt(18): the stage barrier behind a conversion that also copied the operand to HBM - 10 loads + WUNET_H3U_NST = 8 stores may stay in flight)
    is_stage_barrier = lambda t: t is not None and re.match(r"asm s_waitcnt vmcnt\((10|18)\) lgkmcnt\(0\)$", t)
    is_full_wait = lambda t: t is not None and re.search(r"s_waitcnt (lgkmcnt\(0\) )?vmcnt\(0\)", t)
    loads = [i for i, (t, _) in enumerate(ins) if is_load(t)]
    nbar = sum(1 for t, _ in ins if is_stage_barrier(t))
    if not loads or nbar < 3:
        return [f"{name}: expected asm prefetch loads and >= 3 stage barriers, found {len(loads)} / {nbar}"]

    def succ(i):
        t = ins[i][0]
        if t is None:
            return [i + 1]
        u = t[4:] if t.startswith("asm ") else t
        if u.startswith("s_endpgm"):
            return []
        m = re.match(r"(s_c?branch\w*)\s+(\.\w+)", u)
        if m:
            tgt = labels.get(m.group(2))
            if tgt is None:
                return [i + 1]
            return [tgt] if m.group(1) == "s_branch" else [i + 1, tgt]
        return [i + 1]

    # hipcc structurizes the loop exits through flag registers (`s_mov_b64 s[6:7], -1 ... s_and_b64 vcc, exec, s[6:7]; s_cbranch_vccnz <exit>`
    # in a block several paths share): followed blindly such a block links the end of one unrolled stage to the head of the same one.  The walk
    # therefore carries the flag pairs it has seen set to 0 / -1 and vcc derived from them, and takes only the feasible side of such a branch.
    PAIR = re.compile(r"s\[(\d+):(\d+)\]")

    def step_state(u, st):
        """st: tuple of sorted (key, value) - keys 'vcc' or (lo, hi); returns the state behind instruction u."""
        d = dict(st)
        v = u[4:] if u.startswith("asm ") else u
        m = re.match(r"s_mov_b64 s\[(\d+):(\d+)\], (-1|0)$", v)
        if m:
            d[(int(m.group(1)), int(m.group(2)))] = int(m.group(3))
            return tuple(sorted(d.items(), key=str))
        m = re.match(r"s_(and|andn2)_b64 vcc, exec, s\[(\d+):(\d+)\]$", v)
        if m:
            k = (int(m.group(2)), int(m.group(3)))
            if k in d:
                d["vcc"] = (d[k] != 0) if m.group(1) == "and" else (d[k] == 0)
            else:
                d.pop("vcc", None)
            return tuple(sorted(d.items(), key=str))
        if re.match(r"s_cbranch_", v):
            return st
        # exec: known non-zero (a running wave outside a lane-masked region) or unknown
        if re.match(r"s_\w+_saveexec_b64", v) or re.match(r"(s_and_b64|s_andn2_b64|s_xor_b64|s_mov_b64) exec,", v) or v.startswith("v_cmpx"):
            d.pop("exec", None)
        elif re.match(r"s_or_b64 exec, exec, ", v):
            d["exec"] = True                               # (the mask saved before the region is put back)
        if "vcc" in v or re.match(r"v_cmpx?_\w+_e32", v) or v.startswith("v_div_scale") or v.startswith("asm"):
            d.pop("vcc", None)
        sregs = set()
        for m in PAIR.finditer(v):
            sregs.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bs(\d+)\b", v):
            sregs.add(int(m.group(1)))
        for k in [k for k in d if not isinstance(k, str) and (k[0] in sregs or k[1] in sregs)]:
            del d[k]
        return tuple(sorted(d.items(), key=str))

    def succ_state(i, st):
        t = ins[i][0]
        if t is not None:
            u = t[4:] if t.startswith("asm ") else t
            m = re.match(r"s_cbranch_(vcc|exec)(nz|z)\s+(\.\w+)", u)
            d = dict(st)
            if m and m.group(1) in d and m.group(3) in labels:
                taken = d[m.group(1)] if m.group(2) == "nz" else not d[m.group(1)]
                return [labels[m.group(3)]] if taken else [i + 1]
        return succ(i)

    def value_dead(j, sreg, st0):
        """hipcc leaves `v_readfirstlane_b32 sN, vK` of an UNDEFINED operand behind on loop-exit paths (any VGPR serves as "undefined", also one
        with a load in flight).  True if on every feasible path behind instruction j the scalar it wrote - and every plain copy of it - is
        overwritten or the program ends before anything else reads it."""
        seen_, todo = set(), [(k, frozenset([sreg]), step_state(ins[j][0], st0)) for k in succ_state(j, st0)]
        steps = 0
        while todo:
            i, taint, st = todo.pop()
            if i >= len(ins) or not taint or (i, taint, st) in seen_:
                continue
            seen_.add((i, taint, st))
            steps += 1
            if steps > 200000:
                return False
            u = ins[i][0]
            if u is None:
                todo.append((i + 1, taint, st))
                continue
            v = u[4:] if u.startswith("asm ") else u
            ops = v.split(None, 1)[1] if " " in v else ""
            first, rest = (ops.split(",", 1) + [""])[:2]

            def sset(text):
                r = set()
                for m in PAIR.finditer(text):
                    r.update(range(int(m.group(1)), int(m.group(2)) + 1))
                for m in re.finditer(r"\bs(\d+)\b", text):
                    r.add(int(m.group(1)))
                return r
            writes_first = v.startswith(("s_", "v_readfirstlane", "v_readlane")) and not v.startswith(("s_cmp", "s_bitcmp", "s_cbranch", "s_branch", "s_waitcnt", "s_barrier", "s_nop", "s_endpgm"))
            reads = sset(rest) if writes_first else sset(ops)
            if reads & taint:
                m = re.match(r"s_mov_b32 s(\d+), s(\d+)$", v)
                if not m:
                    return False
                taint = taint | {int(m.group(1))}
            elif writes_first:
                taint = taint - sset(first)
            if v.startswith("s_endpgm"):
                continue
            nst = step_state(u, st)
            for k in succ_state(i, st):
                todo.append((k, taint, nst))
        return True

    errs = []
    for x in loads:
        t = ins[x][0]
        dest = regs_of(t[4:].split(",")[0])
        seen = set()
        stack = [(j, 0, (("exec", True),)) for j in succ(x)]     # (the prefetch loads are issued by whole waves)
        bad = {}
        while stack:
            j, nb, st = stack.pop()
            if j >= len(ins) or (j, nb, st) in seen:
                continue
            seen.add((j, nb, st))
            u = ins[j][0]
            nxt = succ(j)
            if u is not None:
                if is_full_wait(u):
                    continue                               # everything this wave issued has landed
                if is_stage_barrier(u):
                    nb += 1
                    if nb == 2:
                        continue
                elif regs_of(u) & dest and j not in bad:
                    m = re.match(r"v_readfirstlane_b32 s(\d+), v\d+$", u)
                    if not (m and value_dead(j, int(m.group(1)), st)):
                        bad[j] = u
                nxt = succ_state(j, st)
                st = step_state(u, st)
            for k in nxt:
                stack.append((k, nb, st))
        for j, u in sorted(bad.items())[:3]:
            errs.append(f"{name}: `{u}` (instruction {j}) touches v{sorted(regs_of(u) & dest)} while `{t}` (instruction {x}) may be in flight")
    return errs


def check_store_counts(name, lines):
    """The loaders' waits allow WUNET_H3U_NST = 8 stores per copying conversion to stay in flight: a conversion that issued FEWER would let
    a wait return before the loads in front of it have landed.  In the ISA the loader side of a kernel holds 8 conversions (3 unrolled stages
    x {row start, elsewhere} + 2 of the prologue); each must hold 8 16-byte and 2 two-byte stores of the copy - counted here over the kernel:
    the two-byte stores only exist in the loaders (16 = 8 x 2), the 16-byte ones are 8 x 8 + the MFMA waves' epilogue rows."""
    body = "\n".join(lines)
    shorts = len(re.findall(r"global_store_short", body))
    x4 = len(re.findall(r"global_store_dwordx4", body))
    copy = "Lb1EE" in name                     # conv_h3u_kernel<M_REP, COPY = true>: the training instantiation
    errs = []
    if copy and shorts != 16:
        errs.append(f"{name}: expected 16 two-byte stores of the loaders' operand copy (8 conversions x 2), found {shorts}")
    if copy and x4 < 8 * 8:
        errs.append(f"{name}: expected at least 64 16-byte stores of the loaders' operand copy, found {x4}")
    if not copy and (shorts != 0 or "vmcnt(18)" in body):
        errs.append(f"{name}: the eval instantiation holds operand-copy code ({shorts} two-byte stores)")
    return errs


def compile_isa():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "h3u.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I.", "-I../../include",
                        "-Wno-unused-result", "-Wno-unused-value", "-S", "--cuda-device-only", "h3u_inst.cpp", "-o", out], cwd=CSRC, check=True)
        return open(out).read()


def main():
    asm = compile_isa()
    errs, n = [], 0
    for name, lines in kernels(asm):
        n += 1
        errs += check(name, lines)
        errs += check_store_counts(name, lines)
    if n == 0:
        errs.append("no conv_h3u_kernel in the ISA")
    for e in errs[:20]:
        print(e)
    print(f"{n} kernels checked, {len(errs)} problems")
    return 1 if errs else 0


if __name__ == "__main__":
    sys.exit(main())
