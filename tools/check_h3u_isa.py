"""conv_h3u_kernel's loader waves issue their prefetch loads from inline asm and order their use with hand-placed s_waitcnt (wunet_h3u.h):
hipcc neither counts those loads nor protects their destination registers.  This checks, in the ISA hipcc generated for THIS build, the one
thing the scheme depends on: between a prefetch load and the wait that covers it (the third `s_waitcnt vmcnt(20 + ...)` after it - the
loads of a tile are consumed two stages later) NO instruction reads or writes the load's destination registers - no compiler-inserted
copy, no re-use as a temporary.  The steady loop is walked cyclically (three unrolled stages, rotating register sets).

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


def _sgpr_dead(ins, labels, j, sreg, depth=0):
    """True if scalar register `sreg`, written by instruction j, is overwritten before it is read on every path (followed through
    up to three conditional branches): the write is dead - hipcc leaves such a v_readfirstlane of an UNDEFINED operand behind (any VGPR
    serves as "undefined", also one with a load in flight); its result reaches nothing."""
    pat = re.compile(r"\b%s\b" % re.escape(sreg))
    k = j + 1
    while k < len(ins) and k < j + 400:
        t = ins[k][0]
        k += 1
        if t is None:
            continue
        m = re.match(r"(asm )?(s_c?branch\w*)\s+(\.LBB\d+_\d+)", t)
        if m:
            tgt = labels.get(m.group(3))
            if m.group(2) == "s_branch":
                if tgt is None:
                    return False
                k = tgt
                continue
            if depth >= 3 or tgt is None or not _sgpr_dead(ins, labels, tgt, sreg, depth + 1):
                return False
            continue
        if not pat.search(t):
            continue
        ops = t.split(None, 1)[1] if " " in t else ""
        first, rest = (ops.split(",", 1) + [""])[:2]
        return bool(pat.search(first)) and not pat.search(rest) and t.startswith(("s_", "v_readfirstlane", "v_readlane"))
    return False


def check(name, lines):
    ins = []                                   # (text, label or None); instructions of inline-asm blocks carry the prefix "asm "
    in_asm = False
    for ln in lines:
        if "#ASMSTART" in ln:
            in_asm = True
        elif "#ASMEND" in ln:
            in_asm = False
        t = ln.split(";")[0].strip()
        if t and in_asm:
            t = "asm " + t
        if not t or t.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", t):
                ins.append((None, t[:-1]))
            continue
        ins.append((t, None))
    labels = {lab: i for i, (t, lab) in enumerate(ins) if lab}
    is_load = lambda t: t is not None and re.match(r"asm global_load_dword(x4)? v", t)
    is_tile_wait = lambda t: t is not None and re.match(r"asm s_waitcnt vmcnt\((2\d|3\d|4\d)\)$", t)
    loads = [i for i, (t, _) in enumerate(ins) if is_load(t)]
    waits = [i for i, (t, _) in enumerate(ins) if is_tile_wait(t)]
    if not loads or len(waits) < 3:
        return [f"{name}: expected asm prefetch loads and >= 3 tile waits, found {len(loads)} / {len(waits)}"]
    # the steady loop: the backward branch behind the last tile wait, to its target label
    back = None
    for i in range(len(ins) - 1, waits[-1], -1):
        t = ins[i][0]
        m = t and re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
        if m and labels.get(m.group(1), 1 << 30) < waits[0 if len(waits) == 3 else -3]:
            back = (labels[m.group(1)], i)
            break
    if back is None:
        return [f"{name}: loop of the loader waves not found"]
    head, tail = back
    body = list(range(head, tail + 1))
    order = list(range(loads[0], head)) + body + body + body          # prologue, then the loop three times round
    errs = []
    seen = set()
    for pos, i in enumerate(order):
        t = ins[i][0]
        if not is_load(t) or (i in seen and i >= head):
            continue
        seen.add(i)
        dest = regs_of(t[4:].split(",")[0])
        nw = 0
        for j in order[pos + 1:]:
            u = ins[j][0]
            if u is None:
                continue
            if is_tile_wait(u):
                nw += 1
                if nw == 3:
                    break
                continue
            if i < head and nw == 0 and u.startswith("asm s_waitcnt vmcnt(0)"):
                break                                                   # (prologue: tile 0 is waited for with vmcnt(0))
            if is_load(u) and regs_of(u[4:].split(",")[0]) & dest:
                errs.append(f"{name}: `{t}` is overwritten by `{u}` before its wait")
            elif not is_load(u) and regs_of(u) & dest:
                m = re.match(r"v_readfirstlane_b32 (s\d+), v\d+$", u)
                if m and _sgpr_dead(ins, labels, j, m.group(1)):
                    continue
                errs.append(f"{name}: `{u}` touches v{sorted(regs_of(u) & dest)} while `{t}` may be in flight")
        else:
            if i >= head:
                errs.append(f"{name}: no third wait behind `{t}`")
    return errs


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "h3u.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I.", "-I../../include",
                        "-Wno-unused-result", "-Wno-unused-value", "-S", "--cuda-device-only", "h3u_inst.cpp", "-o", out], cwd=CSRC, check=True)
        asm = open(out).read()
    errs, n = [], 0
    for name, lines in kernels(asm):
        n += 1
        errs += check(name, lines)
    if n == 0:
        errs.append("no conv_h3u_kernel in the ISA")
    for e in errs[:20]:
        print(e)
    print(f"{n} kernels checked, {len(errs)} problems")
    return 1 if errs else 0


if __name__ == "__main__":
    sys.exit(main())
