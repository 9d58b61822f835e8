#!/usr/bin/env python3
"""Are the kernels of two builds the same machine code?  (round 6: ptw_kernels.hip was cut into one
translation unit per kernel family - the proof that the cut changed nothing.)

    hipcc --offload-arch=gfx950 -std=c++17 -O3 -Icsrc --cuda-device-only -S csrc/<file>.hip -o <file>.s
    python scripts/isa_compare.py old.s new1.s [new2.s ...]

Every kernel of the new files is looked up among the kernels of old.s by its instruction stream (labels
renumbered in order of appearance, comments and directives dropped); prints one line per kernel: identical /
the number of differing instructions against the old kernel of the same demangled name."""
import difflib
import hashlib
import re
import subprocess
import sys


def kernels(path):
    text = open(path).read()
    out = {}
    # a kernel: ".globl sym" ... "sym:" ... ".Lfunc_endN:"; only those listed under amdhsa.kernels
    names = set(re.findall(r"\.amdhsa_kernel\s+(\S+)", text))
    for sym in names:
        m = re.search(r"^" + re.escape(sym) + r":[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M)
        if not m:
            continue
        labels = {}
        body = []
        for line in m.group(1).split("\n"):
            line = line.split(";")[0].strip()
            if not line or line.startswith((".p2align", ".loc", ".file", ".cfi", ".section")):
                continue

            def renum(mm):
                return labels.setdefault(mm.group(0), "L%d" % len(labels))
            line = re.sub(r"\.LBB\d+_\d+", renum, line)
            line = re.sub(r"\.Ltmp\d+", "Ltmp", line)
            body.append(line)
        out[sym] = body
    return out


def demangle(syms):
    r = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True).stdout.split("\n")
    clean = []
    for n in r:
        n = re.sub(r"\(.*", "", n.replace("ptw::(anonymous namespace)::", "").replace("void ", ""))
        clean.append(n)
    return dict(zip(syms, clean))


def canon(name):
    """old names carry the retired PAIR template argument: traceSequential<S, W, lds, reg, M, false, picks>"""
    m = re.match(r"traceSequential<(.*)>$", name)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        if len(a) == 7:
            a = a[:5] + a[6:]
        return "traceSequential<" + ", ".join(a) + ">"
    return name


old = kernels(sys.argv[1])
old_names = {canon(v): k for k, v in demangle(list(old)).items()}
old_hash = {hashlib.sha1("\n".join(b).encode()).hexdigest(): k for k, b in old.items()}
same = diff = 0
for path in sys.argv[2:]:
    new = kernels(path)
    for sym, name in sorted(demangle(list(new)).items(), key=lambda kv: kv[1]):
        body = new[sym]
        h = hashlib.sha1("\n".join(body).encode()).hexdigest()
        if h in old_hash:
            same += 1
            print(f"identical  {len(body):6d} instr  {name}")
            continue
        diff += 1
        ref = old.get(old_names.get(canon(name), ""), None)
        if ref is None:
            print(f"NEW        {len(body):6d} instr  {name}")
            continue
        sm = difflib.SequenceMatcher(a=ref, b=body, autojunk=False)
        changed = sum(max(i2 - i1, j2 - j1) for tag, i1, i2, j1, j2 in sm.get_opcodes() if tag != "equal")
        print(f"DIFFERS    {len(body):6d} instr ({len(ref)} before, {changed} changed)  {name}")
print(f"{same} identical, {diff} not")
