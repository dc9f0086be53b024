#!/usr/bin/env python
"""ISA audit for the asynchronous LDS fragment reads (`lds_read16_async`, leco_prims.h).

hipcc treats an `asm volatile("ds_read_b128 ...")` as one opaque instruction whose destination is written at
the end of the statement: it neither counts the read in its s_waitcnt bookkeeping nor keeps other instructions
(register copies at control-flow joins, spills) away from the destination while the data is still in flight
(cdna_hip_programming.md section 5.7).  The kernels complete those reads themselves (`lds_wait<N>` + `lds_tie`);
this tool proves, on the generated ISA, that no instruction touches a destination register between the read and
the wait that covers it.  It found the round-1 full-size NaN: the `w_last` variant of the steady-state loop of
`gemm_kernel<64,64,plain,TF=1>` kept fragment set A in v[2:21], the drain loop in v[22:41], and the ten
`v_mov_b64` the compiler placed on the loop-exit edge copied registers whose ds_reads had not landed.

    python tools/audit_async_lds.py [file.hip ...]      (default: every csrc/*.hip that uses lds_read16_async)

Method: per kernel, basic blocks + forward data flow.  State = ordered list of in-flight asm ds_reads (their
destination VGPRs).  `s_waitcnt lgkmcnt(n)` retires all but the youngest n LDS operations (LDS returns in order;
scalar loads in flight only make the wait stronger for the LDS reads, so they are ignored).  Any other
instruction naming an in-flight register is a violation.  At joins the states are aligned youngest-first and
merged by union.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "leco_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
         "-I", os.path.join(CSRC, "prims")]

REG_RANGE = re.compile(r"\bv\[(\d+):(\d+)\]")
REG_ONE = re.compile(r"\bv(\d+)\b")


def vregs(text):
    regs = set()
    for a, b in REG_RANGE.findall(text):
        regs.update(range(int(a), int(b) + 1))
    for a in REG_ONE.findall(REG_RANGE.sub(" ", text)):
        regs.add(int(a))
    return regs


def functions(lines):
    cur, name = None, None
    for l in lines:
        m = re.match(r"^(_Z\S+):", l)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if l.startswith(".Lfunc_end"):
                yield name, cur
                cur = None
            else:
                cur.append(l)


def blocks_of(body):
    """-> (blocks: list of (label, [(kind, text)]), succ: list of successor indices)"""
    blocks, cur, label, in_asm = [], [], "<entry>", False
    for l in body:
        s = l.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";") or (s.startswith(".") and not re.match(r"^\.LBB\S+:", s)):
            continue
        m = re.match(r"^(\.LBB\S+):", s)
        if m:
            if cur or not label.startswith("<after"):
                blocks.append((label, cur))
            label, cur = m.group(1), []
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        cur.append(("asm" if in_asm else "cc", s))
        if re.match(r"^s_(cbranch|branch|endpgm)", s):
            blocks.append((label, cur))
            label, cur = f"<after {label} #{len(blocks)}>", []
    if cur:
        blocks.append((label, cur))
    index = {lab: i for i, (lab, _) in enumerate(blocks)}
    succ = []
    for i, (_, ins) in enumerate(blocks):
        out = []
        last = ins[-1][1] if ins else ""
        m = re.match(r"^s_(cbranch\S*|branch)\s+(\.LBB\S+)", last)
        if m:
            out.append(index[m.group(2)])
            if m.group(1) != "branch" and i + 1 < len(blocks):
                out.append(i + 1)
        elif not last.startswith("s_endpgm") and i + 1 < len(blocks):
            out.append(i + 1)
        succ.append(out)
    return blocks, succ


def merge(a, b):
    if a is None:
        return b
    n = max(len(a), len(b))
    pa = (frozenset(),) * (n - len(a)) + a
    pb = (frozenset(),) * (n - len(b)) + b
    return tuple(x | y for x, y in zip(pa, pb))


def audit_function(name, body):
    blocks, succ = blocks_of(body)
    state_in = [None] * len(blocks)
    state_in[0] = ()
    work = [0]
    violations = {}
    rounds = 0
    while work and rounds < 20000:
        rounds += 1
        i = work.pop()
        st = list(state_in[i])
        for kind, s in blocks[i][1]:
            op = s.split()[0]
            if kind == "asm" and op in ("ds_read_b128", "ds_read_b64_tr_b16"):
                dst = s.split(None, 1)[1].split(",")[0]
                st.append(frozenset(vregs(dst)))
                continue
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", s)
                if m:
                    n = int(m.group(1))
                    st = st[len(st) - n:] if n else []
                continue
            if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
                if op.startswith("ds_"):
                    st.append(frozenset())      # a compiler-counted LDS op also occupies a slot of the in-order queue
            if op == "s_barrier" or not st:
                continue
            touched = vregs(s)
            for entry in st:
                hit = touched & entry
                if hit:
                    violations[(blocks[i][0], s)] = sorted(hit)
        st = tuple(st[-16:])
        for j in succ[i]:
            new = merge(state_in[j], st)
            if new != state_in[j]:
                state_in[j] = new
                work.append(j)
    n_reads = sum(1 for _, ins in blocks for k, s in ins if k == "asm" and s.startswith(("ds_read_b128", "ds_read_b64_tr_b16")))
    return n_reads, violations


def audit_source(src):
    out = subprocess.run([HIPCC, *FLAGS, "-x", "hip", "-S", "--cuda-device-only", src, "-o", "-"], check=True,
                         capture_output=True, text=True).stdout
    report = []
    for name, body in functions(out.split("\n")):
        n_reads, viol = audit_function(name, body)
        if n_reads:
            report.append((name, n_reads, viol))
    return report


def pretty(mangled):
    """gemm_kernelILi64ELi64ELb0ELi4ELi2ELi1EE -> gemm_kernel<64,64,0,4,2,1> (enough to name an instantiation)"""
    m = re.search(r"\d+([a-z_]+kernel)I((?:L[ib]\d+E)+)E", mangled)
    if not m:
        return mangled[:80]
    return m.group(1) + "<" + ",".join(re.findall(r"L[ib](\d+)E", m.group(2))) + ">"


def main(argv):
    srcs = argv or [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))
                    if f.endswith(".hip") and re.search(r"lds_read16_async|lds_read16_at|lds_read_tr16_at", open(os.path.join(CSRC, f)).read())]
    bad = 0
    for src in srcs:
        for name, n_reads, viol in audit_source(src):
            short = pretty(name)
            print(f"{'FAIL' if viol else 'ok  '} {short}: {n_reads} async reads, {len(viol)} violations")
            for (lab, ins), regs in list(viol.items())[:12]:
                print(f"       {lab}: {ins}   <- in-flight v{regs}")
            bad += bool(viol)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
