#!/usr/bin/env python3
"""Static check of the wide GEMM kernels' disassembly (csrc/gemm_bf16_wide.hpp).

The kernels issue their matrix instructions as inline assembly, so hipcc's hazard recogniser does not know that an
accumulator register was just written through the matrix pipe.  The one thing it can get wrong is a register COPY it
inserts itself (v_accvgpr_read / v_accvgpr_mov with an AGPR source, on a loop edge or where it splits a live range) too
close behind the v_mfma that wrote the register: the ISA asks for 19 wait states between an 16-pass XDL write and a
VALU read of the result (11 for 8 passes; we demand 19).  This script walks the control-flow graph of every
`gemm_wide` kernel in an assembly file (hipcc -S) and fails if any such read can happen sooner.  It also reports how
many accumulator copies sit inside the steady k-loop (expected: none).

Second rule: the accumulators live in a0..a127 WITHOUT the compiler knowing (they are named only inside the assembly text
and in clobber lists), so the compiler may think those registers are free between two matrix instructions and park a
spilled vector register there.  Any compiler-generated write to a0..a127 (v_accvgpr_write_b32 from a VGPR, v_accvgpr_mov)
fails the check.  The build runs this script on the shipped translation unit (csrc/Makefile, `wide-check`).

usage: check_wide_hazards.py file.s
"""
import re, sys

NEED = 19
re_mfma = re.compile(r"^\s*v_mfma_\S+\s+a\[(\d+):(\d+)\]")
re_acc_src = re.compile(r"^\s*v_accvgpr_(read_b32|mov_b32)\s+\S+,\s*a(\d+)")
re_acc_st = re.compile(r"^\s*ds_write_b128\s+v\d+,\s*a\[(\d+):(\d+)\]")      # panel kernel: the partial sums go to LDS from the accumulation file
re_label = re.compile(r"^(\.LBB\d+_\d+):")
re_branch = re.compile(r"^\s*s_c?branch\S*\s+(\.LBB\d+_\d+)")
re_nop = re.compile(r"^\s*s_nop\s+(\d+)")
re_acc_wr = re.compile(r"^\s*v_accvgpr_(write_b32\s+a(\d+),\s*v\d+|mov_b32\s+a(\d+),)")
N_ACC_REGS = 128


def kernels(text):
    cur, name = None, None
    for line in text.splitlines():
        m = re.match(r"^(_Z\S*(?:gemm_wide|gemm_panel|chain_kernel)\S*):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if re.match(r"^\.Lfunc_end\d+:", line):      # (not the first s_endpgm: an early exit may sit in front of the body)
                yield name, cur
                cur = None
                continue
            cur.append(line)


def is_panel(name):
    """kernels built on the panel body (gemm_bf16_panel.hpp): the one-GEMM kernel and the layer chain (gemm_bf16_chain.hpp)"""
    return "gemm_panel" in name or "chain_kernel" in name


def check(name, lines):
    n_acc_regs = 224 if is_panel(name) else N_ACC_REGS      # the panel body also keeps its weight fragments there (a128..a223)
    # instructions: (kind, payload); blocks split at labels and after branches
    ins = []
    in_asm = False
    asm_flag = []
    # a branch out of the +-32 K-instruction range is expanded into s_getpc / s_add_u32 (label - .Lpost_getpc) / s_addc / s_setpc:
    # to the walk below that is an unconditional branch to the label
    far = None
    fixed = []
    for ln in lines:
        m = re.match(r"^\s*s_add_u32\s+\S+,\s*\S+,\s*\((\.LBB\d+_\d+)-\.Lpost_getpc\d+\)", ln)
        if m:
            far = m.group(1)
        if re.match(r"^\s*s_setpc_b64", ln):
            assert far is not None, "s_setpc_b64 without a recognisable target"
            ln = "\ts_branch " + far
            far = None
        fixed.append(ln)
    lines = fixed
    for ln in lines:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        if not t or t.startswith(";") or t.startswith("."):
            m = re_label.match(ln)
            if m:
                ins.append(("label", m.group(1)))
                asm_flag.append(False)
            continue
        m = re_label.match(ln)
        if m:
            ins.append(("label", m.group(1)))
            asm_flag.append(False)
            continue
        ins.append(("op", ln))
        asm_flag.append(in_asm)
    label_at = {p: i for i, (k, p) in enumerate(ins) if k == "label"}
    # state: dict areg -> wait states since the last matrix write (capped at NEED); dataflow to a fixpoint, min over paths
    n = len(ins)
    state_in = [None] * (n + 1)
    state_in[0] = {}
    work = [0]
    bad = []

    def merge(i, st):
        old = state_in[i]
        if old is None:
            state_in[i] = dict(st)
            return True
        changed = False
        for r, d in st.items():
            if r not in old or d < old[r]:
                old[r] = d
                changed = True
        return changed

    seen_bad = set()
    while work:
        i = work.pop()
        st = dict(state_in[i])
        while i < n:
            kind, p = ins[i]
            nxt_fall = True
            if kind == "op":
                m = re_acc_src.match(p)
                if m:
                    r = int(m.group(2))
                    if r in st and st[r] < NEED and i not in seen_bad:
                        seen_bad.add(i)
                        bad.append((i, p.strip(), st[r]))
                m = re_acc_st.match(p)
                if m:
                    for r in range(int(m.group(1)), int(m.group(2)) + 1):
                        if r in st and st[r] < NEED and i not in seen_bad:
                            seen_bad.add(i)
                            bad.append((i, p.strip(), st[r]))
                adv = 1
                m = re_nop.match(p)
                if m:
                    adv = int(m.group(1)) + 1
                st = {r: d + adv for r, d in st.items() if d + adv < NEED}
                # (matrix instructions the COMPILER emitted -- the chain kernel's attention stages -- are its hazard recogniser's
                #  business: it knows their pass counts; only the inline-assembly ones are invisible to it)
                m = re_mfma.match(p) if asm_flag[i] else None
                if m:
                    for r in range(int(m.group(1)), int(m.group(2)) + 1):
                        st[r] = 0
                m = re_branch.match(p)
                if m:
                    tgt = label_at.get(m.group(1))
                    if tgt is not None and merge(tgt, st):
                        work.append(tgt)
                    if p.strip().startswith("s_branch"):
                        nxt_fall = False
                if "s_endpgm" in p:
                    nxt_fall = False
            i += 1
            if not nxt_fall:
                break
            if i < n and ins[i][0] == "label":
                if merge(i, st):
                    st = dict(state_in[i])
                else:
                    break
    # ---- second rule: compiler writes into a0..a127 from which an asm matrix instruction or the asm read-out is still
    #      reachable (a spill the epilogue makes AFTER the read-out is harmless)
    succ = [[] for _ in range(n)]
    for i, (kind, p) in enumerate(ins):
        fall = True
        if kind == "op":
            m = re_branch.match(p)
            if m:
                tgt = label_at.get(m.group(1))
                if tgt is not None:
                    succ[i].append(tgt)
                if p.strip().startswith("s_branch"):
                    fall = False
            if "s_endpgm" in p:
                fall = False
        if fall and i + 1 < n:
            succ[i].append(i + 1)
    pred = [[] for _ in range(n)]
    for i in range(n):
        for j in succ[i]:
            pred[j].append(i)
    live = [False] * n
    stack = []
    for i, (kind, p) in enumerate(ins):
        if kind == "op" and asm_flag[i] and (re_mfma.match(p) or re.match(r"^\s*v_accvgpr_read_b32", p)):
            live[i] = True
            stack.append(i)
    # the panel body marks where none of its fixed registers holds anything (`panel_state_dead`: on entry and on return):
    # the chain kernel runs compiler-allocated code (its attention stages) between two calls of the body, and what the
    # compiler parks in a0..a223 THERE is harmless -- liveness does not propagate backwards through a marker
    dead_mark = [kind == "op" and "panel_state_dead" in p for kind, p in ins]
    while stack:
        i = stack.pop()
        if dead_mark[i]:
            continue
        for j in pred[i]:
            if not live[j]:
                live[j] = True
                stack.append(j)
    spills = []
    for i, (kind, p) in enumerate(ins):
        if kind != "op" or asm_flag[i]:
            continue
        m = re_acc_wr.match(p)
        if m and int(m.group(2) or m.group(3)) < n_acc_regs and live[i]:
            spills.append(p.strip())
    return bad, spills


def inflight_copies(lines):
    """Registers written by an (inline-assembly) ds_read_b128 hold data only after the next `s_waitcnt lgkmcnt(0)`; the
    compiler does not know the statement is a load, so under register pressure it may spill (v_accvgpr_write) or move
    (v_mov) such a register right behind the read -- saving garbage.  Linear scan (the reads and their waits sit in
    straight-line code): every copy of a register with a read in flight is an error."""
    pending, out = {}, []
    for l in lines:
        t = l.strip()
        m = re.match(r"ds_read_b128 v\[(\d+):(\d+)\]", t)
        if m:
            for r in range(int(m.group(1)), int(m.group(2)) + 1):
                pending[r] = t
            continue
        if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
            pending = {}
            continue
        if re.match(r"^\.LBB\d+_\d+:", t):      # (reads and their waits sit in one basic block: a label starts over)
            pending = {}
            continue
        m = re.match(r"v_accvgpr_write_b32 a\d+, v(\d+)", t) or re.match(r"v_mov_b32_e32 v\d+, v(\d+)", t)
        if m and int(m.group(1)) in pending:
            out.append(t)
    return out


def inflight_vm_copies(lines):
    """Panel kernel: registers written by an inline-assembly global_load hold data only after the counted `s_waitcnt vmcnt`
    that names them -- at the latest the SECOND vmcnt wait behind the load (a chunk requested in step 3 of a trip is named by
    the wait of step 1 of the next one; the row-list entry of step 0 by the wait of step 2).  Any compiler copy of such a
    register before that is an error.  The text is scanned twice so that a load at the end of the loop body meets the waits
    at its top."""
    pending, out = {}, []
    in_asm = False
    for l in lines + lines:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = in_asm and re.match(r"global_load_dword(x4)? v(?:\[(\d+):(\d+)\]|(\d+)),", t)
        if m:
            lo = int(m.group(2) if m.group(2) else m.group(4))
            hi = int(m.group(3) if m.group(3) else m.group(4))
            for r in range(lo, hi + 1):
                pending[r] = 0
            continue
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
        if m:
            if int(m.group(1)) == 0:
                pending = {}
            else:
                pending = {r: c + 1 for r, c in pending.items() if c + 1 < 2}
            continue
        m = re.match(r"v_accvgpr_write_b32 a\d+, v(\d+)", t) or re.match(r"v_mov_b32_e32 v\d+, v(\d+)", t)
        if m and int(m.group(1)) in pending and t not in out:
            out.append(t)
    return out


def loop_copies(lines):
    """accumulator copies between the first two s_barrier of the steady loop body (a rough but stable proxy)"""
    text = "\n".join(lines)
    m = re.search(r"Inner Loop Header.*?\n(.*?)s_cbranch_scc\d \.LBB", text, re.S)
    return None


def main():
    text = open(sys.argv[1]).read()
    total_bad = 0
    found = 0
    for name, lines in kernels(text):
        found += 1
        bad, wr = check(name, lines)
        n_mfma = sum(1 for l in lines if re_mfma.match(l))
        print(f"{name}: {n_mfma} matrix instructions, {len(bad)} accumulator reads closer than {NEED} wait states behind their write, "
              f"{len(wr)} compiler writes into a0..a{(224 if is_panel(name) else N_ACC_REGS) - 1} while the accumulators are live")
        for l in wr[:10]:
            print("   ", l)
        total_bad += len(wr)
        fl = inflight_copies(lines)
        if fl:
            print(f"    {len(fl)} copies of registers whose ds_read has not been waited for:")
            for l in fl[:6]:
                print("   ", l)
        total_bad += len(fl)
        if is_panel(name):
            fv = inflight_vm_copies(lines)
            if fv:
                print(f"    {len(fv)} copies of registers whose global_load has not been waited for:")
                for l in fv[:6]:
                    print("   ", l)
            total_bad += len(fv)
        for i, p, d in bad[:10]:
            print(f"   after {d} wait states: {p}")
        total_bad += len(bad)
    if not found:
        print("no gemm_wide / gemm_panel / chain kernel in", sys.argv[1])
        return 2
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
