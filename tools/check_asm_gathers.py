#!/usr/bin/env python3
"""tools/check_asm_gathers.py <file.s ...> — build-time guard for the hand-issued gathers of the tap loops.

pm_tap_r5.h / pm_core_lut.h issue their buffer_load_dword ... idxen gathers from inline asm and write the s_waitcnt vmcnt(n)
waits out by hand, because no compiler builtin reaches the idxen form.  The compiler therefore does not know that a gather's
destination register is still in flight between the load and its wait: if register allocation ever copied, spilled or reused
such a register there (a new variant under VGPR pressure, a compiler update), the kernel would read stale data with no
diagnostic.  This script reads hipcc's assembly (tools/isa.sh) and checks, kernel by kernel, that between every idxen gather and
the first s_waitcnt that retires it NO instruction reads or writes the destination VGPR.

Model: vector-memory instructions retire in issue order (vmcnt counts loads and stores on gfx9-family targets); an
`s_waitcnt vmcnt(n)` retires all but the youngest n.  The kernel's control-flow graph is rebuilt from labels and branches, and
from every gather every path is followed (depth-first over (instruction, number of vector-memory instructions issued since))
until a wait retires the gather; block layout order does not matter.
Exit status 1 and one line per violation if anything is found (or if no gather was seen at all)."""
import re
import sys

VMEM = re.compile(r"^(buffer_|global_|flat_|scratch_)(load|store|atomic)")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse_kernels(path):
    """{kernel: [blocks]}, a block = (label, [(line number, text)])"""
    kernels, cur, blocks = {}, None, None
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip()
        if raw.startswith("_Z") and ":" in raw:
            cur = raw.split(":")[0]
            blocks = kernels.setdefault(cur, [("entry", [])])
            continue
        if cur is None or not line:
            continue
        if line.startswith(".size"):
            cur = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            blocks.append((m.group(1), []))
            continue
        if line.startswith(".") or line.endswith(":"):
            continue
        blocks[-1][1].append((ln, line))
    return kernels


def vmcnt_of(line):
    """outstanding count an s_waitcnt allows, or None if it does not wait on vmcnt"""
    m = re.search(r"vmcnt\((\d+)\)", line)
    if m:
        return int(m.group(1))
    m = re.match(r"s_waitcnt\s+(0x[0-9a-f]+|\d+)$", line)
    if m:                                   # raw immediate: vmcnt is bits 3:0 and 15:14
        imm = int(m.group(1), 0)
        return (imm & 0xf) | (((imm >> 14) & 0x3) << 4)
    return None


def check(path):
    bad, n_gathers = [], 0
    for kernel, blocks in parse_kernels(path).items():
        index = {lab: k for k, (lab, _) in enumerate(blocks)}

        def successors(b):
            ins = blocks[b][1]
            last = ins[-1][1] if ins else ""
            op = last.split()[0] if last else ""
            if op == "s_endpgm":
                return []
            out = []
            if op == "s_branch" or op.startswith("s_cbranch"):
                tgt = last.split()[-1]
                if tgt in index:
                    out.append(index[tgt])
            if op != "s_branch" and b + 1 < len(blocks):
                out.append(b + 1)
            return out

        for b0, (_, ins0) in enumerate(blocks):
            for i0, (ln0, line0) in enumerate(ins0):
                if not (line0.startswith("buffer_load") and "idxen" in line0):
                    continue
                n_gathers += 1
                dest = regs_of(line0.split()[1].rstrip(","))
                seen, stack, hit = set(), [(b0, i0 + 1, 0)], None
                while stack and hit is None:
                    b, i, k = stack.pop()
                    if (b, i, k) in seen or k > 63:
                        continue
                    seen.add((b, i, k))
                    ins = blocks[b][1]
                    retired = False
                    while i < len(ins):
                        ln, line = ins[i]
                        op = line.split()[0]
                        if op == "s_waitcnt":
                            n = vmcnt_of(line)
                            if n is not None and n <= k:
                                retired = True
                                break
                        elif regs_of(line) & dest:
                            hit = (ln, line)
                            break
                        if VMEM.match(op):
                            k += 1
                        i += 1
                    if hit is None and not retired:
                        for nb in successors(b):
                            stack.append((nb, 0, k))
                if hit is not None:
                    bad.append(f"{path}:{hit[0]}: `{hit[1]}` touches the destination of `{line0}` (line {ln0}) while it is in flight  [{kernel}]")
    return bad, n_gathers


def main():
    total_bad, total = [], 0
    for p in sys.argv[1:]:
        bad, n = check(p)
        total_bad += bad
        total += n
        print(f"{p}: {n} idxen gathers checked, {len(bad)} violations")
    for b in total_bad:
        print(b)
    return 1 if total_bad or total == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
