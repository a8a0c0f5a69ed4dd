#!/usr/bin/env python3
"""Static audit of the gfx950 code objects, no GPU needed: compiles every csrc/*.hip to device assembly (hipcc -S --cuda-device-only, the
library's own flags) and lists, per kernel, registers / LDS / scratch / spills, and — the trap that cost the conv loops 20 % this round
(DESIGN.md §7a "operand-ahead") — scratch loads or `s_waitcnt vmcnt(0)` inside loops that also issue LDS-DMA or global loads.

  python tools/isa_audit.py [--keep DIR] [file.hip ...]        # default: every source of anyedit_amd/build.py
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def audit(asm_path):
    txt = open(asm_path).read()
    meta = {}
    for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
        blk = ".agpr_count:" + blk
        get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        meta[get("name")] = {k: get(k) for k in ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size",
                                                 "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count")}
    # per-kernel body: loops = backward branches; report vmcnt(0) / scratch traffic between a loop label and its backward branch
    bodies = {}
    for m in re.finditer(r"^(\S+):\s*; @\1\n(.*?)^\s+s_endpgm", txt, re.S | re.M):
        bodies[m.group(1)] = m.group(2)
    rows = []
    for name, md in meta.items():
        body = bodies.get(name, "")
        lines = body.split("\n")
        label_at = {}
        for i, l in enumerate(lines):
            mm = re.match(r"^(\.LBB\d+_\d+):", l)
            if mm:
                label_at[mm.group(1)] = i
        loops = []
        for i, l in enumerate(lines):
            mm = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in label_at and label_at[mm.group(1)] < i:
                loops.append((label_at[mm.group(1)], i))
        worst = None
        for a, b in loops:
            seg = lines[a:b]
            n_mfma = sum("v_mfma" in x for x in seg)
            n_vm0 = sum(bool(re.search(r"s_waitcnt.*vmcnt\(0\)", x)) for x in seg)
            n_scr = sum(bool(re.search(r"scratch_(load|store)|buffer_(load|store)_dword.*off, s\[0:3\]", x)) for x in seg)
            n_dma = sum(" lds" in x and "buffer_load" in x for x in seg)
            n_gl = sum(bool(re.search(r"(global|buffer)_load", x)) for x in seg) - n_dma
            n_vmn = sum(bool(re.search(r"s_waitcnt.*vmcnt\([1-9]\d*\)", x)) for x in seg)     # counted waits: loads left in flight
            if n_mfma and (worst is None or n_mfma > worst[0]):
                ops_ = [x.split()[0] for x in (y.strip() for y in seg) if x and x[0] not in ";."]
                n_valu = sum(o.startswith("v_") and not o.startswith("v_mfma") for o in ops_)
                n_salu = sum(o.startswith("s_") and not o.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_cbranch", "s_branch")) for o in ops_)
                n_lds = sum(o.startswith("ds_") for o in ops_)
                n_br = sum(o.startswith(("s_cbranch", "s_branch")) for o in ops_)
                worst = (n_mfma, n_vm0, n_scr, n_dma, n_gl, len(ops_), n_vmn, n_valu, n_salu, n_lds, n_br)
        rows.append((name, md, worst))
    return rows


def _regs(tok):
    """'v[18:33]' / 'v7' -> set of VGPR numbers; 'a[0:3]' / 'a5' -> AGPR numbers offset by 1000 (a separate file: an MFMA's AGPR destination read back by
    v_accvgpr_read is a read of its result like any other); anything else -> empty set."""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        off = 1000 if m.group(1) == "a" else 0
        return set(range(off + int(m.group(2)), off + int(m.group(3)) + 1))
    m = re.fullmatch(r"([va])(\d+)", tok)
    return {(1000 if m.group(1) == "a" else 0) + int(m.group(2))} if m else set()


def mfma_source_overwrites(asm_path, kernel_filter):
    """gfx950 does not interlock a VALU write to a VGPR that an issued-but-not-yet-finished MFMA still reads as SrcA / SrcB (found on the two-query-group
    attention kernel, profiles/r03_attn_qg2_hazard.txt).  The matrix pipe is in order and takes one MFMA at a time, so once TWO later MFMAs have been
    issued the first one has finished.  Returns, for every kernel whose demangled name contains `kernel_filter`, the list of
    (index, mfma, index, instruction) of instructions that write a source register of an MFMA before two further MFMAs were issued, on ANY path
    through the listing from that MFMA (unconditional branches are followed, conditional ones explored both ways; a path ends after 48
    instructions — ~200 issue cycles, beyond the ~128 cycles a queued 32x32x16 MFMA can still be reading)."""
    txt = open(asm_path).read()
    bad = {}
    names = re.findall(r"^(\S+):\s*; @\1\n", txt, re.M)
    dm = demangle(names)
    for m in re.finditer(r"^(\S+):\s*; @\1\n(.*?)^\s+s_endpgm", txt, re.S | re.M):
        if kernel_filter not in dm.get(m.group(1), ""):
            continue
        ins, label = [], {}
        for l in m.group(2).split("\n"):
            t = l.split(";")[0].strip()
            if not t:
                continue
            lm = re.match(r"^(\.LBB\d+_\d+):", t)
            if lm:
                label[lm.group(1)] = len(ins)
                continue
            if t.startswith("."):
                continue
            ins.append(t)
        hits = []
        for i, t in enumerate(ins):
            if not t.startswith("v_mfma"):
                continue
            ops_ = [x.strip() for x in t.split(None, 1)[1].split(",")]
            src = _regs(ops_[1]) | _regs(ops_[2])      # SrcA, SrcB (operand 0 is the destination, 3 the accumulator input)
            dstm0 = _regs(ops_[0])                     # (AGPR destinations parse as empty: no early exit for them — conservative)
            stack, visited = [(i + 1, 0, 0, frozenset(dstm0))], set()
            while stack:
                j, seen, steps, dstm = stack.pop()
                dstm = set(dstm)
                while j < len(ins) and steps < 48 and seen < 2:
                    if (j, seen) in visited:
                        break
                    visited.add((j, seen))
                    u = ins[j]
                    steps += 1
                    if u.startswith("v_mfma"):
                        seen += 1
                        # the matrix pipe is in order: a later reader of THIS MFMA's result also proves the flagged one finished (round 5)
                        dstm |= _regs([x.strip() for x in u.split(None, 1)[1].split(",")][0])
                        j += 1
                        continue
                    bm = re.match(r"^s_branch\s+(\.LBB\d+_\d+)", u)
                    if bm:
                        j = label.get(bm.group(1), len(ins))
                        continue
                    cm = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)", u)
                    if cm:
                        stack.append((label.get(cm.group(1), len(ins)), seen, steps, frozenset(dstm)))
                        j += 1
                        continue
                    # A non-MFMA instruction that READS the MFMA's destination (or that of a later MFMA: the pipe is in order) is issued only once that
                    # MFMA has written its result back, i.e. has finished — with all of its source reads: it, and everything behind it on this path, is
                    # safe.  (An MFMA taking the result as its C operand proves nothing: accumulate chains forward inside the matrix pipe.)  Round 5: without
                    # this the audit flagged the rebase path's rewrite of the Q operand's offset slot, which sits behind the whole maximum tree over the
                    # MFMA's result; AGPR results read back by v_accvgpr_read count the same way.
                    if dstm and " " in u and not u.startswith(("s_", "v_mfma")):
                        rd = u.split(None, 1)[1].split(",")
                        rd = rd[1:] if u.startswith(("v_", "ds_read", "buffer_load", "global_load")) and not u.startswith("v_cmp") else rd
                        if any(_regs(x.strip().split(" ")[0]) & dstm for x in rd):
                            break
                    # VALU writes only: an LDS / VMEM load returns its data 64+ cycles after issue (the reload of a fragment register right behind
                    # its last MFMA is what hipcc emits in every GEMM loop), a VALU result lands a few cycles after issue
                    if u.startswith("v_") and not u.startswith(("v_cmp", "v_nop", "v_readfirstlane", "v_readlane", "v_mfma")):
                        dst = _regs(u.split(None, 1)[1].split(",")[0].strip()) if " " in u else set()
                        if dst & src:
                            hits.append((i, t, j, u))
                    j += 1
        bad[dm[m.group(1)]] = sorted(set(hits))
    return bad


def compile_asm(srcs, outdir):
    """hipcc -S --cuda-device-only of csrc/<src> with the library's own flags -> list of .s paths (compiled in parallel)."""
    from anyedit_amd import build as B
    os.makedirs(outdir, exist_ok=True)
    hipcc = B._hipcc()

    def comp(src):
        out = os.path.join(outdir, os.path.basename(src).replace(".hip", ".s"))
        flags = [f for f in B.FLAGS if f != "-fPIC"] + B.EXTRA.get(os.path.basename(src), [])
        subprocess.run([hipcc] + flags + ["--cuda-device-only", "-S", os.path.join(B.CSRC, os.path.basename(src)), "-o", out], check=True,
                       stderr=subprocess.DEVNULL)
        return out

    with ThreadPoolExecutor(8) as ex:
        return list(ex.map(comp, srcs))


def audit_named(asm_path):
    """audit() with demangled kernel names: list of (name, metadata dict, hottest-MFMA-loop tuple or None)."""
    rows = audit(asm_path)
    dm = demangle([r[0] for r in rows])
    return [(dm[n], md, worst) for n, md, worst in rows]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keep", default=None)
    ap.add_argument("files", nargs="*")
    a = ap.parse_args()
    from anyedit_amd import build as B
    asms = compile_asm(a.files or B.SOURCES, a.keep or tempfile.mkdtemp(prefix="isa_"))
    print(f"{'kernel':100s} vgpr agpr  lds_B scratch_B vspill | hottest MFMA loop: mfma vmcnt0 scratch dma gloads instr vmcntN valu salu lds branch")
    flagged = 0
    for asm in asms:
        for name, md, worst in sorted(audit_named(asm)):
            # a vmcnt(0) inside an LDS-DMA loop is only counted, not flagged: the two-stage loops have one by design, and the operand-ahead
            # loops have it on the branch of the last K step
            bad = md["private_segment_fixed_size"] not in ("0", "?") or md["vgpr_spill_count"] not in ("0", "?") or (worst and worst[2] > 0)
            flagged += bool(bad)
            w = "%4d %6d %7d %3d %6d %5d %6d %4d %4d %3d %6d" % worst if worst else "-"
            print(f"{'!' if bad else ' '} {name[:98]:98s} {md['vgpr_count']:>4s} {md['agpr_count']:>4s} {md['group_segment_fixed_size']:>6s} "
                  f"{md['private_segment_fixed_size']:>9s} {md['vgpr_spill_count']:>6s} | {w}")
    print(f"{flagged} kernel(s) flagged ('!': scratch memory or register spills; in-loop scratch traffic is the 'scratch' column)")


if __name__ == "__main__":
    main()
