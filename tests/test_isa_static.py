"""Static checks on the gfx950 code of the hot-path kernels — no GPU: hipcc cross-compiles the three sources to device assembly and
tools/isa_audit.py reads registers, scratch and the wait structure of the main loops.

Why a test: the largest gain of round 3 came from removing two things no numerics test can see: loader lambdas that hipcc
did not inline kept their state in SCRATCH, and every scratch load is preceded by `s_waitcnt vmcnt(0)` — the whole LDS-DMA queue drained
once per K step, 20 % of the conv time.  These assertions fail if a change brings either back."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    import isa_audit as A
    out = tmp_path_factory.mktemp("isa")
    rows = []
    for asm in A.compile_asm(["gemm_conv.hip", "attention_fast.hip", "gemm_rowpanel.hip", "norm.hip"], str(out)):
        rows += A.audit_named(asm)
    assert len(rows) > 100
    return rows


def _targs(name, kernel):
    m = re.search(kernel + r"<(.*?)>\(", name)
    return [x.strip() for x in m.group(1).split(",")] if m else None


def test_operand_ahead_loops_keep_loads_in_flight(kernels):
    """gemm_kernel<..., WA = 1 | 2>: no scratch, no spills, six unrolled K steps per loop trip, and every step's wait is a COUNTED vmcnt
    (the far operand of the step after next stays in flight); the only vmcnt(0) sits on the branch of the last K step."""
    seen = 0
    for name, md, loop in kernels:
        a = _targs(name, "gemm_kernel")
        if not a or a[10] not in ("1", "2"):      # WA is the eleventh template argument (XE, the LayerNorm-fold epilogue selector, follows it)
            continue
        seen += 1
        assert md["private_segment_fixed_size"] == "0" and md["vgpr_spill_count"] == "0", name
        n_mfma, n_vm0, n_scratch, n_dma, n_gloads, _, n_vmn = loop[:7]
        bm, bn, wm, wn = int(a[0]), int(a[1]), int(a[3]), int(a[4])
        per_step = (bm // wm // 16) * (bn // wn // 16) * 2           # 16x16x32 MFMAs of one wave per 64-deep K tile
        assert n_mfma == 6 * per_step, (name, n_mfma)
        assert n_scratch == 0 and n_gloads == 0 and n_dma > 0, (name, loop)
        assert n_vmn >= 5 and n_vm0 <= 6, (name, loop)     # 5: hipcc rotates some loops, the sixth counted wait then sits above the loop label
    assert seen >= 4, "operand-ahead instantiations not found: did the template signature change?"


def test_ping_pong_loops_wait_by_count_and_alternate_phases(kernels):
    """gemm_kernel<192, 320, ..., WA = 3> (round 4): the six-tile loop body holds 6 x 60 MFMAs for every wave layout, no scratch, no spills,
    its LDS-DMA pieces go through the asm form (the listing shows them, hipcc's bookkeeping does not), and every K tile ends on a COUNTED
    vmcnt (the activations of the tile after next stay in flight) — a vmcnt(0) appears only on the branch of the last tiles."""
    seen = 0
    for name, md, loop in kernels:
        a = _targs(name, "gemm_kernel")
        if not a or a[10] != "3":
            continue
        seen += 1
        assert a[0] == "192" and a[1] == "320", name
        assert md["private_segment_fixed_size"] == "0" and md["vgpr_spill_count"] == "0", name
        n_mfma, n_vm0, n_scratch, n_dma, n_gloads, _, n_vmn = loop[:7]
        assert n_mfma == 360, (name, n_mfma)
        assert n_scratch == 0 and n_gloads == 0 and n_dma >= 48, (name, loop)
        assert n_vmn >= 5 and n_vm0 <= 6, (name, loop)
    assert seen >= 5, "ping-pong instantiations not found (conv with / without statistics, dense 2x4, GEGLU 4x2 without / with the LayerNorm fold)"


def test_slab_ping_pong_loop_structure(kernels):
    """gemm_kernel<192, 320, conv, ..., WA = 4> (round 4: ping-pong over an activation slab per (chunk, ky)): 18 unrolled K tiles of 60 MFMAs, 18 x 5
    weight pieces + 6 x 4 slab pieces per trip (a slab is issued on the kx = 0 tiles only), no scratch, no spills, and — the point of the
    zero-row layout — next to no VALU in the loop (the conv's padding is not computed there)."""
    seen = 0
    for name, md, loop in kernels:
        a = _targs(name, "gemm_kernel")
        if not a or a[10] != "4":
            continue
        seen += 1
        assert a[:3] == ["192", "320", "1"], name
        assert md["private_segment_fixed_size"] == "0" and md["vgpr_spill_count"] == "0", name
        n_mfma, n_vm0, n_scratch, n_dma, n_gloads, _, n_vmn, n_valu = loop[:8]
        assert n_mfma == 18 * 60 and n_dma == 18 * 5 + 6 * 4, (name, loop)
        assert n_scratch == 0 and n_gloads == 0 and n_vmn == 6 and n_valu <= 72, (name, loop)
    assert seen == 2, "slab instantiations (with / without column statistics) not found"


def test_layernorm_fold_epilogues_leave_the_main_loops_alone(kernels):
    """gemm_kernel<..., XE = 1 | 2> (round 4: row statistics out / LayerNorm fold in): five instantiations, none with scratch or spills (the
    192x320 GEGLU form keeps the block's s and c vectors in LDS for that reason), and each one's hottest loop has the MFMA, DMA and wait
    counts of its XE = 0 twin — the two row scalars per fragment row the fold carries through the loop cost it nothing."""
    by_args = {tuple(_targs(name, "gemm_kernel")): (md, loop) for name, md, loop in kernels if _targs(name, "gemm_kernel")}
    seen = 0
    for a, (md, loop) in by_args.items():
        if a[11] == "0":
            continue
        seen += 1
        assert md["private_segment_fixed_size"] == "0" and md["vgpr_spill_count"] == "0", a
        twin = by_args[a[:11] + ("0",)][1]
        assert loop[:5] == twin[:5] and loop[6] == twin[6], (a, loop, twin)
        if a[:2] == ("128", "128") and a[7] == "2":
            # two 8-wave blocks per CU = four waves per SIMD = at most 128 registers: at 129 the qkv / q launches of the 32x32 / 16x16 levels take 47 us
            # instead of 38 (profiles/r04_v35_xe_occ_ab.txt: 0.12 ms per UNet step) — the kernel's launch bound states it, this checks the allocator obeyed
            assert int(md["vgpr_count"]) + int(md["agpr_count"]) <= 128, (a, md["vgpr_count"], md["agpr_count"])
    assert seen == 5, seen


def test_hot_loops_have_no_scratch_traffic(kernels):
    """Every attention, row-panel and GEMM / conv main loop of the denoising path: no scratch loads or stores inside the hottest MFMA loop;
    the fast attention kernels and the single-launch GroupNorm of the bench's shapes do not spill at all."""
    n = {"attn": 0, "rp": 0, "gemm": 0, "slab": 0}
    for name, md, loop in kernels:
        if "attn_fast_kernel<" in name:
            n["attn"] += 1
            assert md["private_segment_fixed_size"] == "0" and md["vgpr_spill_count"] == "0" and loop and loop[2] == 0, name
        elif "gemm_rowpanel_kernel<" in name:
            n["rp"] += 1
            assert loop and loop[2] == 0, name                      # its LayerNorm prologue spills; the chunk loop must not touch scratch
        elif "gemm_kernel<" in name and loop:
            n["gemm"] += 1
            assert loop[2] == 0, name
        elif "gn_slab_kernel<" in name:
            a = _targs(name, "gn_slab_kernel")
            if int(a[0]) <= 8:
                n["slab"] += 1
                assert md["vgpr_spill_count"] == "0", name
    assert n["attn"] >= 10 and n["rp"] >= 3 and n["gemm"] >= 20 and n["slab"] >= 9, n


def test_no_vgpr_write_to_the_sources_of_a_running_mfma(tmp_path):
    """gfx950 does not interlock a VALU write to an MFMA's SrcA / SrcB registers while that MFMA is still reading them
    (profiles/r03_attn_qg2_hazard.txt), and hipcc pads nothing for it.  The attention kernel keeps such registers live by empty asm statements
    (`hold`, `srcring`, the P registers) — which are only as good as the listing hipcc produces from them: read the listing.  In EVERY
    instantiation (two query groups, one group tiled, short K/V with and without the adapter segment, SAM's rel-pos forms, head dims 40 / 80 /
    160) no VALU instruction may write a source register of an MFMA before two further MFMAs have been issued or an instruction has read that
    MFMA's result (either proves the in-order matrix pipe is done with it).  Round 4 asserted the two-query-group instantiation only; round 5
    found v_exp_f32 / address writes one to four instructions behind an MFMA in the head-dim 80 / 160 and SAM-window instantiations and tied
    the fences of the one-group forms to the MFMA results."""
    import isa_audit as A
    asm = A.compile_asm(["attention_fast.hip"], str(tmp_path))[0]
    res = A.mfma_source_overwrites(asm, "attn_fast_kernel<")
    assert len(res) >= 20
    kinds = {"qg2": r"attn_fast_kernel<\d+, \d+, (true|false), 0, 0, 2,", "short_kv": r"attn_fast_kernel<\d+, \d+, (true|false), 0, 0, 1, false, true",
             "short_kv_seg2_d160": r"attn_fast_kernel<160, 1, true, 0, 0, 1, false, true", "sam_window": r"attn_fast_kernel<80, 2, false, 0, 3, 1, false, true, 7, 4>",
             "sam_global": r"attn_fast_kernel<80, 2, false, 0, 2,"}
    for kind, pat in kinds.items():
        assert [k for k in res if re.search(pat, k)], f"{kind}: instantiation not found — did the template signature change?"
    for k, hits in res.items():
        assert not hits, (k, hits[:3])
    # round 5: the software-pipelined kernel — its 14 MFMAs per step are chained into one program order and every fragment is tied to the MFMA two
    # positions behind its last reader (the first listing of that kernel, without the ties, had 186 such writes)
    pipe = A.mfma_source_overwrites(asm, "attn_pipe_kernel<")
    assert len(pipe) == 3                           # keys per tile 64 (round 5), 128 (round 6), and 128 with the 48-row 16x16x32 PV products (round 6, opt-in)
    for k, hits in pipe.items():
        assert not hits, (k, hits[:3])
    rows = [(n, md, loop) for n, md, loop in A.audit_named(asm) if "attn_pipe_kernel<" in n]
    assert len(rows) == 3
    for n, md, loop in rows:
        steps = 4 if "<40, 128" in n else 2         # 32-key blocks (pipeline steps) per loop trip = per tile
        pv16 = "true>" in n                         # 6 logit MFMAs + 12 PV MFMAs of 16x16x32 per step instead of 6 + 8 of 32x32x16; 8 lane swaps more
        assert md["vgpr_spill_count"] == "0" and md["private_segment_fixed_size"] == "0" and int(md["vgpr_count"]) <= 256, (n, md)   # two waves per SIMD
        assert loop[0] == (18 if pv16 else 14) * steps and loop[2] == 0, (n, loop)   # one tile per loop trip, no scratch traffic
        assert loop[7] <= (88 if pv16 else 80) * steps, (n, loop)     # VALU per trip (74 per step at the time of writing: 32 exp2, 16 converts, 17 max3, addresses): register copies would show here


def test_fused_feed_forward_stream_is_the_hand_placed_one(tmp_path):
    """Round 6 (csrc/ff_fused.hip): the fused feed-forward kernel lives at the edge of the register file (one wave per SIMD, 240 output accumulators in
    the accumulator half, ~245 VGPRs) and its MFMAs are asm statements hipcc neither pads nor tracks.  The listing must show: no spills, no scratch;
    the main loop (two steps = four phases per trip) holds 4 x 90 MFMAs, its LDS-DMA pieces (4 x 8) and NOTHING moving between the two register files
    (the compiler-scheduled forms moved ~80 values per step, profiles/r06_ff_fused_notes.txt); only counted DMA waits inside the loop; and no VALU / LDS
    result lands in a source register of an MFMA before two further MFMAs were issued (W fragments are kept live by empty asm uses)."""
    import isa_audit as A
    asm = A.compile_asm(["ff_fused.hip"], str(tmp_path))[0]
    rows = [(n, md, loop) for n, md, loop in A.audit_named(asm) if "ff_fused_kernel<" in n]
    assert len(rows) == 2, [r[0] for r in rows]            # with and without the proj_out tail
    for n, md, loop in rows:
        assert md["vgpr_spill_count"] == "0" and md["private_segment_fixed_size"] == "0", (n, md)
        assert int(md["agpr_count"]) >= 240 and int(md["vgpr_count"]) <= 512, (n, md)
        n_mfma, n_vm0, n_scratch, n_dma, n_gloads, _, n_vmn = loop[:7]
        assert n_mfma == 360 and n_dma == 32 and n_scratch == 0 and n_gloads == 0 and n_vm0 == 0 and n_vmn == 4, (n, loop)
    txt = open(asm).read()
    for m in re.finditer(r"^(\S+ff_fused_kernel\S+):\s*; @\1\n(.*?)^\s+s_endpgm", txt, re.S | re.M):
        body = m.group(2).split("\n")
        # the main loop: the backward branch with the most MFMAs between label and branch
        labels = {mm.group(1): i for i, l in enumerate(body) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
        best = (0, 0, 0)
        for i, l in enumerate(body):
            mm = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                seg = body[labels[mm.group(1)]:i]
                best = max(best, (sum("v_mfma" in x for x in seg), labels[mm.group(1)], i))
        seg = body[best[1]:best[2]]
        assert best[0] == 360
        assert sum("v_accvgpr" in x for x in seg) <= 2, [x for x in seg if "v_accvgpr" in x][:5]
    res = A.mfma_source_overwrites(asm, "ff_fused_kernel<")
    assert len(res) == 2
    for k, hits in res.items():
        assert not hits, (k, hits[:3])


def test_fused_cross_attention_stream_keeps_its_mfma_sources(tmp_path):
    """Round 6 (csrc/xattn_fused.hip, opt-in): the same hand-placed technique as the fused feed-forward — asm MFMAs hipcc neither pads nor tracks, with q, probabilities and
    attention outputs written by VALU and consumed as MFMA operands a few instructions later.  The listing must show no spills / scratch and no VALU result landing in a source
    register of an MFMA before two further MFMAs were issued or a read of that MFMA's (or a later one's) result (the first listing had 26 such writes: pack temporaries in the W
    fragments of a phase's last slots, the row maximum in the q operands)."""
    import isa_audit as A
    asm = A.compile_asm(["xattn_fused.hip"], str(tmp_path))[0]
    rows = [(n, md, loop) for n, md, loop in A.audit_named(asm) if "xattn_fused_kernel" in n]
    assert len(rows) == 1
    _, md, _ = rows[0]
    assert md["vgpr_spill_count"] == "0" and md["private_segment_fixed_size"] == "0" and int(md["agpr_count"]) >= 160, md
    res = A.mfma_source_overwrites(asm, "xattn_fused_kernel")
    assert len(res) == 1
    for k, hits in res.items():
        assert not hits, (k, hits[:3])


def test_attention_backward_keeps_valu_writes_behind_its_mfma_phases(tmp_path):
    """Round 5: the same listing check on EVERY instantiation of attn_bwd_kernel (training step).  The passes are phase-structured — first-product MFMAs,
    the exp2 / P block, second-product MFMAs — and hipcc re-uses the operand registers of a phase's last MFMAs for the first VALU results of the next (the
    first audit of this file: 904 such writes; e.g. `v_exp_f32 v184, ...` six instructions behind `v_mfma ... v[184:187]`).  Each phase now ends in a fence
    that READS the results of its last MFMA group (in-order matrix pipe: everything before is finished), and the two-accumulator dQ pass at head dims 64 / 80
    runs one query fragment per wave (at two it needs 328 registers and hipcc shuffled accumulators through AGPRs between the MFMAs)."""
    import isa_audit as A
    asm = A.compile_asm(["attention_bwd.hip"], str(tmp_path))[0]
    res = A.mfma_source_overwrites(asm, "attn_bwd_kernel<")
    assert len(res) >= 27, sorted(res)
    for want in ("attn_bwd_kernel<40, 2, 0, true>", "attn_bwd_kernel<40, 2, 0, false>", "attn_bwd_kernel<40, 2, 1, false>", "attn_bwd_kernel<80, 1, 0, false>",
                 "attn_bwd_kernel<80, 2, 1, false>", "attn_bwd_kernel<160, 1, 1, false>"):
        assert [k for k in res if want in k], f"{want}: instantiation not found"
    assert not [k for k in res if "attn_bwd_kernel<80, 2, 0, false>" in k or "attn_bwd_kernel<64, 2, 0, false>" in k]
    for k, hits in res.items():
        assert not hits, (k, hits[:3])
    for n, md, loop in A.audit_named(asm):
        if "attn_bwd_kernel<" in n:
            assert md["private_segment_fixed_size"] == "0" and md["vgpr_spill_count"] == "0", (n, md)
