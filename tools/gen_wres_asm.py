#!/usr/bin/env python3
"""Generates multimodal_eeg_fmri_amd/csrc/conv3d_wres_asm.inc: the hand-scheduled gfx950 instruction
streams of the weight-resident 3-D convolution kernel (csrc/conv3d_wres.hip).

Why assembly: one wave per SIMD issues everything itself, so the LDS fragment reads of the next tap, the
sixteen MFMAs of this tap, the previous tile's pack / store / BatchNorm-sum instructions, the next tile's
halo prefetch and its way into LDS have to sit at fixed places in each other's shadow.  hipcc (ROCm 7.2)
re-clusters the ds_reads of an unrolled K loop into bursts next to their consumers and leaves 30-40 % of the
loop waiting on LDS latency, and whatever it schedules at a tile boundary runs with the matrix pipe idle.

MFMA shape: v_mfma_f32_16x16x32_bf16 (K = 32 = all input channels of one tap).  Same FLOP per pipe cycle as
32x32x16, but the chip holds a ~14 % higher clock on it under this kernel's load (1.79 vs 1.57 GHz measured
with the in-kernel stamps, tools/kbench.py stamp) - cdna_hip_programming.md rule 28.

The halo is a RING of six depth planes (10 x 12 rows x 64 B each; plane p of a (b, h, w) tile column sits in
slot p mod 6).  A workgroup walks a run of tiles with the depth index fastest.  The next tile down the column
shares two of its six planes with this one and needs four new ones; a plane dies after the last kd phase that
reads it, so the new planes take the dead ones' slots WHILE the tile computes:
    after kd = 0 (tap 8)   plane d0 dead   -> new plane d0 + 6
    after kd = 1 (tap 17)  plane d0 + 1    -> new plane d0 + 7
    after kd = 2 (tap 26)  planes d0 + 2, 3 -> new planes d0 + 8, 9
one s_barrier each, no second barrier (every write is followed by a later barrier before its first read), and
the tile boundary disappears from the critical path (round-2 stamps: ~1 400 cycles per tile before).  A column
change takes the classic boundary (six planes prefetched during the last tile, written between two barriers).

Register plan (accumulator file, named literally; every statement lists a0-a255 as clobbers):
  a[0:63]     two fragment sets (tap parity): A_i (i = 0..3, the wave's four 4x4-voxel patches) at 32 s + 4 i,
              B_j (j = 0..3, sixteen output channels each) at 32 s + 16 + 4 j; ds_read_b128 straight into AGPRs
  a[64:127]   accumulator set X: MFMA tile (i, j) at 64 + 4 (4 i + j)
  a[128:191]  accumulator set Y
  a[192:239]  prefetched halo planes: plane pp (0..5), chunk j (0..1) of a lane at 192 + 4 (2 pp + j)
  a[240:243]  per-lane constants of the two plane chunks: byte offset in the plane, one-hot (hh, hw) selector
  a[244:251]  BatchNorm sums of the lane's four channels (sum x 4, sum of squares x 4)
  a[252:255]  the lane's four bias values
Scratch VGPRs v[224:255] and SGPRs s[90:93] are named literally too (clobbers; dead outside a statement).
"""
import os
import sys

# ablation builds (tools/abl_build.sh): 1 no epilogue instructions, 2 no ds_reads inside the K loop, 4 no MFMAs,
# 8 no global stores, 64 no halo prefetch, 128 no in-loop barriers, 256 no in-loop vmcnt waits.  0 in the product.
ABL = int(os.environ.get("WRES_ABL", "0"))
# schedule / epilogue experiments (A/B builds, profiles/r03_wres_ab.txt; 0 in the product): 1 packed fp32 math
# (v_pk_*) for the bias add and the BatchNorm sums inside the K loop (measured: +12 % K-loop cycles - a v_pk_* in an
# MFMA's shadow costs more than the two instructions it replaces), 2 a side instruction ALSO in the MFMA gaps that
# carry a ds_read (round 2's schedule; one instruction per gap is 2 % fewer K-loop cycles)
OPT = int(os.environ.get("WRES_OPT", "0"))
# cache policy of the output stores: write-through ("sc0 sc1") - the 16.8 MB leave the XCD's L2 while the kernel
# still computes instead of as one write-back burst at the kernel boundary (measured at C2, graph-replayed:
# plain 17.9 us, sc1 / nt 16.6, sc0 sc1 15.8); the consumer is on another XCD's L2 anyway.
STORE_BITS = os.environ.get("WRES_STORE_BITS", "sc0 sc1")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "multimodal_eeg_fmri_amd", "csrc", "conv3d_wres_asm.inc")
ACC = {"X": 64, "Y": 128}
HALO, GOFFP, SELP, STATS, BIAS = 192, 240, 242, 244, 252
NAGPR = 256
ROWB, WP, BN = 64, 12, 64
TAPS = 27
S1 = ["v%d" % (224 + j) for j in range(4)]          # running BatchNorm sums while a statement runs
S2 = ["v%d" % (228 + j) for j in range(4)]
SH = ["v%d" % (232 + j) for j in range(4)]          # bias
TT = ["v%d" % (236 + j) for j in range(4)]          # epilogue values
PK = "v[240:241]"
VOFA, VOFB = "v242", "v243"                         # output row offsets of patch rows h 0-3 / 4-7
PF0, PF1 = "v244", "v245"                           # prefetch temporaries
PW0, PW1 = "v246", "v247"                           # plane-write addresses
SCRATCH_V = list(range(224, 256))
SCRATCH_S = [90, 91, 92, 93]


def frag(s, which):
    b = 32 * s + (0 if which[0] == "A" else 16) + 4 * int(which[1])
    return f"a[{b}:{b + 3}]"


def acc(cur, i, j):
    b = ACC[cur] + 4 * (4 * i + j)
    return f"a[{b}:{b + 3}]"


def halo(pp, j):
    b = HALO + 4 * (2 * pp + j)
    return f"a[{b}:{b + 3}]"


def a_read(t, i, base=None):
    kd, kh, kw = t // 9, (t // 3) % 3, t % 3
    off = (kh * WP + kw) * ROWB + (i >> 1) * 4 * WP * ROWB + (i & 1) * 4 * ROWB
    return f"ds_read_b128 {frag(t & 1, 'A%d' % i)}, {base or '%%[ab%d%d]' % (kd, kh & 1)} offset:{off}"


def b_read(t, j):
    half = 1 if t >= 13 else 0
    off = (t - 13 * half) * BN * ROWB + j * 16 * ROWB
    return f"ds_read_b128 {frag(t & 1, 'B%d' % j)}, %[bb{half}] offset:{off}"


READ_ORDER = ("A0", "B0", "A1", "B1", "A2", "B2", "A3", "B3")
MFMA_ORDER = ((0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (2, 1), (0, 2), (1, 2), (2, 2), (3, 0), (3, 1), (3, 2), (0, 3), (1, 3),
              (2, 3), (3, 3))


class Stream:
    """instruction list + in-order scoreboards of the two memory counters"""

    def __init__(self):
        self.lines = []
        self.ds, self.ds_done = [], 0         # LDS operations in issue order (reads tagged (tap, which), writes "W")
        self.vm, self.vm_done = [], 0         # vector-memory operations issued by THIS statement
        self.first_wait_done = False

    def emit(self, s):
        self.lines.append(s)
        op = s.split()[0]
        if op.startswith("buffer_load") or op.startswith("global_store") or op.startswith("global_load_lds"):
            self.vm.append(s)
        elif op.startswith("ds_write"):
            self.ds.append("W")

    def read(self, t, which, base=None, force=False):
        if (ABL & 2) and not force:
            return
        self.lines.append(a_read(t, int(which[1]), base) if which[0] == "A" else b_read(t, int(which[1])))
        self.ds.append((t, which))

    def wait_ds(self, idx):
        """every LDS operation up to index idx (inclusive) has completed"""
        if idx < self.ds_done:
            return
        younger = len(self.ds) - 1 - idx
        assert younger <= 15
        if not self.first_wait_done:          # compiler code ran since the tap-0 reads were issued (scalar loads share the
            younger, idx = 0, len(self.ds) - 1    # counter and return out of order): the first wait drains everything
            self.first_wait_done = True
        self.lines.append(f"s_waitcnt lgkmcnt({younger})")
        self.ds_done = idx + 1

    def need(self, t, names):
        idxs = [i for i, e in enumerate(self.ds) if e != "W" and e[0] == t and e[1] in names]
        if idxs:
            self.wait_ds(max(idxs))

    def need_vm(self, pred):
        """every vector-memory operation of this statement matching pred has completed (older ones too)"""
        idxs = [i for i, s in enumerate(self.vm) if pred(s)]
        if not idxs or max(idxs) < self.vm_done:
            return
        idx = max(idxs)
        younger = len(self.vm) - 1 - idx
        assert younger <= 63
        self.lines.append(f"s_waitcnt vmcnt({younger})")
        self.vm_done = idx + 1


def epilogue_group(prev, q, bias=True):
    """store group q = 4 i + r of the previous tile: registers r of the four column tiles of M-tile i ->
    4 channels x 4 voxel rows per lane quad: 19 instructions (15 without a bias)"""
    i, r = q >> 2, q & 3
    out = [f"v_accvgpr_read_b32 {TT[j]}, a{ACC[prev] + 4 * (4 * i + j) + r}" for j in range(4)]
    if bias:
        if OPT & 1:
            out += ["v_pk_add_f32 v[236:237], v[236:237], v[232:233]", "v_pk_add_f32 v[238:239], v[238:239], v[234:235]"]
        else:
            out += [f"v_add_f32 {TT[j]}, {TT[j]}, {SH[j]}" for j in range(4)]
    out += [f"v_cvt_pk_bf16_f32 v240, {TT[0]}, {TT[1]}", f"v_cvt_pk_bf16_f32 v241, {TT[2]}, {TT[3]}"]
    if not (ABL & 8):
        voff = VOFA if (i >> 1) == 0 else VOFB
        out.append(f"global_store_dwordx2 {voff}, {PK}, %[pbase] offset:{(4 * (i & 1) + r) * BN * 2} {STORE_BITS}".rstrip())
    if OPT & 1:
        out += ["v_pk_add_f32 v[224:225], v[224:225], v[236:237]", "v_pk_add_f32 v[226:227], v[226:227], v[238:239]",
                "v_pk_fma_f32 v[228:229], v[236:237], v[236:237], v[228:229]", "v_pk_fma_f32 v[230:231], v[238:239], v[238:239], v[230:231]"]
    else:
        for j in range(4):
            out += [f"v_add_f32 {S1[j]}, {S1[j]}, {TT[j]}", f"v_fmac_f32 {S2[j]}, {TT[j]}, {TT[j]}"]
    return out


def epi_load_state():
    return [f"v_accvgpr_read_b32 {S1[j]}, a{STATS + j}" for j in range(4)] + [f"v_accvgpr_read_b32 {S2[j]}, a{STATS + 4 + j}" for j in range(4)] + \
           [f"v_accvgpr_read_b32 {SH[j]}, a{BIAS + j}" for j in range(4)] + \
           [f"v_mov_b32 {VOFA}, %[voff0]", f"v_add_u32 {VOFB}, %[pitch4], %[voff0]"]


def epi_save_state():
    return [f"v_accvgpr_write_b32 a{STATS + j}, {S1[j]}" for j in range(4)] + [f"v_accvgpr_write_b32 a{STATS + 4 + j}, {S2[j]}" for j in range(4)]


def prefetch_plane(pp):
    """plane pp of the next tile (depth = first plane + pp) -> a[192 + 8 pp ...]: two 16-byte chunks per lane; a plane
    or a lane's (hh, hw) outside the volume gets buffer offset -16 (out of range) and reads zeros"""
    out = [f"s_mul_i32 s90, %[planeb], {pp}", "s_add_u32 s90, s90, %[off0]",
           f"s_bitcmp1_b32 %[pvalid], {pp}", "s_cselect_b32 s91, %[mhw], 0"]
    for j in range(2):
        out += [f"v_accvgpr_read_b32 {PF0}, a{SELP + j}",
                f"v_and_b32 {PF1}, s91, {PF0}",
                f"v_cmp_eq_u32 vcc, {PF1}, {PF0}",
                f"v_accvgpr_read_b32 {PF1}, a{GOFFP + j}",
                f"v_add_u32 {PF1}, s90, {PF1}",
                f"v_cndmask_b32 {PF1}, -16, {PF1}, vcc",
                f"buffer_load_dwordx4 {halo(pp, j)}, {PF1}, %[rsrc], 0 offen"]
    return out


def plane_write(pp, sb):
    """prefetched plane pp -> the LDS slot whose byte base is the scalar operand sb"""
    return [f"v_add_u32 {PW0}, %[{sb}], %[ldsp0]", f"ds_write_b128 {PW0}, {halo(pp, 0)}",
            f"v_add_u32 {PW1}, %[{sb}], %[ldsp1]", f"ds_write_b128 {PW1}, {halo(pp, 1)}"]


def is_plane_load(pp):
    tags = [halo(pp, 0), halo(pp, 1)]
    return lambda s: s.startswith("buffer_load") and any(t in s for t in tags)


def kloop(cur, prev=None, march=True, bias=True, first=False):
    """27 taps of one tile into set `cur`.  march: the next tile is the next one down the column - its four new
    planes are prefetched in taps 0-3, written into the ring after each kd phase, and the statement ends with the
    fragment reads of ITS tap 0.  Otherwise: six planes of the next column (or nothing: all-invalid masks) are
    prefetched in taps 0-5 and WRES_BOUNDARY_* follows."""
    st = Stream()
    if prev is not None:
        for ins in epi_load_state():
            st.emit(ins)
    # the fragments of tap 0 were requested by the statement before (their latency hides behind the scalar set-up
    # between the statements); the scoreboard starts with those 8 reads outstanding
    for w in READ_ORDER:
        st.ds.append((0, w))
    side = []                                        # instructions waiting for a gap
    ngroups = 0
    nplanes = 4 if march else 6
    e0, e1 = nplanes, 25                             # the previous tile's 16 store groups: taps e0 .. e1
    if first:
        # the first tile of a workgroup: taps 9-26 of the weights (18 LDS-DMA pieces of this wave: tap k -> its 1 KB of
        # LDS at %[wlds] + 4096 (k - 9), source {%[whi], %[wlo]} + 64 (k - 9)) are issued HERE, in the MFMA gaps of taps 0-2,
        # instead of before the first boundary: a wave that issues 27 pieces in a row sits ~2 000-3 000 cycles on the
        # full queue (profiles/r03_wres_*) with the matrix pipe idle
        # (m0 - the DMA's LDS base - is a register the compiler manages: saved in s93, restored after the last piece)
        side += ["s_mov_b32 s93, m0"]
        for k in range(9, TAPS):
            # (an SALU write of m0 needs one wait state before the DMA that reads it)
            # (no immediate offset on the DMA: it would be added to the LDS address as well; the per-piece source address is
            # formed in v[244:245] - PF0 / PF1, free until the halo prefetch instructions that follow in the queue)
            side += [f"s_add_i32 m0, %[wlds], {4096 * (k - 9)}",
                     f"v_add_co_u32 v244, vcc, {64 * (k - 9)}, %[wlo]", "v_addc_co_u32 v245, vcc, 0, %[whi], vcc",
                     "global_load_lds_dwordx4 v[244:245], off"]
        side += ["s_mov_b32 m0, s93"]
    for t in range(TAPS):
        s = t & 1
        if not (ABL & 64) and t < nplanes:
            side += prefetch_plane(t)
        if prev is not None and not (ABL & 1) and e0 <= t <= e1:
            want = ((t - e0 + 1) * 16 + (e1 - e0)) // (e1 - e0 + 1)       # groups due by the end of tap t
            while ngroups < min(want, 16):
                side += epilogue_group(prev, ngroups, bias)
                ngroups += 1
        nxt = list(READ_ORDER) if t + 1 < TAPS else []
        for g, (i, j) in enumerate(MFMA_ORDER):
            st.need(t, {"A%d" % i, "B%d" % j})
            c = "0" if t == 0 else acc(cur, i, j)
            if not (ABL & 4):
                st.emit(f"v_mfma_f32_16x16x32_bf16 {acc(cur, i, j)}, {frag(s, 'A%d' % i)}, {frag(s, 'B%d' % j)}, {c}")
            budget = 2
            # the reads of tap t + 1 go to the OTHER fragment set, whose last readers (tap t - 1's MFMAs) have issued
            if nxt and g < 8:
                st.read(t + 1, nxt.pop(0))
                budget -= 1 if (OPT & 2) else 2
            if first and t == 7 and g == 8:
                # the first tile of a workgroup started with only taps 0-8 of the weights in LDS: wait for this wave's 18
                # pieces issued above (the next tile's halo loads issued after them may stay in flight); the barrier
                # publishes every wave's rows before tap 9's B reads (gaps 0-7 of tap 8) are issued
                assert not any("m0" in x for x in side), "weight DMA not issued by tap 7"
                st.need_vm(lambda x: x.startswith("global_load_lds"))
                st.emit("s_barrier")
                budget = 0
            if march and t in (8, 17, 26) and g == 8:
                # end of a kd phase.  Every fragment of this tap has landed (so no wave still reads the dying plane
                # once all have passed the barrier); the new plane's loads have landed; then overwrite.
                mine = [i_ for i_, e in enumerate(st.ds) if e != "W" and e[0] == t]
                if mine:                             # (none in the no-ds_read ablation build)
                    st.wait_ds(max(mine))
                planes = {8: [0], 17: [1], 26: [2, 3]}[t]
                if not (ABL & 64) and not (ABL & 256):
                    st.need_vm(is_plane_load(planes[-1]))
                if not (ABL & 128):
                    st.emit("s_barrier")
                w = []
                for pp in planes:
                    w += plane_write(pp, f"sb{pp}")
                side = w + side                      # ahead of everything else that waits for a gap
                budget = 0
            while side and budget > 0:
                st.emit(side.pop(0))
                budget -= 1
    assert ngroups == 16 or prev is None or (ABL & 1)
    for ins in side:                                 # (the last plane writes of a marching tile may spill over)
        st.emit(ins)
    if march:
        # tap 0 of the next tile: its kd = 0 planes are this tile's planes d0 + 4 .. d0 + 7 (the last two written above,
        # two barriers ago); only this wave's own outstanding reads matter for the counter
        for w in READ_ORDER:
            st.lines.append(a_read(0, int(w[1]), "%[abn]") if w[0] == "A" else b_read(0, int(w[1])))
    else:
        if st.ds_done != len(st.ds):
            assert ABL, "a fragment was read but never waited for"
            st.emit("s_waitcnt lgkmcnt(0)")
    if prev is not None:
        for ins in epi_save_state():
            st.emit(ins)
    return st.lines


def flush_group(prev, q, base):
    """one store group of the exposed last-tile epilogue on fixed registers v[base : base + 6] and packed fp32 math
    (no MFMA beside it: v_pk_* is a win here, unlike in the K loop); sums accumulate in v[224:231]"""
    i, r = q >> 2, q & 3
    t = [f"v{base + k}" for k in range(4)]
    p = f"v[{base + 4}:{base + 5}]"
    out = [f"v_accvgpr_read_b32 {t[j]}, a{ACC[prev] + 4 * (4 * i + j) + r}" for j in range(4)]
    out += [f"v_pk_add_f32 v[{base}:{base + 1}], v[{base}:{base + 1}], v[232:233]",
            f"v_pk_add_f32 v[{base + 2}:{base + 3}], v[{base + 2}:{base + 3}], v[234:235]",
            f"v_cvt_pk_bf16_f32 v{base + 4}, {t[0]}, {t[1]}", f"v_cvt_pk_bf16_f32 v{base + 5}, {t[2]}, {t[3]}"]
    voff = VOFA if (i >> 1) == 0 else VOFB
    out.append(f"global_store_dwordx2 {voff}, {p}, %[pbase] offset:{(4 * (i & 1) + r) * BN * 2} {STORE_BITS}".rstrip())
    out += [f"v_pk_add_f32 v[224:225], v[224:225], v[{base}:{base + 1}]", f"v_pk_add_f32 v[226:227], v[226:227], v[{base + 2}:{base + 3}]",
            f"v_pk_fma_f32 v[228:229], v[{base}:{base + 1}], v[{base}:{base + 1}], v[228:229]",
            f"v_pk_fma_f32 v[230:231], v[{base + 2}:{base + 3}], v[{base + 2}:{base + 3}], v[230:231]"]
    return out


def flush(prev):
    lines = ["s_nop 7", "s_nop 7"] + epi_load_state()     # the last MFMAs of the tile retire before their results are read
    for q in range(0, 16, 2):                         # two groups in flight: their dependent chains interleave
        a, b = flush_group(prev, q, 236), flush_group(prev, q + 1, 248)
        for x, y in zip(a, b):
            lines += [x, y]
    return lines + epi_save_state()


def prefetch_only():
    lines = []
    for pp in range(6):
        lines += prefetch_plane(pp)
    return lines


def boundary(younger, nplanes=6):
    """column change / first tile: six prefetched planes -> LDS slots sb0..sb5, between the two barriers that
    separate the previous tile's LDS reads from the overwrite; then the fragment reads of the new tile's tap 0.
    `younger`: vector-memory operations issued after the plane loads that may still be in flight (the 16 stores of
    an epilogue: vmcnt counts in issue order); 0 = wait for everything (first tile: the weight DMA too)"""
    lines = [f"s_waitcnt vmcnt({younger})", "s_barrier"]
    for pp in range(nplanes):
        lines += plane_write(pp, f"sb{pp}")
    lines += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    for w in READ_ORDER:                              # tap 0 of the tile that starts now
        lines.append(a_read(0, int(w[1]), "%[abn]") if w[0] == "A" else b_read(0, int(w[1])))
    return lines


def init_consts():
    """first thing a wave does: the halo-chunk constants the prefetch statements read, and zeroed BatchNorm sums"""
    return [f"v_accvgpr_write_b32 a{GOFFP + j}, %[g{j}]" for j in range(2)] + [f"v_accvgpr_write_b32 a{SELP + j}, %[s{j}]" for j in range(2)] + \
           [f"v_accvgpr_write_b32 a{STATS + k}, 0" for k in range(8)]


def init_bias():
    """the lane's four bias values (global loads: parked after the halo prefetch and the weight DMA have been issued)"""
    return [f"v_accvgpr_write_b32 a{BIAS + j}, %[sh{j}]" for j in range(4)]


def stats_out():
    return [f"v_accvgpr_read_b32 %{k}, a{STATS + k}" for k in range(8)]


def extract(cur, i):
    """the 16 accumulator registers of M-tile i (4 column tiles x 4 rows) -> sixteen "=v" operands"""
    b = ACC[cur] + 16 * i
    return ["s_nop 7", "s_nop 7"] + [f"v_accvgpr_read_b32 %{k}, a{b + k}" for k in range(16)]


def cstr(lines):
    return "\n".join('    "' + l + '\\n\\t"' for l in lines)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else OUT
    parts = ["// GENERATED by tools/gen_wres_asm.py - do not edit.  See that file for the register plan and schedule.",
             "#pragma once",
             "#define WRES_CLOBBERS " + ", ".join(f'"a{i}"' for i in range(NAGPR)) + ", " + ", ".join(f'"v{i}"' for i in SCRATCH_V) +
             ", " + ", ".join(f'"s{i}"' for i in SCRATCH_S) + ', "vcc", "scc"']

    def define(name, lines):
        parts.append(f"#define {name} \\\n" + cstr(lines).replace("\n", " \\\n"))
    for cur in ("X", "Y"):
        other = "Y" if cur == "X" else "X"
        for march in (True, False):
            sfx = "_MARCH" if march else "_COL"
            define(f"WRES_K_{cur}{sfx}", kloop(cur, march=march))
            define(f"WRES_K_{cur}_EPI{sfx}", kloop(cur, prev=other, march=march))
        define(f"WRES_FLUSH_{cur}", flush(cur))
        for i in range(4):
            define(f"WRES_EXTRACT_{cur}_{i}", extract(cur, i))
    define("WRES_PREFETCH", prefetch_only())
    define("WRES_BOUNDARY_ALL", boundary(0))
    # first tile of a workgroup: only the halo loads, the DMA of taps 0-8 and the bias piece have been issued; taps 9-26
    # are issued and waited for inside WRES_K_X_*_FIRST
    define("WRES_BOUNDARY_FIRST", boundary(0))
    for march in (True, False):
        define("WRES_K_X_" + ("MARCH" if march else "COL") + "_FIRST", kloop("X", march=march, first=True))
    define("WRES_BOUNDARY_EPI", boundary(16))
    define("WRES_INIT", init_consts())
    define("WRES_INIT_BIAS", init_bias())
    define("WRES_STATS_OUT", stats_out())
    with open(out, "w") as f:
        f.write("\n".join(parts) + "\n")
    k = kloop("X", prev="Y", march=True)
    nm = sum(1 for l in k if l.startswith("v_mfma"))
    print(f"wrote {out}: marching K loop with epilogue = {len(k)} instructions ({nm} MFMAs)")


if __name__ == "__main__":
    main()
