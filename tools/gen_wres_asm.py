#!/usr/bin/env python3
"""Generates multimodal_eeg_fmri_amd/csrc/conv3d_wres_asm.inc: the hand-scheduled gfx950 instruction
streams of the weight-resident 3-D convolution kernel (csrc/conv3d_wres.hip).

Why assembly: one wave per SIMD issues everything itself, so the LDS fragment reads of the next tap, the
sixteen MFMAs of this tap, the previous tile's pack / store / BatchNorm-sum instructions and the next tile's
halo prefetch have to sit at fixed places in each other's shadow.  hipcc (ROCm 7.2) re-clusters the ds_reads of
an unrolled K loop into bursts next to their consumers and leaves 30-40 % of the loop waiting on LDS latency,
and everything it schedules at a tile boundary (address arithmetic, loads, LDS stores) runs with the matrix
pipe idle (round-2 stamps: 2 250 of 9 500 cycles per tile).

MFMA shape: v_mfma_f32_16x16x32_bf16 (K = 32 = all input channels of one tap).  Same FLOP per pipe cycle as
32x32x16, but the chip holds a ~14 % higher clock on it under this kernel's load (1.79 vs 1.57 GHz measured
with the in-kernel stamps, tools/kbench.py stamp) - cdna_hip_programming.md rule 28.

Register plan (accumulator file, named literally; every statement lists a0-a251 as clobbers):
  a[0:63]     two fragment sets (tap parity): A_i (i = 0..3, the wave's four 4x4-voxel patches) at 32 s + 4 i,
              B_j (j = 0..3, sixteen output channels each) at 32 s + 16 + 4 j; ds_read_b128 straight into AGPRs
  a[64:127]   accumulator set X: MFMA tile (i, j) at 64 + 4 (4 i + j)
  a[128:191]  accumulator set Y
  a[192:231]  the next tile's halo, ten 16-byte chunks per lane (buffer_load straight into AGPRs, ds_write from them)
  a[232:241]  per-lane constants: byte offset of halo chunk q relative to the tile's first voxel
  a[242:251]  per-lane constants: one-hot (hd, hh, hw) selector of halo chunk q
Scratch VGPRs v[240:255] are named literally too (clobbers; dead outside a statement).  Everything else is a
named operand.

One tap (K-step) of the tile held in set CUR, fragments in set t & 1:
    16 MFMAs in the order their fragments were requested; in the gaps: the 8 ds_reads of tap t + 1 (gaps 0-7),
    the next tile's halo prefetch (taps 0-4: 7 instructions per chunk; out-of-volume chunks get an out-of-range
    buffer offset and come back as zeros) and the previous tile's epilogue (16 store groups of 19 instructions
    over taps 5-25 = 4 accvgpr reads, 4 bias adds, 2 v_cvt_pk_bf16_f32, one global_store_dwordx2 of 4 channels x
    4 voxel rows = four 128-byte lines per wave instruction, 8 BatchNorm-sum updates).  At most 2 instructions
    ride in a gap (1 beside a ds_read): a 16x16x32 MFMA leaves 16 - 8 issue cycles.  The loads are issued BEFORE
    the stores, so the boundary waits with a counted vmcnt(16): the halo has landed, the stores may still fly.
"""
import os
import sys

# ablation builds (tools/abl_build.sh): 1 no epilogue instructions, 2 no ds_reads inside the K loop, 4 no MFMAs,
# 8 no global stores, 64 no halo prefetch.  0 in the product.
ABL = int(os.environ.get("WRES_ABL", "0"))
# cache policy of the output stores: write-through ("sc0 sc1") - the 16.8 MB leave the XCD's L2 while the kernel
# still computes instead of as one write-back burst at the kernel boundary (measured at C2, graph-replayed:
# plain 17.9 us, sc1 / nt 16.6, sc0 sc1 15.8); the consumer is on another XCD's L2 anyway.
STORE_BITS = os.environ.get("WRES_STORE_BITS", "sc0 sc1")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "multimodal_eeg_fmri_amd", "csrc", "conv3d_wres_asm.inc")
ACC = {"X": 64, "Y": 128}
HALO = 192
GOFF = 232
SEL = 242
NAGPR = 252
ROWB, WP, DP, BN = 64, 12, 120, 64
TAPS = 27
PF_TAPS = 5                   # halo prefetch: two chunks per tap in taps 0-4 (lands long before the tile boundary)
EPI_TAP0, EPI_TAP1 = 5, 25    # the previous tile's 16 store groups, spread over taps 5-25
NSTORES = 16                  # global stores an epilogue issues AFTER the prefetch loads (counted vmcnt at the boundary)
T = ["v%d" % i for i in range(240, 256)]           # scratch VGPRs (clobbered, dead outside a statement)


def frag(s, which):
    b = 32 * s + (0 if which[0] == "A" else 16) + 4 * int(which[1])
    return f"a[{b}:{b + 3}]"


def acc(cur, i, j):
    b = ACC[cur] + 4 * (4 * i + j)
    return f"a[{b}:{b + 3}]"


def a_read(t, i):
    kd, kh, kw = t // 9, (t // 3) % 3, t % 3
    off = (kd * DP + kh * WP + kw) * ROWB + (i >> 1) * 4 * WP * ROWB + (i & 1) * 4 * ROWB
    return f"ds_read_b128 {frag(t & 1, 'A%d' % i)}, %[ab{kh & 1}] offset:{off}"


def b_read(t, j):
    half = 1 if t >= 13 else 0
    off = (t - 13 * half) * BN * ROWB + j * 16 * ROWB
    return f"ds_read_b128 {frag(t & 1, 'B%d' % j)}, %[bb{half}] offset:{off}"


READ_ORDER = ("A0", "B0", "A1", "B1", "A2", "B2", "A3", "B3")
MFMA_ORDER = ((0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (2, 1), (0, 2), (1, 2), (2, 2), (3, 0), (3, 1), (3, 2), (0, 3), (1, 3),
              (2, 3), (3, 3))


class Stream:
    def __init__(self):
        self.lines = []
        self.ds = []                 # outstanding LDS reads in issue order: (tap, which)
        self.done = 0
        self.first_wait_done = False

    def emit(self, s):
        self.lines.append(s)

    def read(self, t, which, force=False):
        if (ABL & 2) and not force:
            return
        self.emit(a_read(t, int(which[1])) if which[0] == "A" else b_read(t, int(which[1])))
        self.ds.append((t, which))

    def need(self, t, names):
        idxs = [i for i, (tt, w) in enumerate(self.ds) if tt == t and w in names]
        if not idxs or max(idxs) < self.done:
            return
        idx = max(idxs)
        younger = len(self.ds) - 1 - idx
        assert younger <= 15
        if not self.first_wait_done:                 # compiler code ran since those reads were issued (scalar loads
            younger = 0                              # share the counter and return out of order): drain everything
            idx = len(self.ds) - 1
            self.first_wait_done = True
        self.emit(f"s_waitcnt lgkmcnt({younger})")
        self.done = idx + 1


def epilogue_group(prev, q):
    """store group q = 4 i + r of the previous tile: registers r of the four column tiles of M-tile i ->
    4 channels x 4 voxel rows per lane quad: 19 instructions"""
    i, r = q >> 2, q & 3
    out = []
    t = T[0:4]
    for j in range(4):
        out.append(f"v_accvgpr_read_b32 {t[j]}, a{ACC[prev] + 4 * (4 * i + j) + r}")
    for j in range(4):
        out.append(f"v_add_f32 {t[j]}, {t[j]}, %[sh{j}]")
    out.append(f"v_cvt_pk_bf16_f32 {T[4]}, {t[0]}, {t[1]}")
    out.append(f"v_cvt_pk_bf16_f32 {T[5]}, {t[2]}, {t[3]}")
    if not (ABL & 8):
        voff = T[6] if (i >> 1) == 0 else T[7]
        out.append(f"global_store_dwordx2 {voff}, v[244:245], %[pbase] offset:{(4 * (i & 1) + r) * BN * 2} {STORE_BITS}".rstrip())
    for j in range(4):
        out.append(f"v_add_f32 %[s1{j}], %[s1{j}], {t[j]}")
        out.append(f"v_fmac_f32 %[s2{j}], {t[j]}, {t[j]}")
    return out


EPI_INIT = [f"v_mov_b32 %[s1{j}], 0" for j in range(4)] + [f"v_mov_b32 %[s2{j}], 0" for j in range(4)] + \
           [f"v_mov_b32 {T[6]}, %[voff0]", f"v_add_u32 {T[7]}, %[pitch4], %[voff0]"]      # patch rows h 0-3 / 4-7


def prefetch_chunk(q):
    """halo chunk q of the NEXT tile -> a[192 + 4 q ...]; out-of-volume chunks read zeros (offset -16 is out of range)"""
    d = HALO + 4 * q
    return [f"v_accvgpr_read_b32 {T[8]}, a{SEL + q}",
            f"v_and_b32 {T[9]}, %[vmask], {T[8]}",
            f"v_cmp_eq_u32 vcc, {T[9]}, {T[8]}",
            f"v_accvgpr_read_b32 {T[9]}, a{GOFF + q}",
            f"v_add_u32 {T[9]}, %[toff], {T[9]}",
            f"v_cndmask_b32 {T[9]}, -16, {T[9]}, vcc",
            f"buffer_load_dwordx4 a[{d}:{d + 3}], {T[9]}, %[rsrc], 0 offen"]


def kloop(cur, prev=None, prefetch=True):
    st = Stream()
    if prev is not None:
        for ins in EPI_INIT:
            st.emit(ins)
    # the fragments of tap 0 were requested at the end of the boundary statement (their latency hides behind the
    # scalar set-up between the two statements); the scoreboard starts with those 8 reads outstanding
    for w in READ_ORDER:
        st.ds.append((0, w))
    side = []                                        # instructions waiting for a gap
    ngroups = 0
    for t in range(TAPS):
        s = t & 1
        if prefetch and not (ABL & 64) and t < PF_TAPS:
            side += prefetch_chunk(2 * t) + prefetch_chunk(2 * t + 1)
        if prev is not None and not (ABL & 1) and EPI_TAP0 <= t <= EPI_TAP1:
            want = ((t - EPI_TAP0 + 1) * 16 + (EPI_TAP1 - EPI_TAP0)) // (EPI_TAP1 - EPI_TAP0 + 1)   # groups due by the end of tap t
            while ngroups < min(want, 16):
                side += epilogue_group(prev, ngroups)
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
                budget -= 1
            while side and budget > 0:
                st.emit(side.pop(0))
                budget -= 1
    assert ngroups == 16 or prev is None or (ABL & 1)
    assert not side, f"{len(side)} side instructions did not fit"
    if st.done != len(st.ds):
        assert ABL, "a fragment was read but never waited for"
        st.emit("s_waitcnt lgkmcnt(0)")
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
    voff = T[6] if (i >> 1) == 0 else T[7]
    out.append(f"global_store_dwordx2 {voff}, {p}, %[pbase] offset:{(4 * (i & 1) + r) * BN * 2} {STORE_BITS}".rstrip())
    out += [f"v_pk_add_f32 v[224:225], v[224:225], v[{base}:{base + 1}]", f"v_pk_add_f32 v[226:227], v[226:227], v[{base + 2}:{base + 3}]",
            f"v_pk_fma_f32 v[228:229], v[{base}:{base + 1}], v[{base}:{base + 1}], v[228:229]",
            f"v_pk_fma_f32 v[230:231], v[{base + 2}:{base + 3}], v[{base + 2}:{base + 3}], v[230:231]"]
    return out


FLUSH_REGS = list(range(224, 240))                    # more scratch for the flush statement only (clobbers)


def flush(prev):
    lines = ["s_nop 7", "s_nop 7"]                    # the last MFMAs of the tile retire before their results are read
    lines += [f"v_mov_b32 {T[6]}, %[voff0]", f"v_add_u32 {T[7]}, %[pitch4], %[voff0]"]
    lines += [f"v_mov_b32 v{232 + j}, %[sh{j}]" for j in range(4)] + [f"v_mov_b32 v{224 + k}, 0" for k in range(8)]
    for q in range(0, 16, 2):                         # two groups in flight: their dependent chains interleave
        a, b = flush_group(prev, q, 236), flush_group(prev, q + 1, 248)
        for x, y in zip(a, b):
            lines += [x, y]
    lines += [f"v_mov_b32 %[s1{j}], v{224 + j}" for j in range(4)] + [f"v_mov_b32 %[s2{j}], v{228 + j}" for j in range(4)]
    return lines


def prefetch_only():
    lines = []
    for q in range(10):
        lines += prefetch_chunk(q)
    return lines


def boundary(younger):
    """next halo: AGPRs -> LDS, between the two barriers that separate the tiles' LDS reads from the overwrite.
    `younger`: vector-memory operations issued after the halo loads that may still be in flight (the 16 stores of
    an epilogue: vmcnt counts in issue order); 0 = wait for everything (first tile: the weight DMA too)"""
    lines = [f"s_waitcnt vmcnt({younger})", "s_barrier"]
    for q in range(10):
        lines.append(f"ds_write_b128 %[l{q}], a[{HALO + 4 * q}:{HALO + 4 * q + 3}]")
    lines += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    for w in READ_ORDER:                              # tap 0 of the tile that starts now
        lines.append(a_read(0, int(w[1])) if w[0] == "A" else b_read(0, int(w[1])))
    return lines


def init_consts():
    return [f"v_accvgpr_write_b32 a{GOFF + q}, %[g{q}]" for q in range(10)] + \
           [f"v_accvgpr_write_b32 a{SEL + q}, %[s{q}]" for q in range(10)]


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
             "#define WRES_FLUSH_CLOBBERS " + ", ".join(f'"v{i}"' for i in FLUSH_REGS),
             "#define WRES_CLOBBERS " + ", ".join(f'"a{i}"' for i in range(NAGPR)) + ", " + ", ".join(f'"{v}"' for v in T) + ', "vcc"']

    def define(name, lines):
        parts.append(f"#define {name} \\\n" + cstr(lines).replace("\n", " \\\n"))
    for cur in ("X", "Y"):
        other = "Y" if cur == "X" else "X"
        define(f"WRES_K_{cur}", kloop(cur))
        define(f"WRES_K_{cur}_EPI", kloop(cur, prev=other))
        define(f"WRES_FLUSH_{cur}", flush(cur))
        for i in range(4):
            define(f"WRES_EXTRACT_{cur}_{i}", extract(cur, i))
    define("WRES_PREFETCH", prefetch_only())
    define("WRES_BOUNDARY_ALL", boundary(0))
    define("WRES_BOUNDARY_EPI", boundary(NSTORES))
    define("WRES_INIT", init_consts())
    with open(out, "w") as f:
        f.write("\n".join(parts) + "\n")
    k = kloop("X", prev="Y")
    nm = sum(1 for l in k if l.startswith("v_mfma"))
    print(f"wrote {out}: K loop with epilogue + prefetch = {len(k)} instructions ({nm} MFMAs)")


if __name__ == "__main__":
    main()
