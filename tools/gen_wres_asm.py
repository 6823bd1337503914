#!/usr/bin/env python3
"""Generates multimodal_eeg_fmri_amd/csrc/conv3d_wres_asm.inc: the hand-scheduled gfx950 instruction
streams of the weight-resident 3-D convolution kernel (csrc/conv3d_wres.hip).

Why assembly: one wave per SIMD issues everything itself, so the LDS fragment reads of K-step s + 2, the
four MFMAs of step s and the previous tile's pack / store / BatchNorm-sum instructions have to sit at
fixed places in each other's shadow.  hipcc (ROCm 7.2) re-clusters the ds_reads of a fully unrolled K loop
into bursts next to their consumers (register-pressure mode) and leaves 30-40 % of the loop waiting on
LDS latency (profiles/r01_conv3d_wres_ablation.txt, r02 notes in DESIGN.md).

Register plan (accumulator file, named literally; every statement lists a0-a175 as clobbers):
  a[0:47]     three rotating fragment sets {A0, A1, B0, B1} x 4 registers (ds_read_b128 straight into AGPRs,
              MFMA takes A / B operands from AGPRs)
  a[48:111]   accumulator set X: tile (i, j) at 48 + 16 (2 i + j)
  a[112:175]  accumulator set Y
Everything else (addresses, BatchNorm sums, bias, temporaries) is a named asm operand chosen by the compiler.

K-step s (tap = s >> 1, channel half ks = s & 1) of the tile held in set CUR:
    s_waitcnt lgkmcnt(N)       N from a scoreboard: everything older than the fragments of step s + 1 has landed
    MFMA 00 ; ds_read A0(s+2) ; epilogue piece
    MFMA 01 ; ds_read A1(s+2) ; epilogue piece
    MFMA 10 ; ds_read B0(s+2) ; epilogue piece
    MFMA 11 ; ds_read B1(s+2) ; epilogue piece
The epilogue pieces of step s < 32 are register q = s of the PREVIOUS tile (set PREV): two accvgpr reads,
bias adds, v_cvt_pk_bf16_f32, one global_store_dword (channels 2 lr, 2 lr + 1 of one voxel; a half-wave
writes one 128-byte row) and the four BatchNorm-sum updates.  Stores sit in the first 32 of the 54 steps so
that they have left the vmcnt queue before the compiler's wait for the next halo at the tile boundary.

usage: python tools/gen_wres_asm.py   (writes the .inc; commit it)
"""
import os
import sys

# ablation builds (tools/abl_build.sh): 1 no epilogue instructions, 2 no ds_reads inside the K loop, 4 no MFMAs,
# 8 no global stores, 16 epilogue reads v_mov instead of v_accvgpr_read.  0 in the product.
ABL = int(os.environ.get("WRES_ABL", "0"))
# cache policy of the output stores: write-through ("sc0 sc1") - the 16.8 MB leave the XCD's L2 while the kernel
# still computes instead of as one write-back burst at the kernel boundary (measured at C2, graph-replayed:
# plain 17.9 us, sc1 / nt 16.6, sc0 sc1 15.8); the consumer is on another XCD's L2 anyway.
STORE_BITS = os.environ.get("WRES_STORE_BITS", "sc0 sc1")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "multimodal_eeg_fmri_amd", "csrc", "conv3d_wres_asm.inc")
ACC = {"X": 48, "Y": 112}
NAGPR = 176
ROWB, WP, DP, BN = 64, 12, 120, 64
STEPS = 54
EPI_STEPS = 32


def frag(k, which):
    b = 16 * k + {"A0": 0, "A1": 4, "B0": 8, "B1": 12}[which]
    return f"a[{b}:{b + 3}]"


def acc(cur, i, j):
    b = ACC[cur] + 16 * (2 * i + j)
    return f"a[{b}:{b + 15}]"


def a_read(s, i, dst):
    tap, ks = s >> 1, s & 1
    kd, kh, kw = tap // 9, (tap // 3) % 3, tap % 3
    off = (kd * DP + kh * WP + kw) * ROWB + i * 4 * WP * ROWB
    return f"ds_read_b128 {dst}, %[ab{kh * 2 + ks}] offset:{off}"


def b_read(s, j, dst):
    tap, ks = s >> 1, s & 1
    half = 1 if tap >= 13 else 0
    off = (tap - 13 * half) * BN * ROWB
    return f"ds_read_b128 {dst}, %[bb{(j * 2 + ks) * 2 + half}] offset:{off}"


class Stream:
    def __init__(self):
        self.lines = []
        self.ds = []                 # outstanding LDS reads in issue order: (step, which)
        self.done = 0                # reads [0, done) are known to have landed

    def emit(self, s):
        self.lines.append(s)

    def read(self, step, which, force=False):
        if (ABL & 2) and not force:
            return
        k = step % 3
        if which[0] == "A":
            self.emit(a_read(step, int(which[1]), frag(k, which)))
        else:
            self.emit(b_read(step, int(which[1]), frag(k, which)))
        self.ds.append((step, which))

    def need(self, step):
        """all four fragments of `step` must have landed"""
        if not any(st == step for st, _ in self.ds):
            return
        idx = max(i for i, (st, _) in enumerate(self.ds) if st == step)
        if idx < self.done:
            return
        younger = len(self.ds) - 1 - idx
        assert younger <= 15
        self.emit(f"s_waitcnt lgkmcnt({younger})")
        self.done = idx + 1


def epilogue_pieces(prev, q):
    """register q = 16 i + r of the previous tile: 4 instruction groups, one per MFMA gap"""
    i, r = q >> 4, q & 15
    par = bin(r >> 2).count("1") & 1
    a0 = ACC[prev] + 16 * (2 * i + 0) + r
    a1 = ACC[prev] + 16 * (2 * i + 1) + r
    g = [[], [], [], []]
    if (r & 3) == 0 and q != 0:                       # new h row of the output tile: advance both lane offsets
        g[0] += ["v_add_u32 %[voffe], %[pitch], %[voffe]", "v_add_u32 %[voffo], %[pitch], %[voffo]"]
    t2 = "%[t2]"
    if ABL & 16:
        g[0] += ["v_mov_b32 %[t0], 1.0", "v_mov_b32 %[t1], 1.0"]
    else:
        g[0] += [f"v_accvgpr_read_b32 %[t0], a{a0}", f"v_accvgpr_read_b32 %[t1], a{a1}"]
    g[1] += ["v_add_f32 %[t0], %[t0], %[sh0]", "v_add_f32 %[t1], %[t1], %[sh1]", f"v_cvt_pk_bf16_f32 {t2}, %[t0], %[t1]"]
    if not (ABL & 8):
        g[2] += [f"global_store_dword {'%[voffo]' if par else '%[voffe]'}, {t2}, %[pbase] offset:{(r & 3) * BN * 2} {STORE_BITS}".rstrip()]
    g[2] += ["v_add_f32 %[s10], %[s10], %[t0]", "v_fmac_f32 %[s20], %[t0], %[t0]"]
    g[3] += ["v_add_f32 %[s11], %[s11], %[t1]", "v_fmac_f32 %[s21], %[t1], %[t1]"]
    return g


EPI_INIT = ["v_mov_b32 %[s10], 0", "v_mov_b32 %[s11], 0", "v_mov_b32 %[s20], 0", "v_mov_b32 %[s21], 0",
            "v_mov_b32 %[voffe], %[voff0]", "v_xor_b32 %[voffo], 512, %[voff0]"]   # odd-parity rows: the other half-wave's voxel


def kloop(cur, s0, s1, prev=None):
    st = Stream()
    st.emit("s_waitcnt lgkmcnt(0)")                   # nothing of the compiler's in flight: the counts below are exact
    if prev is not None:
        for ins in EPI_INIT:
            st.emit(ins)
    for s in (s0, s0 + 1):
        if s < s1:
            for w in ("A0", "A1", "B0", "B1"):
                st.read(s, w, force=True)
    for s in range(s0, s1):
        k = s % 3
        n = s + 2 if s + 2 < s1 else None
        ep = epilogue_pieces(prev, s) if (prev is not None and s < EPI_STEPS and not (ABL & 1)) else [[], [], [], []]
        st.need(s)
        for g, (i, j, w) in enumerate(((0, 0, "A0"), (0, 1, "A1"), (1, 0, "B0"), (1, 1, "B1"))):
            c = "0" if s == 0 else acc(cur, i, j)
            if not (ABL & 4):
                st.emit(f"v_mfma_f32_32x32x16_bf16 {acc(cur, i, j)}, {frag(k, 'A%d' % i)}, {frag(k, 'B%d' % j)}, {c}")
            if n is not None:
                st.read(n, w)
            for ins in ep[g]:
                st.emit(ins)
    if st.done != len(st.ds):
        assert ABL, "a fragment was read but never waited for"
        st.emit("s_waitcnt lgkmcnt(0)")
    return st.lines


def flush(prev):
    lines = ["s_nop 7", "s_nop 7", "s_nop 7"] + EPI_INIT   # the last MFMAs of the tile retire before their results are read
    for q in range(32):
        for g in epilogue_pieces(prev, q):
            lines += g
    return lines


def extract(cur, tile):
    """16 accumulator registers of tile (i, j) -> sixteen "=v" operands (ragged tiles are stored by C++ code)"""
    b = ACC[cur] + 16 * tile
    return ["s_nop 7", "s_nop 7", "s_nop 7"] + [f"v_accvgpr_read_b32 %{r}, a{b + r}" for r in range(16)]


def cstr(lines):
    return "\n".join('    "' + l + '\\n\\t"' for l in lines)


def main():
    parts = ["// GENERATED by tools/gen_wres_asm.py - do not edit.  See that file for the register plan and schedule.",
             "#pragma once",
             "#define WRES_AGPR_CLOBBERS " + ", ".join(f'"a{i}"' for i in range(NAGPR))]
    for cur in ("X", "Y"):
        other = "Y" if cur == "X" else "X"
        parts.append(f"#define WRES_K_{cur}_ALL \\\n" + cstr(kloop(cur, 0, STEPS)).replace("\n", " \\\n"))
        parts.append(f"#define WRES_K_{cur}_ALL_EPI \\\n" + cstr(kloop(cur, 0, STEPS, prev=other)).replace("\n", " \\\n"))
        parts.append(f"#define WRES_FLUSH_{cur} \\\n" + cstr(flush(cur)).replace("\n", " \\\n"))
        for t in range(4):
            parts.append(f"#define WRES_EXTRACT_{cur}_{t} \\\n" + cstr(extract(cur, t)).replace("\n", " \\\n"))
    out = sys.argv[1] if len(sys.argv) > 1 else OUT
    with open(out, "w") as f:
        f.write("\n".join(parts) + "\n")
    n = sum(len(kloop("X", 0, STEPS, prev="Y")) for _ in range(1))
    print(f"wrote {out}: K loop with epilogue = {n} instructions")


if __name__ == "__main__":
    main()
