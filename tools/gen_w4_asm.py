"""Writes quantized-training_amd/csrc/gnk_gemm_w4_asm.inc: the hand-scheduled gfx950 main loop of the one-wave-per-SIMD GEMM
(gnk_gemm_w4.h) as inline-asm string macros, one per operand-layout pair and operand class.

Why asm: with 256 accumulator registers (the whole AGPR file) and two fragment sets live, hipcc's register allocator rotates
accumulators through v_accvgpr_mov / read / write copies (about 100 per four stages), waits lgkmcnt(0) right behind the first
fragment read of a stage and rebuilds a 64-bit address per LDS-DMA; the stream below has none of that.

Register plan (fixed; listed as clobbers so the compiler keeps its own values elsewhere):
    a[0:255]    accumulators; 16 x 16 tile (h, i, j) = rows 16 i, columns 64 h + 16 j of the 128 x 128 wave tile at a[4 T .. 4 T + 3],
                T = 32 h + 4 i + j  (fp8, 32 x 32 tiles (h, i, j): a[16 T ..], T = 8 h + 2 i + j)
    v[112:119]  LDS addresses of the A fragments: M-major operand one per fragment (the image's XOR swizzle); K-major operand one per
                (64-k half of the 128-k image, 16-byte chunk pair): the caller's address XOR 64 ks (XOR 32 for fp8's second read);
                v[120:127] same for B
    v[128:255]  two fragment sets: set c at v[128 + 64 c ..]: A fragments first (32 registers), then B (32)
    s[72:73] / s[74:75]  A / B base pointers (advanced by s76 / s77 per issue while s80 / s90 > 0), s78 LDS address of this wave's
                first piece, s79 iterations left (4 stages each), s81 scratch

LDS: 64 KiB per operand.  An M/N-major operand is a ring of four 64-k stages ([64 k][256 B], gnk_mn64_*): 4 pieces per wave and stage,
stage t + 4 issued during stage t.  A K-major operand is a ring of two 128-k "super-stages" ([256 rows][128 B], gnk_kmaj_*: the image of the
8-phase kernel), so that every piece is 8 rows x 128 B = whole 128-byte lines (64-byte row segments cost twice the L2 requests: the
64-k K-major ring ran A@B.T 9 % slower than A.T@B); its 8 pieces per wave go out during the odd stage in which its buffer falls free
(super-stage S + 2 during stage 2 S + 1) and are waited for two stages later.  The vmcnt of every stage is derived from the issue order below.
Run: python tools/gen_w4_asm.py   (the output is committed; __graft_entry__.build() does not run this).
"""
import os
from pathlib import Path

# timing ablations (results WRONG; tools/w4_ablate.sh): 1 = no LDS-DMA in the loop, 2 = no fragment reads, 4 = no barrier, 8 = no MFMAs;
# 16 = clock stamps around the loop (results right; build with -DGNK_W4_STAMP)
ABL = int(os.environ.get("GNK_W4_ABL", "0"))
OUT = Path(__file__).resolve().parent.parent / "quantized-training_amd" / "csrc" / "gnk_gemm_w4_asm.inc"

STAGE_BYTES = 16384      # one operand, one 64-k stage
SS_BYTES = 32768         # one K-major operand, one 128-k super-stage
B_RING = 65536           # B's buffers live behind A's (folded into the B read addresses by the caller)
SET0 = 128               # first VGPR of fragment set 0
OPERANDS = {"a": dict(ptr="s[72:73]", lo="s72", hi="s73", step="s76", cnt="s80", ring=0, abase=112, src="%[va]", off="oa"),
            "b": dict(ptr="s[74:75]", lo="s74", hi="s75", step="s77", cnt="s90", ring=B_RING, abase=120, src="%[vb]", off="ob")}


def frag_reg(cset, operand, f, width):
    """first VGPR of fragment f (width registers) of operand 'a' / 'b' in set cset"""
    return SET0 + 64 * cset + (0 if operand == "a" else 32) + width * f


def rng(r, n):
    return f"[{r}:{r + n - 1}]"


class Gen:
    def __init__(self, op, ak, bk):
        self.op = op
        self.kmaj = {"a": ak, "b": bk}
        self.lines = []
        # int8: 16-row fragments of 4 registers (one 64-k MFMA each); fp8: 32-row fragments of 8 registers
        self.fw = 4 if op == "i8" else 8
        self.nf = 8 if op == "i8" else 4
        self.vm, self.vm_prologue = self.waits()

    def e(self, s):
        self.lines.append(s)

    # ---- LDS fragment reads of stage t (mod 4) into set cset, as a list of instruction strings --------------------------
    def frag_reads(self, cset, t):
        out = []
        for operand in ("b", "a"):
            o = OPERANDS[operand]
            abase = o["abase"]
            for f in range(self.nf):
                r = frag_reg(cset, operand, f, self.fw)
                if self.kmaj[operand]:
                    off, ks = ((t >> 1) & 1) * SS_BYTES, t & 1
                    if self.op == "i8":   # 16 rows x 128 B per fragment: + 2 f KiB; address register ks
                        out.append(f"ds_read_b128 v{rng(r, 4)}, v{abase + ks} offset:{off + f * 2048}")
                    else:                 # 32 rows x 128 B per fragment: + 4 f KiB; chunks (h, 2 + h) of the 64-k half: registers 2 ks, 2 ks + 1
                        out.append(f"ds_read_b128 v{rng(r, 4)}, v{abase + 2 * ks} offset:{off + f * 4096}")
                        out.append(f"ds_read_b128 v{rng(r + 4, 4)}, v{abase + 2 * ks + 1} offset:{off + f * 4096}")
                else:
                    off = (t & 3) * STAGE_BYTES
                    if self.op == "i8":   # two transposing reads: k rows +0..7 and +8..15 (2 KiB apart)
                        out.append(f"ds_read_b64_tr_b8 v{rng(r, 2)}, v{abase + f} offset:{off}")
                        out.append(f"ds_read_b64_tr_b8 v{rng(r + 2, 2)}, v{abase + f} offset:{off + 2048}")
                    else:                 # fragment f covers 16-column chunks 2 f, 2 f + 1 (lane bit 4 picks); k rows +0, +8, +32, +40
                        for q, ko in enumerate((0, 8, 32, 40)):
                            out.append(f"ds_read_b64_tr_b8 v{rng(r + 2 * q, 2)}, v{abase + f} offset:{off + ko * 256}")
        return out

    # ---- LDS-DMA issue plan ---------------------------------------------------------------------------------------------
    # A piece is (operand, LDS offset behind this wave's first piece, offset operand index, deadline): deadline = the stage at whose
    # start it must have landed (the fragments of stage s are read during stage s - 1).
    def pieces_mn(self, operand, stage):
        o = OPERANDS[operand]
        return [(operand, o["ring"] + (stage & 3) * STAGE_BYTES + i * 4096, i, stage - 1) for i in range(4)]

    def pieces_km(self, operand, ss):
        o = OPERANDS[operand]
        return [(operand, o["ring"] + (ss & 1) * SS_BYTES + i * 4096, i, 2 * ss - 1) for i in range(8)]

    def issue_group(self, mn_stage=None, km_ss=None):
        """One group of issues, earliest deadline first; each operand's pointer advances behind its pieces."""
        out = []
        for operand in ("a", "b"):
            if self.kmaj[operand] and km_ss is not None:
                out += self.pieces_km(operand, km_ss)
        for operand in ("a", "b"):
            if not self.kmaj[operand] and mn_stage is not None:
                out += self.pieces_mn(operand, mn_stage)
        return out

    def prologue_groups(self):
        return [self.issue_group(mn_stage=0, km_ss=0), self.issue_group(mn_stage=1), self.issue_group(mn_stage=2, km_ss=1), self.issue_group(mn_stage=3)]

    def stage_group(self, t):
        """issues of (global) stage t: M/N-major stage t + 4; K-major super-stage (t + 3) / 2 during odd stages"""
        return self.issue_group(mn_stage=t + 4, km_ss=(t + 3) // 2 if t & 1 else None)

    def waits(self):
        """vmcnt per stage position (0..3) and for the prologue: how many of the youngest pieces may still be in flight"""
        seq = [pc for g in self.prologue_groups() for pc in g]

        def allowed(w):
            last = max((i for i, pc in enumerate(seq) if pc[3] <= w), default=-1)
            return len(seq) - 1 - last

        pro = allowed(-1)
        vm = [63] * 4
        for t in range(12):
            vm[t & 3] = min(vm[t & 3], allowed(t))
            seq += self.stage_group(t)
        return vm, pro

    def emit_piece(self, pc, out=None):
        operand, lds, i, _ = pc
        o = OPERANDS[operand]
        ins = [f"s_add_u32 m0, s78, {lds}", "s_nop 0", f"global_load_lds_dwordx4 %[{o['off']}{i}], {o['ptr']}"]
        if out is None:
            for x in ins:
                self.e(x)
        return ins

    def advance(self, operand):
        o = OPERANDS[operand]
        self.e(f"s_cmp_gt_i32 {o['cnt']}, 0")
        self.e(f"s_cselect_b32 s81, {o['step']}, 0")
        self.e(f"s_add_u32 {o['lo']}, {o['lo']}, s81")
        self.e(f"s_addc_u32 {o['hi']}, {o['hi']}, 0")
        self.e(f"s_sub_i32 {o['cnt']}, {o['cnt']}, 1")

    def mfmas(self, cset):
        out = []
        if self.op == "i8":
            for s in range(16):
                i, h = s >> 1, s & 1
                for j in range(4):
                    t = 32 * h + 4 * i + j
                    out.append(f"v_mfma_i32_16x16x64_i8 a{rng(4 * t, 4)}, v{rng(frag_reg(cset, 'b', 4 * h + j, 4), 4)}, "
                               f"v{rng(frag_reg(cset, 'a', i, 4), 4)}, a{rng(4 * t, 4)}")
        else:
            for s in range(16):
                i, jj = s >> 2, s & 3
                h, j = jj >> 1, jj & 1
                t = 8 * h + 2 * i + j
                out.append(f"v_mfma_f32_32x32x64_f8f6f4 a{rng(16 * t, 16)}, v{rng(frag_reg(cset, 'b', jj, 8), 8)}, "
                           f"v{rng(frag_reg(cset, 'a', i, 8), 8)}, a{rng(16 * t, 16)} cbsz:%c[fb] blgp:%c[fa]")
        return out

    def stage(self, t, cset):
        if self.vm[t] < 63:
            self.e(f"s_waitcnt vmcnt({self.vm[t]})")
        self.e("s_waitcnt lgkmcnt(0)")
        if not ABL & 4:
            self.e("s_barrier")
        mf = self.mfmas(cset)
        reads = [] if ABL & 2 else self.frag_reads(cset ^ 1, (t + 1) & 3)
        group = self.stage_group(t)
        dmas = [] if ABL & 1 else group
        # auxiliary instructions behind the MFMAs: all fragment reads in the first ~half of the stage (they must have landed by the next
        # stage's lgkmcnt(0)), the DMA pieces spread over the stage behind them
        n = len(mf)
        aux = [[] for _ in range(n)]
        per = max(1, (len(reads) + n // 2 - 1) // (n // 2))        # reads per MFMA over the first half
        k = 0
        for m in range(n):
            for _ in range(per):
                if k < len(reads):
                    aux[m].append(reads[k])
                    k += 1
        assert k == len(reads)
        if n > 16:
            first, last = 2, n - 3
        else:
            first, last = 0, n - 1
        cnt = len(dmas)
        for d, pc in enumerate(dmas):
            m = first if cnt == 1 else first + (d * (last - first)) // (cnt - 1)
            aux[m] += self.emit_piece(pc, out=aux)
        for m in range(n):
            if not ABL & 8:
                self.e(mf[m])
            for a in aux[m]:
                self.e(a)
        for operand in ("a", "b"):
            if any(pc[0] == operand for pc in group):
                self.advance(operand)

    def build(self):
        e = self.e
        e("s_mov_b32 s72, %[alo]"); e("s_mov_b32 s73, %[ahi]"); e("s_mov_b32 s74, %[blo]"); e("s_mov_b32 s75, %[bhi]")
        e("s_mov_b32 s76, %[stepa]"); e("s_mov_b32 s77, %[stepb]"); e("s_mov_b32 s78, %[lds]")
        e("s_mov_b32 s79, %[nit]")
        # advances left: stages - 1 of a M/N-major operand, super-stages - 1 of a K-major one (%[rem] = stages - 1, odd)
        for operand in ("a", "b"):
            o = OPERANDS[operand]
            if self.kmaj[operand]:
                e(f"s_lshr_b32 {o['cnt']}, %[rem], 1")
            else:
                e(f"s_mov_b32 {o['cnt']}, %[rem]")
        # fragment read addresses: M/N-major: chunk index XOR per fragment; K-major: XOR 64 per 64-k half, 32 for fp8's chunk + 2
        for operand in ("a", "b"):
            o = OPERANDS[operand]
            base, src = o["abase"], o["src"]
            if self.kmaj[operand]:
                xs = (0, 64) if self.op == "i8" else (0, 32, 64, 96)
                for i, x in enumerate(xs):
                    e(f"v_xor_b32 v{base + i}, {x}, {src}")
            else:
                for f in range(self.nf):
                    e(f"v_xor_b32 v{base + f}, {(16 if self.op == 'i8' else 32) * f}, {src}")
        # prologue: four stages of every operand in flight
        # the 256 accumulator clears go out between the LDS-DMA issues (each of which holds the wave's next vector-memory issue for
        # ~60-100 cycles while the VALU is free), not behind them
        groups = self.prologue_groups()
        npieces = sum(len(g) for g in groups)
        cleared = 0
        k = 0
        for g in groups:
            for pc in g:
                self.emit_piece(pc)
                k += 1
                upto = 256 * k // npieces
                for r in range(cleared, upto):
                    e(f"v_accvgpr_write_b32 a{r}, 0")
                cleared = upto
            for operand in ("a", "b"):
                if any(pc[0] == operand for pc in g):
                    self.advance(operand)
        assert cleared == 256
        e(f"s_waitcnt vmcnt({self.vm_prologue})")
        e("s_barrier")
        for r in self.frag_reads(0, 0):
            e(r)
        if ABL & 16:    # diagnostic build: shader-clock and 100 MHz real-time stamps around the loop -> s[82:89] (gnk_gemm_w4.h, GNK_W4_STAMP)
            e("s_memtime s[82:83]"); e("s_memrealtime s[84:85]"); e("s_waitcnt lgkmcnt(0)")
        e(".Lgnk_w4_loop_%=:")
        for b in range(4):
            self.stage(b, b & 1)
        e("s_sub_i32 s79, s79, 1")
        e("s_cmp_lg_u32 s79, 0")
        e("s_cbranch_scc1 .Lgnk_w4_loop_%=")
        if ABL & 16:
            e("s_memtime s[86:87]"); e("s_memrealtime s[88:89]"); e("s_waitcnt lgkmcnt(0)")
        e("s_waitcnt vmcnt(0)")
        e("s_waitcnt lgkmcnt(0)")
        e("s_nop 7"); e("s_nop 7"); e("s_nop 7")     # MFMA results -> v_accvgpr_read of the hand-off statements
        e("s_barrier")
        return "".join(f'    "{l}\\n\\t"\n' for l in self.lines)


def main():
    parts = ["// GENERATED by tools/gen_w4_asm.py -- do not edit; see that file for the register plan.\n#pragma once\n"]
    for op in ("i8", "f8"):
        for ak in (True, False):
            for bk in (True, False):
                name = f"GNK_W4_ASM_{op.upper()}_{'K' if ak else 'M'}{'K' if bk else 'M'}"
                g = Gen(op, ak, bk)
                body = g.build()
                print(name, "vmcnt per stage", g.vm, "prologue", g.vm_prologue)
                parts.append(f"#define {name} \\\n" + body.replace("\n", " \\\n").rstrip(" \\\n") + "\n")
    clob = ['"memory"', '"scc"', '"vcc"'] + [f'"a{r}"' for r in range(256)] + [f'"v{r}"' for r in range(112, 256)] + [f'"s{r}"' for r in range(72, 92)]
    parts.append("#define GNK_W4_CLOBBERS " + ", ".join(clob) + "\n")
    for n in range(16):
        body = "".join(f'"v_accvgpr_read_b32 %{k}, a{16 * n + k}\\n\\t" ' for k in range(16))
        parts.append(f"#define GNK_W4_RD_{n} {body}\n")
    OUT.write_text("\n".join(parts))
    print("wrote", OUT, sum(len(p) for p in parts), "bytes")


if __name__ == "__main__":
    main()
