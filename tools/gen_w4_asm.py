"""Writes quantized-training_amd/csrc/gnk_gemm_w4_asm.inc: the hand-scheduled gfx950 main loop of the one-wave-per-SIMD GEMM
(gnk_gemm_w4.h) as inline-asm string macros, one per operand-layout pair and operand class.

Why asm: with 256 accumulator registers (the whole AGPR file) and two fragment sets live, hipcc's register allocator rotates
accumulators through v_accvgpr_mov / read / write copies (about 100 per four stages), waits lgkmcnt(0) right behind the first
fragment read of a stage and rebuilds a 64-bit address per LDS-DMA; the stream below has none of that.

Register plan (fixed; listed as clobbers so the compiler keeps its own values elsewhere):
    a[0:255]    accumulators; 16 x 16 tile (h, i, j) = rows 16 i, columns 64 h + 16 j of the 128 x 128 wave tile at a[4 T .. 4 T + 3],
                T = 32 h + 4 i + j  (fp8, 32 x 32 tiles (h, i, j): a[16 T ..], T = 8 h + 2 i + j)
    v[112:119]  LDS addresses of the A fragments when A is M-major (one per fragment: the image's XOR swizzle), v[120:127] same for B
    v[128:255]  two fragment sets: set c at v[128 + 64 c ..]: A fragments first (32 registers), then B (32)
    s[72:73] / s[74:75]  A / B stage base pointers (advanced by s76 / s77 per stage while s80 > 0), s78 LDS address of this wave's
                first piece, s79 iterations left (4 stages each), s81 scratch
Run: python tools/gen_w4_asm.py   (the output is committed; __graft_entry__.build() does not run this).
"""
import os
from pathlib import Path

# timing ablations (results WRONG; tools/w4_ablate.sh): 1 = no LDS-DMA in the loop, 2 = no fragment reads, 4 = no barrier, 8 = no MFMAs;
# 16 = clock stamps around the loop (results right; build with -DGNK_W4_STAMP)
ABL = int(os.environ.get("GNK_W4_ABL", "0"))
OUT = Path(__file__).resolve().parent.parent / "quantized-training_amd" / "csrc" / "gnk_gemm_w4_asm.inc"

STAGE_BYTES = 16384      # one operand, one 64-k stage
B_RING = 65536           # B stages live behind the four A stages (folded into the B read addresses by the caller)
SET0 = 128               # first VGPR of fragment set 0


def frag_reg(cset, operand, f, width):
    """first VGPR of fragment f (width registers) of operand 'a' / 'b' in set cset"""
    return SET0 + 64 * cset + (0 if operand == "a" else 32) + width * f


def rng(r, n):
    return f"[{r}:{r + n - 1}]"


class Gen:
    def __init__(self, op, ak, bk):
        self.op, self.ak, self.bk = op, ak, bk
        self.lines = []
        # int8: 16-row fragments of 4 registers (one 64-k MFMA each); fp8: 32-row fragments of 8 registers
        self.fw = 4 if op == "i8" else 8
        self.nf = 8 if op == "i8" else 4

    def e(self, s):
        self.lines.append(s)

    # ---- LDS fragment reads of one stage buffer into set cset, as a list of instruction strings -----------------------
    def frag_reads(self, cset, buf):
        out = []
        for operand, kmaj, avgpr, abase in (("b", self.bk, "%[vb]", 120), ("a", self.ak, "%[va]", 112)):
            for f in range(self.nf):
                r = frag_reg(cset, operand, f, self.fw)
                off = buf * STAGE_BYTES
                if self.op == "i8":
                    if kmaj:      # 16 rows x 64 B per fragment: + f KiB
                        out.append(f"ds_read_b128 v{rng(r, 4)}, {avgpr} offset:{off + f * 1024}")
                    else:         # two transposing reads: k rows +0..7 and +8..15 (2 KiB apart)
                        out.append(f"ds_read_b64_tr_b8 v{rng(r, 2)}, v{abase + f} offset:{off}")
                        out.append(f"ds_read_b64_tr_b8 v{rng(r + 2, 2)}, v{abase + f} offset:{off + 2048}")
                else:
                    if kmaj:      # 32 rows x 64 B per fragment: + 2 f KiB; halves are chunk pairs (0, 2) -> the XOR'd address of half 1
                        out.append(f"ds_read_b128 v{rng(r, 4)}, {avgpr} offset:{off + f * 2048}")
                        out.append(f"ds_read_b128 v{rng(r + 4, 4)}, v{abase} offset:{off + f * 2048}")
                    else:         # fragment f covers 16-column chunks 2 f, 2 f + 1 (lane bit 4 picks); k rows +0, +8, +32, +40
                        for q, ko in enumerate((0, 8, 32, 40)):
                            out.append(f"ds_read_b64_tr_b8 v{rng(r + 2 * q, 2)}, v{abase + f} offset:{off + ko * 256}")
        return out

    def dma(self, buf):
        """the 8 LDS-DMA pieces of one stage into buffer buf: A pieces 0-3, then B pieces 0-3"""
        out = []
        for i in range(4):
            out.append((f"s_add_u32 m0, s78, {buf * STAGE_BYTES + i * 4096}", f"global_load_lds_dwordx4 %[o{i}], s[72:73]"))
        for i in range(4):
            out.append((f"s_add_u32 m0, s78, {B_RING + buf * STAGE_BYTES + i * 4096}", f"global_load_lds_dwordx4 %[o{4 + i}], s[74:75]"))
        return out

    def advance(self):
        self.e("s_cmp_gt_i32 s80, 0")
        self.e("s_cselect_b32 s81, s76, 0")
        self.e("s_add_u32 s72, s72, s81")
        self.e("s_addc_u32 s73, s73, 0")
        self.e("s_cmp_gt_i32 s80, 0")
        self.e("s_cselect_b32 s81, s77, 0")
        self.e("s_add_u32 s74, s74, s81")
        self.e("s_addc_u32 s75, s75, 0")
        self.e("s_sub_i32 s80, s80, 1")

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

    def stage(self, buf, cset):
        self.e("s_waitcnt vmcnt(16)")
        self.e("s_waitcnt lgkmcnt(0)")
        if not ABL & 4:
            self.e("s_barrier")
        mf = self.mfmas(cset)
        reads = [] if ABL & 2 else self.frag_reads(cset ^ 1, (buf + 1) & 3)
        dmas = [] if ABL & 1 else self.dma(buf)
        # auxiliary instructions behind the MFMAs: all fragment reads in the first ~half of the stage (they must have landed by the next
        # stage's lgkmcnt(0)), the DMA pieces spread behind them
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
        first = 0 if self.op == "f8" else 2
        stride = max(1, (n - first - (2 if self.op == "i8" else 1)) // 8)
        for d, (setm0, ld) in enumerate(dmas):
            m = min(n - 2, first + d * stride) if n > 16 else min(n - 1, d * 2)
            aux[m] += [setm0, "s_nop 0", ld]
        for m in range(n):
            if not ABL & 8:
                self.e(mf[m])
            for a in aux[m]:
                self.e(a)
        self.advance()

    def build(self):
        e = self.e
        e("s_mov_b32 s72, %[alo]"); e("s_mov_b32 s73, %[ahi]"); e("s_mov_b32 s74, %[blo]"); e("s_mov_b32 s75, %[bhi]")
        e("s_mov_b32 s76, %[stepa]"); e("s_mov_b32 s77, %[stepb]"); e("s_mov_b32 s78, %[lds]")
        e("s_mov_b32 s79, %[nit]"); e("s_mov_b32 s80, %[rem]")
        # per-fragment read addresses of M/N-major operands: chunk index XOR (the caller passes fragment 0's address)
        for kmaj, src, base in ((self.ak, "%[va]", 112), (self.bk, "%[vb]", 120)):
            if self.op == "i8" and not kmaj:
                for f in range(8):
                    e(f"v_xor_b32 v{base + f}, {16 * f}, {src}")
            if self.op == "f8":
                if kmaj:     # half 1 of a 32-row fragment: chunk c + 2 -> slot XOR 2 -> address XOR 32
                    e(f"v_xor_b32 v{base}, 32, {src}")
                else:
                    for f in range(4):
                        e(f"v_xor_b32 v{base + f}, {32 * f}, {src}")
        # prologue: stages 0..3 in flight
        for s in range(4):
            for setm0, ld in self.dma(s):
                e(setm0); e("s_nop 0"); e(ld)
            self.advance()
        for r in range(256):
            e(f"v_accvgpr_write_b32 a{r}, 0")
        e("s_waitcnt vmcnt(24)")
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
                body = Gen(op, ak, bk).build()
                parts.append(f"#define {name} \\\n" + body.replace("\n", " \\\n").rstrip(" \\\n") + "\n")
    clob = ['"memory"', '"scc"', '"vcc"'] + [f'"a{r}"' for r in range(256)] + [f'"v{r}"' for r in range(112, 256)] + [f'"s{r}"' for r in range(72, 90)]
    parts.append("#define GNK_W4_CLOBBERS " + ", ".join(clob) + "\n")
    for n in range(16):
        body = "".join(f'"v_accvgpr_read_b32 %{k}, a{16 * n + k}\\n\\t" ' for k in range(16))
        parts.append(f"#define GNK_W4_RD_{n} {body}\n")
    OUT.write_text("\n".join(parts))
    print("wrote", OUT, sum(len(p) for p in parts), "bytes")


if __name__ == "__main__":
    main()
