// gnk_lds_layout.h -- LDS image index math of the MFMA GEMM, usable from host and device so that
// tests/test_lds_layout.py can simulate stage -> read round trips (including the
// ds_read_b64_tr_b8 gather) on the CPU.
//
// A K-step tile of either operand is ROWS x 128 bytes (int8 / fp8: one byte per element).
//
//  * K-major tile  (the operand's K dim is contiguous in memory):  image [ROWS][128 B].
//    16-byte chunk c (k = 16c .. 16c+15) of row r is stored in slot  c ^ ((r >> 1) & 7).
//    MFMA fragments are read with ds_read_b128; the XOR makes every 16-lane group of a
//    16x16x64 / 32x32x32 fragment read hit 16 distinct 16-B bank slots.
//
//  * MN-major tile (the operand's M or N dim is contiguous):       image [128 k][L B], L = ROWS.
//    16-byte chunk c (n = 16c .. 16c+15) of k-row k is stored in slot c ^ swz(k);
//    fragments are gathered with ds_read_b64_tr_b8 (8 k-rows x 16 columns per 16 lanes).
//
// global_load_lds writes LDS linearly (wave-uniform base + lane * 16), so the swizzle is applied
// to the per-lane SOURCE address when staging and to the address when reading.
#pragma once

#if defined(__HIPCC__)
#define GNK_HD __host__ __device__ __forceinline__
#else
#define GNK_HD inline
#endif

constexpr int GNK_BK = 128;   // bytes (= elements) of K per staged tile

GNK_HD int gnk_kmaj_offset(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

template <int L> GNK_HD int gnk_mnmaj_swz(int k) {
    static_assert(L == 64 || L == 128 || L == 256, "MN-major tiles are 64, 128 or 256 bytes wide");
    if (L == 256) return ((k & 7) << 1) | ((k >> 4) & 1);
    // 64-byte k-rows (64-row tiles, round 3): four k-rows share a 256-byte bank window, and a half-wave of a ds_read_b64_tr_b8 touches
    // k = q, q + 4, 16 + q, 20 + q for each of them -- bits 2 and 4 of k pick four different 16-byte slots
    if (L == 64) return ((k >> 2) & 1) | (((k >> 4) & 1) << 1);
    return (((k >> 1) & 3) << 1) | ((k >> 4) & 1);
}
template <int L> GNK_HD int gnk_mnmaj_offset(int k, int c) { return k * L + ((c ^ gnk_mnmaj_swz<L>(k)) << 4); }

// ---- staging: which (row, chunk) of the tile does LDS 16-byte position p hold? -------------
// K-major:  p = r * 8 + slot
GNK_HD void gnk_kmaj_stage_src(int p, int &r, int &c) {
    r = p >> 3;
    c = (p & 7) ^ ((r >> 1) & 7);
}
// MN-major: p = k * (L/16) + slot
template <int L> GNK_HD void gnk_mnmaj_stage_src(int p, int &k, int &c) {
    k = p / (L / 16);
    c = (p % (L / 16)) ^ gnk_mnmaj_swz<L>(k);
}

// ---- fragment reads for one 16-row (or 16-column) x 64-k MFMA operand ------------------------
// lane l supplies fragment row (l & 15) and k-slot group g = l >> 4; its 16 bytes are
// k = 64*ks + 16*g + (0..15).
// K-major: one ds_read_b128 at this offset (rb = first row of the 16-row fragment in the tile).
GNK_HD int gnk_kmaj_frag_offset(int rb, int ks, int lane) {
    return gnk_kmaj_offset(rb + (lane & 15), 4 * ks + (lane >> 4));
}
// MN-major: two ds_read_b64_tr_b8; the first at this offset (k rows 64ks+16g .. +7), the second
// 8*L further (k rows +8 .. +15).  rb = first column of the fragment in the tile (multiple of 16).
template <int L> GNK_HD int gnk_mnmaj_frag_offset(int rb, int ks, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 1, hb = lane & 1;
    return gnk_mnmaj_offset<L>(64 * ks + 16 * g + q, rb >> 4) + 8 * hb;
}

// =============================================================================================
// 64-k stage images of the 4-stage kernel (gnk_gemm_pp2.h): one stage of an operand is 256 x 64 bytes.
// Two MFMA slot conventions read the same images (both operands use the same one, so the dot product is
// k-permutation safe):
//  * fp8, 32-row tiles (v_mfma_scale_f32_32x32x64_f8f6f4): lane l serves tile row (l & 31) and k-half
//    h = l >> 5; its 32 bytes are [ k = 16h .. 16h+15 | k = 32+16h .. 32+16h+15 ] (chunks h and 2+h);
//  * int8, 16-row tiles (v_mfma_i32_16x16x64_i8): lane l serves tile row (l & 15), k = 16 (l >> 4) .. +15.
//
//  * K-major stage  [256 rows][64 B]: chunk c (0..3) of row r at slot c ^ P[(r >> 2) & 3], P = {0, 2, 3, 1}
//    (the permutation keeps ds_read_b128's lane groups {0-3,12-15,20-27}... conflict free for BOTH conventions).
//  * MN-major stage [64 k][256 B]:    chunk c (0..15) of k-row k at slot c ^ (((k & 7) << 1) | ((k >> 4) & 1)).
GNK_HD int gnk_k64_swz(int r) { return (0x78 >> (2 * ((r >> 2) & 3))) & 3; }
GNK_HD int gnk_k64_offset(int r, int c) { return r * 64 + ((c ^ gnk_k64_swz(r)) << 4); }
GNK_HD void gnk_k64_stage_src(int p, int &r, int &c) {
    r = p >> 2;
    c = (p & 3) ^ gnk_k64_swz(r);
}
// 32-row tiles: ds_read_b128 #half (0, 1) of the fragment whose first row is rb
GNK_HD int gnk_k64_frag_offset(int rb, int half, int lane) { return gnk_k64_offset(rb + (lane & 31), 2 * half + (lane >> 5)); }
// 16-row tiles: the one ds_read_b128 of the fragment whose first row is rb
GNK_HD int gnk_k64_frag16_offset(int rb, int lane) { return gnk_k64_offset(rb + (lane & 15), lane >> 4); }

GNK_HD int gnk_mn64_swz(int k) { return ((k & 7) << 1) | ((k >> 4) & 1); }
GNK_HD int gnk_mn64_offset(int k, int c) { return k * 256 + ((c ^ gnk_mn64_swz(k)) << 4); }
GNK_HD void gnk_mn64_stage_src(int p, int &k, int &c) {
    k = p >> 4;
    c = (p & 15) ^ gnk_mn64_swz(k);
}
// 32-column tiles: ds_read_b64_tr_b8 #(half, t): k rows 32*half + 16h + 8t .. +7 of columns rb + 16*((l>>4)&1) .. +15
GNK_HD int gnk_mn64_frag_offset(int rb, int half, int t, int lane) {
    const int h = lane >> 5, blk = (lane >> 4) & 1, q = (lane & 15) >> 1, hb = lane & 1;
    return gnk_mn64_offset(32 * half + 16 * h + 8 * t + q, (rb >> 4) + blk) + 8 * hb;
}
// 16-column tiles: ds_read_b64_tr_b8 #t (0, 1): k rows 16 (l >> 4) + 8t .. +7 of columns rb .. rb+15
GNK_HD int gnk_mn64_frag16_offset(int rb, int t, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 1, hb = lane & 1;
    return gnk_mn64_offset(16 * g + 8 * t + q, rb >> 4) + 8 * hb;
}
