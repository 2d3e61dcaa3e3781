// Host-side simulation of the GEMM's LDS image: stage (global_load_lds, lane-linear) -> fragment
// reads (ds_read_b128 / ds_read_b64_tr_b8).  Checks, for every tile shape the kernels use, that
//   (1) staging is a bijection between LDS 16-byte positions and (row, chunk) of the tile,
//   (2) every lane of every MFMA fragment read gets exactly the (row, k) elements the MFMA slot
//       convention wants (k = 64*ks + 16*(lane>>4) + byte),
//   (3) the reads are bank-conflict free under the gfx950 banking rules
//       (ds_read_b128: four 16-lane groups, 64 banks x 4 B; ds_read_b64[_tr_b8]: two 32-lane halves).
// ds_read_b64_tr_b8 semantics (per 16 consecutive lanes; lane 2q+p addresses row q, bytes 8p..8p+7):
// lane i receives byte (i & 7) of the 8 bytes addressed by lane 2b + (i >> 3), for b = 0..7.
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "../quantized-training_amd/csrc/gnk_lds_layout.h"

struct Elem { int row, k; };   // tile coordinates: row = m or n index, k = k index
static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails < 20) { printf(__VA_ARGS__); printf("\n"); } ++fails; } } while (0)

static const int B128_GROUPS[4][16] = {
    {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};

template <int ROWS> void sim_kmajor() {
    std::vector<Elem> lds(ROWS * 128, Elem{-1, -1});
    std::set<int> seen;
    for (int p = 0; p < ROWS * 8; ++p) {
        int r, c;
        gnk_kmaj_stage_src(p, r, c);
        CHECK(r >= 0 && r < ROWS && c >= 0 && c < 8, "kmajor<%d>: stage p=%d -> (%d,%d) out of range", ROWS, p, r, c);
        CHECK(seen.insert(r * 8 + c).second, "kmajor<%d>: (%d,%d) staged twice", ROWS, r, c);
        for (int b = 0; b < 16; ++b) lds[p * 16 + b] = Elem{r, 16 * c + b};
    }
    for (int rb = 0; rb < ROWS; rb += 16)
        for (int ks = 0; ks < 2; ++ks) {
            int off[64];
            for (int lane = 0; lane < 64; ++lane) {
                off[lane] = gnk_kmaj_frag_offset(rb, ks, lane);
                CHECK(off[lane] % 16 == 0 && off[lane] >= 0 && off[lane] + 16 <= ROWS * 128, "kmajor: bad offset");
                for (int b = 0; b < 16; ++b) {
                    Elem e = lds[off[lane] + b];
                    CHECK(e.row == rb + (lane & 15) && e.k == 64 * ks + 16 * (lane >> 4) + b,
                          "kmajor<%d> rb=%d ks=%d lane=%d byte=%d: got (%d,%d)", ROWS, rb, ks, lane, b, e.row, e.k);
                }
            }
            for (int g = 0; g < 4; ++g) {
                std::set<int> slots;
                for (int i = 0; i < 16; ++i) slots.insert((off[B128_GROUPS[g][i]] / 16) % 16);
                CHECK(slots.size() == 16, "kmajor<%d> rb=%d ks=%d: b128 group %d has a bank conflict (%zu slots)", ROWS, rb, ks, g, slots.size());
            }
        }
}

template <int L> void sim_mnmajor() {
    std::vector<Elem> lds(128 * L, Elem{-1, -1});
    std::set<int> seen;
    for (int p = 0; p < 128 * L / 16; ++p) {
        int k, c;
        gnk_mnmaj_stage_src<L>(p, k, c);
        CHECK(k >= 0 && k < 128 && c >= 0 && c < L / 16, "mnmajor<%d>: stage p=%d -> (%d,%d) out of range", L, p, k, c);
        CHECK(seen.insert(k * (L / 16) + c).second, "mnmajor<%d>: (%d,%d) staged twice", L, k, c);
        for (int b = 0; b < 16; ++b) lds[p * 16 + b] = Elem{16 * c + b, k};
    }
    for (int rb = 0; rb < L; rb += 16)
        for (int ks = 0; ks < 2; ++ks)
            for (int half = 0; half < 2; ++half) {   // the two tr8 reads of a fragment
                int addr[64];
                for (int lane = 0; lane < 64; ++lane) {
                    addr[lane] = gnk_mnmaj_frag_offset<L>(rb, ks, lane) + half * 8 * L;
                    CHECK(addr[lane] % 8 == 0 && addr[lane] >= 0 && addr[lane] + 8 <= 128 * L, "mnmajor: bad address");
                }
                for (int lane = 0; lane < 64; ++lane) {
                    const int grp = lane & ~15, i = lane & 15;
                    for (int b = 0; b < 8; ++b) {
                        Elem e = lds[addr[grp + 2 * b + (i >> 3)] + (i & 7)];
                        const int want_k = 64 * ks + 16 * (lane >> 4) + 8 * half + b;
                        CHECK(e.row == rb + i && e.k == want_k, "mnmajor<%d> rb=%d ks=%d half=%d lane=%d byte=%d: got (n=%d,k=%d) want (n=%d,k=%d)",
                              L, rb, ks, half, lane, b, e.row, e.k, rb + i, want_k);
                    }
                }
                for (int h = 0; h < 2; ++h) {
                    std::set<int> slots;
                    for (int i = 0; i < 32; ++i) slots.insert((addr[32 * h + i] / 8) % 32);
                    CHECK(slots.size() == 32, "mnmajor<%d> rb=%d ks=%d: b64 half %d has a bank conflict (%zu slots)", L, rb, ks, h, slots.size());
                }
            }
}

// ---- 64-k stage images of the 4-stage kernel (32-row MFMA tiles) ------------------------------------
void sim_k64() {
    const int ROWS = 256;
    std::vector<Elem> lds(ROWS * 64, Elem{-1, -1});
    std::set<int> seen;
    for (int p = 0; p < ROWS * 4; ++p) {
        int r, c;
        gnk_k64_stage_src(p, r, c);
        CHECK(r >= 0 && r < ROWS && c >= 0 && c < 4, "k64: stage p=%d out of range", p);
        CHECK(seen.insert(r * 4 + c).second, "k64: (%d,%d) staged twice", r, c);
        for (int b = 0; b < 16; ++b) lds[p * 16 + b] = Elem{r, 16 * c + b};
    }
    for (int rb = 0; rb < ROWS; rb += 32)
        for (int half = 0; half < 2; ++half) {
            int off[64];
            for (int lane = 0; lane < 64; ++lane) {
                off[lane] = gnk_k64_frag_offset(rb, half, lane);
                for (int b = 0; b < 16; ++b) {
                    Elem e = lds[off[lane] + b];
                    CHECK(e.row == rb + (lane & 31) && e.k == 32 * half + 16 * (lane >> 5) + b, "k64 rb=%d half=%d lane=%d byte=%d: got (%d,%d)", rb, half, lane, b, e.row, e.k);
                }
            }
            for (int g = 0; g < 4; ++g) {
                std::set<int> slots;
                for (int i = 0; i < 16; ++i) slots.insert((off[B128_GROUPS[g][i]] / 16) % 16);
                CHECK(slots.size() == 16, "k64 rb=%d half=%d: b128 group %d has a bank conflict (%zu slots)", rb, half, g, slots.size());
            }
        }
}

void sim_k64_frag16() {
    const int ROWS = 256;
    std::vector<Elem> lds(ROWS * 64, Elem{-1, -1});
    for (int p = 0; p < ROWS * 4; ++p) {
        int r, c;
        gnk_k64_stage_src(p, r, c);
        for (int b = 0; b < 16; ++b) lds[p * 16 + b] = Elem{r, 16 * c + b};
    }
    for (int rb = 0; rb < ROWS; rb += 16) {
        int off[64];
        for (int lane = 0; lane < 64; ++lane) {
            off[lane] = gnk_k64_frag16_offset(rb, lane);
            CHECK(off[lane] - gnk_k64_frag16_offset(0, lane) == rb * 64, "k64/16: row block %d is not a constant byte offset", rb);
            for (int b = 0; b < 16; ++b) {
                Elem e = lds[off[lane] + b];
                CHECK(e.row == rb + (lane & 15) && e.k == 16 * (lane >> 4) + b, "k64/16 rb=%d lane=%d byte=%d: got (%d,%d)", rb, lane, b, e.row, e.k);
            }
        }
        for (int g = 0; g < 4; ++g) {
            std::set<int> slots;
            for (int i = 0; i < 16; ++i) slots.insert((off[B128_GROUPS[g][i]] / 16) % 16);
            CHECK(slots.size() == 16, "k64/16 rb=%d: b128 group %d has a bank conflict (%zu slots)", rb, g, slots.size());
        }
    }
}

void sim_mn64_frag16() {
    std::vector<Elem> lds(64 * 256, Elem{-1, -1});
    for (int p = 0; p < 64 * 16; ++p) {
        int k, c;
        gnk_mn64_stage_src(p, k, c);
        for (int b = 0; b < 16; ++b) lds[p * 16 + b] = Elem{16 * c + b, k};
    }
    for (int rb = 0; rb < 256; rb += 16)
        for (int t = 0; t < 2; ++t) {
            int addr[64];
            for (int lane = 0; lane < 64; ++lane) {
                addr[lane] = gnk_mn64_frag16_offset(rb, t, lane);
                CHECK(addr[lane] == gnk_mn64_frag16_offset(rb, 0, lane) + t * 8 * 256, "mn64/16: second read is not +8 k-rows");
            }
            for (int lane = 0; lane < 64; ++lane) {
                const int grp = lane & ~15, i = lane & 15;
                for (int b = 0; b < 8; ++b) {
                    Elem e = lds[addr[grp + 2 * b + (i >> 3)] + (i & 7)];
                    const int want_n = rb + i, want_k = 16 * (lane >> 4) + 8 * t + b;
                    CHECK(e.row == want_n && e.k == want_k, "mn64/16 rb=%d t=%d lane=%d byte=%d: got (n=%d,k=%d) want (%d,%d)", rb, t, lane, b, e.row, e.k, want_n, want_k);
                }
            }
            for (int h = 0; h < 2; ++h) {
                std::set<int> slots;
                for (int i = 0; i < 32; ++i) slots.insert((addr[32 * h + i] / 8) % 32);
                CHECK(slots.size() == 32, "mn64/16 rb=%d t=%d: b64 half %d has a bank conflict (%zu slots)", rb, t, h, slots.size());
            }
        }
}

void sim_mn64() {
    std::vector<Elem> lds(64 * 256, Elem{-1, -1});
    std::set<int> seen;
    for (int p = 0; p < 64 * 16; ++p) {
        int k, c;
        gnk_mn64_stage_src(p, k, c);
        CHECK(k >= 0 && k < 64 && c >= 0 && c < 16, "mn64: stage p=%d out of range", p);
        CHECK(seen.insert(k * 16 + c).second, "mn64: (%d,%d) staged twice", k, c);
        for (int b = 0; b < 16; ++b) lds[p * 16 + b] = Elem{16 * c + b, k};
    }
    for (int rb = 0; rb < 256; rb += 32)
        for (int half = 0; half < 2; ++half)
            for (int t = 0; t < 2; ++t) {
                int addr[64];
                for (int lane = 0; lane < 64; ++lane) {
                    addr[lane] = gnk_mn64_frag_offset(rb, half, t, lane);
                    CHECK(addr[lane] == gnk_mn64_frag_offset(rb, 0, 0, lane) + (32 * half + 8 * t) * 256, "mn64: read (%d,%d) is not a constant offset from read (0,0)", half, t);
                }
                for (int lane = 0; lane < 64; ++lane) {
                    const int grp = lane & ~15, i = lane & 15;
                    for (int b = 0; b < 8; ++b) {
                        Elem e = lds[addr[grp + 2 * b + (i >> 3)] + (i & 7)];
                        const int want_n = rb + (lane & 31), want_k = 32 * half + 16 * (lane >> 5) + 8 * t + b;
                        CHECK(e.row == want_n && e.k == want_k, "mn64 rb=%d half=%d t=%d lane=%d byte=%d: got (n=%d,k=%d) want (%d,%d)", rb, half, t, lane, b, e.row, e.k, want_n, want_k);
                    }
                }
                for (int h = 0; h < 2; ++h) {
                    std::set<int> slots;
                    for (int i = 0; i < 32; ++i) slots.insert((addr[32 * h + i] / 8) % 32);
                    CHECK(slots.size() == 32, "mn64 rb=%d half=%d t=%d: b64 half %d has a bank conflict (%zu slots)", rb, half, t, h, slots.size());
                }
            }
}

int main() {
    sim_k64();
    sim_mn64();
    sim_k64_frag16();
    sim_mn64_frag16();
    sim_kmajor<128>();
    sim_kmajor<256>();
    sim_mnmajor<64>();
    sim_mnmajor<128>();
    sim_mnmajor<256>();
    if (fails) { printf("FAILED: %d checks\n", fails); return 1; }
    printf("OK\n");
    return 0;
}
