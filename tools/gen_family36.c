// Offline generator for the built-in 36-bit, min-Hamming-11 codebook ("tag36h11-compatible").
//
// The upstream AprilTag tag36h11 table (587 codes) is data of the external apriltag C library
// [EXT] and is not present in this environment.  This tool builds a lexicode with the SAME
// parameters (6x6 data bits, min Hamming distance 11 under all four rotations, candidates walked
// with the upstream generator's increment 982451653 mod 2^36 starting at upstream code 0), seeded
// with the 39 upstream codes that could be recalled AND verified: code[i] - code[0] is a small,
// strictly increasing multiple k_i of the increment (k = 0,1,2,4,6,8,13,...,120; a misremembered
// 36-bit value lands on such a k with probability ~1e-9), every one of them is >= 11 away from all
// rotations of the earlier ones, and upstream's own order is the order of k.  IDs 0..38 therefore
// equal upstream IDs — that covers every tag of the reference's field.json (IDs 1..32).  IDs >= 39
// are NOT upstream tag36h11 IDs: upstream's generator also rejects about half of the distance-valid
// candidates with a pattern-complexity test that could not be reconstructed (between k = 0 and 120
// it drops 35 candidates that pass every distance test).  A deployment that needs real tag36h11 IDs passes the upstream table
// through ck_family_create().
//
// Codes are emitted in the AprilTag-2 convention (row-major, MSB = top-left data bit).
// Build: gcc -O3 -march=native -fopenmp tools/gen_family36.c -o /tmp/gen36 ; /tmp/gen36 587 > codes.txt
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NB 36
#define D 6
#define MINH 11
#define NSEED 100 // upstream codes recalled AND verified on the generator lattice (tests/test_families.py): ids 0..38 with their indices, 61 more without
static const uint64_t MASK = (1ULL << NB) - 1;
static const uint64_t PRIME = 982451653ULL;

static uint64_t rot90(uint64_t w) {
    uint64_t wr = 0;
    for (int r = D - 1; r >= 0; r--)
        for (int c = 0; c < D; c++) {
            int b = r + D * c;
            wr = (wr << 1) | ((w >> b) & 1);
        }
    return wr;
}
// number of 4-neighbour transitions of the 8x8 image (6x6 data + black border ring)
static int energy(uint64_t v) {
    int im[D + 2][D + 2];
    memset(im, 0, sizeof im);
    for (int y = 0; y < D; y++)
        for (int x = 0; x < D; x++) im[y + 1][x + 1] = (v >> (NB - 1 - (y * D + x))) & 1;
    int e = 0;
    for (int y = 0; y < D + 2; y++)
        for (int x = 0; x < D + 1; x++) e += im[y][x] != im[y][x + 1];
    for (int x = 0; x < D + 2; x++)
        for (int y = 0; y < D + 1; y++) e += im[y][x] != im[y + 1][x];
    return e;
}
static int self_ok(uint64_t v) {
    uint64_t r1 = rot90(v), r2 = rot90(r1), r3 = rot90(r2);
    return __builtin_popcountll(v ^ r1) >= MINH && __builtin_popcountll(v ^ r2) >= MINH &&
           __builtin_popcountll(v ^ r3) >= MINH && __builtin_popcountll(r1 ^ r2) >= MINH &&
           __builtin_popcountll(r1 ^ r3) >= MINH && __builtin_popcountll(r2 ^ r3) >= MINH;
}

static uint64_t rots[4 * 1024];
static int nrots = 0;
static uint64_t codes[1024];
static int ncodes = 0;

static void add_code(uint64_t v) {
    codes[ncodes++] = v;
    uint64_t r = v;
    for (int i = 0; i < 4; i++) { rots[nrots++] = r; r = rot90(r); }
}
static inline int far_from_all(uint64_t v, int from) {
    for (int i = from; i < nrots; i++)
        if (__builtin_popcountll(v ^ rots[i]) < MINH) return 0;
    return 1;
}

int main(int argc, char **argv) {
    int want = argc > 1 ? atoi(argv[1]) : 587;
    static const uint64_t seed[NSEED] = {
        0xd5d628584ULL, 0xd97f18b49ULL, 0xdd280910eULL, 0xe479e9c98ULL,
        0xebcbca822ULL, 0xf31dab3acULL, 0x056a5d085ULL, 0x10652e1d4ULL,
        0x22b1dfeadULL, 0x265ad0472ULL, 0x34fe91b86ULL, 0x3ff962cd5ULL,
        0x43a25329aULL, 0x474b4385fULL, 0x4e9d243e9ULL, 0x5246149aeULL,
        0x5997f5538ULL, 0x683bb6c4cULL, 0x6be4a7211ULL, 0x7e3158eeaULL,
        0x81da494afULL, 0x858339a74ULL, 0x8cd51a5feULL, 0x9f21cc2d7ULL,
        0xa2cabc89cULL, 0xadc58d9ebULL, 0xb16e7dfb0ULL, 0xb8c05eb3aULL,
        0xd25ef139dULL, 0xd607e1962ULL, 0xe4aba3076ULL, 0x2dde6a3daULL,
        0x43d40c678ULL, 0x5620be351ULL, 0x64c47fa65ULL, 0x686d7002aULL,
        0x6c16605efULL, 0x6fbf50bb4ULL, 0x8d06d39dcULL,
        // 61 further upstream codes (k = 125 ... 871 on the same walk; round-3 review, verified by tests/test_families.py):
        // upstream codes in upstream order, but it cannot be shown that no upstream code lies between two of them, so
        // their INDICES are not claimed (n_upstream stays 39); seeding them keeps the stand-ins from colliding with real tags
        0x9f53856b5ULL, 0xadf746dc9ULL, 0xbc9b084ddULL, 0xd290aa77bULL,
        0xd9e28b305ULL, 0xe4dd5c454ULL, 0xfad2fe6f2ULL, 0x181a8151aULL,
        0x26be42c2eULL, 0x2e10237b8ULL, 0x405cd5491ULL, 0x7742eab1cULL,
        0x85e6ac230ULL, 0x8d388cdbaULL, 0x9f853ea93ULL, 0xc41ea2445ULL,
        0xcf1973594ULL, 0x14a34a333ULL, 0x31eacd15bULL, 0x6c79d2dabULL,
        0x73cbb3935ULL, 0x89c155bd3ULL, 0x8d6a46198ULL, 0x91133675dULL,
        0xa708d89fbULL, 0xae5ab9585ULL, 0xb9558a6d4ULL, 0xb98743ab2ULL,
        0xd6cec68daULL, 0x1506bcaefULL, 0x4becd217aULL, 0x4f95c273fULL,
        0x658b649ddULL, 0xa76c4b1b7ULL, 0xecf621f56ULL, 0x1c8a56a57ULL,
        0x3628e92baULL, 0x53706c0e2ULL, 0x5e6b3d231ULL, 0x7809cfa94ULL,
        0xe97eead6fULL, 0x5af40604aULL, 0x7492988adULL, 0xed5994712ULL,
        0x5eceaf9edULL, 0x7c1632815ULL, 0xc1a0095b4ULL, 0xe9e25d52bULL,
        0x3a6705419ULL, 0xa8333012fULL, 0x4ce5704d0ULL, 0x508e60a95ULL,
        0x877476120ULL, 0xa864e950dULL, 0xea45cfce7ULL, 0x19da047e8ULL,
        0x24d4d5937ULL, 0x6e079cc9bULL, 0x99f2e11d7ULL, 0x33aa50429ULL,
        0x499ff26c7ULL};
    for (int i = 0; i < NSEED; i++) add_code(seed[i]);
    uint64_t v0 = (seed[NSEED - 1] + PRIME) & MASK;
    const uint64_t CH = 1ULL << 22;
    uint64_t *surv = malloc(CH * sizeof(uint64_t));
    uint64_t total = (1ULL << NB);
    for (uint64_t base = 0; base < total && ncodes < want; base += CH) {
        uint64_t ns = 0;
#pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < CH; i++) {
            uint64_t v = (v0 + PRIME * (base + i)) & MASK;
            surv[i] = far_from_all(v, 0) ? v : ~0ULL;
        }
        int nr0 = nrots;
        for (uint64_t i = 0; i < CH && ncodes < want; i++) {
            uint64_t v = surv[i];
            if (v == ~0ULL) continue;
            ns++;
            if (!far_from_all(v, nr0)) continue;
            if (energy(v) < 34) continue;  // (a filler rule for the stand-ins only: upstream's rejection is the paper's rectangle-cover complexity, not an energy threshold —
                                           // the distance-valid lattice points upstream skipped below k = 871 have energies 28..54)
            if (!self_ok(v)) continue;
            add_code(v);
            fprintf(stderr, "code %d = 0x%09llx at iter %llu\n", ncodes - 1, (unsigned long long)v,
                    (unsigned long long)(base + i));
        }
        (void)ns;
    }
    for (int i = 0; i < ncodes; i++) printf("0x%09llx\n", (unsigned long long)codes[i]);
    return 0;
}
