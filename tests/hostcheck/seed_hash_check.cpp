// Host-side check of the seed table's key functions (slamem_amd/csrc/common.h; compiled with hipcc, runs without a GPU):
//   * seed_mix / seed_mix64 are bijections of the keys of `bits` bits (exhaustive for 16 and 20 bits, no two of 4 M sampled keys
//     of 32 / 34 / 36 bits collide and every image fits the width);
//   * seed_place files a k-mer and its reverse complement in the same bucket under the same tag bits, with opposite orientation
//     bits -- or calls it a palindrome --, for seeds of 8 to 18 letters, and bucket / tag stay inside their widths.
#include "../../slamem_amd/csrc/common.h"

#include <algorithm>
#include <cstdio>
#include <vector>

using namespace slamem;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main() {
    int bad = 0;
    for (uint32_t bits : {16u, 20u}) {
        std::vector<uint8_t> seen32(1u << bits, 0), seen64(1u << bits, 0);
        for (uint32_t x = 0; x < (1u << bits); x++) {
            const uint32_t a = seed_mix(x, bits);
            const uint64_t b = seed_mix64(x, bits);
            if (a >> bits || b >> bits || seen32[a]++ || seen64[b]++) { bad++; break; }
        }
    }
    for (uint32_t bits : {32u, 34u, 36u}) {
        std::vector<uint64_t> im;
        std::vector<uint64_t> keys;
        const uint64_t mask = bits >= 64 ? ~0ull : (1ull << bits) - 1ull;
        for (int i = 0; i < (1 << 22); i++) keys.push_back(rnd() & mask);
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        for (uint64_t x : keys) {
            const uint64_t h = bits <= 32 ? (uint64_t)seed_mix((uint32_t)x, bits) : seed_mix64(x, bits);
            if (h & ~mask) bad++;
            im.push_back(h);
        }
        std::sort(im.begin(), im.end());
        if (std::adjacent_find(im.begin(), im.end()) != im.end()) bad++;
    }
    for (uint32_t k = 8; k <= kSeedMaxK; k++) {
        const uint32_t tb = k <= 10 ? 2u * k - 16u : 7u, log2b = 2u * k - tb;  // any split with at most seven tag bits
        const uint32_t km = (1u << k) - 1u;
        for (int i = 0; i < 200000; i++) {
            uint32_t f0 = (uint32_t)rnd() & km, f1 = (uint32_t)rnd() & km;
            if (i % 1000 == 0) {  // a palindrome: the reverse complement of its own first half behind it (even k)
                if (k % 2 == 0) {
                    const uint32_t h = k / 2, hm = (1u << h) - 1u;
                    const uint32_t a0 = f0 & hm, a1 = f1 & hm;
                    f0 = a0 | (seed_rev_field(a0, h) << h);
                    f1 = a1 | (seed_rev_field(a1, h) << h);
                }
            }
            const uint32_t r0 = seed_rev_field(f0, k), r1 = seed_rev_field(f1, k);
            uint32_t b1, t1, o1, p1, b2, t2, o2, p2;
            seed_place(f0, f1, k, log2b, b1, t1, o1, p1);
            seed_place(r0, r1, k, log2b, b2, t2, o2, p2);
            if (b1 != b2 || t1 != t2 || p1 != p2 || (p1 ? (o1 | o2) != 0u : (o1 ^ o2) != 1u)) bad++;
            if ((uint64_t)b1 >> log2b || t1 >> tb) bad++;
            if (p1 != (uint32_t)(f0 == r0 && f1 == r1)) bad++;
            if (seed_rev_field(r0, k) != f0) bad++;
        }
    }
    printf("%s\n", bad ? "FAILED" : "seed hash ok");
    return bad ? 1 : 0;
}
