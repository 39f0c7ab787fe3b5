// Host-side check of the note a batch leaves for the next one about the seed kernel's form (slamem::seed_words_next in
// slamem_amd/csrc/common.h; compiled with hipcc, runs without a GPU): a run of batches through the rule, as
// tests/test_gpu_seed.py::test_the_form_follows_the_reads_of_the_last_batch sees it on the GPU.
#include "../../slamem_amd/csrc/common.h"

#include <cstdio>

using namespace slamem;

struct Batch { uint32_t avg; uint64_t n150, n250, n380; };  // reads of 150 / 250 / 380 letters

// what the kernel would count in a batch run with `words` plane words (every read sampled), then the rule
static uint32_t step(uint32_t& hint, const Batch& b, uint32_t* ran) {
    const uint32_t words_avg = b.avg > 256u ? 6u : b.avg > 192u ? 4u : 3u;
    const uint32_t words = words_avg > hint ? words_avg : hint;
    const uint64_t sampled = b.n150 + b.n250 + b.n380;
    const uint64_t c4 = words < 4u ? b.n250 : 0u, c6 = words < 6u ? b.n380 : 0u;
    const bool needed = (words == 4u && b.n250 != 0u) || (words == 6u && b.n380 != 0u);
    *ran = words;
    hint = seed_words_next(hint, words, words_avg, c4, c6, sampled, needed);
    return hint;
}

int main() {
    int bad = 0;
    uint32_t hint = 0, ran = 0;
    const Batch mixed = {180, 700, 300, 0}, longer = {219, 700, 0, 300}, shortb = {150, 1000, 0, 0}, few = {152, 990, 10, 0};
    auto expect = [&](const Batch& b, uint32_t want_ran, uint32_t want_hint, const char* what) {
        step(hint, b, &ran);
        if (ran != want_ran || hint != want_hint) { printf("FAIL %s: ran %u (want %u), note %u (want %u)\n", what, ran, want_ran, hint, want_hint); bad++; }
    };
    expect(mixed, 3, 4, "mixed, first batch: three words by the average, the note asks for four");
    expect(mixed, 4, 4, "mixed, second batch: four words, needed: the note stays");
    expect(shortb, 4, 0, "short batch under the note: four words nobody needed: back");
    expect(shortb, 3, 0, "short batch: three words, nothing to note");
    expect(few, 3, 0, "one read in a hundred beyond the form: below an eighth, no note");
    expect(longer, 4, 6, "average 219: four words; 30 % of 380 letters: the note asks for six");
    expect(longer, 6, 6, "six words, needed: stays");
    expect(mixed, 6, 4, "six words from the note, no read beyond 256 letters: one step back");
    expect(mixed, 4, 4, "four words, needed: stays");
    expect(shortb, 4, 0, "back to none");
    // the note never asks for a form narrower than it was for a batch that needs it, and is one of 0, 4, 6
    for (uint32_t h : {0u, 4u, 6u})
        for (uint32_t w : {3u, 4u, 6u})
            for (uint32_t wa : {3u, 4u, 6u})
                for (uint64_t c4 : {0ull, 10ull, 200ull})
                    for (uint64_t c6 : {0ull, 10ull, 200ull})
                        for (int needed = 0; needed < 2; needed++) {
                            const uint32_t n = seed_words_next(h, w, wa, c4, c6, 1000, needed != 0);
                            if (n != 0u && n != 4u && n != 6u) bad++;
                            if (c6 * 8u > 1000u && n != 6u) bad++;
                            if ((c4 + c6) * 8u > 1000u && n < 4u) bad++;
                            if (needed && n < h && (c4 + c6) * 8u <= 1000u) bad++;  // (a form that was needed is kept)
                        }
    if (bad) { printf("%d failures\n", bad); return 1; }
    printf("seed form ok\n");
    return 0;
}
