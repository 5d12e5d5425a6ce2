// mt_state_roundtrip.cpp -- host-only check of include/ldpc/bp_simulation.h's std::mt19937 <-> (624 words, next index) conversion, the
// form in which the exact-replay harness hands the caller's generator to the device and takes it back.  No GPU, no library call.
#include <cstdio>
#include <random>

#include "ldpc/bp_simulation.h"

int main() {
    int bad = 0;
    for (unsigned seed : {1u, 5489u, 4294967295u, 123456789u}) {
        for (int burn : {0, 1, 399, 623, 624, 625, 1024, 100000}) {
            std::mt19937 g(seed);
            g.discard((unsigned long long)burn);
            uint32_t w[624];
            int pos = -1;
            if (!ldpc::mt_export(g, w, pos)) { printf("export failed seed %u burn %d\n", seed, burn); ++bad; continue; }
            // a freshly seeded engine regenerates at its first draw (index 624); afterwards the index is the position in the block
            const int want_pos = burn == 0 ? 624 : (burn % 624 == 0 ? 624 : burn % 624);
            if (pos != want_pos) { printf("position %d, expected %d (seed %u burn %d)\n", pos, want_pos, seed, burn); ++bad; }
            std::mt19937 h(99);
            if (!ldpc::mt_import(h, w, pos)) { printf("import failed seed %u burn %d\n", seed, burn); ++bad; continue; }
            for (int i = 0; i < 2000; ++i)
                if (g() != h()) { printf("streams differ at %d (seed %u burn %d)\n", i, seed, burn); ++bad; break; }
        }
    }
    // a corrupted word list must be refused by the probe of the next word
    {
        std::mt19937 g(7);
        uint32_t w[624];
        int pos = 0;
        ldpc::mt_export(g, w, pos);
        if (ldpc::mt_word_after(w, pos) != (unsigned)std::mt19937(7)()) { printf("mt_word_after wrong\n"); ++bad; }
    }
    printf(bad ? "FAILED %d\n" : "ok\n", bad);
    return bad ? 1 : 0;
}
