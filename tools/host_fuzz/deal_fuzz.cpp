// ASan + UBSan harness of the block deal (evplp_amd/csrc/host/deal.cpp; tests/test_host_sanitizers.py builds and runs it): random cost
// tables -- equal costs, zeros, costs near 2^64, one block, no block, capacities that just fit and that do not -- every deal checked for
// its invariants: every block one owner in range, nobody over capacity, evplp_rank_blocks lists exactly the owner's blocks in falling cost.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
extern "C" int evplp_deal_blocks(const uint64_t *, int32_t, int32_t, int32_t, int32_t *);
extern "C" int evplp_rank_blocks(const uint64_t *, const int32_t *, int32_t, int32_t, int32_t *, int32_t);
static int fail(const char *what, int it) { std::printf("FAIL %s (case %d)\n", what, it); return 1; }
int main(int argc, char **argv) {
    const int cases = argc > 1 ? std::atoi(argv[1]) : 3000;
    std::mt19937_64 rng(argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 5);
    long dealt = 0, refused = 0;
    for (int it = 0; it < cases; it++) {
        const int n = 1 + (int)(rng() % 16), nb = (int)(rng() % 260), cap = (nb + n - 1) / n + (int)(rng() % 4) - (rng() % 9 == 0 ? 1 : 0);
        std::vector<uint64_t> c((size_t)nb);
        const int kind = (int)(rng() % 4);
        for (auto &v : c) v = kind == 0 ? 1000u : kind == 1 ? rng() % 1000 : kind == 2 ? (rng() % 7 == 0 ? 0 : rng() >> (24 + rng() % 36)) : (rng() | (1ull << 63)) >> (rng() % 3);
        std::vector<int32_t> own((size_t)nb + 1, -7), out((size_t)nb + 1, -9);
        const int rc = evplp_deal_blocks(c.data(), nb, n, cap, own.data());
        if (cap < 0 || (long)n * cap < nb) { if (rc >= 0) return fail("a deal that cannot fit was not refused", it); refused++; continue; }
        if (rc < 0) return fail("a deal that fits was refused", it);
        if (own[(size_t)nb] != -7) return fail("wrote past the owner table", it);
        std::vector<int> cnt((size_t)n, 0); int total = 0;
        for (int b = 0; b < nb; b++) { if (own[(size_t)b] < 0 || own[(size_t)b] >= n) return fail("owner out of range", it); cnt[(size_t)own[(size_t)b]]++; }
        for (int r = 0; r < n; r++) {
            if (cnt[(size_t)r] > cap) return fail("over capacity", it);
            const int k = evplp_rank_blocks(c.data(), own.data(), nb, r, out.data(), nb);
            if (k != cnt[(size_t)r] || out[(size_t)nb] != -9) return fail("evplp_rank_blocks count / overrun", it);
            for (int i = 0; i < k; i++) if (own[(size_t)out[(size_t)i]] != r) return fail("evplp_rank_blocks lists another rank's block", it);
            for (int i = 1; i < k; i++) if (c[(size_t)out[(size_t)i - 1]] < c[(size_t)out[(size_t)i]]) return fail("not in falling cost", it);
            if (evplp_rank_blocks(nullptr, own.data(), nb, r, nullptr, 0) != k) return fail("count-only call", it);
            total += k;
        }
        if (total != nb) return fail("blocks lost", it);
        // never worse than the round-robin deal by its own measure (when round robin fits the capacity)
        if (cap >= (nb + n - 1) / n) {
            std::vector<long double> rr((size_t)n, 0), ld((size_t)n, 0);
            for (int b = 0; b < nb; b++) { rr[(size_t)(b % n)] += (long double)c[(size_t)b]; ld[(size_t)own[(size_t)b]] += (long double)c[(size_t)b]; }
            long double mr = 0, ml = 0; for (int r = 0; r < n; r++) { mr = rr[(size_t)r] > mr ? rr[(size_t)r] : mr; ml = ld[(size_t)r] > ml ? ld[(size_t)r] : ml; }
            if (kind != 3 && ml > mr * 1.0000001L) return fail("worse than round robin", it);      // (kind 3: sums beyond 2^64 wrap in the deal's own arithmetic; costs are clock ticks, 2^40 at most)
        }
        dealt++;
    }
    std::printf("ok: %ld deals checked, %ld impossible ones refused\n", dealt, refused);
    return 0;
}
