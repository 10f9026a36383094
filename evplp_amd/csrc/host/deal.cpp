// The deal of row blocks by cost (include/evplp.h): host only, no device, nothing but the standard library -- every process of a multi-process
// run calls it for itself on the same costs, and tests/test_host_sanitizers.py runs it under ASan + UBSan (tools/host_fuzz/deal_fuzz.cpp).
#include "../../../include/evplp.h"

#include <algorithm>
#include <cstdint>
#include <vector>

// ("row strips are load-imbalanced".)  Deterministic: ties go to the lower block index / lower rank, so every process of a
// multi-process run arrives at the same table.
namespace {
// pairwise improvement between the fullest rank and every other one (move a block, or swap two) until the largest load stops falling
uint64_t improve(const uint64_t *cost, int32_t nblocks, int32_t nranks, int32_t cap, int32_t *owner) {
    std::vector<uint64_t> load((size_t)nranks, 0); std::vector<int32_t> count((size_t)nranks, 0);
    for (int b = 0; b < nblocks; b++) { load[(size_t)owner[b]] += cost[b]; count[(size_t)owner[b]]++; }
    for (int pass = 0; pass < 4 * nblocks + 16; pass++) {
        int hi = 0;
        for (int r = 1; r < nranks; r++) if (load[(size_t)r] > load[(size_t)hi]) hi = r;
        // the best single move or swap between `hi` and another rank: the one that leaves the smaller of the pair's two new maxima
        uint64_t best_max = load[(size_t)hi]; int best_a = -1, best_b = -1, best_r = -1;
        for (int a = 0; a < nblocks; a++) {
            if (owner[a] != hi) continue;
            for (int r = 0; r < nranks; r++) {
                if (r == hi) continue;
                if (count[(size_t)r] < cap) {       // move a -> r
                    const uint64_t m = std::max(load[(size_t)hi] - cost[a], load[(size_t)r] + cost[a]);
                    if (m < best_max) { best_max = m; best_a = a; best_b = -1; best_r = r; }
                }
            }
            for (int b = 0; b < nblocks; b++) {   // swap a <-> b
                const int r = owner[b];
                if (r == hi || cost[b] >= cost[a]) continue;
                const uint64_t m = std::max(load[(size_t)hi] - cost[a] + cost[b], load[(size_t)r] + cost[a] - cost[b]);
                if (m < best_max) { best_max = m; best_a = a; best_b = b; best_r = r; }
            }
        }
        if (best_a < 0) break;
        owner[best_a] = best_r; load[(size_t)hi] -= cost[best_a]; load[(size_t)best_r] += cost[best_a];
        if (best_b >= 0) { owner[best_b] = hi; load[(size_t)hi] += cost[best_b]; load[(size_t)best_r] -= cost[best_b]; }
        else { count[(size_t)hi]--; count[(size_t)best_r]++; }
    }
    return nranks ? *std::max_element(load.begin(), load.end()) : 0;
}
} // namespace

// Two starting points -- longest-processing-time-first within the capacity, and the round-robin deal it replaces (when that fits the capacity)
// -- each improved pairwise; the one with the smaller largest load wins (LPT on a tie), so the result is never worse than round robin by its
// own measure.  Costs are clock ticks (< 2^45 per block in practice); sums are taken in 64 bits.
extern "C" int evplp_deal_blocks(const uint64_t *cost, int32_t nblocks, int32_t nranks, int32_t cap, int32_t *owner) {
    if (nblocks < 0 || ((!cost || !owner) && nblocks > 0) || nranks < 1 || cap < 0 || (int64_t)nranks * cap < nblocks) return EVPLP_ERR_INVALID;
    std::vector<int32_t> order((size_t)nblocks);
    for (int b = 0; b < nblocks; b++) order[(size_t)b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return cost[x] > cost[y]; });
    std::vector<uint64_t> load((size_t)nranks, 0); std::vector<int32_t> count((size_t)nranks, 0);
    for (int32_t b : order) {
        int best = -1;
        for (int r = 0; r < nranks; r++) if (count[(size_t)r] < cap && (best < 0 || load[(size_t)r] < load[(size_t)best])) best = r;
        owner[b] = best; load[(size_t)best] += cost[b]; count[(size_t)best]++;
    }
    const uint64_t lpt_max = improve(cost, nblocks, nranks, cap, owner);
    if ((int64_t)cap * nranks >= nblocks && cap >= (nblocks + nranks - 1) / nranks) {
        std::vector<int32_t> rr((size_t)nblocks);
        for (int b = 0; b < nblocks; b++) rr[(size_t)b] = b % nranks;
        if (improve(cost, nblocks, nranks, cap, rr.data()) < lpt_max) for (int b = 0; b < nblocks; b++) owner[b] = rr[(size_t)b];
    }
    return EVPLP_OK;
}

// The order in which a rank STORES the blocks a deal gave it -- and therefore the order in which its kernels launch them: the most expensive
// first.  A strip's gather is a small launch (6-10 ms at eight ranks) whose longest items -- the tiles on depth discontinuities, 2.7-4 ms each
// against a median of 0.1 ms -- sit in the middle of the image: launched in image order they start 4 ms in and the GPU drains for 1-1.5 ms
// behind them (tools/gather_times.py); launched first they are done long before the cheap blocks run out.  Ties by block index.
extern "C" int evplp_rank_blocks(const uint64_t *cost, const int32_t *owner, int32_t nblocks, int32_t rank, int32_t *out_blocks, int32_t capacity) {
    if (nblocks < 0 || (!owner && nblocks > 0)) return EVPLP_ERR_INVALID;
    std::vector<int32_t> mine;
    for (int b = 0; b < nblocks; b++) if (owner[b] == rank) mine.push_back(b);
    if (cost) std::stable_sort(mine.begin(), mine.end(), [&](int32_t x, int32_t y) { return cost[x] > cost[y]; });
    for (size_t i = 0; i < mine.size() && out_blocks && (int32_t)i < capacity; i++) out_blocks[i] = mine[i];
    return (int)mine.size();
}
