// host-only contexts under AddressSanitizer: schedules, work lists and tables on sets of every kind
#include "sa_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
extern "C" int sa_debug_work_items(sa_ctx*, int, uint64_t, uint64_t, uint32_t*, uint64_t, uint64_t*);
extern "C" int sa_debug_bucket_order(sa_ctx*, uint32_t*, uint64_t, uint64_t*);
int main() {
    std::mt19937 rng(7);
    int32_t S[400];
    sa_default_matrix(S);
    int cases = 0;
    for (int t = 0; t < 60; t++) {
        const int npep = (int[]){0, 1, 3, 70, 300, 900}[rng() % 6];
        const int nlong = (int[]){0, 1, 2, 9, 70}[rng() % 5];
        if (npep + nlong < 2) continue;
        const int plo = 1 + rng() % 60, phi = plo + rng() % (65 - plo);
        const int llo = 65 + rng() % 300, lhi = llo + rng() % 700;
        std::vector<std::string> seqs;
        for (int k = 0; k < npep; k++) seqs.push_back(std::string(plo + rng() % (phi - plo + 1), 'A' + 0));
        for (int k = 0; k < nlong; k++) seqs.insert(seqs.begin() + rng() % (seqs.size() + 1), std::string(llo + rng() % (lhi - llo + 1), 'C'));
        uint32_t longest = 0;
        std::vector<uint8_t> codes; std::vector<uint64_t> offs; std::vector<uint32_t> lens;
        for (auto& s : seqs) {
            offs.push_back(codes.size()); lens.push_back((uint32_t)s.size()); longest = std::max<uint32_t>(longest, s.size());
            for (char ch : s) codes.push_back(ch == 'A' ? 0 : 4);
        }
        sa_ctx* c = nullptr;
        if (sa_init(&c, S, -4, std::max<uint32_t>(64, longest), SA_DEVICE_NONE)) { printf("init failed\n"); return 1; }
        if (int rc = sa_upload(c, codes.data(), offs.data(), lens.data(), seqs.size())) { printf("upload %d\n", rc); return 1; }
        if (rng() % 3 == 0) sa_set_bucket_width(c, 1 + rng() % 70);
        if (rng() % 4 == 0) sa_set_pairing(c, 0);
        if (rng() % 4 == 0) sa_set_peptide_tiles(c, 0);
        const uint64_t rows = sa_schedule_rows(c, SA_SCHEDULE_TRIANGULAR);
        for (int part = 0; part < 3; part++) {
            uint64_t r0 = part == 0 ? 0 : rng() % (rows + 1), r1 = part == 0 ? rows : r0 + rng() % (rows - r0 + 1);
            uint64_t count = 0;
            if (int rc = sa_debug_work_items(c, SA_SCHEDULE_TRIANGULAR, r0, r1, nullptr, 0, &count)) { printf("items %d\n", rc); return 1; }
            std::vector<uint32_t> items(count * 8 + 8);
            sa_debug_work_items(c, SA_SCHEDULE_TRIANGULAR, r0, r1, items.data(), count, &count);
            // pairs covered: sum over items of entries (lanes counted on the host side of the kernels' scan)
            (void)sa_cell_count(c, SA_SCHEDULE_TRIANGULAR, r0, r1);
            (void)sa_compact_bytes(c, SA_SCHEDULE_TRIANGULAR, r0, r1);
        }
        uint64_t bounds[9];
        sa_shard_rows(c, SA_SCHEDULE_TRIANGULAR, 8, bounds);
        std::vector<uint32_t> order(seqs.size());
        uint64_t n = 0;
        sa_debug_bucket_order(c, order.data(), order.size(), &n);
        sa_destroy(c);
        cases++;
    }
    printf("host check: %d sets, no sanitizer report\n", cases);
    return 0;
}
