// sa_tune.cpp -- the GPU counterpart of the reference's batch-size tuner
// (src/tuners/batch.c:80-118): the reference sweeps BATCH_SIZE from 4096 to 524288
// over the first 4 000 000 rows with writing disabled and prints size, time and
// rows/s per setting.  Here the quantity that plays the role of the batch is the
// number of pairs per GPU submission (sa_settings.pairs_per_submit); the sweep runs
// the whole CSV driver without writing and prints the same kind of table, then a
// second table for the peptide kernel's own knobs (64-pair tiles per work item,
// workgroups per CU) and a third for the width of the length buckets, both measured
// with the records left on the device.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/sa_hip.h"
#include "../../include/sa_host.h"

int main(int argc, char** argv) {
    sa_settings st;
    sa_settings_default(&st);
    st.mode_write = 0;
    int kernel_rows = 20000;
    for (int k = 1; k < argc; k++) {
        if (!strcmp(argv[k], "--input") && k + 1 < argc) snprintf(st.input_file, sizeof(st.input_file), "%s", argv[++k]);
        else if (!strcmp(argv[k], "--config") && k + 1 < argc) { if (sa_settings_load(&st, argv[++k])) { fprintf(stderr, "cannot read %s\n", argv[k]); return 2; } st.mode_write = 0; }
        else if (!strcmp(argv[k], "--schedule") && k + 1 < argc) st.schedule = !strcmp(argv[++k], "triangular") ? SA_SCHEDULE_TRIANGULAR : SA_SCHEDULE_ADJACENT;
        else if (!strcmp(argv[k], "--max-seq-len") && k + 1 < argc) st.max_seq_len = (uint32_t)atoi(argv[++k]);
        else if (!strcmp(argv[k], "--kernel-rows") && k + 1 < argc) kernel_rows = atoi(argv[++k]);
        else { fprintf(stderr, "usage: %s --input in.csv [--schedule adjacent|triangular] [--max-seq-len N] [--kernel-rows N]\n", argv[0]); return 2; }
    }
    if (!*st.input_file) { fprintf(stderr, "--input is required\n"); return 2; }

    printf("Pairs per submission\tTime (s)\tPairs/s\n");
    double best_t = 1e30;
    uint64_t best = 0;
    for (uint64_t pps = 4096; pps <= (1ull << 23); pps *= 4) {
        st.pairs_per_submit = pps;
        sa_run_report rep;
        const int rc = sa_run_csv(&st, &rep);
        if (rc) { fprintf(stderr, "error %d: %s\n", rc, rep.message); return 1; }
        printf("%llu\t%.4f\t%.0f\n", (unsigned long long)pps, rep.seconds, rep.pairs / rep.seconds);
        if (rep.seconds < best_t) { best_t = rep.seconds; best = pps; }
    }
    printf("Best: %llu pairs per submission (%.4f s)\n", (unsigned long long)best, best_t);

    // Kernel knobs, on synthetic peptides (lengths 6..50, all pairs), records left on the device.
    if (kernel_rows < 2) return 0;
    uint64_t state = 1;
    auto next = [&state]() {
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    std::vector<uint8_t> codes;
    std::vector<uint64_t> offs(kernel_rows);
    std::vector<uint32_t> lens(kernel_rows);
    for (int s = 0; s < kernel_rows; s++) {
        lens[s] = 6 + (uint32_t)(next() % 45);
        offs[s] = codes.size();
        for (uint32_t i = 0; i < lens[s]; i++) codes.push_back((uint8_t)(next() % 20));
        next();
    }
    sa_ctx* ctx = nullptr;
    int rc = sa_init(&ctx, nullptr, st.gap_penalty, 64, st.num_devices > 0 ? st.devices[0] : 0);
    if (!rc) rc = sa_upload(ctx, codes.data(), offs.data(), lens.data(), (uint64_t)kernel_rows);
    const uint64_t rows = ctx ? sa_schedule_rows(ctx, SA_SCHEDULE_TRIANGULAR) : 0;
    if (!rc) rc = sa_prepare_resident(ctx, SA_SCHEDULE_TRIANGULAR, 0, rows);
    if (rc) { fprintf(stderr, "kernel sweep: %s\n", sa_strerror(rc)); return 1; }
    const double cells = (double)sa_cell_count(ctx, SA_SCHEDULE_TRIANGULAR, 0, rows);
    printf("\nTiles per item\tWaves per CU\tKernel (ms)\tGCUPS\n");
    const uint32_t tiles[] = {2, 4, 8, 16, 32};
    const uint32_t waves[] = {12, 15, 16, 17, 20};     // the library's own choice: what is resident (15 or 16) plus one
    for (uint32_t w : waves)
        for (uint32_t t : tiles) {
            sa_set_tuning(ctx, t, w);
            rc = sa_prepare_resident(ctx, SA_SCHEDULE_TRIANGULAR, 0, rows);
            float ms = 0, bestms = 1e30f;
            for (int rep = 0; rep < 3 && !rc; rep++) {
                rc = sa_run_resident(ctx, &ms);
                if (ms < bestms) bestms = ms;
            }
            if (rc) { fprintf(stderr, "kernel sweep: %s\n", sa_strerror(rc)); return 1; }
            printf("%u\t%u\t%.3f\t%.1f\n", t, w, bestms, cells / bestms * 1e-6);
        }
    // Length buckets: the library's automatic width first (0), then fixed widths.
    sa_set_tuning(ctx, 16, 17);
    printf("\nBucket width\tKernel (ms)\tGCUPS\tKernel\n");
    const uint32_t widths[] = {0, 1, 2, 4, 8, 16, 64};
    for (uint32_t bw : widths) {
        sa_set_bucket_width(ctx, bw);
        rc = sa_prepare_resident(ctx, SA_SCHEDULE_TRIANGULAR, 0, rows);
        float ms = 0, bestms = 1e30f;
        for (int rep = 0; rep < 3 && !rc; rep++) {
            rc = sa_run_resident(ctx, &ms);
            if (ms < bestms) bestms = ms;
        }
        if (rc) { fprintf(stderr, "bucket sweep: %s\n", sa_strerror(rc)); return 1; }
        printf("%u\t%.3f\t%.1f\t%s\n", bw, bestms, cells / bestms * 1e-6, sa_last_kernel(ctx));
    }
    sa_destroy(ctx);
    return 0;
}
