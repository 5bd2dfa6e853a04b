// sa_main.cpp -- command-line front end of the CSV-in / CSV-out driver.
// Prints the reference's "Alignment time: %f seconds" line (src/main.c:133), which
// is what the reference's GUI scrapes from the program output, and a JSON line
// with pairs, cells and throughput.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/sa_hip.h"
#include "../../include/sa_host.h"

static void usage(const char* argv0) {
    fprintf(stderr,
            "usage: %s [--config user.h] [--input in.csv] [--output out.csv]\n"
            "          [--schedule adjacent|triangular] [--gap N] [--max-seq-len N]\n"
            "          [--matrix file] [--devices N] [--threads N] [--pairs-per-submit N] [--no-write] [--no-pairing]\n"
            "          [--host-format]\n",
            argv0);
}

int main(int argc, char** argv) {
    sa_settings st;
    sa_settings_default(&st);
    for (int k = 1; k < argc; k++) {
        const char* a = argv[k];
        auto need = [&](const char* what) -> const char* {
            if (k + 1 >= argc) { fprintf(stderr, "%s needs a value\n", what); exit(2); }
            return argv[++k];
        };
        if (!strcmp(a, "--config")) {
            const char* path = need(a);
            if (sa_settings_load(&st, path) != SA_OK) { fprintf(stderr, "cannot read settings from %s\n", path); return 2; }
        } else if (!strcmp(a, "--input")) snprintf(st.input_file, sizeof(st.input_file), "%s", need(a));
        else if (!strcmp(a, "--output")) snprintf(st.output_file, sizeof(st.output_file), "%s", need(a));
        else if (!strcmp(a, "--schedule")) {
            const char* v = need(a);
            st.schedule = !strcmp(v, "triangular") ? SA_SCHEDULE_TRIANGULAR : SA_SCHEDULE_ADJACENT;
        } else if (!strcmp(a, "--gap")) st.gap_penalty = atoi(need(a));
        else if (!strcmp(a, "--max-seq-len")) st.max_seq_len = (uint32_t)atoi(need(a));
        else if (!strcmp(a, "--matrix")) snprintf(st.scoring_matrix_file, sizeof(st.scoring_matrix_file), "%s", need(a));
        else if (!strcmp(a, "--devices")) st.num_devices = atoi(need(a));
        else if (!strcmp(a, "--threads")) st.format_threads = atoi(need(a));
        else if (!strcmp(a, "--pairs-per-submit")) st.pairs_per_submit = strtoull(need(a), nullptr, 10);
        else if (!strcmp(a, "--no-write")) st.mode_write = 0;
        else if (!strcmp(a, "--no-pairing")) st.pairing = 0;
        else if (!strcmp(a, "--host-format")) st.device_format = 0;
        else { usage(argv[0]); return 2; }
    }
    char msg[256];
    if (sa_settings_validate(&st, msg, sizeof(msg)) != SA_OK) { fprintf(stderr, "settings: %s\n", msg); return 2; }
    sa_run_report rep;
    const int rc = sa_run_csv(&st, &rep);
    if (rc != SA_OK) { fprintf(stderr, "error %d: %s\n", rc, rep.message); return 1; }
    printf("Alignment time: %f seconds\n", rep.seconds);
    printf("{\"rows\": %llu, \"pairs\": %llu, \"cells\": %llu, \"bytes_written\": %llu, \"seconds\": %.6f, "
           "\"gcups\": %.3f, \"pairs_per_s\": %.1f, \"devices\": %d, \"seconds_ingest\": %.4f, \"seconds_upload\": %.4f, "
           "\"seconds_gpu_wait\": %.4f, \"seconds_format\": %.4f, \"seconds_write\": %.4f}\n",
           (unsigned long long)rep.rows, (unsigned long long)rep.pairs, (unsigned long long)rep.cells,
           (unsigned long long)rep.bytes_written, rep.seconds, rep.seconds > 0 ? rep.cells / rep.seconds / 1e9 : 0.0,
           rep.seconds > 0 ? rep.pairs / rep.seconds : 0.0, st.num_devices, rep.seconds_ingest, rep.seconds_upload,
           rep.seconds_gpu_wait, rep.seconds_format, rep.seconds_write);
    return 0;
}
