/*
 * sa_dataset.c -- benchmark inputs without Python.
 *
 *   sa_dataset synth N LO HI SEED out.csv
 *       N sequences from one splitmix64(SEED) stream; per sequence
 *       len = LO + next() % (HI-LO+1), residues ALPHABET[next() % 20], label = next() & 1.
 *       (Same stream as sequencealigner_amd/synth.py: both write identical files.)
 *   sa_dataset mega in.csv out.csv [FACTOR=1920]
 *       header + (body + body with its lines in reverse order) x FACTOR: the reference's
 *       enlarged data set (scripts/create_mega_dataset.py:121-167).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const char ALPHABET[] = "ARNDCQEGHILKMFPSTWYV";

static uint64_t sm_state;
static uint64_t sm_next(void) {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int cmd_synth(int argc, char** argv) {
    if (argc != 7) return 2;
    const long n = atol(argv[2]), lo = atol(argv[3]), hi = atol(argv[4]);
    sm_state = strtoull(argv[5], NULL, 10);
    if (n < 1 || lo < 1 || hi < lo) return 2;
    FILE* f = fopen(argv[6], "wb");
    if (!f) { perror(argv[6]); return 1; }
    fputs("sequence,label\n", f);
    char* buf = (char*)malloc((size_t)hi + 8);
    for (long s = 0; s < n; s++) {
        const long len = lo + (long)(sm_next() % (uint64_t)(hi - lo + 1));
        for (long i = 0; i < len; i++) buf[i] = ALPHABET[sm_next() % 20];
        buf[len] = ',';
        buf[len + 1] = (char)('0' + (sm_next() & 1));
        buf[len + 2] = '\n';
        fwrite(buf, 1, (size_t)len + 3, f);
    }
    free(buf);
    return fclose(f) ? 1 : 0;
}

static int cmd_mega(int argc, char** argv) {
    if (argc != 4 && argc != 5) return 2;
    const long factor = argc == 5 ? atol(argv[4]) : 1920;
    FILE* in = fopen(argv[2], "rb");
    if (!in) { perror(argv[2]); return 1; }
    fseek(in, 0, SEEK_END);
    const long size = ftell(in);
    fseek(in, 0, SEEK_SET);
    char* data = (char*)malloc((size_t)size + 1);
    if (fread(data, 1, (size_t)size, in) != (size_t)size) { fclose(in); return 1; }
    fclose(in);
    long hdr = 0;
    while (hdr < size && data[hdr] != '\n') hdr++;
    if (hdr < size) hdr++;
    const char* body = data + hdr;
    const long blen = size - hdr;
    /* body followed by its lines in reverse order (a last line without newline stays as it is) */
    char* rev = (char*)malloc((size_t)blen + 1);
    long w = 0, end = blen;
    while (end > 0) {
        long start = end - 1;
        while (start > 0 && body[start - 1] != '\n') start--;
        memcpy(rev + w, body + start, (size_t)(end - start));
        w += end - start;
        end = start;
    }
    FILE* out = fopen(argv[3], "wb");
    if (!out) { perror(argv[3]); return 1; }
    fwrite(data, 1, (size_t)hdr, out);
    for (long k = 0; k < factor; k++) {
        fwrite(body, 1, (size_t)blen, out);
        fwrite(rev, 1, (size_t)blen, out);
    }
    free(rev);
    free(data);
    return fclose(out) ? 1 : 0;
}

int main(int argc, char** argv) {
    int rc = 2;
    if (argc >= 2 && !strcmp(argv[1], "synth")) rc = cmd_synth(argc, argv);
    else if (argc >= 2 && !strcmp(argv[1], "mega")) rc = cmd_mega(argc, argv);
    if (rc == 2)
        fprintf(stderr, "usage: %s synth N LO HI SEED out.csv\n       %s mega in.csv out.csv [FACTOR]\n", argv[0], argv[0]);
    return rc;
}
