/*
 * ref_config.h -- the settings the reference takes at compile time, supplied
 * from outside its tree so that its sources can be compiled where they lie
 * (read-only) by oracle/Makefile.
 *
 * The reference keeps its settings in include/user.h and expects that file to
 * be rewritten before every build (its GUI does so: scripts/config_schema.py:201-241);
 * the shipped copy leaves INPUT_FILE and OUTPUT_FILE empty (include/user.h:59-60),
 * so it does not compile as it stands.  This header is force-included first and
 * defines that file's include guard, which makes the compiler skip it; every
 * value below is the shipped default unless the build line overrides it with -D.
 * The two paths become run-time values read from the environment.
 *
 * Test infrastructure only: used for oracle/_ref builds, never by the product.
 */
#ifndef SA_REF_CONFIG_H
#define SA_REF_CONFIG_H
#define USER_H

#define _GNU_SOURCE   /* as include/macros.h:32, needed before the first libc header */
#include <stdlib.h>

#ifndef MAX_CSV_LINE
#define MAX_CSV_LINE 256
#endif
#ifndef MAX_SEQ_LEN
#define MAX_SEQ_LEN 64
#endif
#ifndef GAP_PENALTY
#define GAP_PENALTY -4
#endif
#ifndef READ_CSV_HEADER
#define READ_CSV_HEADER "sequence,label\n"
#endif
#ifndef READ_CSV_SEQ_POS
#define READ_CSV_SEQ_POS 0
#endif
#ifndef READ_CSV_COLS
#define READ_CSV_COLS 2
#endif
#ifndef WRITE_CSV_HEADER
#define WRITE_CSV_HEADER "sequence1,sequence2,label1,label2,score,alignment,matches,mismatches,gaps,similarity\n"
#endif
#ifndef WRITE_CSV_SEQ1_POS
#define WRITE_CSV_SEQ1_POS 0
#endif
#ifndef WRITE_CSV_SCORE_POS
#define WRITE_CSV_SCORE_POS 4
#endif
#ifndef WRITE_CSV_ALIGN_POS
#define WRITE_CSV_ALIGN_POS 5
#endif
#ifndef WRITE_CSV_MATCHES_POS
#define WRITE_CSV_MATCHES_POS 6
#endif
#ifndef WRITE_CSV_MISMATCHES_POS
#define WRITE_CSV_MISMATCHES_POS 7
#endif
#ifndef WRITE_CSV_GAPS_POS
#define WRITE_CSV_GAPS_POS 8
#endif
#ifndef WRITE_CSV_SIMILARITY_POS
#define WRITE_CSV_SIMILARITY_POS 9
#endif
#ifndef WRITE_CSV_ALIGN_FMT
#define WRITE_CSV_ALIGN_FMT "\"('%s', '%s')\""
#endif
#define INPUT_FILE getenv("SA_REF_INPUT")
#define OUTPUT_FILE getenv("SA_REF_OUTPUT")
#ifndef MODE_MULTITHREAD
#define MODE_MULTITHREAD 0
#endif
#ifndef SIMILARITY_ANALYSIS
#define SIMILARITY_ANALYSIS 1
#endif
#ifndef MODE_WRITE
#define MODE_WRITE 1
#endif
#ifndef BATCH_SIZE
#define BATCH_SIZE 32768
#endif
#define KiB (1ULL << 10)
#define MiB (KiB << 10)
#define GiB (MiB << 10)
#define WRITE_BUF (128 * KiB)

#endif
