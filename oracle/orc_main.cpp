// oracle/orc_main.cpp -- TEST INFRASTRUCTURE: command-line front end of the CPU restatement.
//   bmbs_oracle index  <genome.fa> [<prefix>]        (prefix defaults to the FASTA path)
//   bmbs_oracle search <prefix> --seq r.fq | --seq1 a.fq --seq2 b.fq  -o out.sam [-e f] [--sensitive] [--unmapped_out] [--ambiguous_out] [--pbat]
//                      [--min n] [--max n] [--phred64] [--mapstats file]
// Option names follow Process_CommandLines.cpp:88-132.
#include "bmbs_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>

static void print_stats(FILE* o, const int64_t st[5])
{
    long long reads = st[0], uniq = st[1], amb = st[2], unm = st[0] - st[1] - st[2];
    // Bitmapper_main.cpp:275-284
    fprintf(o, "%-48s%lld\n", "No. of Reads:", reads);
    fprintf(o, "%-48s%lld (%0.2f%%)\n", "No. of Unique Mapped Reads:", uniq, ((double)uniq / (double)reads) * 100);
    fprintf(o, "%-48s%lld (%0.2f%%)\n", "No. of Ambiguous Mapped Reads:", amb, ((double)amb / (double)reads) * 100);
    fprintf(o, "%-48s%lld (%0.2f%%)\n", "No. of Unmapped Reads:", unm, ((double)unm / (double)reads) * 100);
    fprintf(o, "%-47s %0.2f%%\n", "Mismatch and Indel Rate:", ((double)st[4] / (double)st[3]) * 100);
}

int main(int argc, char** argv)
{
    if (argc >= 3 && !strcmp(argv[1], "index")) {
        const char* prefix = argc >= 4 ? argv[3] : argv[2];
        return orc_index_build(argv[2], prefix) ? 1 : 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "search")) {
        orc_params P; orc_default_params(&P);
        const char *seq = 0, *seq1 = 0, *seq2 = 0, *out = "output", *mapstats = 0, *cl = 0;
        for (int i = 3; i < argc; i++) {
            if (!strcmp(argv[i], "--seq") && i + 1 < argc) seq = argv[++i];
            else if (!strcmp(argv[i], "--seq1") && i + 1 < argc) seq1 = argv[++i];
            else if (!strcmp(argv[i], "--seq2") && i + 1 < argc) seq2 = argv[++i];
            else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
            else if (!strcmp(argv[i], "-e") && i + 1 < argc) P.e_f = atof(argv[++i]);
            else if (!strcmp(argv[i], "--min") && i + 1 < argc) P.min_ins = atoi(argv[++i]);
            else if (!strcmp(argv[i], "--max") && i + 1 < argc) P.max_ins = atoi(argv[++i]);
            else if (!strcmp(argv[i], "--sensitive")) P.sensitive = 1;
            else if (!strcmp(argv[i], "--unmapped_out")) P.unmapped_out = 1;
            else if (!strcmp(argv[i], "--ambiguous_out")) P.ambiguous_out = 1;
            else if (!strcmp(argv[i], "--pbat")) P.pbat = 1;
            else if (!strcmp(argv[i], "--phred64")) P.q_base = 64;
            else if (!strcmp(argv[i], "--mp_max") && i + 1 < argc) P.mp_max = atoi(argv[++i]);        // Process_CommandLines.cpp:126-130
            else if (!strcmp(argv[i], "--mp_min") && i + 1 < argc) P.mp_min = atoi(argv[++i]);
            else if (!strcmp(argv[i], "--np") && i + 1 < argc) P.np = atoi(argv[++i]);
            else if (!strcmp(argv[i], "--gap_open") && i + 1 < argc) P.gap_open = atoi(argv[++i]);
            else if (!strcmp(argv[i], "--gap_extension") && i + 1 < argc) P.gap_ext = atoi(argv[++i]);
            else if (!strcmp(argv[i], "--mapstats") && i + 1 < argc) mapstats = argv[++i];
            else if (!strcmp(argv[i], "--cl") && i + 1 < argc) cl = argv[++i];
            else if (!strcmp(argv[i], "-t") && i + 1 < argc) ++i;
            else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
        }
        orc_index* ix = orc_index_load(argv[2]);
        if (!ix) { fprintf(stderr, "cannot load index %s\n", argv[2]); return 1; }
        int64_t st[5];
        int rc;
        if (seq) rc = orc_search_se(ix, &P, seq, out, cl, st);
        else if (seq1 && seq2) rc = orc_search_pe(ix, &P, seq1, seq2, out, cl, st);
        else { fprintf(stderr, "need --seq or --seq1/--seq2\n"); return 2; }
        if (rc) { fprintf(stderr, "search failed (%d)\n", rc); return 1; }
        print_stats(stderr, st);
        if (mapstats) { FILE* m = fopen(mapstats, "w"); if (m) { print_stats(m, st); fclose(m); } }
        orc_index_free(ix);
        return 0;
    }
    fprintf(stderr, "usage: bmbs_oracle index <fa> [prefix] | search <prefix> --seq r.fq -o out.sam ...\n");
    return 2;
}
