/* abi_driver.c -- a plain C consumer of include/am.h (no Python, no C++): what a foreign-function binding does.
 * Reads a packed automaton (the reference's AcMachine arrays, Automaton.hs:108-123) and a haystack file, then calls
 * am_automaton_create / am_count / am_contains_any / am_run and prints the results as text:
 *     count <n>\n any <0|1>\n records <k>\n <end_pos> <state>\n ...
 * Usage: abi_driver <transitions.u64> <offsets.u32> <root_ascii.u64> <values_len.u32> <haystack> <case_mode>
 * Exit code: 0 ok, 2 no device (AM_ERR_NO_DEVICE), 1 anything else.  Built and run by tests/test_c_driver.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "am.h"

static void* slurp(const char* path, size_t* n)
{
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(1); }
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* p = malloc(len > 0 ? (size_t)len : 1);
    if (len > 0 && fread(p, 1, (size_t)len, f) != (size_t)len) { fprintf(stderr, "short read %s\n", path); exit(1); }
    fclose(f);
    *n = (size_t)len;
    return p;
}

static int check(int rc, const char* what)
{
    if (rc == AM_OK) return 0;
    fprintf(stderr, "%s: %d (%s)\n", what, rc, am_last_error());
    exit(rc == AM_ERR_NO_DEVICE ? 2 : 1);
}

int main(int argc, char** argv)
{
    if (argc != 7) { fprintf(stderr, "usage: %s transitions offsets root_ascii values_len haystack case_mode\n", argv[0]); return 1; }
    size_t nt, no, nr, nv, nh;
    uint64_t* transitions = (uint64_t*)slurp(argv[1], &nt);
    uint32_t* offsets = (uint32_t*)slurp(argv[2], &no);
    uint64_t* root_ascii = (uint64_t*)slurp(argv[3], &nr);
    uint32_t* values_len = (uint32_t*)slurp(argv[4], &nv);
    uint8_t* hay = (uint8_t*)slurp(argv[5], &nh);
    const int case_mode = atoi(argv[6]);
    const size_t n_states = no / 4 - 1;
    if (nr != 128 * 8 || nv != n_states * 4) { fprintf(stderr, "inconsistent array sizes\n"); return 1; }

    am_automaton* a = NULL;
    check(am_automaton_create(transitions, nt / 8, offsets, n_states, root_ascii, values_len, &a), "am_automaton_create");

    am_slice slice = { hay, 0, nh };
    uint64_t count = 0;
    check(am_count(a, case_mode, &slice, 1, &count), "am_count");
    uint8_t any = 0;
    check(am_contains_any(a, case_mode, &slice, 1, &any), "am_contains_any");
    am_matches* m = NULL;
    check(am_run(a, case_mode, &slice, 1, &m), "am_run");
    const uint64_t k = am_matches_size(m);
    const am_match* recs = am_matches_data(m);
    if (k && !recs) { fprintf(stderr, "am_matches_data: %s\n", am_last_error()); return 1; }
    printf("count %llu\nany %u\nrecords %llu\n", (unsigned long long)count, (unsigned)any, (unsigned long long)k);
    for (uint64_t i = 0; i < k; i++) printf("%llu %u\n", (unsigned long long)recs[i].end_pos, recs[i].state);
    am_matches_free(m);
    am_automaton_destroy(a);
    free(transitions); free(offsets); free(root_ascii); free(values_len); free(hay);
    return 0;
}
