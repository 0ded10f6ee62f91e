/* multi_driver.c -- a plain C consumer of the several-GPU entry points of include/am.h (am_multi_*): ONE process drives
 * every visible device (ncclCommInitAll inside libam), the automaton is broadcast over xGMI, a batch of haystacks is
 * scanned in contiguous blocks per device, counts are all-reduced, records concatenated in haystack order.
 * Then the same job with the haystacks RESIDENT on the devices (am_multi_batch_upload + am_multi_count_batch / am_multi_run_batch:
 * records stay in each device's HBM, only counts are all-reduced) must give the same totals, counts and records: "resident ok".
 *     devices <D>\n total <n>\n counts <c0> <c1> ...\n records <k>\n <haystack> <end_pos> <state>\n ... resident ok\n
 * The same output must come out for every number of devices (tests/test_multi.py compares it with the oracle).
 * Usage: multi_driver <transitions.u64> <offsets.u32> <root_ascii.u64> <values_len.u32> <haystacks> <n_hay> <case_mode> [n_devices]
 *        (the haystack file holds n_hay haystacks of equal length)
 * Exit code: 0 ok, 2 no device, 3 fewer devices than requested, 1 anything else. */
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

static void check(int rc, const char* what)
{
    if (rc == AM_OK) return;
    fprintf(stderr, "%s: %d (%s)\n", what, rc, am_last_error());
    exit(rc == AM_ERR_NO_DEVICE ? 2 : 1);
}

int main(int argc, char** argv)
{
    if (argc < 8) { fprintf(stderr, "usage: %s transitions offsets root_ascii values_len haystacks n_hay case_mode [n_devices]\n", argv[0]); return 1; }
    size_t nt, no, nr, nv, nh;
    uint64_t* transitions = (uint64_t*)slurp(argv[1], &nt);
    uint32_t* offsets = (uint32_t*)slurp(argv[2], &no);
    uint64_t* root_ascii = (uint64_t*)slurp(argv[3], &nr);
    uint32_t* values_len = (uint32_t*)slurp(argv[4], &nv);
    uint8_t* text = (uint8_t*)slurp(argv[5], &nh);
    const size_t n_hay = (size_t)atol(argv[6]);
    const int case_mode = atoi(argv[7]);
    const int want = argc > 8 ? atoi(argv[8]) : 0;
    const size_t n_states = no / 4 - 1;
    if (nr != 128 * 8 || nv != n_states * 4 || n_hay == 0 || nh % n_hay) { fprintf(stderr, "inconsistent array sizes\n"); return 1; }

    am_multi* m = NULL;
    int rc = am_multi_create(want, &m);
    if (rc == AM_ERR_INVALID && want > 1) { fprintf(stderr, "fewer than %d devices: %s\n", want, am_last_error()); return 3; }
    check(rc, "am_multi_create");
    const int D = am_multi_local_devices(m);

    /* build on the first device, broadcast the flattened image to all of them */
    am_automaton* a = NULL;
    check(am_automaton_create(transitions, nt / 8, offsets, n_states, root_ascii, values_len, &a), "am_automaton_create");
    am_automaton** autos = (am_automaton**)calloc((size_t)D, sizeof(am_automaton*));
    check(am_multi_broadcast_automaton(m, a, case_mode, 0, autos), "am_multi_broadcast_automaton");

    const size_t hay_len = nh / n_hay;
    am_slice* slices = (am_slice*)malloc(n_hay * sizeof(am_slice));
    for (size_t i = 0; i < n_hay; i++) { slices[i].ptr = text; slices[i].off = i * hay_len; slices[i].len = hay_len; }
    uint64_t* counts = (uint64_t*)calloc(n_hay, sizeof(uint64_t));
    uint64_t total = 0;
    check(am_multi_count(m, autos, case_mode, slices, n_hay, counts, &total), "am_multi_count");
    am_match* recs = NULL; size_t k = 0;
    check(am_multi_run(m, autos, case_mode, slices, n_hay, &recs, &k), "am_multi_run");

    printf("devices %d\ntotal %llu\ncounts", D, (unsigned long long)total);
    for (size_t i = 0; i < n_hay; i++) printf(" %llu", (unsigned long long)counts[i]);
    printf("\nrecords %llu\n", (unsigned long long)k);
    for (size_t i = 0; i < k; i++) printf("%u %llu %u\n", recs[i].haystack, (unsigned long long)recs[i].end_pos, recs[i].state);

    /* device-resident: block i of the haystacks lives on device i; nothing but counts crosses PCIe in the timed calls */
    {
        am_batch** batches = (am_batch**)calloc((size_t)D, sizeof(am_batch*));
        am_matches** res = (am_matches**)calloc((size_t)D, sizeof(am_matches*));
        uint64_t** dcounts = (uint64_t**)calloc((size_t)D, sizeof(uint64_t*));
        uint64_t* local_totals = (uint64_t*)calloc((size_t)D, sizeof(uint64_t));
        for (int i = 0; i < D; i++) {
            const size_t lo = n_hay * (size_t)i / (size_t)D, hi = n_hay * (size_t)(i + 1) / (size_t)D;
            dcounts[i] = (uint64_t*)calloc(hi - lo + 1, sizeof(uint64_t));
            if (hi > lo) check(am_multi_batch_upload(m, i, slices + lo, hi - lo, &batches[i]), "am_multi_batch_upload");
        }
        uint64_t total2 = 0, nrec2 = 0;
        check(am_multi_count_batch(m, autos, case_mode, batches, dcounts, local_totals, &total2), "am_multi_count_batch");
        check(am_multi_run_batch(m, autos, case_mode, batches, res, &nrec2), "am_multi_run_batch");
        int ok = total2 == total && nrec2 == (uint64_t)k;
        uint64_t sum_local = 0;
        size_t at = 0;
        for (int i = 0; i < D && ok; i++) {
            const size_t lo = n_hay * (size_t)i / (size_t)D, hi = n_hay * (size_t)(i + 1) / (size_t)D;
            sum_local += local_totals[i];
            for (size_t j = lo; j < hi; j++) ok = ok && dcounts[i][j - lo] == counts[j];
            const size_t kk = res[i] ? (size_t)am_matches_size(res[i]) : 0;
            const am_match* src = kk ? am_matches_data(res[i]) : NULL;
            for (size_t j = 0; j < kk && ok; j++)
                ok = at + j < k && src[j].haystack + (uint32_t)lo == recs[at + j].haystack && src[j].end_pos == recs[at + j].end_pos && src[j].state == recs[at + j].state;
            at += kk;
        }
        ok = ok && at == k && sum_local == total;
        printf("resident %s\n", ok ? "ok" : "MISMATCH");
        for (int i = 0; i < D; i++) { am_matches_free(res[i]); am_batch_destroy(batches[i]); free(dcounts[i]); }
        free(batches); free(res); free(dcounts); free(local_totals);
        if (!ok) return 1;
    }
    /* ONE haystack on all D devices: the whole text as a single document, cut into D ranges with one maximal match of overlap;
       the ranges' records, concatenated, and the all-reduced count must equal a one-device scan of the same document (autos[0], am_run / am_count) */
    {
        am_slice doc; doc.ptr = text; doc.off = 0; doc.len = nh;
        am_matches* whole = NULL;
        check(am_run(autos[0], case_mode, &doc, 1, &whole), "am_run (whole document)");
        const size_t kw = (size_t)am_matches_size(whole);
        const am_match* wsrc = kw ? am_matches_data(whole) : NULL;
        uint64_t count_whole = 0, count_single = 0, nrec_all = 0;
        check(am_count(autos[0], case_mode, &doc, 1, &count_whole), "am_count (whole document)");
        uint64_t* local_counts = (uint64_t*)calloc((size_t)D, sizeof(uint64_t));
        check(am_multi_count_single(m, autos, case_mode, &doc, local_counts, &count_single), "am_multi_count_single");
        am_match* srecs = NULL; size_t ks = 0;
        check(am_multi_run_single(m, autos, case_mode, &doc, &srecs, &ks, &nrec_all), "am_multi_run_single");
        int ok = count_single == count_whole && ks == kw && nrec_all == (uint64_t)kw;
        uint64_t sum_local = 0;
        for (int i = 0; i < D; i++) sum_local += local_counts[i];
        ok = ok && sum_local == count_whole;
        for (size_t j = 0; j < kw && ok; j++) ok = srecs[j].haystack == 0 && srecs[j].end_pos == wsrc[j].end_pos && srecs[j].state == wsrc[j].state;
        printf("single %s (%llu records, %llu matches)\n", ok ? "ok" : "MISMATCH", (unsigned long long)kw, (unsigned long long)count_whole);
        am_multi_matches_free(srecs); am_matches_free(whole); free(local_counts);
        if (!ok) return 1;
    }
    am_multi_matches_free(recs);
    for (int i = 0; i < D; i++) am_automaton_destroy(autos[i]);
    am_automaton_destroy(a);
    am_multi_destroy(m);
    free(autos); free(slices); free(counts); free(transitions); free(offsets); free(root_ascii); free(values_len); free(text);
    return 0;
}
