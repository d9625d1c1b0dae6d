/*
 * oracle/sanitize_driver.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Runs oracle/maxpath_oracle.c (included below, so the whole restatement is compiled with
 * -fsanitize=address,undefined -fno-sanitize-recover) on cases piped in by tests/test_oracle.py
 * and pipes the paths back.  Every buffer is its own exact-size heap block, so a read or write
 * outside an utterance's arrays aborts the run.  The reference turns these checks off
 * (core.pyx:7-8,38-39 boundscheck/wraparound False); SURVEY.md section 5 asks the build to turn
 * them on in its CPU tests.
 *
 * stdin : int32 n; n x { int32 B,Tx,Ty; float neg; int32 tx[B]; int32 ty[B]; float value[B*Tx*Ty] }
 * stdout: n x { int32 path[B*Tx*Ty]; float q[B*Tx*Ty] }   (q = the scores after the in-place sweep)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "maxpath_oracle.c"

static void need(size_t got, size_t want) {
    if (got != want) { fprintf(stderr, "sanitize_driver: short read/write\n"); exit(3); }
}

int main(void) {
    int32_t n;
    need(fread(&n, 4, 1, stdin), 1);
    for (int32_t c = 0; c < n; c++) {
        int32_t hdr[3];
        float neg;
        need(fread(hdr, 4, 3, stdin), 3);
        need(fread(&neg, 4, 1, stdin), 1);
        const size_t B = (size_t)hdr[0], cells = B * (size_t)hdr[1] * (size_t)hdr[2];
        int32_t *tx = malloc(B * 4), *ty = malloc(B * 4);
        float *value = malloc(cells * 4);
        int32_t *path = calloc(cells, 4);
        if (!tx || !ty || !value || !path) return 4;
        need(fread(tx, 4, B, stdin), B);
        need(fread(ty, 4, B, stdin), B);
        need(fread(value, 4, cells, stdin), cells);
        oracle_maxpath_c(path, value, tx, ty, hdr[0], hdr[1], hdr[2], neg);
        need(fwrite(path, 4, cells, stdout), cells);
        need(fwrite(value, 4, cells, stdout), cells);
        free(tx); free(ty); free(value); free(path);
    }
    return 0;
}
