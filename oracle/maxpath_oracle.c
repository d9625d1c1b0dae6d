/*
 * oracle/maxpath_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the reference's monotonic
 * alignment search.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / the CPU
 * baseline -- the product path (aligner_amd/) never links or calls it.
 *
 * Restates (reference paths relative to /root/reference):
 *   oracle_maxpath_each   <- monotonic_align/core.pyx:7-35  (maximum_path_each)
 *   oracle_maxpath_c      <- monotonic_align/core.pyx:38-45 (maximum_path_c, the
 *                            prange is a plain serial loop: setup.py:5-9 passes
 *                            no -fopenmp, so the shipped build is serial)
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this file bit-for-bit
 * against (a) the golden vectors in tests/golden/ that were produced by the
 * reference's own core.c compiled in the authoring container
 * (tests/golden/make_golden.py) and (b) oracle/_ref when that build is present.
 *
 * Build: see oracle/Makefile (gcc -O2 -fno-fast-math -ffp-contract=off; the
 * reference is built -O2 without fast-math by distutils).
 */
#include <stdint.h>
#include <stddef.h>

/* core.pyx:7-35.  value is [Tx_pad, Ty_pad] row-major (mel axis contiguous),
 * mutated in place into the running score Q; path must be pre-zeroed by the
 * caller (reference __init__.py:15) and only receives the ones. */
static void oracle_maxpath_each(int32_t *path, float *value, int t_x, int t_y,
                                ptrdiff_t row_stride, float max_neg_val)
{
    int x, y;
    float v_prev, v_cur;
    int index = t_x - 1;                                   /* core.pyx:15 */

    for (y = 0; y < t_y; y++) {                            /* core.pyx:17 */
        int x_lo = t_x + y - t_y;                          /* core.pyx:18 */
        int x_hi = (y + 1 < t_x) ? (y + 1) : t_x;
        if (x_lo < 0) x_lo = 0;
        for (x = x_lo; x < x_hi; x++) {
            if (x == y) {                                  /* core.pyx:19-22 */
                v_cur = max_neg_val;
            } else {
                v_cur = value[x * row_stride + (y - 1)];
            }
            if (x == 0) {                                  /* core.pyx:23-29 */
                if (y == 0) v_prev = 0.0f;
                else        v_prev = max_neg_val;
            } else {
                v_prev = value[(x - 1) * row_stride + (y - 1)];
            }
            /* core.pyx:30; max() as Cython lowers it (core.c:19384-19396):
             * (v_prev > v_cur) ? v_prev : v_cur, i.e. NaN picks v_cur. */
            {
                float m = (v_prev > v_cur) ? v_prev : v_cur;
                value[x * row_stride + y] = m + value[x * row_stride + y];
            }
        }
    }

    for (y = t_y - 1; y > -1; y--) {                       /* core.pyx:32-35 */
        path[index * row_stride + y] = 1;
        if (index != 0 &&
            (index == y ||
             value[index * row_stride + (y - 1)] <
                 value[(index - 1) * row_stride + (y - 1)])) {
            index = index - 1;
        }
    }
}

/* core.pyx:40-45: paths[B,Tx,Ty] int32 (pre-zeroed), values[B,Tx,Ty] float32
 * (mutated), t_xs[B], t_ys[B] int32.  Same undefined behaviour as the
 * reference for t_x == 0 / t_x > t_y -- callers in tests keep 1 <= t_x <= t_y
 * unless they are probing the degenerate case on purpose with padded buffers. */
void oracle_maxpath_c(int32_t *paths, float *values, const int32_t *t_xs,
                      const int32_t *t_ys, int b, int tx_pad, int ty_pad,
                      float max_neg_val)
{
    int i;
    for (i = 0; i < b; i++) {                              /* core.pyx:44 */
        size_t off = (size_t)i * (size_t)tx_pad * (size_t)ty_pad;
        oracle_maxpath_each(paths + off, values + off, t_xs[i], t_ys[i],
                            (ptrdiff_t)ty_pad, max_neg_val); /* core.pyx:45 */
    }
}

/* Same loop with the batch split over `nthreads` OpenMP threads when this file
 * is compiled with -fopenmp (mirrors what prange at core.pyx:44 would do if the
 * reference were built with OpenMP).  Used only for the "all cores" CPU line. */
void oracle_maxpath_c_omp(int32_t *paths, float *values, const int32_t *t_xs,
                          const int32_t *t_ys, int b, int tx_pad, int ty_pad,
                          float max_neg_val, int nthreads)
{
    int i;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (i = 0; i < b; i++) {
        size_t off = (size_t)i * (size_t)tx_pad * (size_t)ty_pad;
        oracle_maxpath_each(paths + off, values + off, t_xs[i], t_ys[i],
                            (ptrdiff_t)ty_pad, max_neg_val);
    }
    (void)nthreads;
}

int oracle_has_openmp(void)
{
#ifdef _OPENMP
    return 1;
#else
    return 0;
#endif
}
