/*
 * ORACLE (test infrastructure only -- never linked into or called from the product path).
 *
 * CPU restatement of the reference's only native op on the RPN hot path:
 *   sort_vertices_forward(vertices f32[B,N,24,2], mask bool[B,N,24], num_valid i32[B,N]) -> i32[B,N,9]
 * Follows /root/reference/nerf_rpn/model/rotated_iou/cuda_op/sort_vert_kernel.cu:15-134
 * (comparator :15-40, per-polygon selection loop :53-131) and the host wrapper
 * sort_vert.cpp:6-33 (zero-initialised int32 output, m = vertices.size(2)).
 *
 * The .cu itself cannot be built here (needs ATen + CUDA headers and a CUDA device), so this
 * is a restatement, pinned by known answers and by the reference's Python IoU stack run on top
 * of it (tests/golden/make_golden.py).  Points the CUDA source leaves undefined are fixed here
 * and this file is the spec for them:
 *   - comparator falls off the end when y1 == 0 or y2 == 0  -> returns false;
 *   - `pad` is uninitialised when all 16 intersection slots are valid -> pad = m - 1;
 *   - num_valid > 8 would overflow the 9-slot row -> clamped to 8.
 * Arithmetic is done exactly as the CUDA source types imply: float products/sums, the 1e-8
 * EPSILON is a double literal so `a + EPSILON` is evaluated in double and rounded back to float.
 * Build with -ffp-contract=off (see Makefile) so no FMA contraction changes comparisons.
 */
#include <math.h>
#include <stdint.h>

#define NSLOT 9
#define ISECT0 8
#define EPS 1e-8

static int before(float x1, float y1, float x2, float y2)
{
    if (fabs((double)(x1 - x2)) < EPS && fabs((double)(y2 - y1)) < EPS) return 0;
    if (y1 > 0 && y2 < 0) return 1;
    if (y1 < 0 && y2 > 0) return 0;
    float n1 = (float)((double)(x1 * x1 + y1 * y1) + EPS);
    float n2 = (float)((double)(x2 * x2 + y2 * y2) + EPS);
    float q1 = fabsf(x1) * x1 / n1;
    float q2 = fabsf(x2) * x2 / n2;
    float d = q1 - q2;
    if (y1 > 0 && y2 > 0) return (double)d > EPS;
    if (y1 < 0 && y2 < 0) return (double)d < EPS;
    return 0; /* y == 0 on either side: undefined in the reference, defined false here */
}

static void sort_one(int m, const float *v, const uint8_t *msk, int nv, int32_t *out)
{
    int pad = m - 1;
    for (int j = ISECT0; j < m; ++j)
        if (!msk[j]) { pad = j; break; }
    if (nv < 3) {
        for (int j = 0; j < NSLOT; ++j) out[j] = pad;
        return;
    }
    if (nv > 8) nv = 8;
    for (int j = 0; j < nv; ++j) {
        float bx = 1.0f, by = (float)(-EPS);
        int take = 0;
        for (int k = 0; k < m; ++k) {
            if (!msk[k]) continue;
            float x = v[2 * k], y = v[2 * k + 1];
            if (j == 0) {
                if (before(x, y, bx, by)) { bx = x; by = y; take = k; }
            } else {
                int p = out[j - 1];
                if (before(x, y, bx, by) && before(v[2 * p], v[2 * p + 1], x, y)) { bx = x; by = y; take = k; }
            }
        }
        out[j] = take;
    }
    out[nv] = out[0];
    for (int j = nv + 1; j < NSLOT; ++j) out[j] = pad;
    if (nv == 8) { /* identical boxes: corners of both boxes coincide pairwise */
        int dup = 0;
        for (int j = 0; j < 4; ++j)
            for (int k = 4; k < ISECT0; ++k)
                if (out[k] == out[j]) ++dup;
        if (dup == 4) {
            out[4] = out[0];
            for (int j = 5; j < NSLOT; ++j) out[j] = pad;
        }
    }
}

/* vertices[b*n][m][2], mask[b*n][m] (0/1 bytes), num_valid[b*n] -> idx[b*n][9] */
void oracle_sort_vertices(int64_t bn, int m, const float *vertices, const uint8_t *mask,
                          const int32_t *num_valid, int32_t *idx)
{
    for (int64_t i = 0; i < bn; ++i)
        sort_one(m, vertices + i * m * 2, mask + i * m, num_valid[i], idx + i * NSLOT);
}
