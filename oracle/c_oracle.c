/*
 * TEST INFRASTRUCTURE -- scalar C restatement of the reference's geometry hot path.
 *
 * Second oracle beside oracle/np_oracle.py.  np_oracle.py follows the reference line by line
 * and lets NumPy/OpenBLAS pick the summation order of the 4x4 * 4xN products; this file spells
 * that order out: every row of every matrix product is the chain
 *     acc = m0*x;  acc = fma(m1, y, acc);  acc = fma(m2, z, acc);  acc = fma(m3, w, acc)
 * which is what OpenBLAS' FMA dgemm micro-kernels compute for K = 4 (checked bit-for-bit against
 * NumPy in tests/test_c_oracle.py) and what the HIP kernels compute, so GPU float64 outputs can
 * be compared to this file bit-for-bit while np_oracle.py pins integers/masks and the 1e-5 bound.
 *
 * Reference lines restated (abbreviations as in SURVEY.md):
 *   project_points                  IH:46-72      (mspa_c_project_points)
 *   check_point_in_image_boundary   IH:337-344  \
 *   check_point_visibility_by_depth IH:346-373   > (visibility inside both functions)
 *   check_point_visibility          IH:375-386  /
 *   project_mask_to_3d              OPS:235-329   (first half of mspa_c_frame_pair)
 *   calculate_camera_overlap        CFR:102-137   (mspa_c_pair_overlap)
 *
 * Built by oracle/Makefile into oracle/_build/libmspa_c_oracle.so with -ffp-contract=off so the
 * compiler neither fuses nor splits any operation.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* one row of a 4x4 (row-major) matrix times (x, y, z, w) in BLAS order */
static inline double row4(const double *m, double x, double y, double z, double w) {
    double acc = m[0] * x;
    acc = fma(m[1], y, acc);
    acc = fma(m[2], z, acc);
    acc = fma(m[3], w, acc);
    return acc;
}

/* np.round(v).astype(int) followed by np.clip(.., 0, hi): half-to-even, then the x86-64
 * double->int64 conversion numpy performs (NaN and out-of-range become INT64_MIN), then clip. */
static inline int64_t round_clip(double v, int64_t hi) {
    double r = rint(v);
    int64_t i;
    if (!(r >= -9223372036854775808.0 && r < 9223372036854775808.0))
        i = INT64_MIN;
    else
        i = (int64_t)r;
    if (i < 0) i = 0;
    if (i > hi) i = hi;
    return i;
}

/* IH:46-72 on pre-inverted extrinsics.  points [n,3] (w = 1 appended as in IH:328-330),
 * einv = inv(E) row-major 4x4, k = K row-major 4x4.  uv [n,2], depth [n]. */
void mspa_c_project_points(const double *points, int64_t n, int64_t stride, const double *einv,
                           const double *k, double *uv, double *depth) {
    for (int64_t i = 0; i < n; i++) {
        const double *p = points + i * stride;
        double c[4], im[4];
        for (int r = 0; r < 4; r++) c[r] = row4(einv + 4 * r, p[0], p[1], p[2], 1.0);
        for (int r = 0; r < 4; r++) im[r] = row4(k + 4 * r, c[0], c[1], c[2], c[3]);
        uv[2 * i + 0] = im[0] / im[2];
        uv[2 * i + 1] = im[1] / im[2];
        depth[i] = c[2];
    }
}

/* IH:337-386 for one projected point. */
static inline int visible_at(double u, double v, double d, const uint16_t *depth_img, int dh, int dw,
                             int H, int W, double sx, double sy, int64_t *xi_out, int64_t *yi_out) {
    int inb = (u >= 0) && (u < (double)W) && (v >= 0) && (v < (double)H);
    int64_t xi = round_clip(u * sx, dw - 1);
    int64_t yi = round_clip(v * sy, dh - 1);
    double dv = (double)depth_img[yi * dw + xi] * 0.001;
    if (xi_out) *xi_out = xi;
    if (yi_out) *yi_out = yi;
    return inb && (d > 0) && (d < dv);
}

/* HOT LOOP 1 body (CFR:152-157): vertices -> one image.  mask [n] bytes; uv/depth optional. */
void mspa_c_vertex_visibility(const double *points, int64_t n, int64_t stride, const double *einv,
                              const double *k, const uint16_t *depth_img, int dh, int dw, int H, int W,
                              uint8_t *mask, double *uv, double *depth) {
    double sx = (double)dw / (double)W, sy = (double)dh / (double)H;
    for (int64_t i = 0; i < n; i++) {
        const double *p = points + i * stride;
        double c[4], im[4];
        for (int r = 0; r < 4; r++) c[r] = row4(einv + 4 * r, p[0], p[1], p[2], 1.0);
        for (int r = 0; r < 4; r++) im[r] = row4(k + 4 * r, c[0], c[1], c[2], c[3]);
        double u = im[0] / im[2], v = im[1] / im[2];
        mask[i] = (uint8_t)visible_at(u, v, c[2], depth_img, dh, dw, H, W, sx, sy, 0, 0);
        if (uv) { uv[2 * i] = u; uv[2 * i + 1] = v; }
        if (depth) depth[i] = c[2];
    }
}

/* Composite frame pair: OPS:235-329 over the whole colour grid of frame 1, then IH:313-386 in
 * frame 2.  Matrices row-major 4x4: kinv = inv(K), e1 = E1 (camera->world), a = A,
 * einv2 = inv(A @ E2), k = K.  Dense outputs over P = H*W pixels (any pointer may be NULL):
 *   valid u8, xyz f64x3 (NaN where !valid), uv2 f64x2, depth2 f64, xiyi i64x2, vis u8.      */
void mspa_c_frame_pair(const uint16_t *depth1, const uint16_t *depth2, int dh, int dw, int H, int W,
                       const double *kinv, const double *e1, const double *a, const double *einv2,
                       const double *k, uint8_t *valid, double *xyz, double *uv2, double *depth2_out,
                       int64_t *xiyi, uint8_t *vis, int64_t *counts) {
    double sx = (double)dw / (double)W, sy = (double)dh / (double)H;
    int64_t n_valid = 0, n_vis = 0;
    for (int my = 0; my < H; my++) {
        for (int mx = 0; mx < W; mx++) {
            int64_t i = (int64_t)my * W + mx;
            int64_t dy = round_clip((double)my * sy, dh - 1);       /* OPS:285-290 */
            int64_t dx = round_clip((double)mx * sx, dw - 1);
            double d = (double)depth1[dy * dw + dx] * 0.001;        /* OPS:292-294 */
            int ok = d > 0;                                         /* OPS:297 */
            if (valid) valid[i] = (uint8_t)ok;
            if (!ok) {
                if (xyz) { xyz[3 * i] = xyz[3 * i + 1] = xyz[3 * i + 2] = NAN; }
                if (uv2) { uv2[2 * i] = uv2[2 * i + 1] = NAN; }
                if (depth2_out) depth2_out[i] = NAN;
                if (xiyi) { xiyi[2 * i] = xiyi[2 * i + 1] = 0; }
                if (vis) vis[i] = 0;
                continue;
            }
            n_valid++;
            double px = (double)mx * d, py = (double)my * d;        /* OPS:303-310 */
            double c[4], w[4], al[4], c2[4], im[4];
            for (int r = 0; r < 4; r++) c[r] = row4(kinv + 4 * r, px, py, d, 1.0);          /* OPS:313 */
            for (int r = 0; r < 4; r++) w[r] = row4(e1 + 4 * r, c[0], c[1], c[2], c[3]);    /* OPS:316 */
            for (int r = 0; r < 4; r++) al[r] = row4(a + 4 * r, w[0], w[1], w[2], w[3]);    /* OPS:320 */
            /* IH:328-330 re-homogenises with a fresh 1 */
            for (int r = 0; r < 4; r++) c2[r] = row4(einv2 + 4 * r, al[0], al[1], al[2], 1.0);   /* IH:60 */
            for (int r = 0; r < 4; r++) im[r] = row4(k + 4 * r, c2[0], c2[1], c2[2], c2[3]);     /* IH:66 */
            double u = im[0] / im[2], v = im[1] / im[2];            /* IH:69 */
            int64_t xi, yi;
            int vz = visible_at(u, v, c2[2], depth2, dh, dw, H, W, sx, sy, &xi, &yi);
            n_vis += vz;
            if (xyz) { xyz[3 * i] = al[0]; xyz[3 * i + 1] = al[1]; xyz[3 * i + 2] = al[2]; }
            if (uv2) { uv2[2 * i] = u; uv2[2 * i + 1] = v; }
            if (depth2_out) depth2_out[i] = c2[2];
            if (xiyi) { xiyi[2 * i] = xi; xiyi[2 * i + 1] = yi; }
            if (vis) vis[i] = (uint8_t)vz;
        }
    }
    if (counts) { counts[0] = n_valid; counts[1] = n_vis; }
}

/* CFR:102-137 over byte masks: returns overlap in percent, NaN for an empty union. */
double mspa_c_pair_overlap(const uint8_t *m1, const uint8_t *m2, int64_t n, int64_t *inter_out,
                           int64_t *union_out) {
    int64_t inter = 0, uni = 0;
    for (int64_t i = 0; i < n; i++) {
        inter += (m1[i] && m2[i]);
        uni += (m1[i] || m2[i]);
    }
    if (inter_out) *inter_out = inter;
    if (union_out) *union_out = uni;
    return (double)inter / (double)uni * 100;
}

/* 4x4 row-major product in the same chain order (used for inv(E1) @ E2, CME:186). */
void mspa_c_matmul4(const double *a, const double *b, double *out) {
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            double acc = a[4 * r] * b[c];
            acc = fma(a[4 * r + 1], b[4 + c], acc);
            acc = fma(a[4 * r + 2], b[8 + c], acc);
            acc = fma(a[4 * r + 3], b[12 + c], acc);
            out[4 * r + c] = acc;
        }
}
